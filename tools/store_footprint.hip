// store_footprint.hip -- how fast can a launch WRITE a footprint that fits the 256 MiB Infinity Cache?  (round 5)
// grad at E = 1e5 writes 84 MB per launch, launch after launch to the same arrays; its loop phase moves ~5.3 TB/s of stores
// (profiles/r05/tiles_grad_100000_before.txt).  If a cache policy let the Infinity Cache absorb such a footprint faster than HBM
// takes it, short launches would have a lever; this probe writes 84 MB / 168 MB / 840 MB with every policy, back to back.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_footprint.hip -o build/store_footprint && build/store_footprint
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

// POLICY 0 plain, 1 nt, 2 sc0 sc1 (write-through), 3 sc1, 4 sc0 sc1 nt.  Each wave writes 4480-byte tiles (grad's plane tile), cyclic.
template <int POLICY>
__global__ __launch_bounds__(256, 2) void writer(char* out, long tiles) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    const v2d v = {1.0, 2.0};
    for (long t = wave; t < tiles; t += nw) {
        char* p = out + t * 4480;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if (c == 4 && lane >= 24) continue;
            v2d* q = reinterpret_cast<v2d*>(p + c * 1024 + lane * 16);
            if (POLICY == 0) *q = v;
            else if (POLICY == 1) __builtin_nontemporal_store(v, q);
            else if (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory");
            else if (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(q), "v"(v) : "memory");
        }
    }
}

template <int POLICY>
int run(char* buf, long bytes, const char* name) {
    const long tiles = bytes / 4480;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(writer<POLICY>, dim3(512), dim3(256), 0, 0, buf, tiles);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(writer<POLICY>, dim3(512), dim3(256), 0, 0, buf, tiles);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double us = best / 50 * 1e3;
    printf("  %-22s %8.2f us per launch = %5.2f TB/s (launch boundaries included)\n", name, us, tiles * 4480.0 / us * 1e-6);
    return 0;
}

int main() {
    char* buf;
    CK(hipMalloc(&buf, 900l << 20));
    for (long mb : {42l, 84l, 168l, 420l, 840l}) {
        printf("# %ld MB written per launch, 50 launches back to back on the same array\n", mb);
        run<0>(buf, mb * 1000000, "plain");
        run<1>(buf, mb * 1000000, "nt");
        run<2>(buf, mb * 1000000, "sc0 sc1 (write-through)");
        run<3>(buf, mb * 1000000, "sc1");
        run<4>(buf, mb * 1000000, "sc0 sc1 nt");
    }
    return 0;
}
