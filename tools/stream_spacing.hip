// stream_spacing.hip -- does the bandwidth of S concurrent sequential streams depend on how far
// apart they lie?  S streams of `len` bytes each start `delta` bytes apart inside one allocation;
// the persistent grid walks all of them in lockstep (wave w reads -- or writes -- piece w, w + W,
// ... of EVERY stream), like the DG kernels walk their 13-26 operand streams.
//   build/stream_spacing <S> <len MiB> <mode r|w|rw> <delta_start MiB> <delta_stop MiB> <delta_step MiB> [base MiB [base_stop base_step]]
// One line per (base, delta): GB/s (median of 5 x 10 launches).  With a base range the layout is moved through ONE
// allocation: where in (physical) memory a stream lies matters as much as how far apart the streams are.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// piece = 4 KiB per wave and stream (4 x 1 KiB wave-instructions), as a DG tile
template <int MODE>   // 0 read, 1 write, 2 read stream s and write stream s + S/2
__global__ __launch_bounds__(256, 2) void streams_kernel(char* base, long delta, int S, long pieces, double* sink) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    v2d acc = {0.0, 0.0};
    for (long p = wave; p < pieces; p += nw) {
        for (int s = 0; s < S; ++s) {
            char* q = base + s * delta + p * 4096 + lane * 16;
            if (MODE == 0 || (MODE == 2 && s < S / 2)) {
#pragma unroll
                for (int c = 0; c < 4; ++c) acc += __builtin_nontemporal_load(reinterpret_cast<const v2d*>(q + c * 1024));
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) __builtin_nontemporal_store(v2d{(double)p, (double)s}, reinterpret_cast<v2d*>(q + c * 1024));
            }
        }
    }
    if (acc[0] + acc[1] == 1.2345e-300) sink[0] = acc[0];
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: stream_spacing S lenMiB r|w|rw d0 d1 dstep [baseMiB]\n"); return 1; }
    const int S = atoi(argv[1]);
    const long MIB = 1l << 20, len = atol(argv[2]) * MIB;
    const char* mode = argv[3];
    const long d0 = atol(argv[4]) * MIB, d1 = atol(argv[5]) * MIB, ds = atol(argv[6]) * MIB;
    const long base0 = argc > 7 ? atol(argv[7]) * MIB : 0;
    const long base1 = argc > 9 ? atol(argv[8]) * MIB : base0, bstep = argc > 9 ? atol(argv[9]) * MIB : MIB;
    char* arena;
    const size_t total = (size_t)base1 + (size_t)(S - 1) * d1 + len + 4 * MIB;
    CK(hipMalloc(&arena, total));
    CK(hipMemset(arena, 0, total));
    double* sink;
    CK(hipMalloc(&sink, 64));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = 2 * prop.multiProcessorCount;
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const long pieces = len / 4096;
    long base_off = base0;
    auto launch = [&](long delta) {
        if (!strcmp(mode, "r")) hipLaunchKernelGGL(streams_kernel<0>, dim3(grid), dim3(256), 0, 0, arena + base_off, delta, S, pieces, sink);
        else if (!strcmp(mode, "w")) hipLaunchKernelGGL(streams_kernel<1>, dim3(grid), dim3(256), 0, 0, arena + base_off, delta, S, pieces, sink);
        else hipLaunchKernelGGL(streams_kernel<2>, dim3(grid), dim3(256), 0, 0, arena + base_off, delta, S, pieces, sink);
    };
    printf("# S=%d streams of %ld MiB, mode %s, arena %p, %d blocks\n", S, len / MIB, mode, (void*)arena, grid);
    for (base_off = base0; base_off <= base1; base_off += bstep)
    for (long delta = std::max(d0, len); delta <= d1; delta += ds) {
        for (int i = 0; i < 5; ++i) launch(delta);
        std::vector<float> ts;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(t0));
            for (int i = 0; i < 10; ++i) launch(delta);
            CK(hipEventRecord(t1));
            CK(hipEventSynchronize(t1));
            float ms; CK(hipEventElapsedTime(&ms, t0, t1));
            ts.push_back(ms / 10);
        }
        std::sort(ts.begin(), ts.end());
        printf("delta %6ld MiB  %8.1f GB/s  (%.4f ms)  base %6ld MiB\n", delta / MIB, (double)S * len / (ts[2] * 1e-3) * 1e-9, ts[2],
               base_off / MIB);
        fflush(stdout);
    }
    return 0;
}
