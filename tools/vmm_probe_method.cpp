// vmm_probe_method.cpp -- which probe classifies a 32 MiB handle reliably and cheaply?  (fe_split_alloc's first version
// misclassified: profiles/r03/split_alloc_check_v2_misclassified.txt.)  References: 2 GiB handles whose superclass
// relation is known from the robust probe (2 x 128 MiB, 23 launches).  Every small handle is then measured
//   R  robust: against both references' first 32 MiB, 32 MiB x 8 passes, 3 + 5 x 4 launches (median)
//   Q  quick : the same with 1 + 2 launches
//   A  quick against 32 MiB ANCHOR handles (small handles that R put into either class)
//   L  long  : 32 MiB x 32 passes in ONE launch (1 warm-up + 1 timed)
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/vmm_probe_method.cpp -o build/vmm_probe_method
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
static const size_t MIB = 1ull << 20, GIB = 1ull << 30;
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256, 2) void wprobe_kernel(char* a, char* b, long pieces, int passes) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    for (int r = 0; r < passes; ++r)
        for (long p = wave; p < pieces; p += nw) {
#pragma unroll
            for (int c = 0; c < 4; ++c) __builtin_nontemporal_store(v2d{0.0, 0.0}, reinterpret_cast<v2d*>(a + p * 4096 + c * 1024 + lane * 16));
#pragma unroll
            for (int c = 0; c < 4; ++c) __builtin_nontemporal_store(v2d{0.0, 0.0}, reinterpret_cast<v2d*>(b + p * 4096 + c * 1024 + lane * 16));
        }
}
static hipStream_t s;
static hipEvent_t e0, e1;
static double gbps(char* a, char* b, size_t bytes, int passes, int warm, int reps, int n) {
    const long pieces = (long)(bytes / 4096);
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(wprobe_kernel, dim3(512), dim3(256), 0, s, a, b, pieces, passes);
    std::vector<double> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(wprobe_kernel, dim3(512), dim3(256), 0, s, a, b, pieces, passes);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms / n);
    }
    std::sort(ts.begin(), ts.end());
    return 2.0 * bytes * passes / (ts[ts.size() / 2] * 1e-3) * 1e-9;
}
int main() {
    CK(hipSetDevice(0));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // references: 2 GiB handles
    const int NB = 12;
    char* bva;
    CK(hipMemAddressReserve((void**)&bva, 2 * GIB * NB, 2 * MIB, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> bh(NB);
    for (int i = 0; i < NB; ++i) {
        CK(hipMemCreate(&bh[i], 2 * GIB, &prop, 0));
        CK(hipMemMap(bva + 2 * GIB * i, 2 * GIB, 0, bh[i], 0));
        CK(hipMemSetAccess(bva + 2 * GIB * i, 2 * GIB, &acc, 1));
    }
    int other = -1;
    printf("# 2 GiB handles against handle 0, robust probe (GB/s):");
    for (int i = 1; i < NB; ++i) {
        const double r = gbps(bva, bva + 2 * GIB * i, 128 * MIB, 2, 3, 5, 4);
        printf(" %.0f", r);
        if (other < 0 && r > 6500) other = i;
    }
    printf("\n");
    if (other < 0) { printf("# no other superclass among the references\n"); return 0; }
    char* ref[2] = {bva, bva + 2 * GIB * other};
    printf("# references: handle 0 and handle %d (other superclass)\n", other);
    // small handles, one every 1 GiB of allocation order
    const int NS = 40;
    const size_t Q = 32 * MIB;
    char* sva;
    CK(hipMemAddressReserve((void**)&sva, Q * NS, 2 * MIB, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> sh(NS);
    for (int i = 0; i < NS; ++i) {
        CK(hipMemCreate(&sh[i], Q, &prop, 0));
        CK(hipMemMap(sva + Q * i, Q, 0, sh[i], 0));
        CK(hipMemSetAccess(sva + Q * i, Q, &acc, 1));
        hipMemGenericAllocationHandle_t sp;
        CK(hipMemCreate(&sp, GIB - Q, &prop, 0));
    }
    std::vector<double> R0(NS), R1(NS);
    for (int i = 0; i < NS; ++i) { R0[i] = gbps(ref[0], sva + Q * i, Q, 8, 3, 5, 4); R1[i] = gbps(ref[1], sva + Q * i, Q, 8, 3, 5, 4); }
    int anchor[2] = {-1, -1};
    for (int i = 0; i < NS; ++i) {
        if (anchor[0] < 0 && R0[i] < 5700 && R1[i] > 6500) anchor[0] = i;
        if (anchor[1] < 0 && R1[i] < 5700 && R0[i] > 6500) anchor[1] = i;
    }
    printf("# anchors (32 MiB handles): %d (class of reference 0), %d (class of reference 1)\n", anchor[0], anchor[1]);
    printf("# handle:  R vs ref0 / ref1 |  Q vs ref0 / ref1 |  L vs ref0 / ref1 |  A (quick) vs anchor0 / anchor1 | A robust vs anchor0 / anchor1\n");
    for (int i = 0; i < NS; ++i) {
        char* c = sva + Q * i;
        const double q0 = gbps(ref[0], c, Q, 8, 1, 1, 2), q1 = gbps(ref[1], c, Q, 8, 1, 1, 2);
        const double l0 = gbps(ref[0], c, Q, 32, 1, 1, 1), l1 = gbps(ref[1], c, Q, 32, 1, 1, 1);
        double a0 = 0, a1 = 0, ar0 = 0, ar1 = 0;
        if (anchor[0] >= 0 && anchor[0] != i) { a0 = gbps(sva + Q * anchor[0], c, Q, 8, 1, 1, 2); ar0 = gbps(sva + Q * anchor[0], c, Q, 8, 3, 5, 4); }
        if (anchor[1] >= 0 && anchor[1] != i) { a1 = gbps(sva + Q * anchor[1], c, Q, 8, 1, 1, 2); ar1 = gbps(sva + Q * anchor[1], c, Q, 8, 3, 5, 4); }
        printf("%3d:  %5.0f / %5.0f  |  %5.0f / %5.0f  |  %5.0f / %5.0f  |  %5.0f / %5.0f  |  %5.0f / %5.0f\n", i, R0[i], R1[i], q0, q1, l0, l1, a0, a1, ar0, ar1);
    }
    return 0;
}
