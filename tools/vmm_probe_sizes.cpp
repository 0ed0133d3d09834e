// vmm_probe_sizes.cpp -- can a handle of 128 / 256 / 512 MiB be classified by the two-stream write probe, against a 2 GiB
// reference handle and against a handle of its own size?  (fe_split_alloc with 128 MiB pieces saw ONE class across the
// whole memory: profiles/r03/split_alloc_check_v4_whole_memory_one_class.txt.)  Handles of every size are created along
// one walk through the allocator's memory (a spacer behind each, so that consecutive handles lie 2 GiB apart), each probed
// (2 x 128 MiB x 2 passes, 3 + 5 x 4 launches, GB/s) against references X and Y (2 GiB handles of different superclasses)
// and against the first handle of its own size.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/vmm_probe_sizes.cpp -o build/vmm_probe_sizes
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
static const size_t MIB = 1ull << 20, GIB = 1ull << 30;
typedef double v2d __attribute__((ext_vector_type(2)));
template <bool ZERO>
__global__ __launch_bounds__(256, 2) void wprobe_kernel(char* a, char* b, long pieces, int passes) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    for (int r = 0; r < passes; ++r)
        for (long p = wave; p < pieces; p += nw) {
            const v2d val = ZERO ? v2d{0.0, 0.0} : v2d{(double)p, (double)r};
#pragma unroll
            for (int c = 0; c < 4; ++c) __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(a + p * 4096 + c * 1024 + lane * 16));
#pragma unroll
            for (int c = 0; c < 4; ++c) __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(b + p * 4096 + c * 1024 + lane * 16));
        }
}
static hipStream_t s;
static hipEvent_t e0, e1;
template <bool ZERO = false>
static double gbps(char* a, char* b, int warm = 3, int reps = 5, int n = 4) {
    const size_t bytes = 128 * MIB;
    const int passes = 2;
    const long pieces = (long)(bytes / 4096);
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(wprobe_kernel<ZERO>, dim3(512), dim3(256), 0, s, a, b, pieces, passes);
    std::vector<double> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(wprobe_kernel<ZERO>, dim3(512), dim3(256), 0, s, a, b, pieces, passes);
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
    char* va;
    CK(hipMemAddressReserve((void**)&va, 2 * GIB * 200, 2 * MIB, nullptr, 0));
    size_t slot = 0;
    auto make = [&](size_t size, size_t spacer) {
        hipMemGenericAllocationHandle_t h, sp;
        CK(hipMemCreate(&h, size, &prop, 0));
        char* at;      // a reservation of its own (hipMemSetAccess refused a 128 MiB mapping inside a reservation that
        (void)slot;    // already held 2 GiB mappings: "invalid argument")
        CK(hipMemAddressReserve((void**)&at, size, 2 * MIB, nullptr, 0));
        CK(hipMemMap(at, size, 0, h, 0));
        CK(hipMemSetAccess(at, size, &acc, 1));
        if (spacer) CK(hipMemCreate(&sp, spacer, &prop, 0));
        return at;
    };
    // references
    std::vector<char*> big;
    for (int i = 0; i < 16; ++i) big.push_back(make(2 * GIB, 0));
    char* X = big[0];
    char* Y = nullptr;
    printf("# 2 GiB handles against handle 0 (GB/s; values / zeros written):");
    for (int i = 1; i < 16; ++i) {
        const double r = gbps(X, big[i]), z = gbps<true>(X, big[i]);
        printf(" %.0f/%.0f", r, z);
        if (!Y && r > 6500) Y = big[i];
    }
    printf("\n");
    if (!Y) { printf("# no second superclass among the references\n"); return 0; }
    const size_t sizes[] = {128 * MIB, 256 * MIB, 512 * MIB, GIB};
    const int N = 14;
    std::vector<std::vector<char*>> hs(4);
    for (int k = 0; k < N; ++k)                // interleaved along ONE walk: 128, 256, 512, 1024, 128, ...
        for (int q = 0; q < 4; ++q) hs[q].push_back(make(sizes[q], 2 * GIB - sizes[q]));
    for (int q = 0; q < 4; ++q) {
        printf("# handles of %4zu MiB: against X / against Y / against the first handle of this size (GB/s)   [zeros written: X / first]\n", sizes[q] >> 20);
        for (int k = 0; k < N; ++k) {
            char* c = hs[q][k];
            printf("  %2d: %5.0f / %5.0f / %5.0f     [%5.0f / %5.0f]   quick (1 + 3 x 1 launches) vs first: %5.0f\n", k, gbps(X, c), gbps(Y, c),
                   k ? gbps(hs[q][0], c) : 0.0, gbps<true>(X, c), k ? gbps<true>(hs[q][0], c) : 0.0, k ? gbps<true>(hs[q][0], c, 1, 3, 1) : 0.0);
        }
    }
    return 0;
}
