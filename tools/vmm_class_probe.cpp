// vmm_class_probe.cpp -- which pieces of physical memory are of the same "class" for concurrent write streams?
//
// N physical handles of S GiB each are created one after the other (hipMemCreate) and mapped; face-mass x 4 is then timed
// with outputs 0 and 1 in handle 0 and outputs 2 and 3 in handle i, for every i (i = 0: all four in handle 0).  A pair of
// handles whose split launch is as slow as the unsplit one is of the same class (tools/split_probe.py: it is the split of the
// written arrays between two classes of physical memory that makes a launch fast).  With handles created back to back the
// index is (presumably) a physical-address axis.
//
//   vmm_class_probe [N=48] [S GiB=2] [ref=0]
//
// Build: hipcc -O2 -std=c++17 tools/vmm_class_probe.cpp -Lfeinsum_amd -lfeinsum_hip -Wl,-rpath,'$ORIGIN/../feinsum_amd' -o build/vmm_class_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/feinsum_hip.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(2);                                                               \
        }                                                                          \
    } while (0)
#define FE(x)                                                                      \
    do {                                                                           \
        int r_ = (x);                                                              \
        if (r_ != 0) {                                                             \
            fprintf(stderr, "%s -> %d: %s\n", #x, r_, fe_last_error());            \
            exit(3);                                                               \
        }                                                                          \
    } while (0)

__global__ void fill_kernel(double* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (x & 0xFFFFFF) * (1.0 / 16777216.0);
    }
}
static double* dev_random(size_t n, unsigned seed) {
    double* d;
    CK(hipMalloc(&d, n * 8));
    fill_kernel<<<2048, 256>>>(d, n, seed);
    CK(hipDeviceSynchronize());
    return d;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 48;
    const size_t S = (size_t)(argc > 2 ? atoi(argv[2]) : 2) << 30;
    const int ref = argc > 3 ? atoi(argv[3]) : 0;
    const int64_t E = 1000000;
    const int Np = 35, Nfp = 15, nf = 4, nb = 4;
    CK(hipSetDevice(0));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double* J = dev_random(E * nf, 1);
    double* R = dev_random((size_t)nf * Np * Nfp, 2);
    const double* vv[4];
    for (int k = 0; k < nb; ++k) vv[k] = dev_random((size_t)nf * E * Nfp, 10 + k);

    char* va;
    CK(hipMemAddressReserve((void**)&va, S * N, 2 << 20, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(N);
    for (int i = 0; i < N; ++i) {
        CK(hipMemCreate(&h[i], S, &prop, 0));
        CK(hipMemMap(va + S * i, S, 0, h[i], 0));
    }
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, S * N, &acc, 1));
    const size_t W = ((size_t)E * Np * 8 + (2 << 20) - 1) / (2 << 20) * (2 << 20);     // one output, 2 MiB rounded
    if (4 * W > S) { fprintf(stderr, "handles too small\n"); return 1; }
    printf("# %d handles of %zu GiB, reference handle %d; one output = %zu MiB\n", N, S >> 30, ref, W >> 20);

    auto time_it = [&](double* const* oo) {
        for (int i = 0; i < 10; ++i) FE(fe_facemass_f64(J, R, vv, oo, E, Np, nf, Nfp, nb, 0, 0, s));
        CK(hipStreamSynchronize(s));
        std::vector<double> ts;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 10; ++i) FE(fe_facemass_f64(J, R, vv, oo, E, Np, nf, Nfp, nb, 0, 0, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms / 10);
        }
        std::sort(ts.begin(), ts.end());
        return ts[1];
    };
    auto at = [&](int handle, int slot) { return reinterpret_cast<double*>(va + S * handle + W * slot); };
    for (int i = 0; i < N; ++i) {
        double* both[4] = {at(ref, 0), at(ref, 1), at(i, 2), at(i, 3)};
        double* own[4] = {at(i, 0), at(i, 1), at(i, 2), at(i, 3)};
        printf("handle %3d (+%3zu GiB): 2 in handle %d + 2 here %.4f ms    all four here %.4f ms\n", i, (S * i) >> 30, ref,
               time_it(both), time_it(own));
        fflush(stdout);
    }
    return 0;
}
