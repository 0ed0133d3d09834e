// vmm_factor_probe.cpp -- fourth placement experiment: WHAT makes "two outputs in one 2 GiB handle + two in another"
// fast (0.50 ms, tools/vmm_class_probe.cpp) when four individually classified 268 MiB handles of the same probe classes
// are slow (0.57 ms, tools/vmm_piece_probe.cpp)?  One factor at a time, face-mass x 4 at E = 1e6 (and grad where it applies):
//   handle size (2 GiB / 1 GiB / 536 MiB / 268 MiB), offsets inside the handles, order of the outputs, virtual
//   adjacency, re-mapping of a handle to another virtual address.
// Classes are taken with the two-stream write probe against the first 2 GiB handle: "same" (slow probe: > 0.85 of the
// slowest) or "other" superclass (fast probe); see profiles/r03/vmm_piece_probe.txt for the six-class picture.
//
//   vmm_factor_probe
//
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/vmm_factor_probe.cpp -Lfeinsum_amd -lfeinsum_hip
//        -Wl,-rpath,'$ORIGIN/../feinsum_amd' -o build/vmm_factor_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
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

static const size_t MIB = 1ull << 20, GIB = 1ull << 30;
typedef double v2d __attribute__((ext_vector_type(2)));

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
__global__ __launch_bounds__(256, 2) void wprobe_kernel(char* a, char* b, long pieces, int passes) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    for (int r = 0; r < passes; ++r)
        for (long p = wave; p < pieces; p += nw) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                __builtin_nontemporal_store(v2d{(double)p, (double)r}, reinterpret_cast<v2d*>(a + p * 4096 + c * 1024 + lane * 16));
#pragma unroll
            for (int c = 0; c < 4; ++c)
                __builtin_nontemporal_store(v2d{(double)p, (double)r}, reinterpret_cast<v2d*>(b + p * 4096 + c * 1024 + lane * 16));
        }
}

static hipStream_t s;
static hipEvent_t e0, e1;
static hipMemAllocationProp prop;
static hipMemAccessDesc acc;

static double time_batches(const std::function<void()>& launch, int warm, int reps, int n) {
    for (int i = 0; i < warm; ++i) launch();
    CK(hipStreamSynchronize(s));
    std::vector<double> ts;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) launch();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms / n);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}
static double wprobe_ms(char* a, char* b) {
    const long pieces = (long)(128 * MIB / 4096);
    return time_batches([&] { hipLaunchKernelGGL(wprobe_kernel, dim3(512), dim3(256), 0, s, a, b, pieces, 2); }, 3, 5, 4);
}

// n handles of `size` bytes, handle i mapped at va + i * stride (stride >= size), `spacer` bytes of unmapped physical
// memory created behind each one
struct Set {
    size_t size = 0, stride = 0;
    char* va = nullptr;
    std::vector<hipMemGenericAllocationHandle_t> h;
    std::vector<double> t;       // probe against the reference
    std::vector<int> other;      // 1: other superclass than the reference
    void create(int n, size_t bytes, size_t stride_, size_t spacer) {
        size = bytes;
        stride = stride_;
        CK(hipMemAddressReserve((void**)&va, stride * n, 2 * MIB, nullptr, 0));
        h.resize(n);
        for (int i = 0; i < n; ++i) {
            CK(hipMemCreate(&h[i], size, &prop, 0));
            CK(hipMemMap(va + stride * i, size, 0, h[i], 0));
            CK(hipMemSetAccess(va + stride * i, size, &acc, 1));
            if (spacer) {
                hipMemGenericAllocationHandle_t sp;
                CK(hipMemCreate(&sp, spacer, &prop, 0));   // held to the end of the process
            }
        }
    }
    char* at(int i, size_t off = 0) const { return va + stride * i + off; }
    void classify(char* ref, double t_same, double t_other, const char* what) {
        t.resize(h.size());
        other.resize(h.size());
        printf("# %s against the reference (x 0.1 ms; o = other superclass):", what);
        for (size_t i = 0; i < h.size(); ++i) {
            t[i] = wprobe_ms(ref, at((int)i) == ref ? ref + 128 * MIB : at((int)i));
            other[i] = t[i] < t_other + 0.3 * (t_same - t_other);
            printf(" %.3f%c", t[i] * 10, other[i] ? 'o' : ' ');
        }
        printf("\n");
    }
    // grow the set (one handle every 2 GiB of allocation order) until `need` handles of either kind are present
    void create_until(size_t bytes, size_t stride_, int need, int max_n, char* ref, double t_same, double t_other, const char* what) {
        size = bytes;
        stride = stride_;
        CK(hipMemAddressReserve((void**)&va, stride * max_n, 2 * MIB, nullptr, 0));
        int have[2] = {0, 0};
        printf("# %s against the reference (x 0.1 ms; o = other superclass):", what);
        while ((int)h.size() < max_n && (have[0] < need || have[1] < need)) {
            const int i = (int)h.size();
            hipMemGenericAllocationHandle_t hh, sp;
            CK(hipMemCreate(&hh, size, &prop, 0));
            CK(hipMemMap(va + stride * i, size, 0, hh, 0));
            CK(hipMemSetAccess(va + stride * i, size, &acc, 1));
            CK(hipMemCreate(&sp, 2 * GIB - size, &prop, 0));
            h.push_back(hh);
            t.push_back(wprobe_ms(ref, at(i)));
            other.push_back(t[i] < t_other + 0.3 * (t_same - t_other));
            ++have[other[i]];
            printf(" %.3f%c", t[i] * 10, other[i] ? 'o' : ' ');
        }
        printf("\n");
    }
    int find(int want_other, int skip = 0) const {
        for (size_t i = 0; i < h.size(); ++i)
            if (other[i] == want_other && skip-- == 0) return (int)i;
        return -1;
    }
};

int main() {
    const int64_t E = 1000000;
    const int Np = 35, Nfp = 15, nf = 4, nb = 4;
    CK(hipSetDevice(0));
    prop = hipMemAllocationProp{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc = hipMemAccessDesc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipStreamCreate(&s));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double* gJ = dev_random(9 * E, 1);
    double* gD = dev_random(3 * Np * Np, 2);
    double* gu = dev_random(E * Np, 3);
    double* fJ = dev_random(E * nf, 5);
    double* fR = dev_random((size_t)nf * Np * Nfp, 6);
    const double* fv[4];
    for (int k = 0; k < nb; ++k) fv[k] = dev_random((size_t)nf * E * Nfp, 10 + k);
    const size_t plane = (size_t)E * Np * 8;
    const size_t W = (plane + 2 * MIB - 1) / (2 * MIB) * (2 * MIB);   // 268 MiB
    auto fm = [&](const char* what, char* o0, char* o1, char* o2, char* o3) {
        double* oo[4] = {(double*)o0, (double*)o1, (double*)o2, (double*)o3};
        const double t = time_batches([&] { FE(fe_facemass_f64(fJ, fR, fv, oo, E, Np, nf, Nfp, nb, 0, 0, s)); }, 10, 5, 10);
        printf("face-mass x4  %-92s %.4f ms (%.1f %%)\n", what, t, 3072.0168e6 / (t * 1e-3) / 8e12 * 100);
        fflush(stdout);
    };
    auto grad = [&](const char* what, char* out) {
        const double t = time_batches([&] { FE(fe_grad3d_f64(gJ, gD, gu, (double*)out, E, Np, 0, s)); }, 30, 5, 20);
        printf("grad          %-92s %.4f ms (%.1f %%)\n", what, t, 1192.0294e6 / (t * 1e-3) / 8e12 * 100);
        fflush(stdout);
    };

    // ---- 2 GiB handles, as tools/vmm_class_probe.cpp
    Set B;
    B.create(16, 2 * GIB, 2 * GIB, 0);
    char* ref = B.at(0);
    double t_same = wprobe_ms(ref, B.at(0, 128 * MIB)), t_other = t_same;
    for (int i = 1; i < 16; ++i) t_other = std::min(t_other, wprobe_ms(ref, B.at(i)));
    printf("# write probe 2 x 128 MiB: same region %.4f ms, fastest pair %.4f ms\n", t_same, t_other);
    if (t_same / t_other < 1.1) { printf("# no second class among 16 handles of 2 GiB\n"); return 0; }
    B.classify(ref, t_same, t_other, "2 GiB handles (first 128 MiB)");
    {   // the second GiB of every handle (address bit 30?)
        printf("# second GiB of the 2 GiB handles against the reference:");
        for (int i = 0; i < 16; ++i) printf(" %.3f", wprobe_ms(ref, B.at(i, GIB)) * 10);
        printf("\n");
    }
    const int Y = B.find(1), X2 = B.find(0, 1);   // (handle 0 itself is 'same' by construction)
    if (Y < 0) { printf("# no handle of another superclass\n"); return 0; }
    printf("# X = handle 0, Y = handle %d (other superclass), X2 = handle %d (same superclass as X)\n", Y, X2);
    for (int rep = 0; rep < 2; ++rep) {
        fm("2 GiB handles: X+0, X+W, Y+2W, Y+3W  [the known fast case]", B.at(0), B.at(0, W), B.at(Y, 2 * W), B.at(Y, 3 * W));
        fm("2 GiB handles: X+0, X+W, Y+0, Y+W", B.at(0), B.at(0, W), B.at(Y), B.at(Y, W));
        fm("2 GiB handles: X+0, Y+0, X+W, Y+W  (order)", B.at(0), B.at(Y), B.at(0, W), B.at(Y, W));
        fm("2 GiB handles: X+0, X+W, X+2W, X+3W  (one handle)", B.at(0), B.at(0, W), B.at(0, 2 * W), B.at(0, 3 * W));
        fm("2 GiB handles: X+0, X+W, X+1GiB, X+1GiB+W  (one handle, both GiB halves)", B.at(0), B.at(0, W), B.at(0, GIB), B.at(0, GIB + W));
        if (X2 >= 0) fm("2 GiB handles: X+0, X+W, X2+0, X2+W  (two handles of one superclass)", B.at(0), B.at(0, W), B.at(X2), B.at(X2, W));
        fm("2 GiB handles: X+0, X+W, Y+1GiB, Y+1GiB+W", B.at(0), B.at(0, W), B.at(Y, GIB), B.at(Y, GIB + W));
        grad("2 GiB handle X, offset 0 (one handle)", B.at(0));
        grad("2 GiB handle X, offset 600 MiB (planes cross the 1 GiB line)", B.at(0, 600 * MIB));
    }

    // ---- smaller handles, each at a virtual address of its own (1 GiB apart), never re-mapped
    Set H1, H2, H4;
    H1.create_until(GIB, GIB, 2, 30, ref, t_same, t_other, "1 GiB handles");
    H2.create_until(2 * W, GIB, 2, 30, ref, t_same, t_other, "536 MiB handles");
    H4.create_until(W, GIB, 2, 30, ref, t_same, t_other, "268 MiB handles");
    for (int rep = 0; rep < 2; ++rep) {
        {
            const int p = H1.find(0), q = H1.find(1), p2 = H1.find(0, 1), q2 = H1.find(1, 1);
            if (p >= 0 && q >= 0) fm("1 GiB handles: P+0, P+W, Q+0, Q+W  (P same superclass as X, Q other)", H1.at(p), H1.at(p, W), H1.at(q), H1.at(q, W));
            if (p >= 0 && p2 >= 0) fm("1 GiB handles: P+0, P+W, P2+0, P2+W  (both of X's superclass)", H1.at(p), H1.at(p, W), H1.at(p2), H1.at(p2, W));
            if (q >= 0) fm("mixed: X+0, X+W (2 GiB handle), Q+0, Q+W (1 GiB handle of the other superclass)", B.at(0), B.at(0, W), H1.at(q), H1.at(q, W));
            if (p >= 0 && p2 >= 0 && q >= 0 && q2 >= 0) {
                fm("1 GiB handles, one output each, all at offset 0: P, P2, Q, Q2", H1.at(p), H1.at(p2), H1.at(q), H1.at(q2));
                fm("1 GiB handles, one output each, offsets 0, W, 2W, 3W - 1 GiB wrap: P, P2+W, Q+2W, Q2+512MiB", H1.at(p), H1.at(p2, W), H1.at(q, 2 * W),
                   H1.at(q2, 512 * MIB));
                fm("1 GiB handles, one output each, offsets 0, 33 MiB, 66 MiB, 99 MiB: P, P2, Q, Q2", H1.at(p), H1.at(p2, 33 * MIB), H1.at(q, 66 * MIB),
                   H1.at(q2, 99 * MIB));
                fm("1 GiB handles, one output each, offsets 0, 1 MiB, 2 MiB, 3 MiB: P, P2, Q, Q2", H1.at(p), H1.at(p2, 1 * MIB), H1.at(q, 2 * MIB),
                   H1.at(q2, 3 * MIB));
            }
        }
        {
            const int p = H2.find(0), q = H2.find(1), p2 = H2.find(0, 1);
            if (p >= 0 && q >= 0) fm("536 MiB handles: P+0, P+W, Q+0, Q+W", H2.at(p), H2.at(p, W), H2.at(q), H2.at(q, W));
            if (p >= 0 && p2 >= 0) fm("536 MiB handles: P+0, P+W, P2+0, P2+W  (one superclass)", H2.at(p), H2.at(p, W), H2.at(p2), H2.at(p2, W));
            if (q >= 0) fm("mixed: X+0, X+W (2 GiB handle), Q+0, Q+W (536 MiB handle of the other superclass)", B.at(0), B.at(0, W), H2.at(q), H2.at(q, W));
        }
        {
            const int p = H4.find(0), p2 = H4.find(0, 1), q = H4.find(1), q2 = H4.find(1, 1);
            if (p >= 0 && p2 >= 0 && q >= 0 && q2 >= 0) {
                fm("268 MiB handles at their own addresses: P, P2, Q, Q2", H4.at(p), H4.at(p2), H4.at(q), H4.at(q2));
                fm("268 MiB handles at their own addresses: P, Q, P2, Q2", H4.at(p), H4.at(q), H4.at(p2), H4.at(q2));
            }
            if (q >= 0 && q2 >= 0) fm("mixed: X+0, X+W (2 GiB handle), Q, Q2 (268 MiB handles of the other superclass)", B.at(0), B.at(0, W), H4.at(q), H4.at(q2));
            if (q >= 0 && q2 >= 0) fm("mixed: X+0, X+33MiB+W (2 GiB handle), Q, Q2 (268 MiB handles of the other superclass)", B.at(0), B.at(0, W + 33 * MIB), H4.at(q), H4.at(q2));
            if (p >= 0 && p2 >= 0) fm("mixed: Y+0, Y+W (2 GiB handle), P, P2 (268 MiB handles of X's superclass)", B.at(Y), B.at(Y, W), H4.at(p), H4.at(p2));
        }
    }
    // ---- the same four 268 MiB handles re-mapped side by side (as tools/vmm_piece_probe.cpp did)
    {
        const int p = H4.find(0), p2 = H4.find(0, 1), q = H4.find(1), q2 = H4.find(1, 1);
        if (p >= 0 && p2 >= 0 && q >= 0 && q2 >= 0) {
            char* ov;
            CK(hipMemAddressReserve((void**)&ov, 4 * W, 2 * MIB, nullptr, 0));
            CK(hipDeviceSynchronize());
            const int idx[4] = {p, p2, q, q2};
            for (int k = 0; k < 4; ++k) {
                CK(hipMemUnmap(H4.at(idx[k]), W));
                CK(hipMemMap(ov + W * k, W, 0, H4.h[idx[k]], 0));
            }
            CK(hipMemSetAccess(ov, 4 * W, &acc, 1));
            for (int rep = 0; rep < 2; ++rep) {
                fm("268 MiB handles re-mapped side by side: P, P2, Q, Q2", ov, ov + W, ov + 2 * W, ov + 3 * W);
                grad("268 MiB handles re-mapped side by side: planes P, P2, Q", ov);
                grad("268 MiB handles re-mapped side by side: planes P2, Q, Q2", ov + W);
            }
        }
    }
    return 0;
}
