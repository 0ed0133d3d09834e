// vmm_piece_probe.cpp -- classes of physical memory, third experiment (after tools/vmm_interleave_probe.cpp found that
// arrays composed of 2 MiB handles are slow whatever "class" a group probe gave the pieces).
//
// Here every piece is a handle large enough to be classified ON ITS OWN by the two-stream write probe:
// Phase 1  K chunks of 268 MiB (one output plane, rounded to 2 MiB) are created with a 1.75 GiB spacer behind each (so the
//          chunks sample the allocator's memory every 2 GiB).  Each is probed against chunk 0, then against the first chunk
//          that differs from chunk 0, then against the first that differs from both: HOW MANY classes are there?
// Phase 2  the DG launches with whole chunks as output planes / output arrays in all class combinations.
// Phase 3  half chunks (134 MiB handles): div cut in the middle, face-mass outputs cut in the middle (same / alternating order).
// Phase 4  32 MiB handles classified one by one; grad / face-mass outputs alternating every 32 MiB.
//
//   vmm_piece_probe [K=40]
//
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/vmm_piece_probe.cpp -Lfeinsum_amd -lfeinsum_hip
//        -Wl,-rpath,'$ORIGIN/../feinsum_amd' -o build/vmm_piece_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
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
static double wprobe_ms(char* a, char* b, size_t bytes_per_stream, int passes) {
    const long pieces = (long)(bytes_per_stream / 4096);
    return time_batches([&] { hipLaunchKernelGGL(wprobe_kernel, dim3(512), dim3(256), 0, s, a, b, pieces, passes); }, 3, 5, 4);
}

// A set of handles of one size, each mapped at its own place in one reserved range.
struct Pieces {
    size_t size = 0;
    char* va = nullptr;
    std::vector<hipMemGenericAllocationHandle_t> h;
    std::vector<int> cls;
    void create(int n, size_t bytes, size_t spacer, std::vector<hipMemGenericAllocationHandle_t>& spacers) {
        size = bytes;
        CK(hipMemAddressReserve((void**)&va, size * n, 2 * MIB, nullptr, 0));
        h.resize(n);
        for (int i = 0; i < n; ++i) {
            CK(hipMemCreate(&h[i], size, &prop, 0));
            CK(hipMemMap(va + size * i, size, 0, h[i], 0));
            if (spacer) {
                hipMemGenericAllocationHandle_t sp;
                CK(hipMemCreate(&sp, spacer, &prop, 0));
                spacers.push_back(sp);
            }
        }
        CK(hipMemSetAccess(va, size * n, &acc, 1));
        cls.assign(n, -1);
    }
    char* at(int i) const { return va + size * i; }
    void unmap_all() {
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(va, size * h.size()));
    }
};

// Classify pieces: class 0 = same as piece 0; class 1 = same as the first piece that differs from class 0; ...
// `probe(i, j)` -> true when pieces i and j are of the SAME class (the probe is slow).
static int classify(int n, std::vector<int>& cls, const std::function<double(int, int)>& probe_ms, double& t_same, double& t_diff,
                    bool verbose) {
    std::vector<std::vector<double>> rows;
    std::vector<int> refs;
    cls.assign(n, -1);
    // calibrate: piece 0 against every piece
    std::vector<double> t0(n);
    for (int i = 0; i < n; ++i) t0[i] = probe_ms(0, i);
    t_same = *std::max_element(t0.begin(), t0.end());
    t_diff = *std::min_element(t0.begin(), t0.end());
    const double mid = 0.5 * (t_same + t_diff);
    if (t_same / t_diff < 1.08) {   // everything looks alike: one class
        cls.assign(n, 0);
        return 1;
    }
    int ncls = 0;
    while (true) {
        int ref = -1;
        for (int i = 0; i < n; ++i)
            if (cls[i] < 0) { ref = i; break; }
        if (ref < 0 || ncls >= 6) break;
        std::vector<double> t(n);
        for (int i = 0; i < n; ++i) t[i] = (ref == 0) ? t0[i] : probe_ms(ref, i);
        for (int i = 0; i < n; ++i)
            if (cls[i] < 0 && t[i] > mid) cls[i] = ncls;
        cls[ref] = ncls;
        if (verbose) {
            printf("  against piece %2d (class %d):", ref, ncls);
            for (int i = 0; i < n; ++i) printf(" %.3f%c", t[i] * 10, t[i] > mid ? '*' : ' ');
            printf("   (x 0.1 ms, * = same class)\n");
        }
        ++ncls;
    }
    return ncls;
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 40;
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
    double* dv = dev_random(3 * E * Np, 4);
    double* fJ = dev_random(E * nf, 5);
    double* fR = dev_random((size_t)nf * Np * Nfp, 6);
    const double* fv[4];
    for (int k = 0; k < nb; ++k) fv[k] = dev_random((size_t)nf * E * Nfp, 10 + k);
    const size_t plane = (size_t)E * Np * 8;
    const size_t W = (plane + 2 * MIB - 1) / (2 * MIB) * (2 * MIB);   // 268 MiB

    std::vector<hipMemGenericAllocationHandle_t> spacers;
    auto pct = [](double bytes, double ms) { return bytes / (ms * 1e-3) / 8e12 * 100; };
    const double GB = 1192.0294e6, FB = 3072.0168e6;

    // ---- phase 1: chunks of one plane, classified pairwise
    Pieces C;
    C.create(K, W, 2 * GIB - W, spacers);
    double ts, td;
    const int ncls = classify(K, C.cls, [&](int i, int j) { return wprobe_ms(C.at(i), C.at(j) + (i == j ? 128 * MIB : 0), 128 * MIB, 2); }, ts, td, true);
    printf("# phase 1: %d chunks of %zu MiB, one every 2 GiB of allocation order: %d class(es); probe same %.4f / other %.4f ms\n# classes: ", K,
           W >> 20, ncls, ts, td);
    for (int i = 0; i < K; ++i) printf("%c", 'A' + C.cls[i]);
    printf("\n");
    fflush(stdout);
    std::vector<std::vector<int>> by(ncls);
    for (int i = 0; i < K; ++i) by[C.cls[i]].push_back(i);
    if (ncls < 2 || by[0].size() < 1) { printf("# one class only\n"); return 0; }

    // ---- phase 2: whole chunks as planes / arrays.  A composition names the class of each plane; chunks are taken in
    //      order from the class lists and mapped contiguously into `ov`.
    char* ov;
    CK(hipMemAddressReserve((void**)&ov, 4 * W, 2 * MIB, nullptr, 0));
    C.unmap_all();
    auto pick = [&](const std::string& pat, std::vector<int>& out) {
        std::vector<size_t> used(ncls, 0);
        out.clear();
        for (char ch : pat) {
            const int c = ch - 'A';
            if (c >= ncls || used[c] >= by[c].size()) return false;
            out.push_back(by[c][used[c]++]);
        }
        return true;
    };
    auto with_chunks = [&](const std::string& pat, const std::function<void()>& body) {
        std::vector<int> idx;
        if (!pick(pat, idx)) return false;
        for (size_t q = 0; q < idx.size(); ++q) CK(hipMemMap(ov + W * q, W, 0, C.h[idx[q]], 0));
        CK(hipMemSetAccess(ov, W * idx.size(), &acc, 1));
        body();
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(ov, W * idx.size()));
        return true;
    };
    auto grad_ms = [&] { return time_batches([&] { FE(fe_grad3d_f64(gJ, gD, gu, (double*)ov, E, Np, 0, s)); }, 30, 5, 20); };
    auto div_ms = [&](int variant) { return time_batches([&] { FE(fe_div3d_f64(gJ, gD, dv, (double*)ov, E, Np, variant, s)); }, 30, 5, 20); };
    auto fm_ms = [&](size_t stride) {
        double* oo[4] = {(double*)ov, (double*)(ov + stride), (double*)(ov + 2 * stride), (double*)(ov + 3 * stride)};
        return time_batches([&] { FE(fe_facemass_f64(fJ, fR, fv, oo, E, Np, nf, Nfp, nb, 0, 0, s)); }, 10, 5, 10);
    };
    printf("# phase 2: whole chunks (268 MiB handles) as output planes / arrays; class of each plane\n");
    for (int rep = 0; rep < 2; ++rep) {
        for (const char* pat : {"AAA", "BBB", "CCC", "ABA", "AAB", "ABB", "BAB", "ABC", "ACB", "BCC", "ACC"})
            with_chunks(pat, [&] { const double t = grad_ms(); printf("grad planes %-4s  %.4f ms (%.1f %%)\n", pat, t, pct(GB, t)); });
        for (const char* pat : {"AAAA", "BBBB", "CCCC", "AABB", "ABAB", "ABBA", "AAAB", "ABBB", "AABC", "ABCA", "ABCB", "BBCC", "BCBC"})
            with_chunks(pat, [&] { const double t = fm_ms(W); printf("face-mass outputs %-4s  %.4f ms (%.1f %%)\n", pat, t, pct(FB, t)); });
        for (const char* pat : {"A", "B", "C"})
            with_chunks(pat, [&] {
                const double t = div_ms(FE_VARIANT_MFMA), t2 = div_ms(FE_VARIANT_MFMA_SPLIT);
                printf("div output %-4s  %.4f ms (%.1f %%)   two-window walk %.4f ms (%.1f %%)\n", pat, t, pct(GB, t), t2, pct(GB, t2));
            });
        fflush(stdout);
    }

    // ---- phase 3: half chunks
    const size_t H = W / 2;   // 134 MiB
    Pieces Hh;
    Hh.create(32, H, 1 * GIB - H, spacers);
    {
        // class of each half chunk relative to the chunk classes: map one chunk of each class back as references
        std::vector<int> ref_idx;
        char* rv;
        CK(hipMemAddressReserve((void**)&rv, W * ncls, 2 * MIB, nullptr, 0));
        for (int c = 0; c < ncls; ++c) CK(hipMemMap(rv + W * c, W, 0, C.h[by[c][0]], 0));
        CK(hipMemSetAccess(rv, W * ncls, &acc, 1));
        printf("# phase 3: 32 half chunks of %zu MiB; probe against one chunk of every class (x 0.1 ms):\n", H >> 20);
        for (int i = 0; i < 32; ++i) {
            double best = 0;
            int bc = -1;
            printf("  half %2d:", i);
            for (int c = 0; c < ncls; ++c) {
                const double t = wprobe_ms(rv + W * c, Hh.at(i), 128 * MIB, 2);
                printf(" %.3f", t * 10);
                if (t > best) { best = t; bc = c; }
            }
            Hh.cls[i] = best > 0.5 * (ts + td) ? bc : -1;
            printf("  -> %c\n", Hh.cls[i] < 0 ? '?' : 'A' + Hh.cls[i]);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(rv, W * ncls));
    }
    Hh.unmap_all();
    std::vector<std::vector<int>> hby(ncls);
    for (int i = 0; i < 32; ++i)
        if (Hh.cls[i] >= 0) hby[Hh.cls[i]].push_back(i);
    auto with_halves = [&](const std::string& pat, const std::function<void()>& body) {
        std::vector<size_t> used(ncls, 0);
        std::vector<int> idx;
        for (char ch : pat) {
            const int c = ch - 'A';
            if (c >= ncls || used[c] >= hby[c].size()) return false;
            idx.push_back(hby[c][used[c]++]);
        }
        for (size_t q = 0; q < idx.size(); ++q) CK(hipMemMap(ov + H * q, H, 0, Hh.h[idx[q]], 0));
        CK(hipMemSetAccess(ov, H * idx.size(), &acc, 1));
        body();
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(ov, H * idx.size()));
        return true;
    };
    for (int rep = 0; rep < 2; ++rep) {
        for (const char* pat : {"AA", "AB", "BA", "BB", "BC"})
            with_halves(pat, [&] {
                const double t = div_ms(FE_VARIANT_MFMA), t2 = div_ms(FE_VARIANT_MFMA_SPLIT);
                printf("div output halves %-3s  %.4f ms (%.1f %%)   two-window walk %.4f ms (%.1f %%)\n", pat, t, pct(GB, t), t2, pct(GB, t2));
            });
        for (const char* pat : {"AAAAAA", "ABABAB", "AABBAB", "ABBAAB", "AAABBB", "ABCABC", "ABBCCA"})
            with_halves(pat, [&] { const double t = grad_ms(); printf("grad halves %-7s  %.4f ms (%.1f %%)\n", pat, t, pct(GB, t)); });
        for (const char* pat : {"AAAAAAAA", "ABABABAB", "ABBAABBA", "AABBAABB", "AAAABBBB", "ABBCCAAB"})
            with_halves(pat, [&] { const double t = fm_ms(W); printf("face-mass halves %-9s  %.4f ms (%.1f %%)\n", pat, t, pct(FB, t)); });
        fflush(stdout);
    }

    // ---- phase 4: 32 MiB handles, classified one by one
    const size_t Q = 32 * MIB;
    Pieces Sm;
    Sm.create(160, Q, 224 * MIB, spacers);       // one every 256 MiB of allocation order: 40 GiB walked
    {
        char* rv;
        CK(hipMemAddressReserve((void**)&rv, W * ncls, 2 * MIB, nullptr, 0));
        for (int c = 0; c < ncls; ++c) CK(hipMemMap(rv + W * c, W, 0, C.h[by[c][0]], 0));
        CK(hipMemSetAccess(rv, W * ncls, &acc, 1));
        // calibration of the 32 MiB probe on chunks of known class
        double cal_same = 0, cal_diff = 1e9;
        for (int c = 0; c < ncls; ++c) {
            char* other = rv + W * c + 128 * MIB;
            cal_same = std::max(cal_same, wprobe_ms(rv + W * c, other, Q, 8));
            if (c > 0) cal_diff = std::min(cal_diff, wprobe_ms(rv, rv + W * c, Q, 8));
        }
        printf("# phase 4: 160 handles of 32 MiB; 32 MiB probe calibration: same class %.4f, other %.4f ms\n#", cal_same, cal_diff);
        const double mid = 0.5 * (cal_same + cal_diff);
        for (int i = 0; i < 160; ++i) {
            int bc = -1;
            double best = 0;
            for (int c = 0; c < ncls; ++c) {
                const double t = wprobe_ms(rv + W * c, Sm.at(i), Q, 8);
                if (t > best) { best = t; bc = c; }
            }
            Sm.cls[i] = best > mid ? bc : -1;
            printf("%c", Sm.cls[i] < 0 ? '?' : 'A' + Sm.cls[i]);
        }
        printf("\n");
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(rv, W * ncls));
    }
    Sm.unmap_all();
    std::vector<std::vector<int>> sby(ncls);
    for (int i = 0; i < 160; ++i)
        if (Sm.cls[i] >= 0) sby[Sm.cls[i]].push_back(i);
    auto with_small = [&](int npieces, const std::function<int(int)>& cls_of, const std::function<void()>& body) {
        std::vector<size_t> used(ncls, 0);
        std::vector<int> idx;
        for (int q = 0; q < npieces; ++q) {
            const int c = cls_of(q);
            if (c >= ncls || used[c] >= sby[c].size()) return false;
            idx.push_back(sby[c][used[c]++]);
        }
        for (int q = 0; q < npieces; ++q) CK(hipMemMap(ov + Q * q, Q, 0, Sm.h[idx[q]], 0));
        CK(hipMemSetAccess(ov, Q * npieces, &acc, 1));
        body();
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(ov, Q * npieces));
        return true;
    };
    const int gp = (int)((3 * plane + Q - 1) / Q), dp = (int)((plane + Q - 1) / Q);
    for (int rep = 0; rep < 2; ++rep) {
        if (!with_small(gp, [](int) { return 0; }, [&] { const double t = grad_ms(); printf("grad 32 MiB pieces all A           %.4f ms (%.1f %%)\n", t, pct(GB, t)); }))
            printf("grad all A: not enough A pieces\n");
        with_small(gp, [](int) { return 1; }, [&] { const double t = grad_ms(); printf("grad 32 MiB pieces all B           %.4f ms (%.1f %%)\n", t, pct(GB, t)); });
        with_small(gp, [](int q) { return q & 1; }, [&] { const double t = grad_ms(); printf("grad 32 MiB pieces alternating     %.4f ms (%.1f %%)\n", t, pct(GB, t)); });
        with_small(gp, [&](int q) { return (q + (int)((size_t)q * Q / plane)) & 1; },
                   [&] { const double t = grad_ms(); printf("grad 32 MiB pieces alt., planes shifted %.4f ms (%.1f %%)\n", t, pct(GB, t)); });
        with_small(gp, [&](int q) { return q < gp / 2 ? 0 : 1; }, [&] { const double t = grad_ms(); printf("grad 32 MiB pieces one cut         %.4f ms (%.1f %%)\n", t, pct(GB, t)); });
        with_small(dp, [](int q) { return q & 1; }, [&] {
            const double t = div_ms(FE_VARIANT_MFMA), t2 = div_ms(FE_VARIANT_MFMA_SPLIT);
            printf("div 32 MiB pieces alternating      %.4f ms (%.1f %%)   two-window walk %.4f ms (%.1f %%)\n", t, pct(GB, t), t2, pct(GB, t2));
        });
        with_small(dp, [&](int q) { return q < dp / 2 ? 0 : 1; }, [&] {
            const double t = div_ms(FE_VARIANT_MFMA), t2 = div_ms(FE_VARIANT_MFMA_SPLIT);
            printf("div 32 MiB pieces one cut          %.4f ms (%.1f %%)   two-window walk %.4f ms (%.1f %%)\n", t, pct(GB, t), t2, pct(GB, t2));
        });
        fflush(stdout);
    }
    return 0;
}
