// split_alloc_check.cpp -- the split allocator (fe_split_alloc) from plain C++: grad / div / face-mass x 4 / div + grad at
// E = 1e6 with their OUTPUTS in allocator arrays against one hipMalloc per array, in one process: kernel times, bitwise equal
// results, what the allocator did (classes per piece, milliseconds, pool statistics), and allocate / free cycles.
//
//   split_alloc_check [E=1000000] [cycles=3]
//
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/split_alloc_check.cpp -Lfeinsum_amd -lfeinsum_hip
//        -Wl,-rpath,'$ORIGIN/../feinsum_amd' -o build/split_alloc_check
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
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
        if (r_ < 0) {                                                              \
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
__global__ void diff_kernel(const unsigned long long* a, const unsigned long long* b, size_t n, unsigned long long* count) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(count, c);
}
static double* dev_random(size_t n, unsigned seed) {
    double* d;
    CK(hipMalloc(&d, n * 8));
    fill_kernel<<<2048, 256>>>(d, n, seed);
    CK(hipDeviceSynchronize());
    return d;
}
static hipStream_t s;
static hipEvent_t e0, e1;
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
static unsigned long long* d_count;
static unsigned long long differing(const double* a, const double* b, size_t n) {
    CK(hipMemset(d_count, 0, 8));
    diff_kernel<<<2048, 256>>>((const unsigned long long*)a, (const unsigned long long*)b, n, d_count);
    unsigned long long c;
    CK(hipMemcpy(&c, d_count, 8, hipMemcpyDeviceToHost));
    return c;
}
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void show_info(const char* what, const void* p) {
    char buf[4096];
    FE(fe_split_info(p, buf, sizeof buf));
    printf("  %-14s %s\n", what, buf);
}

int main(int argc, char** argv) {
    const int64_t E = argc > 1 ? atoll(argv[1]) : 1000000;
    const int cycles = argc > 2 ? atoi(argv[2]) : 3;
    const int Np = 35, Nfp = 15, nf = 4, nb = 4;
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&s));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMalloc(&d_count, 8));
    double* gJ = dev_random(9 * E, 1);
    double* gD = dev_random(3 * Np * Np, 2);
    double* gu = dev_random(E * Np, 3);
    double* dv = dev_random(3 * E * Np, 4);
    double* fJ = dev_random(E * nf, 5);
    double* fR = dev_random((size_t)nf * Np * Nfp, 6);
    const double* fv[4];
    for (int k = 0; k < nb; ++k) fv[k] = dev_random((size_t)nf * E * Nfp, 10 + k);
    const size_t plane = (size_t)E * Np * 8;
    const double GB = 8.0 * (149.0 * E + 3675.0), FB = 8.0 * ((4 + 240 + 140) * (double)E + 2100.0), GDB = 8.0 * (289.0 * E + 3675.0);
    auto pct = [](double bytes, double ms) { return bytes / (ms * 1e-3) / 8e12 * 100; };

    // one hipMalloc per array
    double *mg, *md, *mf[4];
    CK(hipMalloc(&mg, 3 * plane));
    CK(hipMalloc(&md, plane));
    for (auto& p : mf) CK(hipMalloc(&p, plane));

    for (int cyc = 0; cyc < cycles; ++cyc) {
        printf("== cycle %d\n", cyc);
        double t0 = now_ms();
        void *sg, *sd, *sf[4];
        FE(fe_split_alloc(&sg, 3 * plane, 0));
        const double t_g = now_ms() - t0;
        t0 = now_ms();
        FE(fe_split_alloc(&sd, plane, 0));
        for (auto& p : sf) FE(fe_split_alloc(&p, plane, 0));
        const double t_rest = now_ms() - t0;
        printf("fe_split_alloc: grad output (%.0f MB) %.1f ms, div + 4 face-mass outputs (5 x %.0f MB) %.1f ms\n", 3 * plane / 1e6, t_g,
               plane / 1e6, t_rest);
        show_info("grad out", sg);
        show_info("div out", sd);
        for (int k = 0; k < 4; ++k) show_info("face-mass out", sf[k]);
        char buf[4096];
        FE(fe_split_stats(buf, sizeof buf));
        printf("  pool: %s\n", buf);

        double* fo[4] = {(double*)sf[0], (double*)sf[1], (double*)sf[2], (double*)sf[3]};
        const double g_m = time_batches([&] { FE(fe_grad3d_f64(gJ, gD, gu, mg, E, Np, 0, s)); }, 30, 5, 20);
        const double g_s = time_batches([&] { FE(fe_grad3d_f64(gJ, gD, gu, (double*)sg, E, Np, 0, s)); }, 30, 5, 20);
        const double d_m = time_batches([&] { FE(fe_div3d_f64(gJ, gD, dv, md, E, Np, FE_VARIANT_MFMA, s)); }, 30, 5, 20);
        const double d_s = time_batches([&] { FE(fe_div3d_f64(gJ, gD, dv, (double*)sd, E, Np, FE_VARIANT_MFMA, s)); }, 30, 5, 20);
        const double d_m2 = time_batches([&] { FE(fe_div3d_f64(gJ, gD, dv, md, E, Np, FE_VARIANT_MFMA_SPLIT, s)); }, 30, 5, 20);
        const double d_s2 = time_batches([&] { FE(fe_div3d_f64(gJ, gD, dv, (double*)sd, E, Np, FE_VARIANT_MFMA_SPLIT, s)); }, 30, 5, 20);
        const double f_m = time_batches([&] { FE(fe_facemass_f64(fJ, fR, fv, mf, E, Np, nf, Nfp, nb, 0, 0, s)); }, 10, 5, 10);
        const double f_s = time_batches([&] { FE(fe_facemass_f64(fJ, fR, fv, fo, E, Np, nf, Nfp, nb, 0, 0, s)); }, 10, 5, 10);
        const double gd_m = time_batches([&] { FE(fe_graddiv3d_f64(gJ, gD, gu, dv, mg, md, E, Np, 0, s)); }, 20, 5, 10);
        const double gd_s = time_batches([&] { FE(fe_graddiv3d_f64(gJ, gD, gu, dv, (double*)sg, (double*)sd, E, Np, 0, s)); }, 20, 5, 10);
        printf("grad            hipMalloc %.4f ms (%.1f %%)   split allocator %.4f ms (%.1f %%)\n", g_m, pct(GB, g_m), g_s, pct(GB, g_s));
        printf("div             hipMalloc %.4f ms (%.1f %%)   split allocator %.4f ms (%.1f %%)\n", d_m, pct(GB, d_m), d_s, pct(GB, d_s));
        printf("div two-window  hipMalloc %.4f ms (%.1f %%)   split allocator %.4f ms (%.1f %%)\n", d_m2, pct(GB, d_m2), d_s2, pct(GB, d_s2));
        printf("face-mass x4    hipMalloc %.4f ms (%.1f %%)   split allocator %.4f ms (%.1f %%)\n", f_m, pct(FB, f_m), f_s, pct(FB, f_s));
        printf("div + grad      hipMalloc %.4f ms (%.1f %%)   split allocator %.4f ms (%.1f %%)\n", gd_m, pct(GDB, gd_m), gd_s, pct(GDB, gd_s));
        // results: bitwise those of the plain allocations (last writers: graddiv for grad / div outputs, face-mass)
        CK(hipDeviceSynchronize());
        unsigned long long bad = differing(mg, (double*)sg, 3 * (size_t)E * Np) + differing(md, (double*)sd, (size_t)E * Np);
        for (int k = 0; k < 4; ++k) bad += differing(mf[k], fo[k], (size_t)E * Np);
        printf("differing output words between the two placements: %llu\n", bad);
        fflush(stdout);
        t0 = now_ms();
        FE(fe_split_free(sg));
        FE(fe_split_free(sd));
        for (auto& p : sf) FE(fe_split_free(p));
        printf("fe_split_free of the six arrays: %.1f ms\n", now_ms() - t0);
        if (bad) return 1;
    }
    char buf[4096];
    FE(fe_split_stats(buf, sizeof buf));
    printf("pool at the end: %s\n", buf);
    FE(fe_split_trim());
    FE(fe_split_stats(buf, sizeof buf));
    printf("pool after fe_split_trim: %s\n", buf);
    return 0;
}
