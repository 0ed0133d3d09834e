// fe_check.cpp -- standalone driver for libfeinsum_hip.so: correctness against a
// host loop nest and device timing, without Python.  Used for quick GPU
// iterations and as the program under rocprofv3 (fast start-up).
//
//   fe_check <family:grad|div|facemass|graddiv> <E> [variant=0] [launches=20] [check=1]      (FE_PREPARED=1: prepared operators)
//
// Build: hipcc -O2 tools/fe_check.cpp -Lfeinsum_amd -lfeinsum_hip -Wl,-rpath,'$ORIGIN/../feinsum_amd' -o build/fe_check
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../include/feinsum_hip.h"

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
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

static std::vector<double> rnd(size_t n, uint64_t seed) {
    std::mt19937_64 g(seed);
    std::uniform_real_distribution<double> d(0.0, 1.0);
    std::vector<double> v(n);
    for (auto& x : v) x = d(g);
    return v;
}
static double* to_dev(const std::vector<double>& h) {
    double* d;
    CK(hipMalloc(&d, std::max<size_t>(h.size(), 1) * 8));
    CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    return d;
}

// In-process A/B: interleaved rounds over several variants (methodology rule: compare in ONE
// process, report median and min).   fe_check ab <grad|div|facemass> E rounds launches v1,v2,...
static int ab_main(int argc, char** argv) {
    const std::string fam = argv[2];
    const int64_t E = atoll(argv[3]);
    const int rounds = atoi(argv[4]), launches = atoi(argv[5]);
    std::vector<int> variants;
    for (char* tok = strtok(argv[6], ","); tok; tok = strtok(nullptr, ",")) variants.push_back(atoi(tok));
    const int Np = getenv("FE_NP") ? atoi(getenv("FE_NP")) : 35, nf = 4,
              Nfp = Np == 35 ? 15 : Np == 20 ? 10 : Np == 10 ? 6 : Np == 4 ? 3 : 15, b = 4;
    fe_argpack a;
    memset(&a, 0, sizeof a);
    a.E = E; a.Np = Np; a.nf = nf; a.Nfp = Nfp; a.b = b;
    int family;
    double flops, bytes;
    std::vector<const double*> dv(b);
    std::vector<double*> dout(b);
    if (fam == "grad" || fam == "div") {
        a.J = to_dev(rnd(9 * E, 1)); a.D = to_dev(rnd(3 * Np * Np, 2));
        a.u = to_dev(rnd((fam == "grad" ? 1 : 3) * E * Np, 3));
        double* o; CK(hipMalloc(&o, (fam == "grad" ? 3 : 1) * E * Np * 8)); a.out = o;
        family = fam == "grad" ? FE_FAMILY_GRAD : FE_FAMILY_DIV;
        a.b = 1;
        flops = 7980.0 * E; bytes = 1192.0 * E;
    } else {
        a.J = to_dev(rnd(E * nf, 1)); a.D = to_dev(rnd(nf * Np * Nfp, 2));
        for (int k = 0; k < b; ++k) { dv[k] = to_dev(rnd(nf * E * Nfp, 10 + k)); CK(hipMalloc(&dout[k], E * Np * 8)); }
        a.v = dv.data(); a.outs = dout.data();
        family = FE_FAMILY_FACEMASS; flops = 17040.0 * E; bytes = 3072.0 * E;
    }
    // variant codes >= 100000: the same variant (code - 100000) with the operator in prepared form
    void* prepared = nullptr;
    CK(hipMalloc(&prepared, FE_PREPARED_OPERATOR_BYTES));
    if (fe_prepare_operator(family, a.D, Np, family == FE_FAMILY_FACEMASS ? nf : 0, family == FE_FAMILY_FACEMASS ? Nfp : 0, 0,
                            prepared, nullptr) != 0)
        prepared = nullptr;   // (no prepared form for this shape, e.g. p = 5: variant codes >= 100000 then launch plainly)
    auto set_variant = [&](int code) {
        a.prepared = code >= 100000 ? prepared : nullptr;
        a.variant = code >= 100000 ? code - 100000 : code;
    };
    std::vector<std::vector<float>> t(variants.size());
    float ms;
    for (size_t v = 0; v < variants.size(); ++v) { set_variant(variants[v]); FE(fe_time_launches(family, &a, 3, nullptr, &ms)); }
    for (int r = 0; r < rounds; ++r)
        for (size_t v = 0; v < variants.size(); ++v) {
            set_variant(variants[v]);
            FE(fe_time_launches(family, &a, launches, nullptr, &ms));
            t[v].push_back(ms / launches);
        }
    typedef int (*clk_fn)(unsigned long long*);
    clk_fn rd = (clk_fn)dlsym(RTLD_DEFAULT, "fe_dbg_read_clock");
    for (size_t v = 0; v < variants.size() && rd; ++v)
        if (variants[v] % 100000 >= 1000 && ((variants[v] % 100000 - 1000) & (fam == "div" ? 128 : 32))) {
            set_variant(variants[v]);
            FE(fe_time_launches(family, &a, launches, nullptr, &ms));
            unsigned long long c[2];
            rd(c);
            if (fam != "div")
                printf("variant %d: wave 0 main loop %llu shader cycles in %.1f us -> in-kernel clock %.0f MHz\n",
                       variants[v], c[0], c[1] / 100.0, (double)c[0] / (double)c[1] * 100.0);
            typedef int (*st_fn)(unsigned long long*, int);
            st_fn rs = (st_fn)dlsym(RTLD_DEFAULT, "fe_dbg_read_stamps");
            if (rs) {
                FE(fe_time_launches(family, &a, 1, nullptr, &ms));   // one isolated launch
                std::vector<unsigned long long> st4(2048 * 4), st(2048 * 3);
                rs(st4.data(), 2048);
                for (int w = 0; w < 2048; ++w) for (int k = 0; k < 3; ++k) st[3 * w + k] = st4[4 * w + k];
                if (getenv("FE_DUMP_STAMPS")) {
                    FILE* f = fopen(getenv("FE_DUMP_STAMPS"), "w");
                    unsigned long long tmin = ~0ull;
                    for (int w = 0; w < 2048; ++w) tmin = std::min(tmin, st4[4 * w]);
                    std::vector<unsigned long long> ph(2048 * 4, 0);
                    st_fn rp = (st_fn)dlsym(RTLD_DEFAULT, "fe_dbg_read_phase");
                    if (rp) rp(ph.data(), 2048);
                    fprintf(f, "wave,xcc,hw_id,entry_us,loop_start_us,loop_end_us,tiles,op_landed_us,barrier1_us,frags_built_us,barrier2_us\n");
                    for (int w = 0; w < 2048; ++w) {
                        fprintf(f, "%d,%llu,%llu,%.2f,%.2f,%.2f,%llu", w, st4[4 * w + 3] & 0xff, (st4[4 * w + 3] >> 8) & 0xffffffffull,
                                (st4[4 * w] - tmin) / 100.0, (st4[4 * w + 1] - tmin) / 100.0, (st4[4 * w + 2] - tmin) / 100.0,
                                st4[4 * w + 3] >> 40);
                        for (int k = 0; k < 4; ++k) fprintf(f, ",%.2f", ph[4 * w + k] >= tmin ? (ph[4 * w + k] - tmin) / 100.0 : -1.0);
                        fprintf(f, "\n");
                    }
                    fclose(f);
                    // per-tile stamps (fe_dbg_tile): <file>.tiles.csv, microseconds after the first wave's entry, -1 = not reached
                    st_fn rt = (st_fn)dlsym(RTLD_DEFAULT, "fe_dbg_read_tiles");
                    if (rt) {
                        std::vector<unsigned long long> tl(2048 * 16, 0);
                        rt(tl.data(), 2048);
                        const std::string name = std::string(getenv("FE_DUMP_STAMPS")) + ".tiles.csv";
                        FILE* g = fopen(name.c_str(), "w");
                        fprintf(g, "wave,xcc,hw_id,entry_us,loop_end_us,tiles");
                        for (int k = 0; k < 16; ++k) fprintf(g, ",t%d_%d", k / 4, k % 4);
                        fprintf(g, "\n");
                        for (int w = 0; w < 2048; ++w) {
                            fprintf(g, "%d,%llu,%llu,%.2f,%.2f,%llu", w, st4[4 * w + 3] & 0xff, (st4[4 * w + 3] >> 8) & 0xffffffffull,
                                    (st4[4 * w] - tmin) / 100.0, (st4[4 * w + 2] - tmin) / 100.0, st4[4 * w + 3] >> 40);
                            for (int k = 0; k < 16; ++k) fprintf(g, ",%.2f", tl[16 * w + k] >= tmin ? (tl[16 * w + k] - tmin) / 100.0 : -1.0);
                            fprintf(g, "\n");
                        }
                        fclose(g);
                    }
                }
                unsigned long long t0 = ~0ull, e_max = 0, l_min = ~0ull, l_max = 0, end_min = ~0ull, end_max = 0;
                std::vector<double> loop_us, end_us;
                for (int w = 0; w < 2048; ++w) t0 = std::min(t0, st[3 * w]);
                for (int w = 0; w < 2048; ++w) {
                    e_max = std::max(e_max, st[3 * w] - t0);
                    l_min = std::min(l_min, st[3 * w + 1] - t0); l_max = std::max(l_max, st[3 * w + 1] - t0);
                    end_min = std::min(end_min, st[3 * w + 2] - t0); end_max = std::max(end_max, st[3 * w + 2] - t0);
                    end_us.push_back((st[3 * w + 2] - t0) / 100.0);
                }
                std::sort(end_us.begin(), end_us.end());
                printf("  single launch %.1f us by events; waves: last entry +%.1f us, loop start +%.1f..%.1f us, "
                       "loop end +%.1f (first) %.1f (median) %.1f (p90) %.1f (last) us\n", ms * 1e3, e_max / 100.0,
                       l_min / 100.0, l_max / 100.0, end_min / 100.0, end_us[1024], end_us[1843], end_max / 100.0);
            }
        }
    for (size_t v = 0; v < variants.size(); ++v) {
        std::sort(t[v].begin(), t[v].end());
        const float med = t[v][t[v].size() / 2], mn = t[v].front();
        printf("%s E=%lld variant %6d: median %.4f ms (%.0f GFLOP/s, %.0f GB/s)  min %.4f ms  max %.4f ms\n", fam.c_str(),
               (long long)E, variants[v], med, flops / med * 1e-6, bytes / med * 1e-6, mn, t[v].back());
    }
    return 0;
}

// A/B between BUILDS in one process: the same launches through several copies of the library
// (dlopen), interleaved.   fe_check abl <grad|div|facemass> E rounds launches libA.so libB.so ...
typedef int (*time_fn)(int32_t, const fe_argpack*, int32_t, void*, float*);
static int abl_main(int argc, char** argv) {
    const std::string fam = argv[2];
    const int64_t E = atoll(argv[3]);
    const int rounds = atoi(argv[4]), launches = atoi(argv[5]);
    std::vector<time_fn> fns;
    std::vector<std::string> names;
    for (int i = 6; i < argc; ++i) {
        void* h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "dlopen %s: %s\n", argv[i], dlerror()); return 2; }
        fns.push_back((time_fn)dlsym(h, "fe_time_launches"));
        names.push_back(argv[i]);
    }
    const int Np = getenv("FE_NP") ? atoi(getenv("FE_NP")) : 35, nf = 4,
              Nfp = Np == 35 ? 15 : Np == 20 ? 10 : Np == 10 ? 6 : Np == 4 ? 3 : 15, b = 4;
    fe_argpack a;
    memset(&a, 0, sizeof a);
    a.E = E; a.Np = Np; a.nf = nf; a.Nfp = Nfp; a.b = b;
    if (getenv("FE_AB_VARIANT")) a.variant = atoi(getenv("FE_AB_VARIANT"));   // experiment builds only
    int family;
    double flops, bytes;
    std::vector<const double*> dv(b);
    std::vector<double*> dout(b);
    if (fam == "grad" || fam == "div") {
        a.J = to_dev(rnd(9 * E, 1)); a.D = to_dev(rnd(3 * Np * Np, 2));
        a.u = to_dev(rnd((fam == "grad" ? 1 : 3) * E * Np, 3));
        double* o; CK(hipMalloc(&o, (fam == "grad" ? 3 : 1) * E * Np * 8)); a.out = o;
        family = fam == "grad" ? FE_FAMILY_GRAD : FE_FAMILY_DIV;
        a.b = 1;
        flops = 7980.0 * E; bytes = 1192.0 * E;
    } else {
        a.J = to_dev(rnd(E * nf, 1)); a.D = to_dev(rnd(nf * Np * Nfp, 2));
        for (int k = 0; k < b; ++k) { dv[k] = to_dev(rnd(nf * E * Nfp, 10 + k)); CK(hipMalloc(&dout[k], E * Np * 8)); }
        a.v = dv.data(); a.outs = dout.data();
        family = FE_FAMILY_FACEMASS; flops = 17040.0 * E; bytes = 3072.0 * E;
    }
    std::vector<std::vector<float>> t(fns.size());
    float ms;
    for (auto f : fns) if (f(family, &a, 3, nullptr, &ms)) { fprintf(stderr, "launch failed\n"); return 3; }
    for (int r = 0; r < rounds; ++r)
        for (size_t v = 0; v < fns.size(); ++v) {
            if (fns[v](family, &a, launches, nullptr, &ms)) return 3;
            t[v].push_back(ms / launches);
        }
    for (size_t v = 0; v < fns.size(); ++v) {
        std::sort(t[v].begin(), t[v].end());
        const float med = t[v][t[v].size() / 2];
        printf("%s E=%lld %-40s median %.4f ms (%.0f GFLOP/s, %.0f GB/s)  min %.4f  max %.4f\n", fam.c_str(),
               (long long)E, names[v].c_str(), med, flops / med * 1e-6, bytes / med * 1e-6, t[v].front(), t[v].back());
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 7 && std::string(argv[1]) == "ab") return ab_main(argc, argv);
    if (argc >= 8 && std::string(argv[1]) == "abl") return abl_main(argc, argv);
    if (argc < 3) {
        fprintf(stderr, "usage: fe_check family E [variant] [launches] [check]\n");
        return 1;
    }
    const std::string fam = argv[1];
    const int64_t E = atoll(argv[2]);
    const int variant = argc > 3 ? atoi(argv[3]) : 0;
    const int launches = argc > 4 ? atoi(argv[4]) : 20;
    const int check = argc > 5 ? atoi(argv[5]) : 1;
    const int Np = getenv("FE_NP") ? atoi(getenv("FE_NP")) : 35, nf = 4,
              Nfp = Np == 35 ? 15 : Np == 20 ? 10 : Np == 10 ? 6 : Np == 4 ? 3 : 15, b = 4;

    char name[256];
    double pf, pb;
    FE(fe_device_info(0, name, sizeof name, &pf, &pb));
    printf("device: %s  peak_f64=%.0f GFLOP/s  peak_bw=%.0f GB/s\n", name, pf, pb);

    const int64_t nchk = std::min<int64_t>(E, 4096);  // elements checked on host: first+last
    double maxrel = 0.0;
    float ms = 0.f;
    double flops = 0, bytes = 0;

    if (fam == "grad" || fam == "div" || fam == "graddiv") {
        auto hJ = rnd(9 * E, 1), hD = rnd(3 * Np * Np, 2);
        auto hu = rnd(E * Np, 3), hv = rnd(3 * E * Np, 4);
        double *J = to_dev(hJ), *D = to_dev(hD), *u = to_dev(hu), *v = to_dev(hv);
        double *og, *od;
        CK(hipMalloc(&og, std::max<int64_t>(3 * E * Np, 1) * 8));
        CK(hipMalloc(&od, std::max<int64_t>(E * Np, 1) * 8));
        CK(hipMemset(og, 0xff, 3 * E * Np * 8));
        CK(hipMemset(od, 0xff, E * Np * 8));
        fe_argpack a;
        memset(&a, 0, sizeof a);
        a.J = J; a.D = D; a.E = E; a.Np = Np; a.variant = variant;
        if (getenv("FE_PREPARED") && atoi(getenv("FE_PREPARED"))) {
            void* prepared = nullptr;
            CK(hipMalloc(&prepared, FE_PREPARED_OPERATOR_BYTES));
            FE(fe_prepare_operator(FE_FAMILY_GRADDIV, D, Np, 0, 0, 0, prepared, nullptr));
            a.prepared = prepared;
        }
        int family;
        if (fam == "grad") { family = FE_FAMILY_GRAD; a.u = u; a.out = og; }
        else if (fam == "div") { family = FE_FAMILY_DIV; a.u = v; a.out = od; }
        else { family = FE_FAMILY_GRADDIV; a.u = u; a.v_div = v; a.out = og; a.out2 = od; }
        flops = (double)fe_flops_per_element(family, Np, 0, 0, 0) * E;
        bytes = (family == FE_FAMILY_GRADDIV ? 8.0 * (9 + 8 * Np) : 8.0 * (9 + 4 * Np)) * E +
                8.0 * 3 * Np * Np;
        float w;
        FE(fe_time_launches(family, &a, 3, nullptr, &w));  // warm-up
        FE(fe_time_launches(family, &a, launches, nullptr, &ms));
        ms /= launches;
        if (check) {
            std::vector<double> hg(3 * E * Np), hd(E * Np);
            CK(hipMemcpy(hg.data(), og, hg.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hd.data(), od, hd.size() * 8, hipMemcpyDeviceToHost));
            auto chk_elem = [&](int64_t e) {
                for (int i = 0; i < Np; ++i) {
                    if (fam != "div") {
                        long double t[3] = {0, 0, 0};
                        for (int r = 0; r < 3; ++r)
                            for (int j = 0; j < Np; ++j)
                                t[r] += (long double)hD[(r * Np + i) * Np + j] * hu[e * Np + j];
                        for (int x = 0; x < 3; ++x) {
                            long double ref = 0;
                            for (int r = 0; r < 3; ++r) ref += (long double)hJ[(x * 3 + r) * E + e] * t[r];
                            double got = hg[(x * E + e) * Np + i];
                            double rel = std::fabs((double)(got - ref)) / std::fabs((double)ref);
                            if (!(rel <= maxrel)) maxrel = std::isnan(rel) ? 1e300 : std::max(rel, maxrel);
                        }
                    }
                    if (fam != "grad") {
                        long double ref = 0;
                        for (int x = 0; x < 3; ++x)
                            for (int r = 0; r < 3; ++r)
                                for (int j = 0; j < Np; ++j)
                                    ref += (long double)hJ[(x * 3 + r) * E + e] *
                                           hD[(r * Np + i) * Np + j] * hv[(x * E + e) * Np + j];
                        double got = hd[e * Np + i];
                        double rel = std::fabs((double)(got - ref)) / std::fabs((double)ref);
                        if (!(rel <= maxrel)) maxrel = std::isnan(rel) ? 1e300 : std::max(rel, maxrel);
                    }
                }
            };
            for (int64_t e = 0; e < nchk / 2; ++e) chk_elem(e);
            for (int64_t e = std::max<int64_t>(nchk / 2, E - (nchk - nchk / 2)); e < E; ++e) chk_elem(e);
        }
    } else if (fam == "facemass") {
        auto hJ = rnd(E * nf, 1), hR = rnd(nf * Np * Nfp, 2);
        std::vector<std::vector<double>> hv(b);
        std::vector<const double*> dv(b);
        std::vector<double*> dout(b);
        for (int k = 0; k < b; ++k) {
            hv[k] = rnd(nf * E * Nfp, 10 + k);
            dv[k] = to_dev(hv[k]);
            CK(hipMalloc(&dout[k], std::max<int64_t>(E * Np, 1) * 8));
            CK(hipMemset(dout[k], 0xff, E * Np * 8));
        }
        double *J = to_dev(hJ), *R = to_dev(hR);
        fe_argpack a;
        memset(&a, 0, sizeof a);
        a.J = J; a.D = R; a.v = dv.data(); a.outs = dout.data();
        a.E = E; a.Np = Np; a.nf = nf; a.Nfp = Nfp; a.b = b; a.variant = variant;
        if (getenv("FE_PREPARED") && atoi(getenv("FE_PREPARED"))) {
            void* prepared = nullptr;
            CK(hipMalloc(&prepared, FE_PREPARED_OPERATOR_BYTES));
            FE(fe_prepare_operator(FE_FAMILY_FACEMASS, R, Np, nf, Nfp, 0, prepared, nullptr));
            a.prepared = prepared;
        }
        flops = (double)fe_flops_per_element(FE_FAMILY_FACEMASS, Np, nf, Nfp, b) * E;
        bytes = 8.0 * (nf + b * nf * Nfp + b * Np) * E + 8.0 * nf * Np * Nfp;
        float w;
        FE(fe_time_launches(FE_FAMILY_FACEMASS, &a, 3, nullptr, &w));
        FE(fe_time_launches(FE_FAMILY_FACEMASS, &a, launches, nullptr, &ms));
        ms /= launches;
        if (check) {
            for (int k = 0; k < b; ++k) {
                std::vector<double> ho(E * Np);
                CK(hipMemcpy(ho.data(), dout[k], ho.size() * 8, hipMemcpyDeviceToHost));
                auto chk_elem = [&](int64_t e) {
                    for (int i = 0; i < Np; ++i) {
                        long double ref = 0;
                        for (int f = 0; f < nf; ++f)
                            for (int j = 0; j < Nfp; ++j)
                                ref += (long double)hJ[e * nf + f] * hR[(f * Np + i) * Nfp + j] *
                                       hv[k][(f * E + e) * Nfp + j];
                        double rel = std::fabs((double)(ho[e * Np + i] - ref)) / std::fabs((double)ref);
                        if (!(rel <= maxrel)) maxrel = std::isnan(rel) ? 1e300 : std::max(rel, maxrel);
                    }
                };
                for (int64_t e = 0; e < nchk / 2; ++e) chk_elem(e);
                for (int64_t e = std::max<int64_t>(nchk / 2, E - (nchk - nchk / 2)); e < E; ++e) chk_elem(e);
            }
        }
    } else {
        fprintf(stderr, "unknown family %s\n", fam.c_str());
        return 1;
    }
    const double gflops = flops / (ms * 1e-3) * 1e-9, gbs = bytes / (ms * 1e-3) * 1e-9;
    printf("%s E=%lld variant=%d: %.4f ms/launch  %.1f GFLOP/s  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)"
           "  max_rel_err=%.3e %s\n",
           fam.c_str(), (long long)E, variant, ms, gflops, gbs, gbs / 80.0, maxrel,
           check ? (maxrel <= 1e-12 ? "PASS" : "FAIL") : "(unchecked)");
    return (check && !(maxrel <= 1e-12)) ? 4 : 0;
}
