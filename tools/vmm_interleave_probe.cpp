// vmm_interleave_probe.cpp -- do written arrays composed of ALTERNATING-class pieces run as fast as arrays split once
// across a class boundary?  (task 2: the experiment profiles/r02/placement_vmm_compositions.txt never made.)
//
// Phase 1  N physical handles of 2 GiB are created back to back and classified against handle 0, twice: with the known
//          classifier (face-mass x 4, two outputs in handle 0 + two in handle i: tools/vmm_class_probe.cpp) and with a
//          bare two-stream write probe (microseconds instead of a DG launch).
// Phase 2  how small may a probed region be?  The write probe on a same-class and on a different-class pair at 1 ... 256 MiB
//          per stream.
// Phase 3  hipMemMap takes no offset, so small pieces are handles of their own.  Does the driver hand out the hole a released
//          2 GiB handle leaves?  One handle of either class is released, 2048 handles of 2 MiB are created (timed), mapped
//          in creation order and classified in groups of 512 MiB.
// Phase 4  grad / div / face-mass x 4 at E = 1e6 with their OUTPUTS composed of those 2 MiB pieces: all of one class, two
//          halves, alternating every k pieces (k = 1 ... 64, i.e. 2 ... 128 MiB), with and without a phase shift between
//          planes / arrays.  Inputs are plain hipMalloc (their position does not matter: placement_joint_probe.txt).
//
//   vmm_interleave_probe [N=40]
//
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/vmm_interleave_probe.cpp -Lfeinsum_amd -lfeinsum_hip
//        -Wl,-rpath,'$ORIGIN/../feinsum_amd' -o build/vmm_interleave_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <thread>
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

// Two write streams walked in lockstep by a persistent grid, 4 KiB per wave and stream and step (as a DG tile: four
// 1-KiB non-temporal wave stores), `passes` times over `pieces` pieces of each stream.
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

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// median ms of `reps` batches of `n` calls
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

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 40;
    const size_t S = 2 * GIB;
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
    size_t gran_min = 0, gran_rec = 0;
    CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("# granularity min %zu KiB, recommended %zu KiB; device memory %.1f GiB free of %.1f\n", gran_min >> 10, gran_rec >> 10,
           free_b / (double)GIB, total_b / (double)GIB);

    // ---- operands (inputs: plain allocations)
    double* gJ = dev_random(9 * E, 1);
    double* gD = dev_random(3 * Np * Np, 2);
    double* gu = dev_random(E * Np, 3);
    double* dv = dev_random(3 * E * Np, 4);
    double* fJ = dev_random(E * nf, 5);
    double* fR = dev_random((size_t)nf * Np * Nfp, 6);
    const double* fv[4];
    for (int k = 0; k < nb; ++k) fv[k] = dev_random((size_t)nf * E * Nfp, 10 + k);
    const size_t plane = (size_t)E * Np * 8;                                // 280 MB
    const size_t W = (plane + 2 * MIB - 1) / (2 * MIB) * (2 * MIB);        // one output, 2 MiB rounded: 268 MiB

    // ---- phase 1: big handles, two classifiers
    char* va;
    CK(hipMemAddressReserve((void**)&va, S * N, 2 * MIB, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(N);
    double t0 = now_ms();
    for (int i = 0; i < N; ++i) {
        CK(hipMemCreate(&h[i], S, &prop, 0));
        CK(hipMemMap(va + S * i, S, 0, h[i], 0));
    }
    CK(hipMemSetAccess(va, S * N, &acc, 1));
    printf("# phase 1: %d handles of 2 GiB created + mapped in %.1f ms\n", N, now_ms() - t0);
    auto at = [&](int handle, int slot) { return va + S * handle + W * slot; };
    auto facemass_ms = [&](double* const* oo) {
        return time_batches([&] { FE(fe_facemass_f64(fJ, fR, fv, oo, E, Np, nf, Nfp, nb, 0, 0, s)); }, 10, 3, 10);
    };
    std::vector<double> fm(N), wp(N);
    for (int i = 0; i < N; ++i) {
        double* both[4] = {(double*)at(0, 0), (double*)at(0, 1), (double*)at(i, 2), (double*)at(i, 3)};
        fm[i] = facemass_ms(both);
        wp[i] = wprobe_ms(at(0, 0), at(i, 2), 256 * MIB, 1);
    }
    const double fm_mid = 0.5 * (*std::min_element(fm.begin(), fm.end()) + *std::max_element(fm.begin(), fm.end()));
    const double wp_mid = 0.5 * (*std::min_element(wp.begin(), wp.end()) + *std::max_element(wp.begin(), wp.end()));
    std::vector<int> cls(N);      // 0 = same class as handle 0 ("S"), 1 = the other ("F": the split launch is fast)
    int agree = 0;
    for (int i = 0; i < N; ++i) {
        cls[i] = fm[i] < fm_mid ? 1 : 0;
        const int w = wp[i] < wp_mid ? 1 : 0;
        agree += (w == cls[i]);
        printf("handle %3d (+%3zu GiB): face-mass 2+2 %.4f ms -> %c    write probe 2 x 256 MiB %.4f ms (%.0f GB/s) -> %c\n", i,
               (S * i) >> 30, fm[i], cls[i] ? 'F' : 'S', wp[i], 2 * 256 * MIB / (wp[i] * 1e-3) * 1e-9, w ? 'F' : 'S');
        fflush(stdout);
    }
    printf("# write probe agrees with the face-mass classifier on %d of %d handles (face-mass spread %.4f..%.4f, probe %.4f..%.4f)\n",
           agree, N, *std::min_element(fm.begin(), fm.end()), *std::max_element(fm.begin(), fm.end()),
           *std::min_element(wp.begin(), wp.end()), *std::max_element(wp.begin(), wp.end()));
    int same = -1, diff = -1;
    for (int i = N - 1; i > 0; --i) {
        if (cls[i] == 0 && same < 0) same = i;
        if (cls[i] == 1 && diff < 0) diff = i;
    }
    if (same < 0 || diff < 0) {
        printf("# only one class among the handles: nothing to interleave in this process\n");
        return 0;
    }

    // ---- phase 2: probe size
    printf("# phase 2: write probe, stream A in handle 0, stream B in handle %d (same class) / %d (other class)\n", same, diff);
    for (size_t mib : {(size_t)1, (size_t)2, (size_t)4, (size_t)8, (size_t)16, (size_t)32, (size_t)64, (size_t)128, (size_t)256}) {
        const int passes = (int)std::max<size_t>(1, 256 / mib);
        const double ts = wprobe_ms(at(0, 0), at(same, 2), mib * MIB, passes);
        const double td = wprobe_ms(at(0, 0), at(diff, 2), mib * MIB, passes);
        printf("probe %4zu MiB per stream x %3d passes: same class %.4f ms   other class %.4f ms   ratio %.3f\n", mib, passes, ts, td,
               ts / td);
    }
    fflush(stdout);

    // ---- phase 3: small pieces of known class.  hipMemMap takes no offset, so a piece is a handle of its own; a 2 MiB
    //      handle cannot be classified alone (phase 2), so handles are created in GROUPS of 64 (128 MiB), mapped in
    //      creation order and probed as one region against handle 0.  The big S handles (except the reference) are
    //      released first -- whether the driver reuses their memory shows in the class sequence.
    double ts128 = 0, td128 = 0;
    {
        ts128 = wprobe_ms(at(0, 0), at(same, 2), 128 * MIB, 2);
        td128 = wprobe_ms(at(0, 0), at(diff, 2), 128 * MIB, 2);
    }
    CK(hipDeviceSynchronize());
    int released = 0;
    for (int i = 1; i < N; ++i)
        if (cls[i] == 0) {
            CK(hipMemUnmap(va + S * i, S));
            CK(hipMemRelease(h[i]));
            ++released;
        }
    std::this_thread::sleep_for(std::chrono::milliseconds(500));
    const size_t P = 2 * MIB;
    const int group = 64;                                       // pieces per probed group: 128 MiB
    const int max_groups = 384;                                 // at most 48 GiB of small handles
    std::vector<hipMemGenericAllocationHandle_t> sh;
    std::vector<int> scls;
    char* sv;
    CK(hipMemAddressReserve((void**)&sv, P * group * (size_t)max_groups, 2 * MIB, nullptr, 0));
    size_t have[3] = {0, 0, 0};
    std::string seq;
    double t_create = 0, t_map = 0, t_probe = 0;
    int ngroups = 0;
    for (; ngroups < max_groups && (have[0] < 560 || have[1] < 560); ++ngroups) {
        const size_t base = sh.size();
        t0 = now_ms();
        for (int i = 0; i < group; ++i) {
            hipMemGenericAllocationHandle_t hh;
            CK(hipMemCreate(&hh, P, &prop, 0));
            sh.push_back(hh);
        }
        t_create += now_ms() - t0;
        t0 = now_ms();
        for (int i = 0; i < group; ++i) CK(hipMemMap(sv + P * (base + i), P, 0, sh[base + i], 0));
        CK(hipMemSetAccess(sv + P * base, P * group, &acc, 1));
        t_map += now_ms() - t0;
        t0 = now_ms();
        const double t = wprobe_ms(at(0, 0), sv + P * base, 128 * MIB, 2);
        t_probe += now_ms() - t0;
        const double lo = td128 + 0.25 * (ts128 - td128), hi = ts128 - 0.25 * (ts128 - td128);
        const int c = t >= hi ? 0 : t <= lo ? 1 : 2;
        seq += c == 0 ? 'S' : c == 1 ? 'F' : 'm';
        for (int i = 0; i < group; ++i) scls.push_back(c);
        have[c] += group;
    }
    printf("# phase 3: released %d big S handles; %d groups of 64 x 2 MiB handles: create %.1f ms, map + access %.1f ms, probe %.1f ms "
           "(thresholds from 128 MiB probes: same %.4f, other %.4f ms)\n# class of the groups in creation order (m = between):\n# %s\n",
           released, ngroups, t_create, t_map, t_probe, ts128, td128, seq.c_str());
    fflush(stdout);
    CK(hipDeviceSynchronize());
    CK(hipMemUnmap(sv, P * sh.size()));
    std::vector<int> pool[2];
    for (size_t i = 0; i < sh.size(); ++i)
        if (scls[i] < 2) pool[scls[i]].push_back((int)i);
    printf("# small pieces by class: S %zu, F %zu, unclear %zu\n", pool[0].size(), pool[1].size(), have[2]);
    if (pool[0].size() < 540 || pool[1].size() < 540) {
        printf("# not enough pieces of both classes for phase 4\n");
        return 0;
    }

    // ---- phase 4: outputs composed of 2 MiB pieces
    // pattern(plane, piece within plane) -> class
    struct Pattern { std::string name; std::function<int(int, int)> f; };
    std::vector<Pattern> patterns;
    patterns.push_back({"all S", [](int, int) { return 0; }});
    patterns.push_back({"all F", [](int, int) { return 1; }});
    for (int k : {1, 2, 4, 8, 16, 32, 64}) {
        char nm[64];
        snprintf(nm, sizeof nm, "alternating every %3d MiB", 2 * k);
        patterns.push_back({nm, [k](int, int q) { return (q / k) & 1; }});
        snprintf(nm, sizeof nm, "alternating every %3d MiB, planes shifted", 2 * k);
        patterns.push_back({nm, [k](int pl, int q) { return ((q / k) + pl) & 1; }});
    }
    const int ppl = (int)(W / P);                          // pieces per 268 MiB plane slot
    // EVERY composition gets an address range that was never mapped before: on this stack a range that is unmapped and
    // mapped again keeps translating to its first handles (tools/vmm_remap_test.cpp), which is why the first version of
    // this phase timed every pattern alike (profiles/r03/vmm_interleave_probe_run2.txt)
    char* ov = nullptr;                                     // 4 plane slots of W bytes
    auto compose = [&](int nplanes, bool contiguous_planes, const std::function<int(int, int)>& f) {
        CK(hipMemAddressReserve((void**)&ov, 4 * W, 2 * MIB, nullptr, 0));   // never freed
        // contiguous_planes (grad): one array [3][E][Np], plane x starts at byte x * plane (not piece aligned)
        size_t used[2] = {0, 0};
        const int npieces = contiguous_planes ? (int)((nplanes * plane + P - 1) / P) : nplanes * ppl;
        for (int q = 0; q < npieces; ++q) {
            int pl, qi;
            if (contiguous_planes) { pl = (int)((size_t)q * P / plane); qi = q - (int)((size_t)pl * plane / P); }
            else { pl = q / ppl; qi = q % ppl; }
            const int c = f(pl, qi) & 1;
            CK(hipMemMap(ov + P * q, P, 0, sh[pool[c][used[c]++]], 0));
        }
        CK(hipMemSetAccess(ov, P * npieces, &acc, 1));
        return npieces;
    };
    auto run_all = [&](const char* name, const std::function<int(int, int)>& f, const std::function<int(int, int)>* halves3,
                       const std::function<int(int, int)>* halves1, const std::function<int(int, int)>* halves4) {
        double tg, td, tds, tf;
        {
            int n = compose(3, true, halves3 ? *halves3 : f);
            tg = time_batches([&] { FE(fe_grad3d_f64(gJ, gD, gu, (double*)ov, E, Np, 0, s)); }, 30, 5, 20);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(ov, P * n));
        }
        {
            int n = compose(1, true, halves1 ? *halves1 : f);
            td = time_batches([&] { FE(fe_div3d_f64(gJ, gD, dv, (double*)ov, E, Np, FE_VARIANT_MFMA, s)); }, 30, 5, 20);
            tds = time_batches([&] { FE(fe_div3d_f64(gJ, gD, dv, (double*)ov, E, Np, FE_VARIANT_MFMA_SPLIT, s)); }, 30, 5, 20);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(ov, P * n));
        }
        {
            int n = compose(4, false, halves4 ? *halves4 : f);
            double* oo[4] = {(double*)ov, (double*)(ov + W), (double*)(ov + 2 * W), (double*)(ov + 3 * W)};
            tf = facemass_ms(oo);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(ov, P * n));
        }
        printf("%-46s grad %.4f ms (%.1f %%)  div %.4f (%.1f %%)  div two-window %.4f (%.1f %%)  face-mass x4 %.4f (%.1f %%)\n", name,
               tg, 1192.0294e6 / (tg * 1e-3) / 8e12 * 100, td, 1192.0294e6 / (td * 1e-3) / 8e12 * 100, tds,
               1192.0294e6 / (tds * 1e-3) / 8e12 * 100, tf, 3072.0168e6 / (tf * 1e-3) / 8e12 * 100);
        fflush(stdout);
    };
    printf("# phase 4: outputs composed of 2 MiB pieces (%% of the 8 TB/s roofline)\n");
    {   // plain allocations
        double *o3, *o1, *o4[4];
        CK(hipMalloc(&o3, 3 * plane));
        CK(hipMalloc(&o1, plane));
        for (auto& p : o4) CK(hipMalloc(&p, plane));
        const double tg = time_batches([&] { FE(fe_grad3d_f64(gJ, gD, gu, o3, E, Np, 0, s)); }, 30, 5, 20);
        const double td = time_batches([&] { FE(fe_div3d_f64(gJ, gD, dv, o1, E, Np, FE_VARIANT_MFMA, s)); }, 30, 5, 20);
        const double tf = facemass_ms(o4);
        printf("%-46s grad %.4f ms (%.1f %%)  div %.4f (%.1f %%)  face-mass x4 %.4f (%.1f %%)\n", "hipMalloc per array", tg,
               1192.0294e6 / (tg * 1e-3) / 8e12 * 100, td, 1192.0294e6 / (td * 1e-3) / 8e12 * 100, tf,
               3072.0168e6 / (tf * 1e-3) / 8e12 * 100);
        CK(hipFree(o3)); CK(hipFree(o1));
        for (auto& p : o4) CK(hipFree(p));
    }
    for (int rep = 0; rep < 2; ++rep) {
        for (auto& p : patterns) run_all(p.name.c_str(), p.f, nullptr, nullptr, nullptr);
        // one cut: grad cut in the middle of the array (piece 200 of 401), div in the middle of its array, face-mass 2 + 2
        std::function<int(int, int)> h3 = [&](int pl, int q) { return ((size_t)pl * plane / P + q) < 200 ? 0 : 1; };
        std::function<int(int, int)> h1 = [&](int, int q) { return q < 67 ? 0 : 1; };
        std::function<int(int, int)> h4 = [&](int pl, int) { return pl < 2 ? 0 : 1; };
        run_all("one cut (halves)", h4, &h3, &h1, &h4);
    }
    return 0;
}
