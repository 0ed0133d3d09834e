// vmm_probe.cpp -- does the launch time depend on how the OUTPUT array is composed of physical allocations?
//
// The grad output [3][E][Np] (one virtually contiguous array) is built with HIP's virtual-memory API from several
// physical handles (hipMemCreate / hipMemMap), optionally with "filler" handles created between them so that consecutive
// pieces lie further apart physically.  Inputs are plain hipMalloc (their position does not matter:
// profiles/r02/placement_joint_probe.txt).  One line per composition: median / min ms of fe_grad3d_f64 (or face-mass x 4).
//
//   vmm_probe <grad|facemass> [E=1000000]
//
// Build: hipcc -O2 -std=c++17 tools/vmm_probe.cpp -Lfeinsum_amd -lfeinsum_hip -Wl,-rpath,'$ORIGIN/../feinsum_amd' -o build/vmm_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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

static const size_t MIB = 1ull << 20;
static size_t gran = 2 * MIB;
static hipMemAllocationProp prop;

static hipMemGenericAllocationHandle_t create(size_t bytes) {
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, bytes, &prop, 0));
    return h;
}

// A virtually contiguous range of `total` bytes composed of pieces of the given sizes (the last piece takes the rest);
// `filler` bytes of physical memory are allocated (and kept until release) between consecutive pieces.
struct Composed {
    char* va = nullptr;
    size_t total = 0;
    std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t>> pieces;
    std::vector<hipMemGenericAllocationHandle_t> fillers;
    void build(size_t total_, const std::vector<size_t>& sizes, size_t filler, bool reverse_va = false) {
        total = (total_ + gran - 1) / gran * gran;
        CK(hipMemAddressReserve((void**)&va, total, gran, nullptr, 0));
        size_t used = 0;
        std::vector<size_t> sz;
        for (size_t k = 0; k < sizes.size() && used < total; ++k) {
            size_t s = (k + 1 == sizes.size()) ? total - used : std::min(total - used, sizes[k] / gran * gran);
            if (s == 0) continue;
            sz.push_back(s);
            used += s;
        }
        if (used < total) sz.push_back(total - used);
        for (size_t k = 0; k < sz.size(); ++k) {
            pieces.push_back({create(sz[k]), sz[k]});
            if (filler && k + 1 < sz.size()) fillers.push_back(create(filler));
        }
        // map in creation order (or reversed: the piece created last comes first in the virtual range)
        size_t off = 0;
        for (size_t k = 0; k < pieces.size(); ++k) {
            auto& p = pieces[reverse_va ? pieces.size() - 1 - k : k];
            CK(hipMemMap(va + off, p.second, 0, p.first, 0));
            off += p.second;
        }
        hipMemAccessDesc acc{};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = 0;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, total, &acc, 1));
    }
    void release() {
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(va, total));
        for (auto& p : pieces) CK(hipMemRelease(p.first));
        for (auto& f : fillers) CK(hipMemRelease(f));
        CK(hipMemAddressFree(va, total));
        pieces.clear();
        fillers.clear();
        va = nullptr;
    }
};

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
    const std::string fam = argc > 1 ? argv[1] : "grad";
    const int64_t E = argc > 2 ? atoll(argv[2]) : 1000000;
    const int Np = 35, Nfp = 15, nf = 4, nb = 4;
    CK(hipSetDevice(0));
    prop = hipMemAllocationProp{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("# %s E=%lld, allocation granularity %zu KiB\n", fam.c_str(), (long long)E, gran >> 10);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    const bool grad = fam == "grad";
    double *J, *D, *u = nullptr, *vbuf[4] = {nullptr, nullptr, nullptr, nullptr};
    if (grad) {
        J = dev_random(9 * E, 1);
        D = dev_random(3 * Np * Np, 2);
        u = dev_random(E * Np, 3);
    } else {
        J = dev_random(E * nf, 1);
        D = dev_random((size_t)nf * Np * Nfp, 2);
        for (int k = 0; k < nb; ++k) vbuf[k] = dev_random((size_t)nf * E * Nfp, 10 + k);
    }
    const size_t plane = (size_t)E * Np * 8;                 // one output plane / one face-mass output
    const int nplanes = grad ? 3 : 4;
    const size_t out_bytes = plane * nplanes;

    auto time_it = [&](double* out, double& med, double& mn) {
        const double* vv[4] = {vbuf[0], vbuf[1], vbuf[2], vbuf[3]};
        double* oo[4] = {out, out + plane / 8, out + 2 * (plane / 8), out + 3 * (plane / 8)};
        auto launch = [&]() {
            if (grad) FE(fe_grad3d_f64(J, D, u, out, E, Np, 0, s));
            else FE(fe_facemass_f64(J, D, vv, oo, E, Np, nf, Nfp, nb, 0, 0, s));
        };
        for (int i = 0; i < 30; ++i) launch();
        CK(hipStreamSynchronize(s));
        std::vector<double> ts;
        for (int r = 0; r < 7; ++r) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ts.push_back(ms / 20);
        }
        std::sort(ts.begin(), ts.end());
        med = ts[ts.size() / 2];
        mn = ts[0];
    };

    auto run = [&](const char* what, const std::vector<size_t>& sizes, size_t filler, bool rev = false) {
        Composed c;
        c.build(out_bytes, sizes, filler, rev);
        double med, mn;
        time_it(reinterpret_cast<double*>(c.va), med, mn);
        printf("%-58s pieces %3zu  filler %6zu MiB%s  median %.4f ms  min %.4f ms\n", what, c.pieces.size(), filler / MIB,
               rev ? " (reversed)" : "", med, mn);
        fflush(stdout);
        c.release();
    };

    // baseline: plain hipMalloc
    {
        double* out;
        CK(hipMalloc(&out, out_bytes));
        double med, mn;
        time_it(out, med, mn);
        printf("%-58s                                  median %.4f ms  min %.4f ms\n", "hipMalloc", med, mn);
        CK(hipFree(out));
    }
    const size_t P = plane / gran * gran;                     // a plane, rounded down to the granularity
    run("one handle", {out_bytes}, 0);
    for (int rep = 0; rep < 2; ++rep) {
        run("one piece per plane", std::vector<size_t>(nplanes, P), 0);
        run("one piece per plane", std::vector<size_t>(nplanes, P), 0, true);
    }
    for (size_t f : {(size_t)2, (size_t)64, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)32768})
        run("one piece per plane", std::vector<size_t>(nplanes, P), f * MIB);
    run("two pieces (cut in the middle)", {out_bytes / 2}, 0);
    for (size_t f : {(size_t)1024, (size_t)16384})
        run("two pieces (cut in the middle)", {out_bytes / 2}, f * MIB);
    for (size_t piece : {(size_t)128, (size_t)64, (size_t)32, (size_t)16, (size_t)8, (size_t)4, (size_t)2}) {
        std::vector<size_t> sz((out_bytes + piece * MIB - 1) / (piece * MIB), piece * MIB);
        char name[64];
        snprintf(name, sizeof name, "pieces of %zu MiB", piece);
        run(name, sz, 0);
    }
    // pieces of 64 MiB with fillers of 64 MiB between them (every second 64 MiB of a physical range)
    {
        std::vector<size_t> sz((out_bytes + 64 * MIB - 1) / (64 * MIB), 64 * MIB);
        run("pieces of 64 MiB", sz, 64 * MIB);
        run("pieces of 64 MiB", sz, 1024 * MIB);
    }
    return 0;
}
