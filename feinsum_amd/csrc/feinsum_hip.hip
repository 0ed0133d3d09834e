// feinsum_hip.hip -- C ABI of libfeinsum_hip.so (see include/feinsum_hip.h).
// gfx950 only.  Host side: argument validation, kernel-variant dispatch (the
// build's replacement for feinsum's transform archive lookup,
// sql_utils.py:247-294) and asynchronous launches on the caller's stream.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/feinsum_hip.h"
#include "fe_common.h"
#include "fe_div.h"
#include "fe_einsum.h"
#include "fe_facemass.h"
#include "fe_fused.h"
#include "fe_generic.h"
#include "fe_grad.h"
#include "fe_grad_f32.h"
#include "fe_div_f32.h"
#include "fe_facemass_f32.h"
#include "fe_tiled.h"

namespace {

thread_local char g_err[512] = "";

// Counts the FE_EHIP returns of this process: after a HIP error a launch may not have run to completion, and a dynamic launch
// that stopped half way leaves tickets in its counter group (later launches through that group would skip tiles).  The first
// dynamic launch of every stream after such a return verifies its group once (tail_slot below).
std::atomic<unsigned> g_hip_error_epoch{0};

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    if (code == FE_EHIP) g_hip_error_epoch.fetch_add(1, std::memory_order_relaxed);
    return code;
}

// (a failed call is reported once: HIP's per-thread "last error" is cleared here, or the launch after it -- which ends with
//  FE_HIP_CHECK(hipGetLastError()) -- would report the same, already handled error again)
#define FE_HIP_CHECK(expr)                                                             \
    do {                                                                               \
        hipError_t _e = (expr);                                                        \
        if (_e != hipSuccess) {                                                        \
            (void)hipGetLastError();                                                   \
            return fail(FE_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e));       \
        }                                                                              \
    } while (0)

// Persistent-style grid: enough blocks to fill every CU at the kernel's
// residency (2 blocks of 256 threads per CU), capped by the work available.
std::atomic<int> g_cu_limit{[] { const char* e = getenv("FEINSUM_CU_LIMIT"); return e ? atoi(e) : 0; }()};
int device_cu_count() {
    // fe_set_cu_limit / FEINSUM_CU_LIMIT: size the persistent grids as if the device had fewer CUs (a CPX / QPX partition
    // of MI355X reports 32 / 64) -- a test hook for the small-grid paths, and a way to leave part of the device to others
    const int limit = g_cu_limit.load(std::memory_order_relaxed);
    static int cus[64];
    static std::once_flag once[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return limit > 0 && limit < 256 ? limit : 256;
    std::call_once(once[dev], [dev] {
        hipDeviceProp_t p;
        cus[dev] = (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
                       ? p.multiProcessorCount
                       : 256;
    });
    return limit > 0 && limit < cus[dev] ? limit : cus[dev];
}

// Resources of every kernel configured so far in this process (fe_kernel_resources).
std::mutex g_resources_mutex;
std::string g_resources;

// Once per (kernel, device): raise the dynamic-LDS limit, then check that the compiled kernel really
// has the residency its launch geometry assumes.  The persistent grids are sized as
// blocks_per_cu x #CUs; if a compiler change pushed the kernel over a register step (> 256 VGPRs
// at two blocks of four waves per CU) the grid's second half would silently queue behind the
// first and every test would still pass at half the speed.
template <typename K>
int configure_kernel(K kernel, const char* what, int lds_bytes, int threads, int blocks_per_cu) {
    const void* fn = reinterpret_cast<const void*>(kernel);
    FE_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    int resident = 0;
    FE_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, fn, threads, (size_t)lds_bytes));
    hipFuncAttributes attr;
    FE_HIP_CHECK(hipFuncGetAttributes(&attr, fn));
    {
        char line[256];
        snprintf(line, sizeof(line), "%-34s threads %4d  VGPRs %3d  scratch %4zu B  LDS %6d B dynamic + %5zu B static  "
                 "blocks/CU %d (geometry needs %d)\n", what, threads, attr.numRegs, (size_t)attr.localSizeBytes,
                 lds_bytes, (size_t)attr.sharedSizeBytes, resident, blocks_per_cu);
        std::lock_guard<std::mutex> lock(g_resources_mutex);
        g_resources += line;
    }
    if (resident < blocks_per_cu) {
        // The launch is still correct (the surplus blocks queue behind the resident ones), only slower: by default this
        // is a WARNING line in fe_kernel_resources().  FEINSUM_STRICT_RESIDENCY=1 (the test suite and bench.py set it)
        // turns it into FE_EHIP, so that a compiler regression cannot pass with green tests at half the speed.
        char line[320];
        snprintf(line, sizeof(line), "WARNING %s: %d block(s) of %d threads fit a CU but the launch geometry assumes %d "
                 "(%d VGPRs, %d B of LDS per block): the compiled kernel's register or LDS use grew\n",
                 what, resident, threads, blocks_per_cu, attr.numRegs, lds_bytes);
        {
            std::lock_guard<std::mutex> lock(g_resources_mutex);
            g_resources += line;
        }
        const char* strict = getenv("FEINSUM_STRICT_RESIDENCY");
        if (strict && strict[0] == '1') return fail(FE_EHIP, "%s", line + 8);
    }
    return FE_OK;
}

// configure_kernel under the caller's once-flag: every kernel INSTANTIATION has a flag of its own, so that a
// shortfall (or any other failure) of one instantiation -- say the opt-in prepared-operator one -- cannot fail the
// launches of its siblings for the rest of the process.
template <typename Once, typename K>
int configured(Once& once, K kernel, const char* what, int lds_bytes, int threads, int blocks_per_cu) {
    return once.run([&] { return configure_kernel(kernel, what, lds_bytes, threads, blocks_per_cu); });
}

// Kernel attributes are per device: a `static std::once_flag` per kernel would configure only
// the first device a process uses.  One flag per (call site, device).
struct PerDeviceOnce {
    std::once_flag flags[64];
    int rc[64] = {};
    template <typename F>
    int run(F&& f) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        std::call_once(flags[dev], [&] { rc[dev] = f(); });
        return rc[dev];
    }
};

}  // namespace
#include "fe_split_alloc.h"
namespace {

int check_common(const void* J, const void* D, const void* u, const void* out, int64_t E,
                 int32_t Np) {
    if (E < 0) return fail(FE_EINVAL, "E must be >= 0 (got %lld)", (long long)E);
    if (Np <= 0) return fail(FE_EINVAL, "Np must be positive (got %d)", Np);
    if (E > 0 && (!J || !D || !u || !out)) return fail(FE_EINVAL, "null device pointer");
    if ((reinterpret_cast<uintptr_t>(J) | reinterpret_cast<uintptr_t>(D) | reinterpret_cast<uintptr_t>(u) |
         reinterpret_cast<uintptr_t>(out)) & 7u)
        return fail(FE_EINVAL, "device pointers must be 8-byte aligned (float64 arrays)");
    if (E * (int64_t)Np >= (int64_t)1 << 39)
        return fail(FE_EINVAL, "E*Np too large (%lld)", (long long)(E * Np));
    return FE_OK;
}

unsigned generic_grid(int64_t E, int Np) { return (unsigned)((E * Np + 255) / 256); }

// ---- dynamic walk (fe_common.h): the ticket counters of a launch.
// Who may share counters: nobody who can run at the same time.  A launch's counters are zero before and after it (the
// kernels clean up behind themselves), so launches that the device SERIALISES may use the same ones; launches that can
// overlap must not (two launches drawing from one counter take each other's tiles -- silently).  Ownership therefore
// follows what orders launches:
//  * an eager launch uses the counter group of ITS STREAM (a per-device map stream -> group; the per-thread default
//    stream is a different stream in every thread and is keyed by the thread).  Launches on one stream run one after the
//    other; two streams -- driven from one host thread or from several -- never share a group;
//  * a launch recorded during stream capture gets a group of ITS OWN, for good: the graph node bakes the pointer in and
//    may be replayed on any stream beside any eager launch (launches of one executable graph are ordered by HIP);
//  * a group is four consecutive counter sets (a fused launch uses one set per body);
//  * groups come from chunks of kTailChunkGroups, allocated and zeroed outside stream capture only (a capture that finds
//    no spare group walks statically, and so does everything once kTailMaxGroups exist: the static walk needs no state);
//  * fe_stream_retired() returns a destroyed stream's group; fe_tail_check() verifies the zero-between-launches invariant
//    on an idle device (and repairs it), FEINSUM_TAIL_CHECK=1 does so before every dynamic launch (debugging aid: it
//    synchronises the stream).
// Round 5 closes three edges of that scheme:
//  * ONE EXECUTABLE PER CAPTURE.  The group pointer is baked into the graph NODE: every executable instantiated from one
//    captured graph uses the same counters, and HIP orders only the launches of one executable.  Two executables of one
//    capture must not run at the same time (include/feinsum_hip.h says so; BoundOperator.capture() instantiates one).
//  * captured groups come back: the groups a capture took are recorded under the capture's id (fe_capture_id), and
//    fe_graph_retired(id) returns them once the caller has destroyed the graph -- an application that re-captures every
//    step no longer runs out of groups after kTailMaxGroups captures and falls back to the static walk for good;
//    fe_tail_stats reports `exhausted` and `static_fallbacks` so that the fallback is visible;
//  * a failed chunk allocation is retried (not latched), and after any FE_EHIP return of the process the first dynamic
//    launch of every stream verifies (and repairs) its group once -- the checked mode is no longer opt-in where it matters.
constexpr int kTailSetsPerGroup = 4;
constexpr int kTailChunkGroups = 16;                     // 8.9 MB per chunk
constexpr int kTailMaxGroups = 256;                      // per device
constexpr size_t kTailGroupWords = (size_t)kTailSetsPerGroup * fe::kTailWords;
struct TailPool {
    std::mutex lock;
    std::unordered_map<uintptr_t, unsigned*> by_stream;   // eager launches
    std::unordered_map<uintptr_t, unsigned> verified_at;  // per stream: the FE_EHIP epoch its group was last verified at
    std::unordered_map<unsigned long long, std::vector<unsigned*>> by_capture;   // capture id -> the groups its nodes own
    std::vector<unsigned*> spare;                          // zeroed groups nobody owns
    std::vector<unsigned*> chunks;                         // every allocation (fe_tail_check walks them)
    int groups = 0, captured = 0;
    long long exhausted = 0;          // launches that found no group (all kTailMaxGroups owned, or none spare during capture)
    long long static_fallbacks = 0;   // launches that wanted tickets and walked statically (exhausted, failed allocation, HIP errors)
    long long grow_failures = 0;      // chunk allocations that failed (retried by later launches)
    long long verified_after_error = 0, repaired_after_error = 0;
    int failed_recently = 0;          // launches to wait before the next allocation attempt
    hipStream_t zero_stream = nullptr;
    // a fresh chunk: zeroed through a stream of our own and waited for, so that whoever takes a group later -- on any
    // stream -- finds zeros without being ordered behind anything
    bool grow() {
        if (groups + kTailChunkGroups > kTailMaxGroups) return false;
        // a failed allocation (say during another thread's global-mode capture) is not final: the launches walk statically
        // meanwhile and the 64th of them tries again
        if (failed_recently > 0) { --failed_recently; return false; }
        unsigned* p = nullptr;
        const size_t bytes = (size_t)kTailChunkGroups * kTailGroupWords * sizeof(unsigned);
        bool ok = zero_stream || hipStreamCreateWithFlags(&zero_stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipMalloc(&p, bytes) == hipSuccess;
        ok = ok && hipMemsetAsync(p, 0, bytes, zero_stream) == hipSuccess && hipStreamSynchronize(zero_stream) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            if (p) (void)hipFree(p);
            ++grow_failures;
            failed_recently = 64;
            return false;
        }
        chunks.push_back(p);
        for (int g = kTailChunkGroups - 1; g >= 0; --g) spare.push_back(p + (size_t)g * kTailGroupWords);
        groups += kTailChunkGroups;
        return true;
    }
};
TailPool g_tail[64];
std::atomic<int> g_tail_check{[] { const char* e = getenv("FEINSUM_TAIL_CHECK"); return e && e[0] == '1' ? 1 : 0; }()};

uintptr_t tail_stream_key(hipStream_t s) {
    if (s == hipStreamPerThread) {   // one handle value, a different stream in every thread
        static thread_local char marker;
        return reinterpret_cast<uintptr_t>(&marker) | 1u;
    }
    return reinterpret_cast<uintptr_t>(s);
}

// counts (and clears) the non-zero words of a group; the caller has made sure nothing is running on it
int tail_group_dirty(unsigned* group, bool repair, long long* dirty) {
    static thread_local std::vector<unsigned> host;
    host.resize(kTailGroupWords);
    FE_HIP_CHECK(hipMemcpy(host.data(), group, kTailGroupWords * sizeof(unsigned), hipMemcpyDeviceToHost));
    long long n = 0;
    for (unsigned w : host) n += w != 0;
    if (n && repair) FE_HIP_CHECK(hipMemset(group, 0, kTailGroupWords * sizeof(unsigned)));
    *dirty += n;
    return FE_OK;
}

// The counters for a launch on stream `s` that needs `sets` consecutive sets; null = walk statically.
unsigned* tail_slot(hipStream_t s, int sets = 1) {
    int dev = 0;
    if (sets < 1 || sets > kTailSetsPerGroup) return nullptr;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    TailPool& pool = g_tail[dev];
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long capture_id = 0;
    if (hipStreamGetCaptureInfo(s, &st, &capture_id) != hipSuccess) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> guard(pool.lock);
        ++pool.static_fallbacks;
        return nullptr;
    }
    std::lock_guard<std::mutex> guard(pool.lock);
    if (st != hipStreamCaptureStatusNone) {   // a graph node: a group of its own until fe_graph_retired(capture id)
        if (pool.spare.empty()) {              // (no allocation during capture)
            ++pool.exhausted;
            ++pool.static_fallbacks;
            return nullptr;
        }
        unsigned* g = pool.spare.back();
        pool.spare.pop_back();
        ++pool.captured;
        pool.by_capture[capture_id].push_back(g);
        return g;
    }
    const uintptr_t key = tail_stream_key(s);
    auto it = pool.by_stream.find(key);
    if (it == pool.by_stream.end()) {
        // keep a chunk's worth of spares behind the eager streams, so that a later capture finds some
        if (pool.spare.size() <= 1 && !pool.grow() && pool.spare.empty()) {
            if (pool.groups + kTailChunkGroups > kTailMaxGroups) ++pool.exhausted;
            ++pool.static_fallbacks;
            return nullptr;
        }
        unsigned* g = pool.spare.back();
        pool.spare.pop_back();
        it = pool.by_stream.emplace(key, g).first;
        pool.verified_at[key] = g_hip_error_epoch.load(std::memory_order_relaxed);   // fresh from the spares: zero
    }
    // after any FE_EHIP return of the process a launch may have stopped half way: the first dynamic launch of every stream
    // behind it verifies its group once (it waits for the stream: rare, and only after an error); FEINSUM_TAIL_CHECK=1 does
    // so before EVERY dynamic launch (debugging aid)
    const unsigned epoch = g_hip_error_epoch.load(std::memory_order_relaxed);
    unsigned& seen = pool.verified_at[key];
    const bool after_error = seen != epoch;
    if (after_error || g_tail_check.load(std::memory_order_relaxed)) {
        long long dirty = 0;
        const bool ok = hipStreamSynchronize(s) == hipSuccess && tail_group_dirty(it->second, true, &dirty) == FE_OK;
        if (!ok) (void)hipGetLastError();
        if (after_error) {
            ++pool.verified_after_error;
            if (dirty) ++pool.repaired_after_error;
            if (ok) seen = epoch;
        }
        if (!ok || dirty)
            fprintf(stderr, "feinsum_hip: ticket counters of stream %p held %lld non-zero words before a launch (%s)\n",
                    (void*)s, dirty, ok ? "repaired" : "could not be verified: static walk");
        if (!ok) {
            ++pool.static_fallbacks;
            return nullptr;
        }
    }
    return it->second;
}
// Number of statically walked tiles of a launch of `blocks` blocks of `waves / blocks` waves: two rounds, the rest by
// tickets (FEINSUM_TAIL_ROUNDS / fe_set_tail_rounds: at most so many full rounds by tickets; negative: none); launches of
// fewer than four and a half rounds walk statically (E = 1e5 on 2048 waves: three rounds, measured slower with tickets), and so do
// grids of fewer than 8 blocks per pool: a block's pool is (bid / 8) % kTailPools and nobody steals, so that a pool
// without blocks would keep its tiles (a 32-CU partition launches 64 blocks).
std::atomic<int> g_tail_rounds{[] { const char* e = getenv("FEINSUM_TAIL_ROUNDS"); return e ? atoi(e) : (1 << 20); }()};
// (experiments: FEINSUM_TAIL_MIN_ROUNDS lowers the four-round rule below, to re-measure tickets in shorter launches)
std::atomic<int> g_tail_min_rounds_v{[] { const char* e = getenv("FEINSUM_TAIL_MIN_ROUNDS"); return e ? atoi(e) : 4; }()};
#define g_tail_min_rounds g_tail_min_rounds_v.load(std::memory_order_relaxed)
// `fused`: a body of a fused launch (div + grad, div + grad + lift) walks dynamically from THREE rounds on -- the bodies of a block
// follow each other, so a block that is late in one body starts the next late, and tickets absorb that: div + grad at E = 1e5
// 46.1 -> 43.4 us, the pipeline 103.9 -> 97.5, -1 ... -6 % at every size between three and four and a half rounds, level at exact
// multiples (profiles/r05/fused_tickets3_ab.txt); single launches lose up to 8 % there (single_tickets3_ab.txt) and keep four.
// `units`: a face-mass launch of four fields (p = 4) walks four units per tile, so a ticket's round trip hides behind more work: it
// gains from tickets from three rounds on as the fused launches do -- E = 1e5 52.3 -> 51.0 us, 1.2e5 -3.6 %, 1.4e5 -2.3 %, level
// at 1.1e5 -- except where the tiles fill whole rounds, which are balanced as they are (98 304: +3.6 % with tickets;
// profiles/r05/facemass_tickets_from_three_rounds_ab.txt).
int64_t tail_static_tiles(int64_t nTiles, int64_t blocks, int wavesPerBlock, bool fused = false, bool units = false) {
    const int dyn_rounds = g_tail_rounds.load(std::memory_order_relaxed);
    const int64_t waves = blocks * wavesPerBlock;
    const int64_t rounds = waves > 0 ? nTiles / waves : 0;
    const int min_rounds = ((fused || units) && g_tail_min_rounds == 4) ? 3 : g_tail_min_rounds;
    if (blocks < 8 * fe::kTailPools) return nTiles;
    if (dyn_rounds < 0 || rounds < min_rounds || nTiles >= ((int64_t)1 << 29)) return nTiles;   // (32-bit ticket arithmetic: fe_common.h)
    if (units && !fused && rounds <= 4 && nTiles == rounds * waves) return nTiles;
    // four rounds and a bit: tickets pay once the partial fifth round is at least half a round -- a static walk leaves those waves
    // a tile behind the rest (div E = 163 000: 41.2 -> 37.7 us; grad 150 000: 34.4 -> 33.5), while four EXACT rounds are perfectly
    // balanced as they are (grad 131 072: 28.6 static, 30.7 with tickets; profiles/r04/dynamic_walk_from_four_and_a_half_rounds.txt)
    if (rounds == 4 && min_rounds == 4 && (nTiles - 4 * waves) * 2 < waves) return nTiles;
    int64_t ks = rounds - dyn_rounds;
    if (ks < 2) ks = 2;
    return ks * waves;
}

// kOpLoadsTemporal (fe_common.h) for a launch that reads `input_bytes`: plain instead of non-temporal loads of the streamed
// operand while the launch's inputs fit the Infinity Cache ($FEINSUM_TEMPORAL_LOADS_MIB / fe_set_temporal_loads_mib; 0 = never).
// `min_bytes`: div and face-mass launches of fewer than about three rounds measured 3 - 8 % SLOWER with plain loads (div
// E = 4e4 ... 8e4, face-mass x 4 E = 2e4 ... 3e4; grad and the fused launches gain or stay level at every small size:
// profiles/r04/temporal_loads_ab.txt), so those families switch only above a measured floor.
std::atomic<long long> g_temporal_input_bytes{[] {
    const char* e = getenv("FEINSUM_TEMPORAL_LOADS_MIB");
    return e ? (long long)atoll(e) << 20 : fe::kTemporalInputBytes;
}()};
// `cap_mib`: up to which size of its inputs a family's launches gain from plain loads -- more than the cache holds for the
// families whose footprint is mostly inputs: what the cache holds of them is found again.  Measured (profiles/r05/
// temporal_loads_fused_ab.txt, back-to-back launches on the same arrays): grad gains up to 235 - 245 MiB and loses 1.5 ... 12 % from
// 245 - 270 on (kTemporalInputBytes = 248 MiB); div gains up to 270 MiB (-4 %), is level at 287 and loses 6 % at 304; div + grad gains
// up to 301 MiB (-2 %; 253 MiB: -8 %) and loses at 326; launches with face-mass in them -- face-mass alone, the wave operator --
// gain up to 316 - 325 MiB (the pipeline at E = 1e5, 307 MiB: 99.6 -> 94.8 us; face-mass x 4 at 1.5e5, 279 MiB: 80.7 -> 72.3) and lose
// 2 ... 4 % at 335 - 370.  The caller's setting (fe_set_temporal_loads_mib; 0 = never) scales every family's cap alike.
constexpr long long kTemporalCapGradMib = 248, kTemporalCapDivMib = 280, kTemporalCapGradDivMib = 310, kTemporalCapFaceMassMib = 320;
int temporal_flag(int64_t input_bytes, int64_t min_bytes = 0, long long cap_mib = kTemporalCapGradMib) {
    long long cap = g_temporal_input_bytes.load(std::memory_order_relaxed);
    if (cap >= (1ll << 40)) return fe::kOpLoadsTemporal;        // "always" (A/B runs)
    cap = cap / kTemporalCapGradMib * cap_mib;
    return input_bytes <= cap && input_bytes >= min_bytes ? fe::kOpLoadsTemporal : 0;
}
constexpr int64_t kTemporalFloorDiv = 80ll << 20, kTemporalFloorFaceMass = 64ll << 20;

// kOpStoresWriteThrough (fe_common.h) for a launch that writes `output_bytes` ($FEINSUM_WRITE_THROUGH_MIB /
// fe_set_write_through_mib; 0 = never)
std::atomic<long long> g_write_through_output_bytes{[] {
    const char* e = getenv("FEINSUM_WRITE_THROUGH_MIB");
    return e ? (long long)atoll(e) << 20 : fe::kWriteThroughOutputBytes;
}()};
int write_through_flag(int64_t output_bytes) {
    return output_bytes <= g_write_through_output_bytes.load(std::memory_order_relaxed) ? fe::kOpStoresWriteThrough : 0;
}

// short div launches (static walk) on the kernel whose B build is interleaved into the matrix phase (fe_div.h, kIlv): launches of
// at most so many tiles ($FEINSUM_DIV_INTERLEAVE_TILES / fe_set_div_interleave; 0 = never)
// Measured (profiles/r05/div_interleave_ab.txt, div_interleave_small.txt; same arrays, in-process): -3 ... -6 % at E = 8e4 ... 3e5 under
// either walk, -2 % at 4e5, level from 6e5 on (the launch is memory bound there) -- hence the default of 37 500 tiles (E = 6e5).
constexpr long long kDivInterleaveTiles = 37500;
std::atomic<long long> g_div_interleave_tiles{[] { const char* e = getenv("FEINSUM_DIV_INTERLEAVE_TILES"); return e ? atoll(e) : kDivInterleaveTiles; }()};

// ... and the tiles behind the last full round of such a launch as quarter tiles (fe_div.h, kOpQuarterTail): $FEINSUM_DIV_QUARTER_TAIL /
// fe_set_div_quarter_tail
std::atomic<int> g_div_quarter_tail{[] { const char* e = getenv("FEINSUM_DIV_QUARTER_TAIL"); return e ? atoi(e) : 1; }()};

// ... and of short grad launches (fe_grad.h; one field, static walk, one sub-tile per wave tile): $FEINSUM_GRAD_QUARTER_TAIL /
// fe_set_grad_quarter_tail.  The rule is the kernel's own (at least one full round, the ragged one at most an eighth full).
constexpr int kGradQuarterDen = 8;
std::atomic<int> g_grad_quarter_tail{[] { const char* e = getenv("FEINSUM_GRAD_QUARTER_TAIL"); return e ? atoi(e) : 1; }()};
int grad_quarter_flag(int m, int nb, int64_t nTiles, int64_t waves) {
    const int64_t ragged = nTiles % waves;
    const int setting = g_grad_quarter_tail.load(std::memory_order_relaxed);   // 0 off, 1 the default rule, n >= 4: ragged <= waves / n
    const int64_t den = setting >= 4 ? setting : kGradQuarterDen;
    return (m == 1 && nb == 1 && setting != 0 && nTiles > waves && ragged > 0 && den * ragged <= waves) ? fe::kOpQuarterTail : 0;
}

// ... and whether the odd CUs of every XCD start such a launch half a tile period late (fe_common.h, kOpStaggeredStart):
// $FEINSUM_GRAD_STAGGERED_START / fe_set_grad_staggered_start.  The rule: tetrahedra p = 4, one field, a full grid, at least 2.5
// rounds of tiles (the static walk ends at 4.5 rounds by itself).
std::atomic<int> g_grad_staggered_start{[] { const char* e = getenv("FEINSUM_GRAD_STAGGERED_START"); return e ? atoi(e) : 1; }()};
int grad_stagger_flag(int np, int m, int nb, int64_t nTiles, int64_t waves) {
    return (np == 35 && m == 1 && nb == 1 && g_grad_staggered_start.load(std::memory_order_relaxed) && 2 * nTiles >= 5 * waves) ? fe::kOpStaggeredStart : 0;
}

// the same flag for the eight-wave p = 5 kernels (compute bound at every size): $FEINSUM_PHASE_PRIORITY_P5 / fe_set_phase_priority_p5
std::atomic<int> g_phase_priority_p5{[] { const char* e = getenv("FEINSUM_PHASE_PRIORITY_P5"); return e ? atoi(e) : 0; }()};
int phase_priority_flag_p5() { return g_phase_priority_p5.load(std::memory_order_relaxed) ? fe::kOpPhasePriority : 0; }

// What the launcher decided for the MFMA launch enqueued last by this thread (fe_last_launch_info): bench.py and the tools report
// these instead of re-deriving the rules (round 4's report recomputed them in Python and could disagree with the kernel).
struct LastLaunch {
    int valid = 0, dynamic_walk = 0, temporal_loads = 0, write_through = 0, blocks = 0, waves_per_block = 0, kind = 0, bodies = 0;
    long long tiles = 0, static_tiles = 0;
};
thread_local LastLaunch g_last_launch;
void note_launch(bool dynamic, int flags, unsigned blocks, int waves_per_block, int64_t tiles, int64_t static_tiles, int kind = 0, int bodies = 1) {
    LastLaunch& L = g_last_launch;
    L.valid = 1;
    L.dynamic_walk = dynamic ? 1 : 0;
    L.temporal_loads = (flags & fe::kOpLoadsTemporal) ? 1 : 0;
    L.write_through = (flags & fe::kOpStoresWriteThrough) ? 1 : 0;
    L.blocks = (int)blocks;
    L.waves_per_block = waves_per_block;
    L.kind = kind;
    L.bodies = bodies;
    L.tiles = tiles;
    L.static_tiles = dynamic ? static_tiles : tiles;
}

// Persistent-style grid for the per-wave-tile kernels: 2 blocks of 4 waves per
// CU (their VGPR / LDS residency), fewer when there is less work.
// (a per-wave-equal grid -- every wave the same number of tiles, on fewer waves -- was measured in round 3 and is slower at
// every size: profiles/r03/balanced_grid_ab.txt)
unsigned persistent_grid(int64_t nTiles, int wavesPerBlock) {
    const int64_t blocks = (nTiles + wavesPerBlock - 1) / wavesPerBlock;
    const int64_t cap = (8 / wavesPerBlock) * (int64_t)device_cu_count();   // 8 waves per CU
    return (unsigned)(blocks < cap ? blocks : cap);
}

// grad of tetrahedra p = 5 (Np = 56): grad by components with the A fragments in LDS, one block per CU
int launch_grad_p5(const double* J, const double* D, const fe::FieldPtrs& P, int nb, int64_t E, int opT,
                   hipStream_t s, bool* launched, int dbg = 0) {
    using G = fe::DivGeom<56, 1, 4, 3, true, true>;   // eight waves per block, one block per CU
    const int64_t nTiles = E / G::TEL;
    *launched = nTiles > 0;   // the launch covers the elements behind the last tile too
    if (nTiles == 0) return FE_OK;
    static PerDeviceOnce once;
    const int attr_rc =
        once.run([] {
        return configure_kernel(fe::div3d_mfma_kernel<56, 1, 0, 4, 3, true, true>, "grad p5 (components, A in LDS)", G::LDS_BYTES,
                                G::THREADS, G::BLOCKS_PER_CU);
    });
    if (attr_rc != FE_OK) return attr_rc;
    opT |= phase_priority_flag_p5();
    const int64_t blocks = (nTiles + G::WAVES - 1) / G::WAVES, cap = device_cu_count();
#ifdef FE_EXPERIMENTS
#define FE_P5_CASE(DBG)                                                                                                   \
    case DBG: {                                                                                                           \
        static PerDeviceOnce once_dbg;                                                                                    \
        once_dbg.run([] {                                                                                                 \
            return configure_kernel(fe::div3d_mfma_kernel<56, 1, DBG, 4, 3, true, true>, "experiment", G::LDS_BYTES,      \
                                    G::THREADS, G::BLOCKS_PER_CU);                                                        \
        });                                                                                                               \
        hipLaunchKernelGGL((fe::div3d_mfma_kernel<56, 1, DBG, 4, 3, true, true>),                                         \
                           dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(G::THREADS), G::LDS_BYTES, s, J, D,        \
                           nullptr, P, nb, E, nTiles, opT, 0);                                                            \
        return FE_OK;                                                                                                     \
    }
    switch (dbg) {
        FE_P5_CASE(1) FE_P5_CASE(2) FE_P5_CASE(3) FE_P5_CASE(4) FE_P5_CASE(5) FE_P5_CASE(6) FE_P5_CASE(8) FE_P5_CASE(9) FE_P5_CASE(10) FE_P5_CASE(11) FE_P5_CASE(32) FE_P5_CASE(16) FE_P5_CASE(64) FE_P5_CASE(80)
        default: break;
    }
#undef FE_P5_CASE
#else
    (void)dbg;
#endif
    const unsigned grid = (unsigned)(blocks < cap ? blocks : cap);
    if (nb == 1) {   // behind two static rounds the tiles come by tickets (fe_common.h, dynamic walk)
        const int64_t t_static = tail_static_tiles(nTiles, grid, G::WAVES);
        unsigned* tail = t_static < nTiles ? tail_slot(s) : nullptr;
        if (tail) {
            static PerDeviceOnce once_tail;
            if (int rc = configured(once_tail, fe::grad_w8_tail_kernel<56>, "grad p5 (components, A in LDS), dynamic walk", G::LDS_BYTES,
                                    G::THREADS, G::BLOCKS_PER_CU))
                return rc;
            hipLaunchKernelGGL((fe::grad_w8_tail_kernel<56>), dim3(grid), dim3(G::THREADS), G::LDS_BYTES, s, J, D, P, nb, E, nTiles, opT, tail,
                               t_static);
            return FE_OK;
        }
    }
    hipLaunchKernelGGL((fe::div3d_mfma_kernel<56, 1, 0, 4, 3, true, true>), dim3(grid), dim3(G::THREADS), G::LDS_BYTES, s, J, D, nullptr,
                       P, nb, E, nTiles, opT, 0);
    return FE_OK;
}

// grad-type planes of tetrahedra p = 5 (MODE 5 of the div template): D u once per field, eight waves per block
int launch_gradplanes_p5(const fe::GradFields& Q, const double* D, int nb, int64_t E, int opT, hipStream_t s,
                         bool* launched) {
    using G = fe::DivGeom<56, 1, 5, 3, true, true>;
    const int64_t nTiles = E / G::TEL;
    *launched = nTiles > 0;   // the launch covers the elements behind the last tile too
    if (nTiles == 0) return FE_OK;
    static PerDeviceOnce once;
    const int attr_rc = once.run([] {
        return configure_kernel(fe::gradplanes_bycomp_kernel<56>, "grad planes p5 (components, A in LDS)", G::LDS_BYTES,
                                G::THREADS, G::BLOCKS_PER_CU);
    });
    if (attr_rc != FE_OK) return attr_rc;
    fe::FieldPtrs P = {};
    for (int k = 0; k < nb; ++k) P.v[k] = Q.u[k];
    const int64_t blocks = (nTiles + G::WAVES - 1) / G::WAVES, cap = device_cu_count();
    hipLaunchKernelGGL(fe::gradplanes_bycomp_kernel<56>, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(G::THREADS),
                       G::LDS_BYTES, s, Q, D, P, nb, E, nTiles, opT);
    return FE_OK;
}

// div of tetrahedra p = 5: the A fragments in LDS and the u planes streamed -- eight waves per block with one plane buffer each
// (round 4; FEINSUM_DIV_P5_WAVES=4: round 2's four waves with two buffers each, kept for the A/B)
int launch_div_p5(const double* J, const double* D, const fe::FieldPtrs& P, int nb, int64_t E, int opT,
                  hipStream_t s, bool* launched) {
    static const bool four_waves = [] { const char* e = getenv("FEINSUM_DIV_P5_WAVES"); return e && atoi(e) == 4; }();
    if (!four_waves) {
        using G = fe::DivGeom<56, 1, 0, 3, true, true>;
        const int64_t nTiles = E / G::TEL;
        *launched = nTiles > 0;   // the launch covers the elements behind the last tile too
        if (nTiles == 0) return FE_OK;
        static PerDeviceOnce once;
        const int attr_rc = once.run([] {
            return configure_kernel(fe::div3d_mfma_kernel<56, 1, 0, 0, 3, true, true>, "div p5 (A in LDS, planes streamed, eight waves)", G::LDS_BYTES,
                                    G::THREADS, G::BLOCKS_PER_CU);
        });
        if (attr_rc != FE_OK) return attr_rc;
        opT |= phase_priority_flag_p5();
        const int64_t blocks = (nTiles + G::WAVES - 1) / G::WAVES, cap = device_cu_count();
        const unsigned grid = (unsigned)(blocks < cap ? blocks : cap);
        if (nb == 1) {   // behind two static rounds the tiles come by tickets (fe_common.h, dynamic walk)
            const int64_t t_static = tail_static_tiles(nTiles, grid, G::WAVES);
            unsigned* tail = t_static < nTiles ? tail_slot(s) : nullptr;
            if (tail) {
                static PerDeviceOnce once_tail;
                if (int rc = configured(once_tail, fe::div_w8_tail_kernel<56>, "div p5 (A in LDS, planes streamed, eight waves), dynamic walk",
                                        G::LDS_BYTES, G::THREADS, G::BLOCKS_PER_CU))
                    return rc;
                hipLaunchKernelGGL((fe::div_w8_tail_kernel<56>), dim3(grid), dim3(G::THREADS), G::LDS_BYTES, s, J, D, P, nb, E, nTiles, opT, tail,
                                   t_static);
                return FE_OK;
            }
        }
        hipLaunchKernelGGL((fe::div3d_mfma_kernel<56, 1, 0, 0, 3, true, true>), dim3(grid), dim3(G::THREADS), G::LDS_BYTES, s, J, D, nullptr, P,
                           nb, E, nTiles, opT, 0);
        return FE_OK;
    }
    using G = fe::DivGeom<56, 1, 0, 3, true>;
    const int64_t nTiles = E / G::TEL;
    *launched = nTiles > 0;   // the launch covers the elements behind the last tile too
    if (nTiles == 0) return FE_OK;
    static PerDeviceOnce once;
    const int attr_rc = once.run([] {
        return configure_kernel(fe::div3d_mfma_kernel<56, 1, 0, 0, 3, true>, "div p5 (A in LDS, planes streamed)", G::LDS_BYTES, 256,
                                G::BLOCKS_PER_CU);
    });
    if (attr_rc != FE_OK) return attr_rc;
    const int64_t blocks = (nTiles + G::WAVES - 1) / G::WAVES, cap = device_cu_count();
    const unsigned grid = (unsigned)(blocks < cap ? blocks : cap);
    hipLaunchKernelGGL((fe::div3d_mfma_kernel<56, 1, 0, 0, 3, true>), dim3(grid), dim3(256), G::LDS_BYTES, s, J, D, nullptr, P, nb, E,
                       nTiles, opT, 0);
    return FE_OK;
}

// ---- the LDS-tiled VALU kernel (fe_tiled.h): any shape whose operator fits in LDS
constexpr int64_t kTiledMaxLds = fe::kTiledLdsBudget;

bool tiled_fits(fe::TiledArgs a) { return a.Np >= 1 && a.Np <= fe::kTiledThreads && fe::tiled_plan(a) <= kTiledMaxLds; }

template <typename T>
int launch_tiled_t(fe::TiledArgs a, hipStream_t s) {
    const int64_t lds = fe::tiled_plan(a, (int)sizeof(T));
    if (a.Np > fe::kTiledThreads || lds > kTiledMaxLds)
        return fail(FE_EUNSUPPORTED, "tiled kernel: operator and tile need %lld bytes of LDS (limit %lld)",
                    (long long)lds, (long long)kTiledMaxLds);
    static PerDeviceOnce once;
    const int attr_rc = once.run([] {
        const int rc = configure_kernel(fe::tiled_apply_kernel<8, T>, sizeof(T) == 8 ? "tiled<8>" : "tiled<8> float32", (int)kTiledMaxLds,
                                        fe::kTiledThreads, 1);
        return rc != FE_OK ? rc : configure_kernel(fe::tiled_apply_kernel<4, T>, sizeof(T) == 8 ? "tiled<4>" : "tiled<4> float32",
                                                   (int)kTiledMaxLds, fe::kTiledThreads, 1);
    });
    if (attr_rc != FE_OK) return attr_rc;
    const int64_t nTiles = (a.E + a.TE - 1) / a.TE;
    int64_t per_cu = kTiledMaxLds / lds;   // resident blocks per CU: LDS, and 2048 threads
    if (per_cu > 2048 / fe::kTiledThreads) per_cu = 2048 / fe::kTiledThreads;
    const int64_t cap = per_cu * device_cu_count();
    const dim3 grid((unsigned)(nTiles < cap ? nTiles : cap)), block(fe::kTiledThreads);
    if (a.EB == 8) hipLaunchKernelGGL((fe::tiled_apply_kernel<8, T>), grid, block, (size_t)lds, s, a);
    else hipLaunchKernelGGL((fe::tiled_apply_kernel<4, T>), grid, block, (size_t)lds, s, a);
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}
int launch_tiled(fe::TiledArgs a, hipStream_t s) { return launch_tiled_t<double>(a, s); }

// Which kernel class serves a call.  AUTO: MFMA where compiled, else the tiled kernel where the
// operator fits in LDS, else the plain generic kernels.
enum KernelPath { kPathGeneric, kPathMfma, kPathTiled };
int choose_path(int variant, bool mfma_ok, bool tiled_ok, const char* what, int Np, KernelPath* path) {
    if (variant == FE_VARIANT_GENERIC) { *path = kPathGeneric; return FE_OK; }
    if (variant == FE_VARIANT_TILED) {
        if (!tiled_ok)
            return fail(FE_EUNSUPPORTED, "%s: the tiled kernel does not serve this call (operator too large for LDS, or "
                        "separate output planes; Np=%d)", what, Np);
        *path = kPathTiled;
        return FE_OK;
    }
    if (variant == FE_VARIANT_MFMA && !mfma_ok)
        return fail(FE_EUNSUPPORTED, "%s: MFMA variant is not compiled for this shape (Np=%d)", what, Np);
    *path = mfma_ok ? kPathMfma : tiled_ok ? kPathTiled : kPathGeneric;
    return FE_OK;
}

fe::TiledArgs tiled_args(int family, const double* J, const double* A, const fe::FieldPtrs& P, int nb, int64_t E,
                         int ndim, int Np, int nf, int Nfp, int opT, int jlayout, int rlayout) {
    fe::TiledArgs a = {};
    a.J = J; a.A = A; a.P = P; a.E = E; a.family = family;
    a.ndim = ndim; a.Np = Np; a.nf = nf; a.Nfp = Nfp; a.nb = nb;
    a.opT = opT; a.jlayout = jlayout; a.rlayout = rlayout;
    return a;
}

template <int NP, int M>
int launch_grad(const fe::GradFields& P, bool plain, const double* D, const void* prep, int nb, int nx, int64_t E,
                int dbg, int opT, hipStream_t s, int64_t* e_done) {
    using G = fe::GradGeom<NP, M>;
    const int64_t nTiles = E / G::TEL;   // full wave tiles; the remainder goes to the generic kernel
    *e_done = nTiles > 0 ? E : 0;   // the launch covers the elements behind the last tile too (remainder_items)
    if (nTiles == 0) return FE_OK;
    opT |= temporal_flag((9 + (int64_t)nb * NP) * E * 8);
    static PerDeviceOnce once_plain, once_prepared, once_planes;
    char what[64];
    const void* gsec = prep ? static_cast<const char*>(prep) + fe::kPrepGradOff : nullptr;
    int attr_rc;
    if (!plain) {
        snprintf(what, sizeof(what), "grad planes Np=%d M=%d", NP, M);
        attr_rc = configured(once_planes, fe::grad3d_mfma_kernel<NP, M, 0, false>, what, G::LDS_BYTES, 256, 2);
    } else if (gsec) {
        snprintf(what, sizeof(what), "grad Np=%d M=%d, prepared operator", NP, M);
        attr_rc = configured(once_prepared, fe::grad3d_mfma_kernel<NP, M, 0, true, true>, what, G::LDS_BYTES, 256, 2);
    } else {
        snprintf(what, sizeof(what), "grad Np=%d M=%d", NP, M);
        attr_rc = configured(once_plain, fe::grad3d_mfma_kernel<NP, M, 0>, what, G::LDS_BYTES, 256, 2);
    }
#ifdef FE_EXPERIMENTS
    static PerDeviceOnce once_exp;
    once_exp.run([] {
        if (NP == 35) {
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 1>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 2>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 4>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 16>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 20>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 32>, "experiment", G::LDS_BYTES + fe::kDbgTileLdsBytes, 256, 1);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 64>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 96>, "experiment", G::LDS_BYTES + fe::kDbgTileLdsBytes, 256, 1);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 128>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 32, true, true>, "experiment", G::LDS_BYTES + fe::kDbgTileLdsBytes, 256, 1);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 0>, "grad (experiments build)", G::LDS_BYTES, 256, 2);
            configure_kernel(fe::grad3d_mfma_kernel<NP, M, 0, true, true>, "grad prepared (experiments build)", G::LDS_BYTES, 256, 2);
        }
        return FE_OK;
    });
#endif
    if (attr_rc != FE_OK) return attr_rc;
    const dim3 g(persistent_grid(nTiles, G::WAVES)), b(256);
    if (!plain) {   // general planes: per-plane geometry-factor and output pointers
        hipLaunchKernelGGL((fe::grad3d_mfma_kernel<NP, M, 0, false>), g, b, G::LDS_BYTES, s, P, D, nullptr, nb, nx, E,
                           nTiles, opT);
        return FE_OK;
    }
#ifdef FE_EXPERIMENTS
#define FE_GRAD_LDS(DBG) (G::LDS_BYTES + (((DBG) & 32) ? fe::kDbgTileLdsBytes : 0))   // (the per-tile stamps live behind the kernel's own LDS)
#else
#define FE_GRAD_LDS(DBG) G::LDS_BYTES
#endif
#define FE_GRAD_CASE(DBG) \
    hipLaunchKernelGGL((fe::grad3d_mfma_kernel<NP, M, DBG>), g, b, FE_GRAD_LDS(DBG), s, P, D, nullptr, nb, nx, E, nTiles, opT)
    switch (NP == 35 ? dbg : 0) {
#ifdef FE_EXPERIMENTS
        case 1: FE_GRAD_CASE(1); break;
        case 2: FE_GRAD_CASE(2); break;
        case 3: FE_GRAD_CASE(3); break;     // data movement in: loads only
        case 4: FE_GRAD_CASE(4); break;     // temporal stores
        case 8: FE_GRAD_CASE(8); break;     // no loads
        case 9: FE_GRAD_CASE(9); break;     // stores + stage 2
        case 10: FE_GRAD_CASE(10); break;   // arithmetic and LDS only
        case 11: FE_GRAD_CASE(11); break;   // stage 2 and LDS only
        case 16: FE_GRAD_CASE(16); break;   // temporal loads
        case 20: FE_GRAD_CASE(20); break;   // both
        case 32:
            if (gsec) {
                hipLaunchKernelGGL((fe::grad3d_mfma_kernel<NP, M, 32, true, true>), g, b, G::LDS_BYTES + fe::kDbgTileLdsBytes, s, P, D, gsec, nb, nx,
                                   E, nTiles, opT);
                break;
            }
            if (nb == 1) {   // per-wave time stamps of the dynamic walk
                const int64_t t_static = tail_static_tiles(nTiles, g.x, G::WAVES);
                unsigned* tail = t_static < nTiles ? tail_slot(s) : nullptr;
                if (tail) {
                    static PerDeviceOnce once_stamps;
                    once_stamps.run([] { return configure_kernel(fe::grad3d_mfma_tail_kernel<NP, M, 32>, "experiment", G::LDS_BYTES + fe::kDbgTileLdsBytes, 256, 1); });
                    hipLaunchKernelGGL((fe::grad3d_mfma_tail_kernel<NP, M, 32>), g, b, G::LDS_BYTES + fe::kDbgTileLdsBytes, s, P, D, nb, E, nTiles, opT, tail, t_static);
                    break;
                }
            }
            if (nb == 1) opT |= write_through_flag(3 * (int64_t)NP * E * 8) | grad_quarter_flag(M, nb, nTiles, (int64_t)g.x * G::WAVES);   // (as the product's static walk)
            hipLaunchKernelGGL((fe::grad3d_mfma_kernel<NP, M, 32>), g, b, G::LDS_BYTES + fe::kDbgTileLdsBytes, s, P, D, nullptr, nb, nx, E, nTiles, opT);
            break;
        case 64: FE_GRAD_CASE(64); break;
        case 96: FE_GRAD_CASE(96); break;
        case 128: FE_GRAD_CASE(128); break;
#endif
        default:
            if (gsec) {
                hipLaunchKernelGGL((fe::grad3d_mfma_kernel<NP, M, 0, true, true>), g, b, G::LDS_BYTES, s, P, D, gsec, nb, nx,
                                   E, nTiles, opT);
                break;
            }
            {
                {   // behind two static rounds the tiles come by tickets (fe_common.h: dynamic walk); any number of fields
                    const int64_t t_static = tail_static_tiles(nTiles, g.x, G::WAVES);
                    unsigned* tail = t_static < nTiles ? tail_slot(s) : nullptr;
                    if (tail) {
                        static PerDeviceOnce once_tail;
                        if (nb == 1) {
                            snprintf(what, sizeof(what), "grad Np=%d M=%d, dynamic walk", NP, M);
                            if (int rc = configured(once_tail, fe::grad3d_mfma_tail_kernel<NP, M>, what, G::LDS_BYTES, 256, 2)) return rc;
                            opT |= write_through_flag(3 * (int64_t)NP * E * 8);   // (short launches: the same rule as the static walk)
                            hipLaunchKernelGGL((fe::grad3d_mfma_tail_kernel<NP, M>), g, b, G::LDS_BYTES, s, P, D, nb, E, nTiles, opT, tail, t_static);
                            note_launch(true, opT, g.x, G::WAVES, nTiles, t_static);
                            break;
                        }
                        {   // b fields
                            static PerDeviceOnce once_tail_b;
                            snprintf(what, sizeof(what), "grad Np=%d M=%d x b, dynamic walk", NP, M);
                            if (int rc = configured(once_tail_b, fe::grad3d_mfma_tail_kernel<NP, M, 0, true>, what, G::LDS_BYTES, 256, 2)) return rc;
                            hipLaunchKernelGGL((fe::grad3d_mfma_tail_kernel<NP, M, 0, true>), g, b, G::LDS_BYTES, s, P, D, nb, E, nTiles, opT, tail,
                                               t_static);
                            note_launch(true, opT, g.x, G::WAVES, nTiles, t_static);
                            break;
                        }
                    }
                }
            }
            if (nb == 1)   // (static walk, one field: a short launch)
                opT |= write_through_flag(3 * (int64_t)NP * E * 8) | grad_quarter_flag(M, nb, nTiles, (int64_t)g.x * G::WAVES) |
                       grad_stagger_flag(NP, M, nb, nTiles, (int64_t)g.x * G::WAVES);
            FE_GRAD_CASE(0);
            note_launch(false, opT, g.x, G::WAVES, nTiles, nTiles, ((opT & fe::kOpQuarterTail) ? 8 : 0) | ((opT & fe::kOpStaggeredStart) ? 16 : 0));
            break;
    }
#undef FE_GRAD_CASE
#undef FE_GRAD_LDS
    return FE_OK;
}

template <int NP, int M>
int launch_div(const double* J, const double* D, const void* prep, const fe::FieldPtrs& P, int nb, int64_t E, int dbg,
               int opT, hipStream_t s, int64_t* e_done) {
    using G = fe::DivGeom<NP, M>;
    const int64_t nTiles = E / G::TEL;
    *e_done = nTiles > 0 ? E : 0;   // the launch covers the elements behind the last tile too (remainder_items)
    if (nTiles == 0) return FE_OK;
    opT |= temporal_flag((9 + 3 * (int64_t)nb * NP) * E * 8, kTemporalFloorDiv, kTemporalCapDivMib);
    static PerDeviceOnce once_plain, once_prepared;
    char what[64];
    int attr_rc;
    if (prep) {
        snprintf(what, sizeof(what), "div Np=%d M=%d, prepared operator", NP, M);
        attr_rc = configured(once_prepared, fe::div3d_mfma_kernel<NP, M, 0, 0, 3, false, false, true>, what, G::LDS_BYTES, 256,
                             G::BLOCKS_PER_CU);
    } else {
        snprintf(what, sizeof(what), "div Np=%d M=%d", NP, M);
        attr_rc = configured(once_plain, fe::div3d_mfma_kernel<NP, M, 0>, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU);
    }
#ifdef FE_EXPERIMENTS
    static PerDeviceOnce once_exp;
    once_exp.run([] {
        if (NP == 35) {
            configure_kernel(fe::div3d_mfma_kernel<NP, M, 1>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::div3d_mfma_kernel<NP, M, 2>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::div3d_mfma_kernel<NP, M, 3>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::div3d_mfma_kernel<NP, M, 8>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::div3d_mfma_kernel<NP, M, 32>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::div3d_mfma_kernel<NP, M, 64>, "experiment", G::LDS_BYTES, 256, 1);
            configure_kernel(fe::div3d_mfma_kernel<NP, M, 128>, "experiment", G::LDS_BYTES + fe::kDbgTileLdsBytes, 256, 1);
        }
        return FE_OK;
    });
#endif
    if (attr_rc != FE_OK) return attr_rc;
    const dim3 g(persistent_grid(nTiles, G::WAVES)), b(256);
#define FE_DIV_CASE(DBG) \
    hipLaunchKernelGGL((fe::div3d_mfma_kernel<NP, M, DBG>), g, b, G::LDS_BYTES, s, J, D, nullptr, P, nb, E, nTiles, opT, 0)
    switch (NP == 35 ? dbg : 0) {
#ifdef FE_EXPERIMENTS
        case 1: FE_DIV_CASE(1); break;
        case 2: FE_DIV_CASE(2); break;
        case 3: FE_DIV_CASE(3); break;
        case 8: FE_DIV_CASE(8); break;
        case 32: FE_DIV_CASE(32); break;   // one u plane loaded instead of three (timing only)
        case 64: FE_DIV_CASE(64); break;   // the tile after next touched line by line (L2 prefetch)
        case 128:                          // per-wave, per-tile time stamps (fe_dbg_tile); FE_DIV_ILV=1: of the interleaved kernel
            if constexpr (NP == 35 && M == 1) {
                if (getenv("FE_DIV_ILV")) {
                    static PerDeviceOnce once_ilv_dbg;
                    once_ilv_dbg.run([] { return configure_kernel(fe::div3d_mfma_ilv_kernel<NP, 128>, "experiment", G::LDS_BYTES + fe::kDbgTileLdsBytes, 256, 1); });
                    hipLaunchKernelGGL((fe::div3d_mfma_ilv_kernel<NP, 128>), g, b, G::LDS_BYTES + fe::kDbgTileLdsBytes, s, J, D, P, nb, E, nTiles, opT);
                    break;
                }
            }
            hipLaunchKernelGGL((fe::div3d_mfma_kernel<NP, M, 128>), g, b, G::LDS_BYTES + fe::kDbgTileLdsBytes, s, J, D, nullptr, P, nb, E, nTiles, opT, 0);
            break;
#endif
        default:
            if (prep) {
                hipLaunchKernelGGL((fe::div3d_mfma_kernel<NP, M, 0, 0, 3, false, false, true>), g, b, G::LDS_BYTES, s, J, D,
                                   prep, P, nb, E, nTiles, opT, 0);
                break;
            }
            if constexpr (NP == 35 && M == 1) {
                if (!(opT & fe::kDivWalkSplit) && nTiles <= g_div_interleave_tiles.load(std::memory_order_relaxed) &&
                    tail_static_tiles(nTiles, g.x, G::WAVES) == nTiles) {   // a short launch (static walk): the interleaved form
                    static PerDeviceOnce once_ilv;
                    if (int rc = configured(once_ilv, fe::div3d_mfma_ilv_kernel<NP>, "div Np=35, B build interleaved", G::LDS_BYTES, 256, G::BLOCKS_PER_CU)) return rc;
                    const int64_t waves = (int64_t)g.x * G::WAVES, ragged = nTiles % waves;   // (the kernel's own rule: fe_div.h)
                    const int qflag = (nb == 1 && ragged > 0 && 8 * ragged <= waves && g_div_quarter_tail.load(std::memory_order_relaxed)) ? fe::kOpQuarterTail : 0;
                    hipLaunchKernelGGL((fe::div3d_mfma_ilv_kernel<NP>), g, b, G::LDS_BYTES, s, J, D, P, nb, E, nTiles, opT | qflag);
                    note_launch(false, opT & ~fe::kOpStoresWriteThrough, g.x, G::WAVES, nTiles, nTiles, 4 | (qflag ? 8 : 0));
                    break;
                }
            }
            {
                if (!(opT & fe::kDivWalkSplit)) {   // behind two static rounds the tiles come by tickets (fe_common.h); any number of fields
                    const int64_t t_static = tail_static_tiles(nTiles, g.x, G::WAVES);
                    unsigned* tail = t_static < nTiles ? tail_slot(s) : nullptr;
                    if (tail) {
                        static PerDeviceOnce once_tail;
                        if constexpr (NP == 35 && M == 1) {
                            if (nb == 1 && nTiles <= g_div_interleave_tiles.load(std::memory_order_relaxed)) {   // the interleaved form, dynamic walk
                                static PerDeviceOnce once_tail_ilv;
                                if (int rc = configured(once_tail_ilv, fe::div3d_mfma_tail_kernel<NP, M, false, true>, "div Np=35, B build interleaved, dynamic walk",
                                                        G::LDS_BYTES, 256, G::BLOCKS_PER_CU))
                                    return rc;
                                hipLaunchKernelGGL((fe::div3d_mfma_tail_kernel<NP, M, false, true>), g, b, G::LDS_BYTES, s, J, D, P, nb, E, nTiles, opT, tail, t_static);
                                note_launch(true, opT, g.x, G::WAVES, nTiles, t_static, 4);
                                break;
                            }
                        }
                        if (nb == 1) {
                            snprintf(what, sizeof(what), "div Np=%d M=%d, dynamic walk", NP, M);
                            if (int rc = configured(once_tail, fe::div3d_mfma_tail_kernel<NP, M>, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU)) return rc;
                            hipLaunchKernelGGL((fe::div3d_mfma_tail_kernel<NP, M>), g, b, G::LDS_BYTES, s, J, D, P, nb, E, nTiles, opT, tail, t_static);
                            note_launch(true, opT, g.x, G::WAVES, nTiles, t_static);
                            break;
                        }
                        {   // b fields
                            static PerDeviceOnce once_tail_b;
                            snprintf(what, sizeof(what), "div Np=%d M=%d x b, dynamic walk", NP, M);
                            if (int rc = configured(once_tail_b, fe::div3d_mfma_tail_kernel<NP, M, true>, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU))
                                return rc;
                            hipLaunchKernelGGL((fe::div3d_mfma_tail_kernel<NP, M, true>), g, b, G::LDS_BYTES, s, J, D, P, nb, E, nTiles, opT, tail,
                                               t_static);
                            note_launch(true, opT, g.x, G::WAVES, nTiles, t_static);
                            break;
                        }
                    }
                }
            }
            FE_DIV_CASE(0);
            note_launch(false, opT & ~fe::kOpStoresWriteThrough, g.x, G::WAVES, nTiles, nTiles);
            break;
    }
#undef FE_DIV_CASE
    return FE_OK;
}

template <int NP, int M, bool ALDS = false>
int launch_divcomp(const double* J, const double* D, const double* u, double* out, int64_t E,
                   int opT, int jes, hipStream_t s, int64_t* e_done) {
    using G = fe::DivGeom<NP, M, 1, 3, ALDS>;
    const int64_t nTiles = E / G::TEL;
    *e_done = nTiles > 0 ? E : 0;   // the launch covers the elements behind the last tile too (remainder_items)
    if (nTiles == 0) return FE_OK;
    static PerDeviceOnce once;
    const int attr_rc =
        once.run([] {
            char what[64];
            snprintf(what, sizeof(what), "div component Np=%d M=%d", NP, M);
            return configure_kernel(fe::div3d_mfma_kernel<NP, M, 0, 1, 3, ALDS>, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU);
        });
    if (attr_rc != FE_OK) return attr_rc;
    fe::FieldPtrs P = {};
    P.v[0] = u;
    P.out[0] = out;
    unsigned grid = persistent_grid(nTiles, G::WAVES);
    if (G::BLOCKS_PER_CU == 1 && grid > (unsigned)device_cu_count()) grid = (unsigned)device_cu_count();
    hipLaunchKernelGGL((fe::div3d_mfma_kernel<NP, M, 0, 1, 3, ALDS>), dim3(grid), dim3(256), G::LDS_BYTES, s, J, D,
                       nullptr, P, 1, E, nTiles, opT, jes);
    return FE_OK;
}

// 'e,ij,ej->ei' (MODE 2) / 'ij,ej->ei' (MODE 3): the one-component instances of the div template
template <int NP, int M, int MODE>
int launch_matapply_mode(const double* J, const double* D, const fe::FieldPtrs& P, int nb, int64_t E, int opT,
                         hipStream_t s, int64_t* e_done) {
    using G = fe::DivGeom<NP, M, MODE>;
    const int64_t nTiles = E / G::TEL;
    *e_done = nTiles > 0 ? E : 0;   // the launch covers the elements behind the last tile too (remainder_items)
    if (nTiles == 0) return FE_OK;
    static PerDeviceOnce once;
    const int attr_rc = once.run([] {
        char what[64];
        snprintf(what, sizeof(what), "matapply Np=%d M=%d mode %d", NP, M, MODE);
        return configure_kernel(fe::div3d_mfma_kernel<NP, M, 0, MODE>, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU);
    });
    if (attr_rc != FE_OK) return attr_rc;
    hipLaunchKernelGGL((fe::div3d_mfma_kernel<NP, M, 0, MODE>), dim3(persistent_grid(nTiles, G::WAVES)), dim3(256),
                       G::LDS_BYTES, s, J, D, nullptr, P, nb, E, nTiles, opT, 0);
    return FE_OK;
}

template <int NP, int M>
int launch_matapply(const double* J, const double* D, const fe::FieldPtrs& P, int nb, int64_t E, int opT,
                    hipStream_t s, int64_t* e_done) {
    return J ? launch_matapply_mode<NP, M, 2>(J, D, P, nb, E, opT, s, e_done)
             : launch_matapply_mode<NP, M, 3>(J, D, P, nb, E, opT, s, e_done);
}

template <int NP, int NFP, int M, int NB, int NF = fe::kFmNf, bool ALDS = false>
int launch_fm_nb(const double* J, const double* R, const void* prep, const fe::FieldPtrs& P, int64_t E, int64_t nTiles,
                 int jfe, int rifj, hipStream_t s) {
    constexpr bool W8 = ALDS;   // fragments in LDS: eight waves per block share them, one block per CU
    constexpr bool kCanPrep = !ALDS && NF == fe::kFmNf;   // prepared operators: tetrahedra p = 1..4
    using G = fe::FmGeom<NP, NFP, M, NF, ALDS, W8>;
    jfe = (jfe ? 1 : 0) | temporal_flag((NF + (int64_t)NB * NF * NFP) * E * 8, kTemporalFloorFaceMass, kTemporalCapFaceMassMib);
    static PerDeviceOnce once_plain, once_prepared;
    char what[96];
    int attr_rc = FE_OK;
    bool use_prep = false;
    if constexpr (kCanPrep) use_prep = prep != nullptr;
    if (use_prep) {
        if constexpr (kCanPrep) {
            snprintf(what, sizeof(what), "face-mass Np=%d Nfp=%d M=%d b=%d, prepared operator", NP, NFP, M, NB);
            attr_rc = configured(once_prepared, fe::facemass_mfma_kernel<NP, NFP, M, NB, NF, false, false, true>, what, G::LDS_BYTES,
                                 G::THREADS, G::BLOCKS_PER_CU);
        }
    } else {
        snprintf(what, sizeof(what), "face-mass Np=%d Nfp=%d nf=%d M=%d b=%d", NP, NFP, NF, M, NB);
        attr_rc = configured(once_plain, fe::facemass_mfma_kernel<NP, NFP, M, NB, NF, ALDS, W8>, what, G::LDS_BYTES, G::THREADS,
                             G::BLOCKS_PER_CU);
    }
    if (attr_rc != FE_OK) return attr_rc;
    int64_t blocks = (nTiles + G::WAVES - 1) / G::WAVES;
    const int64_t cap = (int64_t)G::BLOCKS_PER_CU * device_cu_count();
    if (blocks > cap) blocks = cap;
    if constexpr (kCanPrep) {
        if (prep) {
            hipLaunchKernelGGL((fe::facemass_mfma_kernel<NP, NFP, M, NB, NF, false, false, true>), dim3((unsigned)blocks),
                               dim3(G::THREADS), G::LDS_BYTES, s, J, R, prep, P, E, nTiles, jfe, rifj);
            return FE_OK;
        }
    }
    if constexpr (NP == 56 && NFP == 21 && M == 1 && NF == fe::kFmNf && ALDS && NB == 4) {
        const int64_t t_static = tail_static_tiles(nTiles, blocks, G::WAVES);
        unsigned* tail = t_static < nTiles ? tail_slot(s) : nullptr;
        if (tail) {
            static PerDeviceOnce once_tail;
            snprintf(what, sizeof(what), "face-mass Np=56 b=%d (A in LDS), dynamic walk", NB);
            if (int rc = configured(once_tail, fe::facemass_w8_tail_kernel<NP, NFP, NB>, what, G::LDS_BYTES, G::THREADS, G::BLOCKS_PER_CU))
                return rc;
            hipLaunchKernelGGL((fe::facemass_w8_tail_kernel<NP, NFP, NB>), dim3((unsigned)blocks), dim3(G::THREADS), G::LDS_BYTES, s, J, R, P,
                               E, nTiles, jfe, rifj, tail, t_static);
            return FE_OK;
        }
    }
    if constexpr (!ALDS && NB >= 3) {   // tetrahedra p = 1 .. 4 and triangles, three or more fields
        // behind two static rounds the tiles come by tickets (fe_common.h, dynamic walk)
        const int64_t t_static = tail_static_tiles(nTiles, blocks, G::WAVES, false, NP == 35 && NF == fe::kFmNf && NB == 4);
        unsigned* tail = t_static < nTiles ? tail_slot(s) : nullptr;
        if (tail) {
            static PerDeviceOnce once_tail;
            snprintf(what, sizeof(what), "face-mass Np=%d nf=%d M=%d b=%d, dynamic walk", NP, NF, M, NB);
            if (int rc = configured(once_tail, fe::facemass_mfma_tail_kernel<NP, NFP, M, NB, NF>, what, G::LDS_BYTES, G::THREADS, G::BLOCKS_PER_CU))
                return rc;
            hipLaunchKernelGGL((fe::facemass_mfma_tail_kernel<NP, NFP, M, NB, NF>), dim3((unsigned)blocks), dim3(G::THREADS), G::LDS_BYTES, s,
                               J, R, P, E, nTiles, jfe, rifj, tail, t_static);
            note_launch(true, jfe, (unsigned)blocks, G::WAVES, nTiles, t_static);
            return FE_OK;
        }
    }
    hipLaunchKernelGGL((fe::facemass_mfma_kernel<NP, NFP, M, NB, NF, ALDS, W8>), dim3((unsigned)blocks), dim3(G::THREADS),
                       G::LDS_BYTES, s, J, R, nullptr, P, E, nTiles, jfe, rifj);
    note_launch(false, jfe & ~fe::kOpStoresWriteThrough, (unsigned)blocks, G::WAVES, nTiles, nTiles);
    return FE_OK;
}

// One MFMA launch for a group of nb fields (2 <= nb <= kMaxGroup of the geometry).
template <int NP, int NFP, int M, int NF = fe::kFmNf, bool ALDS = false>
int launch_fm(const double* J, const double* R, const void* prep, const fe::FieldPtrs& P, int nb, int64_t E,
              int64_t nTiles, int jfe, int rifj, hipStream_t s) {
    switch (nb) {
        case 2: return launch_fm_nb<NP, NFP, M, 2, NF, ALDS>(J, R, prep, P, E, nTiles, jfe, rifj, s);
        case 3: return launch_fm_nb<NP, NFP, M, 3, NF, ALDS>(J, R, prep, P, E, nTiles, jfe, rifj, s);
        case 4: return launch_fm_nb<NP, NFP, M, 4, NF, ALDS>(J, R, prep, P, E, nTiles, jfe, rifj, s);
        default: break;
    }
    if constexpr (NP == 35 && NF == fe::kFmNf) {   // p = 4, the headline order: groups of up to 8 fields
        switch (nb) {
            case 5: return launch_fm_nb<NP, NFP, M, 5>(J, R, prep, P, E, nTiles, jfe, rifj, s);
            case 6: return launch_fm_nb<NP, NFP, M, 6>(J, R, prep, P, E, nTiles, jfe, rifj, s);
            case 7: return launch_fm_nb<NP, NFP, M, 7>(J, R, prep, P, E, nTiles, jfe, rifj, s);
            case 8: return launch_fm_nb<NP, NFP, M, 8>(J, R, prep, P, E, nTiles, jfe, rifj, s);
            default: break;
        }
    }
    return fail(FE_EINVAL, "face-mass: internal field grouping error (nb=%d)", nb);
}

// 'xre,rij,ej->xei' operands as grad-type planes: j[x] = J[x], out[k][x] = out_k[x]
fe::GradFields grad_fields(const double* J, const double* const* u, double* const* out, int nb, int64_t E,
                           int Np) {
    fe::GradFields P = {};
    for (int x = 0; x < 3; ++x) P.j[x] = J + (int64_t)x * 3 * E;
    for (int k = 0; k < nb; ++k) {
        P.u[k] = u[k];
        for (int x = 0; x < 3; ++x) P.out[k][x] = out[k] + (int64_t)x * E * Np;
    }
    return P;
}

// MFMA launch over the full tiles + generic kernels for the rest.  Jfull != nullptr: the planes
// are those of a plain grad (one [3][3][E] array, [3][E][Np] outputs).
int grad_fields_launch(const fe::GradFields& P, const double* Jfull, const double* D, const void* prep, int nb, int nx,
                       int64_t E, int Np, int opT, int variant, hipStream_t s) {
    const bool mfma_ok = Np == 35 || Np == 20 || Np == 10 || Np == 4 || (Np == 56 && Jfull);
    fe::FieldPtrs Pt = {};
    for (int k = 0; k < nb; ++k) { Pt.v[k] = P.u[k]; Pt.out[k] = P.out[k][0]; }
    const fe::TiledArgs ta = tiled_args(FE_FAMILY_GRAD, Jfull, D, Pt, nb, E, 3, Np, 0, 0, opT, 0, 0);
    KernelPath path;
    if (int rc = choose_path(variant >= 1000 ? FE_VARIANT_MFMA : variant, mfma_ok, Jfull && tiled_fits(ta), "grad", Np,
                             &path))
        return rc;
    if (path == kPathTiled) return launch_tiled(ta, s);
    if (path == kPathMfma && Np == 56) {   // p = 5: too many A fragments for the row-permuted kernel
        bool launched = false;
        int dbg5 = 0;
#ifdef FE_EXPERIMENTS
        if (variant >= 1000) dbg5 = (variant - 1000) & 127;   // experiment flags, see fe_div.h
#endif
        if (int rc = launch_grad_p5(Jfull, D, Pt, nb, E, opT, s, &launched, dbg5)) return rc;
        if (launched) {
            FE_HIP_CHECK(hipGetLastError());
            return FE_OK;
        }
        return tiled_fits(ta) ? launch_tiled(ta, s) : fail(FE_EUNSUPPORTED, "grad: no kernel for this size");
    }
    int64_t e_done = 0;
    if (path == kPathMfma) {
        int dbg = 0;
#ifdef FE_EXPERIMENTS
        if (variant >= 1000) dbg = (variant - 1000) & 255;   // experiment flags, see fe_grad.h
#endif
        int rc = FE_OK;
        switch (Np) {   // wave tile = 16 M elements
            case 35: rc = launch_grad<35, 1>(P, Jfull != nullptr, D, prep, nb, nx, E, dbg, opT, s, &e_done); break;
            case 20: rc = launch_grad<20, 2>(P, Jfull != nullptr, D, prep, nb, nx, E, dbg, opT, s, &e_done); break;
            case 10: rc = launch_grad<10, 3>(P, Jfull != nullptr, D, prep, nb, nx, E, dbg, opT, s, &e_done); break;
            default: rc = launch_grad<4, 5>(P, Jfull != nullptr, D, prep, nb, nx, E, dbg, opT, s, &e_done); break;
        }
        if (rc != FE_OK) return rc;
    }
    if (e_done < E) {
        const dim3 grid(generic_grid(E - e_done, Np)), block(256);
        for (int k = 0; k < nb; ++k) {
            if (Jfull) {
                hipLaunchKernelGGL(fe::grad3d_generic_kernel, grid, block, 0, s, Jfull, D, P.u[k], P.out[k][0], E,
                                   Np, e_done, opT);
                continue;
            }
            for (int x = 0; x < 3; ++x)
                if (P.out[k][x])
                    hipLaunchKernelGGL(fe::divcomp3d_generic_kernel, grid, block, 0, s, P.j[x], D, P.u[k],
                                       P.out[k][x], E, Np, e_done, opT, 0);
        }
    }
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

// body order of the fused launches (fe_fused.h): chosen by measurement (tools/fused_order_ab.py)
constexpr int kFusedOrderGradDiv = 1;   // the younger half of the grid runs grad first: -1.5 ... -3.7 % in every placement
constexpr int kFusedOrderWaveOp = 3;    // the younger half runs grad, div, lift (rotating all three bodies by thirds of the grid cost time)

// div then grad in one persistent launch (full tiles only; e_done_* report what was covered)
template <int NP, int MG, int MD>
int launch_graddiv(const double* J, const double* D, const void* prep, const fe::GradFields& Pg,
                   const fe::FieldPtrs& Pd, int64_t E, hipStream_t s, int64_t* e_done_g, int64_t* e_done_d) {
    using GG = fe::GradGeom<NP, MG>;
    using GD = fe::DivGeom<NP, MD>;
    using G = fe::GradDivGeom<NP, MG, MD>;
    const int64_t nTilesG = E / GG::TEL, nTilesD = E / GD::TEL;
    *e_done_g = *e_done_d = (nTilesG > 0 || nTilesD > 0) ? E : 0;   // remainders included
    if (nTilesG == 0 && nTilesD == 0) return FE_OK;
    static PerDeviceOnce once_plain, once_prepared;
    char what[64];
    int attr_rc;
    constexpr bool kDyn = NP == 35 && MG == 1 && MD == 1;   // bodies with a dynamic walk (fe_common.h)
    if (prep) {
        snprintf(what, sizeof(what), "div + grad Np=%d, prepared operator", NP);
        attr_rc = configured(once_prepared, fe::graddiv3d_mfma_kernel<NP, MG, MD, true>, what, G::LDS_BYTES, 256, 2);
    } else {
        snprintf(what, sizeof(what), "div + grad Np=%d", NP);
        attr_rc = configured(once_plain, fe::graddiv3d_mfma_kernel<NP, MG, MD, false, kDyn>, what, G::LDS_BYTES, 256, 2);
    }
    if (attr_rc != FE_OK) return attr_rc;
    const int64_t nTiles = nTilesG > nTilesD ? nTilesG : nTilesD;
    const unsigned grid = persistent_grid(nTiles, 4);
    fe::FusedTail ft = {nullptr, nTilesD, nTilesG, 0};
    if (kDyn && !prep) {
        ft.static_d = tail_static_tiles(nTilesD, grid, 4, true);
        ft.static_g = tail_static_tiles(nTilesG, grid, 4, true);
        if (ft.static_d < nTilesD || ft.static_g < nTilesG) ft.tail = tail_slot(s, 2);
    }
    // body order (fe_fused.h): with the static walk the younger half of the grid runs grad first; with tickets every block
    // runs div, then grad (profiles/r03/dynamic_walk_fused.txt: 77.9 - 78.1 against 76.9 - 77.6 %)
    int op_arg = ((ft.tail ? 0 : kFusedOrderGradDiv) << 8) | temporal_flag((9 + 4 * (int64_t)NP) * E * 8, 0, kTemporalCapGradDivMib);
#ifdef FE_EXPERIMENTS
    if (const char* o = getenv("FE_FUSED_ORDER")) op_arg = atoi(o) << 8;
#endif
    if (prep)
        hipLaunchKernelGGL((fe::graddiv3d_mfma_kernel<NP, MG, MD, true>), dim3(grid), dim3(256), G::LDS_BYTES, s, J, D, prep,
                           Pg, Pd, E, nTilesG, nTilesD, op_arg, ft);
    else
        hipLaunchKernelGGL((fe::graddiv3d_mfma_kernel<NP, MG, MD, false, kDyn>), dim3(grid), dim3(256), G::LDS_BYTES, s, J, D, nullptr,
                           Pg, Pd, E, nTilesG, nTilesD, op_arg, ft);
    note_launch(ft.tail != nullptr, op_arg & fe::kOpLoadsTemporal, grid, 4, nTilesG + nTilesD, ft.static_d + ft.static_g, 0, 2);
    return FE_OK;
}

// div, grad and face-mass x nb (2..4) in one persistent launch (full tiles only)
template <int NP, int NFP, int MG, int MD, int MF, int NB>
int launch_waveop_nb(const fe::WaveOpArgs& a, const fe::GradFields& Pg, const fe::FieldPtrs& Pd,
                     const fe::FieldPtrs& Pf, hipStream_t s) {
    using G = fe::WaveOpGeom<NP, NFP, MG, MD, MF>;
    constexpr bool kDyn = NP == 35 && NFP == 15 && MG == 1 && MD == 1 && MF == 1;   // bodies with a dynamic walk (fe_common.h)
    static PerDeviceOnce once_plain, once_prepared;
    char what[80];
    int attr_rc;
    if (a.prepD && a.prepR) {
        snprintf(what, sizeof(what), "div + grad + face-mass Np=%d b=%d, prepared operators", NP, NB);
        attr_rc = configured(once_prepared, fe::waveop3d_mfma_kernel<NP, NFP, MG, MD, MF, NB, true>, what, G::LDS_BYTES, 256, 2);
    } else {
        snprintf(what, sizeof(what), "div + grad + face-mass Np=%d b=%d", NP, NB);
        attr_rc = configured(once_plain, fe::waveop3d_mfma_kernel<NP, NFP, MG, MD, MF, NB, false, kDyn>, what, G::LDS_BYTES, 256, 2);
    }
    if (attr_rc != FE_OK) return attr_rc;
    int64_t nTiles = a.nTilesG > a.nTilesD ? a.nTilesG : a.nTilesD;
    if (a.nTilesF > nTiles) nTiles = a.nTilesF;
    const unsigned grid = persistent_grid(nTiles, 4);
    fe::FusedTail ft = {nullptr, a.nTilesD, a.nTilesG, a.nTilesF};
    if (kDyn && !(a.prepD && a.prepR)) {
        ft.static_d = tail_static_tiles(a.nTilesD, grid, 4, true);
        ft.static_g = tail_static_tiles(a.nTilesG, grid, 4, true);
        ft.static_f = NB >= 3 ? tail_static_tiles(a.nTilesF, grid, 4, true) : a.nTilesF;
        if (ft.static_d < a.nTilesD || ft.static_g < a.nTilesG || ft.static_f < a.nTilesF) ft.tail = tail_slot(s, 3);
    }
    fe::WaveOpArgs args = a;
    args.load_flags = temporal_flag((9 + 4 * (int64_t)NP + 4 + (int64_t)NB * 4 * NFP) * a.E * 8, 0, kTemporalCapFaceMassMib);
    if (ft.tail) args.order = 0;   // with tickets every block runs div, grad, lift (see launch_graddiv)
#ifdef FE_EXPERIMENTS
    if (const char* o = getenv("FE_FUSED_ORDER")) args.order = atoi(o);
#endif
    if (a.prepD && a.prepR)
        hipLaunchKernelGGL((fe::waveop3d_mfma_kernel<NP, NFP, MG, MD, MF, NB, true>), dim3(grid), dim3(256), G::LDS_BYTES, s, args, Pg,
                           Pd, Pf, ft);
    else
        hipLaunchKernelGGL((fe::waveop3d_mfma_kernel<NP, NFP, MG, MD, MF, NB, false, kDyn>), dim3(grid), dim3(256), G::LDS_BYTES, s,
                           args, Pg, Pd, Pf, ft);
    note_launch(ft.tail != nullptr, args.load_flags & fe::kOpLoadsTemporal, grid, 4, a.nTilesG + a.nTilesD + a.nTilesF,
                ft.static_d + ft.static_g + ft.static_f, 0, 3);
    return FE_OK;
}

template <int NP, int NFP, int MG, int MD, int MF>
int launch_waveop(fe::WaveOpArgs a, const fe::GradFields& Pg, const fe::FieldPtrs& Pd, const fe::FieldPtrs& Pf,
                  int nb, hipStream_t s, bool* launched) {
    a.nTilesG = a.E / (16 * MG);
    a.nTilesD = a.E / (16 * MD);
    a.nTilesF = a.E / (16 * MF);
    a.order = kFusedOrderWaveOp;
#ifdef FE_EXPERIMENTS
    if (const char* o = getenv("FE_FUSED_ORDER")) a.order = atoi(o);
#endif
    *launched = a.nTilesG > 0 || a.nTilesD > 0 || a.nTilesF > 0;   // remainders included
    if (!*launched) return FE_OK;
    switch (nb) {
        case 2: return launch_waveop_nb<NP, NFP, MG, MD, MF, 2>(a, Pg, Pd, Pf, s);
        case 3: return launch_waveop_nb<NP, NFP, MG, MD, MF, 3>(a, Pg, Pd, Pf, s);
        case 4: return launch_waveop_nb<NP, NFP, MG, MD, MF, 4>(a, Pg, Pd, Pf, s);
        default: return fail(FE_EINVAL, "waveop: internal field count error (nb=%d)", nb);
    }
}

// triangles (ND = 2): grad by components (MODE 4) and div (MODE 0) instances of the div template
template <int NP, int M, int MODE>
int launch_nd2(const double* J, const double* D, const fe::FieldPtrs& P, int nb, int64_t E, int opT,
               hipStream_t s, bool* launched, int jes = 0) {
    using G = fe::DivGeom<NP, M, MODE, 2>;
    const int64_t nTiles = E / G::TEL;
    *launched = nTiles > 0;   // the launch covers the elements behind the last tile too
    if (nTiles == 0) return FE_OK;
    static PerDeviceOnce once;
    const int attr_rc = once.run([] {
        char what[64];
        snprintf(what, sizeof(what), "triangles %s Np=%d M=%d", MODE == 4 ? "grad" : MODE == 1 ? "div component" : "div", NP, M);
        return configure_kernel(fe::div3d_mfma_kernel<NP, M, 0, MODE, 2>, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU);
    });
    if (attr_rc != FE_OK) return attr_rc;
    const unsigned grid = persistent_grid(nTiles, G::WAVES);
    if constexpr (MODE == 0 || MODE == 4) {   // behind two static rounds the tiles come by tickets (fe_common.h, dynamic walk)
        const int64_t t_static = tail_static_tiles(nTiles, grid, G::WAVES);
        unsigned* tail = t_static < nTiles ? tail_slot(s) : nullptr;
        if (tail) {
            static PerDeviceOnce once_tail;
            char what[64];
            snprintf(what, sizeof(what), "triangles %s Np=%d M=%d, dynamic walk", MODE == 4 ? "grad" : "div", NP, M);
            if (int rc = configured(once_tail, fe::nd2_mfma_tail_kernel<NP, M, MODE>, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU)) return rc;
            hipLaunchKernelGGL((fe::nd2_mfma_tail_kernel<NP, M, MODE>), dim3(grid), dim3(256), G::LDS_BYTES, s, J, D, P, nb, E, nTiles, opT,
                               tail, t_static);
            return FE_OK;
        }
    }
    hipLaunchKernelGGL((fe::div3d_mfma_kernel<NP, M, 0, MODE, 2>), dim3(grid), dim3(256),
                       G::LDS_BYTES, s, J, D, nullptr, P, nb, E, nTiles, opT, jes);
    return FE_OK;
}

template <int MODE>
int launch_nd2_np(const double* J, const double* D, const fe::FieldPtrs& P, int nb, int64_t E, int Np, int opT,
                  hipStream_t s, bool* launched, int jes = 0) {
    switch (Np) {   // triangles p = 1..5; a wave tile of 16 M elements moves a few KB
        case 21: return launch_nd2<21, (MODE == 0 ? 2 : 3), MODE>(J, D, P, nb, E, opT, s, launched, jes);
        case 15: return launch_nd2<15, 4, MODE>(J, D, P, nb, E, opT, s, launched, jes);
        case 10: return launch_nd2<10, 6, MODE>(J, D, P, nb, E, opT, s, launched, jes);
        case 6: return launch_nd2<6, 8, MODE>(J, D, P, nb, E, opT, s, launched, jes);
        default: return launch_nd2<3, 8, MODE>(J, D, P, nb, E, opT, s, launched, jes);
    }
}

// ---- prepared operators: what fe_prepare_operator wrote where (host-side record, so that a launcher
// can refuse a buffer prepared for another shape without reading device memory)
struct PreparedInfo { int kind, Np, nf, Nfp, flags; const void* op; int device; };   // op: the array it is a snapshot of
std::mutex g_prepared_mutex;
std::unordered_map<const void*, PreparedInfo> g_prepared;

// null: not usable (with *err set when the caller passed a buffer that does not fit the call)
const void* usable_prepared(const void* prepared, const void* op, int kind, int Np, int nf, int Nfp, int flags,
                            const char* what, int* err) {
    *err = FE_OK;
    if (!prepared) return nullptr;
    PreparedInfo info;
    {
        std::lock_guard<std::mutex> lock(g_prepared_mutex);
        auto it = g_prepared.find(prepared);
        if (it == g_prepared.end()) {
            *err = fail(FE_EINVAL, "%s: the prepared-operator buffer %p was not written by fe_prepare_operator in this "
                        "process", what, prepared);
            return nullptr;
        }
        info = it->second;
    }
    if (info.kind != kind || info.Np != Np || info.nf != nf || info.Nfp != Nfp || info.flags != flags) {
        *err = fail(FE_EINVAL, "%s: the prepared operator is for another call (kind %d Np %d nf %d Nfp %d flags %d; "
                    "this call: kind %d Np %d nf %d Nfp %d flags %d)", what, info.kind, info.Np, info.nf, info.Nfp,
                    info.flags, kind, Np, nf, Nfp, flags);
        return nullptr;
    }
    int dev = -1;
    (void)hipGetDevice(&dev);
    if (info.op != op || info.device != dev) {   // a snapshot of ANOTHER operator array of the same shape would compute silently with it
        *err = fail(FE_EINVAL, "%s: the prepared operator %p is a snapshot of the operator array %p on device %d, this call "
                    "passes the operator array %p on device %d", what, prepared, info.op, info.device, op, dev);
        return nullptr;
    }
    return prepared;
}
constexpr int kPreparedD = 1, kPreparedR = 2;   // D[3][Np][Np] (grad + div sections) / face-mass R

template <int NP, int MG, int MD>
void prepare_d(const double* D, void* prepared, int opT, hipStream_t s) {
    using GG = fe::GradGeom<NP, MG>;
    using GD = fe::DivGeom<NP, MD>;
    static_assert(fe::kPrepGradOff + ((GG::RT * GG::KS + 1) / 2) * 1024 <= fe::kPrepDivOff, "grad section");
    static_assert(fe::kPrepDivOff + ((GD::BT * GD::KSJ * GD::NC + 1) / 2) * 1024 <= fe::kPrepDivSmallOff, "div section");
    static_assert(fe::kPrepDivSmallOff + GD::ASMALL_D * 8 <= fe::kPreparedBytes, "div 4-row groups");
    hipLaunchKernelGGL((fe::grad_prepare_kernel<NP, MG>), dim3(GG::RT * GG::KS), dim3(64), 0, s, D, 
                       static_cast<char*>(prepared) + fe::kPrepGradOff, opT);
    hipLaunchKernelGGL((fe::div_prepare_kernel<NP, MD>), dim3(GD::BT * GD::KSJ * GD::NC + (GD::ASMALL_D + 63) / 64),
                       dim3(64), 0, s, D, prepared, opT);
}

template <int NP, int NFP, int M>
void prepare_r(const double* R, void* prepared, int rlayout, hipStream_t s) {
    using G = fe::FmGeom<NP, NFP, M>;
    static_assert(fe::kPrepFmOff + (((G::BT + G::NS) * G::KS + 1) / 2) * 1024 <= fe::kPreparedBytes, "face-mass section");
    hipLaunchKernelGGL((fe::facemass_prepare_kernel<NP, NFP, M>), dim3((G::BT + G::NS) * G::KS), dim3(64), 0, s, R,
                       prepared, rlayout);
}

struct FmChoice { int max_group, tel; };
// (Np, Nfp) pairs of tetrahedral orders p = 1..4 with nf = 4
inline bool fm_mfma_geometry(int Np, int nf, int Nfp, FmChoice* c) {
    if (nf == 3) {   // triangles p = 1..5
        if (Np == 21 && Nfp == 6) { *c = {4, 48}; return true; }
        if (Np == 15 && Nfp == 5) { *c = {4, 64}; return true; }
        if (Np == 10 && Nfp == 4) { *c = {4, 80}; return true; }
        if (Np == 6 && Nfp == 3) { *c = {4, 96}; return true; }
        if (Np == 3 && Nfp == 2) { *c = {4, 128}; return true; }
        return false;
    }
    if (nf != fe::kFmNf) return false;
    if (Np == 56 && Nfp == 21) { *c = {4, 16}; return true; }   // p = 5: A fragments in LDS
    if (Np == 35 && Nfp == 15) { *c = {8, 16}; return true; }
    if (Np == 20 && Nfp == 10) { *c = {4, 16}; return true; }
    if (Np == 10 && Nfp == 6) { *c = {4, 32}; return true; }
    if (Np == 4 && Nfp == 3) { *c = {4, 64}; return true; }
    return false;
}

}  // namespace

extern "C" {

int fe_version(void) { return 1000; }

const char* fe_last_error(void) { return g_err; }

int fe_device_count(void) {
    int n = 0;
    FE_HIP_CHECK(hipGetDeviceCount(&n));
    return n;
}

int fe_device_info(int dev, char* name, size_t name_len, double* peak_f64_gflops,
                   double* peak_gbps) {
    hipDeviceProp_t p;
    FE_HIP_CHECK(hipGetDeviceProperties(&p, dev));
    const bool gfx950 = strstr(p.gcnArchName, "gfx950") != nullptr;
    if (name && name_len) {
        // ROCm often reports a generic marketing name ("AMD Radeon Graphics") or none at
        // all; the roofline tables are keyed by part, so gfx950 is named canonically.
        const char* nm = gfx950 ? "AMD Instinct MI355X" : (p.name[0] != 0 ? p.name : p.gcnArchName);
        strncpy(name, nm, name_len - 1);
        name[name_len - 1] = 0;
    }
    // fp64: vector = matrix = 32 FMA-flop/clk/SIMD.. i.e. 128 flop/clk/CU (MI355X: 256 CUs
    // x 2.4 GHz -> 78.6 TFLOP/s); HBM3E 8 TB/s (datasheet).  gfx950 only.
    const bool mi355 = gfx950;
    if (peak_f64_gflops)
        *peak_f64_gflops = mi355 ? 128.0 * p.multiProcessorCount * (p.clockRate * 1e-6) : 0.0;
    if (peak_gbps) *peak_gbps = mi355 ? 8000.0 : 0.0;
    return FE_OK;
}

int fe_prepare_operator(int32_t family, const double* op, int32_t Np, int32_t nf, int32_t Nfp, int32_t flags,
                        void* prepared, void* stream) {
    if (!op || !prepared) return fail(FE_EINVAL, "prepare: null pointer");
    if ((reinterpret_cast<uintptr_t>(op) & 7u) || (reinterpret_cast<uintptr_t>(prepared) & 15u))
        return fail(FE_EINVAL, "prepare: the operator must be 8-byte and the prepared buffer 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    PreparedInfo info{0, Np, 0, 0, flags, op, -1};
    (void)hipGetDevice(&info.device);
    if (family == FE_FAMILY_GRAD || family == FE_FAMILY_DIV || family == FE_FAMILY_GRADDIV) {
        if (flags & ~FE_OP_TRANSPOSED) return fail(FE_EINVAL, "prepare: bad operator flags %d", flags);
        const int opT = (flags & FE_OP_TRANSPOSED) ? 1 : 0;
        info.kind = kPreparedD;
        switch (Np) {   // the (Np, M) geometries of launch_grad / launch_div
            case 35: prepare_d<35, 1, 1>(op, prepared, opT, s); break;
            case 20: prepare_d<20, 2, 1>(op, prepared, opT, s); break;
            case 10: prepare_d<10, 3, 3>(op, prepared, opT, s); break;
            case 4: prepare_d<4, 5, 5>(op, prepared, opT, s); break;
            default: return fail(FE_EUNSUPPORTED, "prepare: grad / div operators of tetrahedra p = 1..4 only (Np=%d)", Np);
        }
    } else if (family == FE_FAMILY_FACEMASS) {
        if (flags & ~(FE_FM_R_IFJ | FE_FM_R_T)) return fail(FE_EINVAL, "prepare: bad face-mass operator flags %d", flags);
        const int rlayout = ((flags & FE_FM_R_IFJ) ? 1 : 0) + ((flags & FE_FM_R_T) ? 2 : 0);
        info.kind = kPreparedR;
        info.nf = nf;
        info.Nfp = Nfp;
        if (nf != fe::kFmNf) return fail(FE_EUNSUPPORTED, "prepare: face-mass operators of tetrahedra only (nf=%d)", nf);
        if (Np == 35 && Nfp == 15) prepare_r<35, 15, 1>(op, prepared, rlayout, s);
        else if (Np == 20 && Nfp == 10) prepare_r<20, 10, 1>(op, prepared, rlayout, s);
        else if (Np == 10 && Nfp == 6) prepare_r<10, 6, 2>(op, prepared, rlayout, s);
        else if (Np == 4 && Nfp == 3) prepare_r<4, 3, 4>(op, prepared, rlayout, s);
        else return fail(FE_EUNSUPPORTED, "prepare: face-mass operators of tetrahedra p = 1..4 only (Np=%d Nfp=%d)", Np, Nfp);
    } else {
        return fail(FE_EUNSUPPORTED, "prepare: family %d has no prepared form", family);
    }
    FE_HIP_CHECK(hipGetLastError());
    std::lock_guard<std::mutex> lock(g_prepared_mutex);
    g_prepared[prepared] = info;
    return FE_OK;
}

int fe_release_prepared(const void* prepared) {
    std::lock_guard<std::mutex> lock(g_prepared_mutex);
    return g_prepared.erase(prepared) ? FE_OK : fail(FE_EINVAL, "fe_release_prepared: %p is not a prepared-operator buffer of this process", prepared);
}

int fe_split_alloc(void** ptr, size_t bytes, int32_t flags) {
    SplitPool* pool = split_pool_of_current_device();
    std::lock_guard<std::mutex> lock(pool->mu);
    return pool->alloc(ptr, bytes, flags);
}

int fe_split_free(void* ptr) {
    if (!ptr) return FE_OK;
    const int dev = split_owner_device(ptr);
    if (dev < 0) return fail(FE_EINVAL, "fe_split_free: %p is not an array of the split allocator (on any device)", ptr);
    SplitDeviceScope scope(dev);
    SplitPool* pool = &g_split_pools[dev];
    std::lock_guard<std::mutex> lock(pool->mu);
    return pool->free_array(ptr);
}

int fe_split_info(const void* ptr, char* buf, size_t buf_len) {
    if (!buf || buf_len == 0) return fail(FE_EINVAL, "fe_split_info: no buffer");
    const int dev = split_owner_device(ptr);
    if (dev < 0) return fail(FE_EINVAL, "fe_split_info: %p is not an array of the split allocator (on any device)", ptr);
    SplitPool* pool = &g_split_pools[dev];
    std::lock_guard<std::mutex> lock(pool->mu);
    return pool->info(ptr, buf, buf_len);
}

int fe_split_stats(char* buf, size_t buf_len) {
    if (!buf || buf_len == 0) return fail(FE_EINVAL, "fe_split_stats: no buffer");
    SplitPool* pool = split_pool_of_current_device();
    std::lock_guard<std::mutex> lock(pool->mu);
    return pool->stats(buf, buf_len);
}

int fe_split_reserve(size_t bytes) {
    SplitPool* pool = split_pool_of_current_device();
    std::lock_guard<std::mutex> lock(pool->mu);
    return pool->reserve(bytes);
}

int fe_split_trim(void) {
    SplitPool* pool = split_pool_of_current_device();
    std::lock_guard<std::mutex> lock(pool->mu);
    return pool->trim();
}

int fe_kernel_resources(char* buf, size_t buf_len) {
    if (!buf || buf_len == 0) return fail(FE_EINVAL, "fe_kernel_resources: no buffer");
    std::lock_guard<std::mutex> lock(g_resources_mutex);
    strncpy(buf, g_resources.c_str(), buf_len - 1);
    buf[buf_len - 1] = 0;
    return (int)g_resources.size();
}

int64_t fe_flops_per_element(int32_t family, int32_t Np, int32_t nf, int32_t Nfp, int32_t b) {
    const int64_t np = Np;
    switch (family) {
        case FE_FAMILY_GRAD:
        case FE_FAMILY_DIV: return 2 * 3 * np * np + 2 * 9 * np;
        case FE_FAMILY_GRADDIV: return 2 * (2 * 3 * np * np + 2 * 9 * np);
        case FE_FAMILY_FACEMASS: return (int64_t)b * ((int64_t)nf * Nfp + 2 * np * nf * Nfp);
        case FE_FAMILY_DIVCOMP: return 3 * np + 2 * 3 * np * np;
        case FE_FAMILY_GRADPLANES: return (int64_t)(b > 0 ? b : 1) * (3 * np + 2 * 3 * np * np);   // b = output planes
        case FE_FAMILY_MATAPPLY: return (int64_t)(b > 0 ? b : 1) * (2 * np * np + np);                // with the J[e] factor
        default: return -1;
    }
}

int fe_grad3d_f64(const double* J, const double* D, const double* u, double* out, int64_t E,
                  int32_t Np, int32_t variant, void* stream) {
    return fe_grad3d_f64_ex(J, D, u, out, E, Np, 0, variant, stream);
}

int fe_grad3d_f64_ex(const double* J, const double* D, const double* u, double* out, int64_t E,
                     int32_t Np, int32_t op_flags, int32_t variant, void* stream) {
    return fe_grad3d_batched_f64(J, D, &u, &out, E, Np, 1, op_flags, variant, stream);
}

int fe_grad3d_batched_f64(const double* J, const double* D, const double* const* u,
                          double* const* out, int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                          int32_t variant, void* stream) {
    return fe_grad3d_prepared_f64(J, D, nullptr, u, out, E, Np, b, op_flags, variant, stream);
}

int fe_grad3d_prepared_f64(const double* J, const double* D, const void* D_prepared, const double* const* u,
                           double* const* out, int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                           int32_t variant, void* stream) {
    if (!u || !out) return fail(FE_EINVAL, "grad: null pointer table");
    if (b < 1) return fail(FE_EINVAL, "grad: b=%d, need at least one field", b);
    if (b > FE_MAX_FIELDS) {   // FE_MAX_FIELDS fields per launch
        if (int rc = fe_grad3d_prepared_f64(J, D, D_prepared, u, out, E, Np, FE_MAX_FIELDS, op_flags, variant, stream))
            return rc;
        return fe_grad3d_prepared_f64(J, D, D_prepared, u + FE_MAX_FIELDS, out + FE_MAX_FIELDS, E, Np, b - FE_MAX_FIELDS,
                                      op_flags, variant, stream);
    }
    int perr;
    const void* prep = usable_prepared(D_prepared, D, kPreparedD, Np, 0, 0, op_flags, "grad", &perr);
    if (perr != FE_OK) return perr;
    for (int k = 0; k < b; ++k)
        if (int rc = check_common(J, D, u[k], out[k], E, Np)) return rc;
    if (op_flags & ~FE_OP_TRANSPOSED) return fail(FE_EINVAL, "grad: bad operator flags %d", op_flags);
    const int opT = (op_flags & FE_OP_TRANSPOSED) ? 1 : 0;
#ifdef FE_EXPERIMENTS
    if (variant < FE_VARIANT_AUTO || (variant > FE_VARIANT_TILED && variant < 1000))
#else
    if (variant < FE_VARIANT_AUTO || variant > FE_VARIANT_TILED)
#endif
        return fail(FE_EUNSUPPORTED, "grad: unknown variant %d", variant);
    if (E == 0) return FE_OK;
    return grad_fields_launch(grad_fields(J, u, out, b, E, Np), J, D, prep, b, 3, E, Np, opT, variant,
                              static_cast<hipStream_t>(stream));
}

int fe_gradplanes3d_f64(const double* const* J3, const double* D, const double* const* u,
                        double* const* out, int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                        int32_t variant, void* stream) {
    if (!J3 || !u || !out) return fail(FE_EINVAL, "grad planes: null pointer table");
    if (b < 1) return fail(FE_EINVAL, "grad planes: b=%d, need at least one field", b);
    if (b > FE_MAX_FIELDS) {   // FE_MAX_FIELDS fields per launch
        if (int rc = fe_gradplanes3d_f64(J3, D, u, out, E, Np, FE_MAX_FIELDS, op_flags, variant, stream)) return rc;
        return fe_gradplanes3d_f64(J3, D, u + FE_MAX_FIELDS, out + 3 * FE_MAX_FIELDS, E, Np, b - FE_MAX_FIELDS,
                                   op_flags, variant, stream);
    }
    if (op_flags & ~FE_OP_TRANSPOSED) return fail(FE_EINVAL, "grad planes: bad operator flags %d", op_flags);
    if (variant < FE_VARIANT_AUTO || variant > FE_VARIANT_TILED)
        return fail(FE_EUNSUPPORTED, "grad planes: unknown variant %d", variant);
    if (E < 0) return fail(FE_EINVAL, "E must be >= 0 (got %lld)", (long long)E);
    if (E == 0) return FE_OK;   // empty arrays have no addresses to tell wanted planes from unwanted ones
    fe::GradFields P = {};
    int nx = -1;
    for (int k = 0; k < b; ++k) {
        int planes = 0;
        for (int x = 0; x < 3; ++x) {
            double* o = out[3 * k + x];
            if (!o) continue;
            if (int rc = check_common(J3[x], D, u[k], o, E, Np)) return rc;
            P.out[k][x] = o;
            ++planes;
        }
        if (planes == 0) return fail(FE_EINVAL, "grad planes: field %d has no output plane", k);
        if (nx >= 0 && planes != nx)
            return fail(FE_EINVAL, "grad planes: every field of a call needs the same number of planes (%d vs %d)",
                        planes, nx);
        nx = planes;
        P.u[k] = u[k];
    }
    for (int x = 0; x < 3; ++x) {
        if (!J3[x]) return fail(FE_EINVAL, "grad planes: null geometry-factor array %d", x);
        if (reinterpret_cast<uintptr_t>(J3[x]) & 7u) return fail(FE_EINVAL, "device pointers must be 8-byte aligned");
        P.j[x] = J3[x];
    }
    // The planes kernel is the row-permuted MFMA grad kernel (tetrahedra p = 1..4).  Whatever it does
    // not serve -- another order, or an explicitly requested tiled / generic variant -- runs plane by
    // plane through the div-component launcher, which has its own MFMA (p = 5), tiled and generic paths.
    const bool planes_kernel = (Np == 35 || Np == 20 || Np == 10 || Np == 4) &&
                               (variant == FE_VARIANT_AUTO || variant == FE_VARIANT_MFMA);
    if (Np == 56 && (variant == FE_VARIANT_AUTO || variant == FE_VARIANT_MFMA)) {   // p = 5: grad by components
        bool launched = false;
        if (int rc = launch_gradplanes_p5(P, D, b, E, (op_flags & FE_OP_TRANSPOSED) ? 1 : 0, static_cast<hipStream_t>(stream),
                                          &launched))
            return rc;
        if (launched) {
            FE_HIP_CHECK(hipGetLastError());
            return FE_OK;
        }
    }
    if (!planes_kernel) {
        for (int k = 0; k < b; ++k)
            for (int x = 0; x < 3; ++x)
                if (P.out[k][x])
                    if (int rc = fe_divcomp3d_f64(P.j[x], D, P.u[k], P.out[k][x], E, Np, op_flags & FE_OP_TRANSPOSED,
                                                  variant, stream))
                        return rc;
        return FE_OK;
    }
    return grad_fields_launch(P, nullptr, D, nullptr, b, nx, E, Np, (op_flags & FE_OP_TRANSPOSED) ? 1 : 0, variant,
                              static_cast<hipStream_t>(stream));
}

int fe_div3d_f64(const double* J, const double* D, const double* u, double* out, int64_t E,
                 int32_t Np, int32_t variant, void* stream) {
    return fe_div3d_f64_ex(J, D, u, out, E, Np, 0, variant, stream);
}

int fe_div3d_f64_ex(const double* J, const double* D, const double* u, double* out, int64_t E,
                    int32_t Np, int32_t op_flags, int32_t variant, void* stream) {
    return fe_div3d_batched_f64(J, D, &u, &out, E, Np, 1, op_flags, variant, stream);
}

int fe_div3d_batched_f64(const double* J, const double* D, const double* const* u,
                         double* const* out, int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                         int32_t variant, void* stream) {
    return fe_div3d_prepared_f64(J, D, nullptr, u, out, E, Np, b, op_flags, variant, stream);
}

int fe_div3d_prepared_f64(const double* J, const double* D, const void* D_prepared, const double* const* u,
                          double* const* out, int64_t E, int32_t Np, int32_t b, int32_t op_flags,
                          int32_t variant, void* stream) {
    if (!u || !out) return fail(FE_EINVAL, "div: null pointer table");
    if (b < 1) return fail(FE_EINVAL, "div: b=%d, need at least one field", b);
    if (b > FE_MAX_FIELDS) {   // FE_MAX_FIELDS fields per launch
        if (int rc = fe_div3d_prepared_f64(J, D, D_prepared, u, out, E, Np, FE_MAX_FIELDS, op_flags, variant, stream))
            return rc;
        return fe_div3d_prepared_f64(J, D, D_prepared, u + FE_MAX_FIELDS, out + FE_MAX_FIELDS, E, Np, b - FE_MAX_FIELDS,
                                     op_flags, variant, stream);
    }
    int perr;
    const void* prep = usable_prepared(D_prepared, D, kPreparedD, Np, 0, 0, op_flags, "div", &perr);
    if (perr != FE_OK) return perr;
    fe::FieldPtrs P = {};
    for (int k = 0; k < b; ++k) {
        if (int rc = check_common(J, D, u[k], out[k], E, Np)) return rc;
        P.v[k] = u[k];
        P.out[k] = out[k];
    }
    if (op_flags & ~FE_OP_TRANSPOSED) return fail(FE_EINVAL, "div: bad operator flags %d", op_flags);
    const int opT = (op_flags & FE_OP_TRANSPOSED) ? 1 : 0;
#ifdef FE_EXPERIMENTS
    if (variant < FE_VARIANT_AUTO || (variant > FE_VARIANT_MFMA_SPLIT && variant < 1000))
#else
    if (variant < FE_VARIANT_AUTO || variant > FE_VARIANT_MFMA_SPLIT)
#endif
        return fail(FE_EUNSUPPORTED, "div: unknown variant %d", variant);
    if (E == 0) return FE_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool mfma_ok = Np == 56 || Np == 35 || Np == 20 || Np == 10 || Np == 4;
    const bool split_walk = variant == FE_VARIANT_MFMA_SPLIT;
    if (split_walk && !(Np == 35 || Np == 20 || Np == 10 || Np == 4))
        return fail(FE_EUNSUPPORTED, "div: the split walk is a flavour of the p = 1..4 MFMA kernels (Np=%d)", Np);
    const fe::TiledArgs ta = tiled_args(FE_FAMILY_DIV, J, D, P, b, E, 3, Np, 0, 0, opT, 0, 0);
    KernelPath path;
    if (int rc = choose_path((variant >= 1000 || split_walk) ? FE_VARIANT_MFMA : variant, mfma_ok, tiled_fits(ta), "div", Np,
                             &path))
        return rc;
    if (path == kPathTiled) return launch_tiled(ta, s);
    if (path == kPathMfma && Np == 56) {   // p = 5
        bool launched = false;
        if (int rc = launch_div_p5(J, D, P, b, E, opT, s, &launched)) return rc;
        if (launched) {
            FE_HIP_CHECK(hipGetLastError());
            return FE_OK;
        }
        return tiled_fits(ta) ? launch_tiled(ta, s) : fail(FE_EUNSUPPORTED, "div: no kernel for this size");
    }
    int64_t e_done = 0;
    if (path == kPathMfma) {
        const int dbg = variant >= 1000 ? (variant - 1000) & 255 : 0;   // experiment builds only
        int rc = FE_OK;
        const int opf = opT | (split_walk ? fe::kDivWalkSplit : 0);
        switch (Np) {   // wave tile = 16 M elements
            case 35: rc = launch_div<35, 1>(J, D, prep, P, b, E, dbg, opf, s, &e_done); break;
            case 20: rc = launch_div<20, 1>(J, D, prep, P, b, E, dbg, opf, s, &e_done); break;
            case 10: rc = launch_div<10, 3>(J, D, prep, P, b, E, dbg, opf, s, &e_done); break;
            default: rc = launch_div<4, 5>(J, D, prep, P, b, E, dbg, opf, s, &e_done); break;
        }
        if (rc != FE_OK) return rc;
    }
    if (e_done < E)
        for (int k = 0; k < b; ++k)
            hipLaunchKernelGGL(fe::div3d_generic_kernel, dim3(generic_grid(E - e_done, Np)), dim3(256),
                               0, s, J, D, P.v[k], P.out[k], E, Np, e_done, opT);
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

// grad / div of ndim-dimensional elements: ndim = 3 is the 3d entry point; ndim = 2 (triangles)
// runs on the MFMA instances above for p = 1..5, else on the tiled kernel.
static int nd_launch(int family, const char* what, const double* J, const double* D, const double* const* u,
                     double* const* out, int64_t E, int32_t ndim, int32_t Np, int32_t b, int32_t op_flags,
                     int32_t variant, void* stream) {
    if (ndim != 2) return fail(FE_EUNSUPPORTED, "%s: ndim must be 2 or 3 (got %d)", what, ndim);
    if (!u || !out) return fail(FE_EINVAL, "%s: null pointer table", what);
    if (b < 1) return fail(FE_EINVAL, "%s: b=%d, need at least one field", what, b);
    if (op_flags & ~FE_OP_TRANSPOSED) return fail(FE_EINVAL, "%s: bad operator flags %d", what, op_flags);
    if (variant < FE_VARIANT_AUTO || variant > FE_VARIANT_TILED || variant == FE_VARIANT_GENERIC)
        return fail(FE_EUNSUPPORTED, "%s: ndim = 2 has the MFMA and the tiled kernels (variant %d)", what, variant);
    const bool mfma_ok = Np == 3 || Np == 6 || Np == 10 || Np == 15 || Np == 21;
    const int opT = (op_flags & FE_OP_TRANSPOSED) ? 1 : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int k0 = 0; k0 < b; k0 += fe::kMaxFields) {
        const int nb = b - k0 < fe::kMaxFields ? b - k0 : fe::kMaxFields;
        fe::FieldPtrs P = {};
        for (int k = 0; k < nb; ++k) {
            if (int rc = check_common(J, D, u[k0 + k], out[k0 + k], E, Np)) return rc;
            P.v[k] = u[k0 + k];
            P.out[k] = out[k0 + k];
        }
        if (E == 0) continue;
        const fe::TiledArgs ta = tiled_args(family, J, D, P, nb, E, ndim, Np, 0, 0, opT, 0, 0);
        KernelPath path;
        if (int rc = choose_path(variant, mfma_ok, tiled_fits(ta), what, Np, &path)) return rc;
        bool launched = false;
        if (path == kPathMfma) {
            const int rc = family == FE_FAMILY_GRAD ? launch_nd2_np<4>(J, D, P, nb, E, Np, opT, s, &launched)
                                                    : launch_nd2_np<0>(J, D, P, nb, E, Np, opT, s, &launched);
            if (rc != FE_OK) return rc;
        }
        if (!launched) {   // no MFMA geometry, or fewer elements than one wave tile
            if (!tiled_fits(ta)) return fail(FE_EUNSUPPORTED, "%s: no kernel for ndim = 2, Np = %d", what, Np);
            if (int rc = launch_tiled(ta, s)) return rc;
        }
    }
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

int fe_grad_f64(const double* J, const double* D, const double* const* u, double* const* out, int64_t E,
                int32_t ndim, int32_t Np, int32_t b, int32_t op_flags, int32_t variant, void* stream) {
    if (ndim == 3) return fe_grad3d_batched_f64(J, D, u, out, E, Np, b, op_flags, variant, stream);
    return nd_launch(FE_FAMILY_GRAD, "grad", J, D, u, out, E, ndim, Np, b, op_flags, variant, stream);
}

int fe_div_f64(const double* J, const double* D, const double* const* u, double* const* out, int64_t E,
               int32_t ndim, int32_t Np, int32_t b, int32_t op_flags, int32_t variant, void* stream) {
    if (ndim == 3) return fe_div3d_batched_f64(J, D, u, out, E, Np, b, op_flags, variant, stream);
    return nd_launch(FE_FAMILY_DIV, "div", J, D, u, out, E, ndim, Np, b, op_flags, variant, stream);
}

int fe_divcomp_f64(const double* J, const double* D, const double* u, double* out, int64_t E, int32_t ndim,
                   int32_t Np, int32_t op_flags, int32_t variant, void* stream) {
    if (ndim == 3) return fe_divcomp3d_f64(J, D, u, out, E, Np, op_flags, variant, stream);
    if (ndim != 2) return fail(FE_EUNSUPPORTED, "div component: ndim must be 2 or 3 (got %d)", ndim);
    if (int rc = check_common(J, D, u, out, E, Np)) return rc;
    if (op_flags & ~(FE_OP_TRANSPOSED | FE_OP_J_ES))
        return fail(FE_EINVAL, "div component: bad operator flags %d", op_flags);
    if (variant < FE_VARIANT_AUTO || variant > FE_VARIANT_TILED || variant == FE_VARIANT_GENERIC)
        return fail(FE_EUNSUPPORTED, "div component: ndim = 2 has the MFMA and the tiled kernels (variant %d)", variant);
    if (E == 0) return FE_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int opT = (op_flags & FE_OP_TRANSPOSED) ? 1 : 0, jes = (op_flags & FE_OP_J_ES) ? 1 : 0;
    const bool mfma_ok = Np == 3 || Np == 6 || Np == 10 || Np == 15 || Np == 21;
    fe::FieldPtrs P = {};
    P.v[0] = u;
    P.out[0] = out;
    const fe::TiledArgs ta = tiled_args(FE_FAMILY_DIVCOMP, J, D, P, 1, E, 2, Np, 0, 0, opT, jes, 0);
    KernelPath path;
    if (int rc = choose_path(variant, mfma_ok, tiled_fits(ta), "div component", Np, &path)) return rc;
    bool launched = false;
    if (path == kPathMfma)
        if (int rc = launch_nd2_np<1>(J, D, P, 1, E, Np, opT, s, &launched, jes)) return rc;
    if (!launched) {   // no MFMA geometry, or fewer elements than one wave tile
        if (!tiled_fits(ta)) return fail(FE_EUNSUPPORTED, "div component: no kernel for ndim = 2, Np = %d", Np);
        if (int rc = launch_tiled(ta, s)) return rc;
    }
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

int fe_divcomp3d_f64(const double* J, const double* D, const double* u, double* out, int64_t E,
                     int32_t Np, int32_t op_flags, int32_t variant, void* stream) {
    if (int rc = check_common(J, D, u, out, E, Np)) return rc;
    if (op_flags & ~(FE_OP_TRANSPOSED | FE_OP_J_ES))
        return fail(FE_EINVAL, "div component: bad operator flags %d", op_flags);
    if (variant < FE_VARIANT_AUTO || variant > FE_VARIANT_TILED)
        return fail(FE_EUNSUPPORTED, "div component: unknown variant %d", variant);
    if (E == 0) return FE_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int opT = (op_flags & FE_OP_TRANSPOSED) ? 1 : 0, jes = (op_flags & FE_OP_J_ES) ? 1 : 0;
    const bool mfma_ok = Np == 56 || Np == 35 || Np == 20 || Np == 10 || Np == 4;
    fe::FieldPtrs Pt = {};
    Pt.v[0] = u;
    Pt.out[0] = out;
    const fe::TiledArgs ta = tiled_args(FE_FAMILY_DIVCOMP, J, D, Pt, 1, E, 3, Np, 0, 0, opT, jes, 0);
    KernelPath path;
    if (int rc = choose_path(variant, mfma_ok, tiled_fits(ta), "div component", Np, &path)) return rc;
    if (path == kPathTiled) return launch_tiled(ta, s);
    int64_t e_done = 0;
    if (path == kPathMfma) {
        int rc = FE_OK;
        switch (Np) {   // wave tile = 16 M elements
            case 56: rc = launch_divcomp<56, 1, true>(J, D, u, out, E, opT, jes, s, &e_done); break;   // p = 5: A in LDS
            case 35: rc = launch_divcomp<35, 1>(J, D, u, out, E, opT, jes, s, &e_done); break;
            case 20: rc = launch_divcomp<20, 2>(J, D, u, out, E, opT, jes, s, &e_done); break;
            case 10: rc = launch_divcomp<10, 4>(J, D, u, out, E, opT, jes, s, &e_done); break;
            default: rc = launch_divcomp<4, 6>(J, D, u, out, E, opT, jes, s, &e_done); break;
        }
        if (rc != FE_OK) return rc;
    }
    if (e_done < E)
        hipLaunchKernelGGL(fe::divcomp3d_generic_kernel, dim3(generic_grid(E - e_done, Np)), dim3(256), 0, s,
                           J, D, u, out, E, Np, e_done, opT, jes);
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

int fe_matapply_f64(const double* J, const double* D, const double* const* u, double* const* out, int64_t E,
                    int32_t Np, int32_t b, int32_t op_flags, int32_t variant, void* stream) {
    if (!u || !out) return fail(FE_EINVAL, "matapply: null pointer table");
    if (b < 1) return fail(FE_EINVAL, "matapply: b=%d, need at least one field", b);
    if (b > FE_MAX_FIELDS) {   // FE_MAX_FIELDS fields per launch
        if (int rc = fe_matapply_f64(J, D, u, out, E, Np, FE_MAX_FIELDS, op_flags, variant, stream)) return rc;
        return fe_matapply_f64(J, D, u + FE_MAX_FIELDS, out + FE_MAX_FIELDS, E, Np, b - FE_MAX_FIELDS, op_flags,
                               variant, stream);
    }
    fe::FieldPtrs P = {};
    for (int k = 0; k < b; ++k) {
        if (int rc = check_common(J ? J : D, D, u[k], out[k], E, Np)) return rc;   // J is optional
        P.v[k] = u[k];
        P.out[k] = out[k];
    }
    if (op_flags & ~FE_OP_TRANSPOSED) return fail(FE_EINVAL, "matapply: bad operator flags %d", op_flags);
    if (variant < FE_VARIANT_AUTO || variant > FE_VARIANT_TILED)
        return fail(FE_EUNSUPPORTED, "matapply: unknown variant %d", variant);
    if (E == 0) return FE_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int opT = (op_flags & FE_OP_TRANSPOSED) ? 1 : 0;
    const bool mfma_ok = Np == 56 || Np == 35 || Np == 20 || Np == 15 || Np == 10 || Np == 6 || Np == 4 || Np == 3;
    const fe::TiledArgs ta = tiled_args(FE_FAMILY_MATAPPLY, J, D, P, b, E, 1, Np, 0, 0, opT, 0, 0);
    KernelPath path;
    if (int rc = choose_path(variant, mfma_ok, tiled_fits(ta), "matapply", Np, &path)) return rc;
    if (path == kPathTiled) return launch_tiled(ta, s);
    int64_t e_done = 0;
    if (path == kPathMfma) {
        int rc = FE_OK;
        switch (Np) {   // wave tile = 16 M elements: a few KB per tile at every order
            case 56: rc = launch_matapply<56, 1>(J, D, P, b, E, opT, s, &e_done); break;   // p = 5
            case 35: rc = launch_matapply<35, 2>(J, D, P, b, E, opT, s, &e_done); break;
            case 20: rc = launch_matapply<20, 4>(J, D, P, b, E, opT, s, &e_done); break;
            case 15: rc = launch_matapply<15, 4>(J, D, P, b, E, opT, s, &e_done); break;
            case 10: rc = launch_matapply<10, 6>(J, D, P, b, E, opT, s, &e_done); break;
            case 6: rc = launch_matapply<6, 8>(J, D, P, b, E, opT, s, &e_done); break;
            case 4: rc = launch_matapply<4, 8>(J, D, P, b, E, opT, s, &e_done); break;
            default: rc = launch_matapply<3, 8>(J, D, P, b, E, opT, s, &e_done); break;
        }
        if (rc != FE_OK) return rc;
    }
    if (e_done < E)
        for (int k = 0; k < b; ++k)
            hipLaunchKernelGGL(fe::matapply_generic_kernel, dim3(generic_grid(E - e_done, Np)), dim3(256), 0, s,
                               J, D, P.v[k], P.out[k], E, Np, e_done, opT);
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

int fe_graddiv3d_f64(const double* J, const double* D, const double* u_grad, const double* v_div,
                     double* grad_out, double* div_out, int64_t E, int32_t Np, int32_t variant,
                     void* stream) {
    return fe_graddiv3d_prepared_f64(J, D, nullptr, u_grad, v_div, grad_out, div_out, E, Np, variant, stream);
}

int fe_graddiv3d_prepared_f64(const double* J, const double* D, const void* D_prepared, const double* u_grad,
                              const double* v_div, double* grad_out, double* div_out, int64_t E, int32_t Np,
                              int32_t variant, void* stream) {
    int perr;
    const void* prep = usable_prepared(D_prepared, D, kPreparedD, Np, 0, 0, 0, "graddiv", &perr);
    if (perr != FE_OK) return perr;
    if (int rc = check_common(J, D, u_grad, grad_out, E, Np)) return rc;
    if (int rc = check_common(J, D, v_div, div_out, E, Np)) return rc;
    if (variant < FE_VARIANT_AUTO || variant > FE_VARIANT_TILED)
        return fail(FE_EUNSUPPORTED, "graddiv: unknown variant %d", variant);
    const bool mfma_ok = Np == 35 || Np == 20 || Np == 10 || Np == 4;
    if ((variant != FE_VARIANT_AUTO && variant != FE_VARIANT_MFMA) || !mfma_ok) {
        // two launches back to back on the stream
        if (int rc = fe_div3d_prepared_f64(J, D, D_prepared, &v_div, &div_out, E, Np, 1, 0, variant, stream)) return rc;
        return fe_grad3d_prepared_f64(J, D, D_prepared, &u_grad, &grad_out, E, Np, 1, 0, variant, stream);
    }
    if (E == 0) return FE_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const fe::GradFields Pg = grad_fields(J, &u_grad, &grad_out, 1, E, Np);
    fe::FieldPtrs Pd = {};
    Pd.v[0] = v_div;  Pd.out[0] = div_out;
    int64_t done_g = 0, done_d = 0;
    int rc = FE_OK;
    switch (Np) {   // same (Np, M) geometries as the separate launches
        case 35: rc = launch_graddiv<35, 1, 1>(J, D, prep, Pg, Pd, E, s, &done_g, &done_d); break;
        case 20: rc = launch_graddiv<20, 2, 1>(J, D, prep, Pg, Pd, E, s, &done_g, &done_d); break;
        case 10: rc = launch_graddiv<10, 3, 3>(J, D, prep, Pg, Pd, E, s, &done_g, &done_d); break;
        default: rc = launch_graddiv<4, 5, 5>(J, D, prep, Pg, Pd, E, s, &done_g, &done_d); break;
    }
    if (rc != FE_OK) return rc;
    if (done_d < E)
        hipLaunchKernelGGL(fe::div3d_generic_kernel, dim3(generic_grid(E - done_d, Np)), dim3(256), 0, s,
                           J, D, v_div, div_out, E, Np, done_d, 0);
    if (done_g < E)
        hipLaunchKernelGGL(fe::grad3d_generic_kernel, dim3(generic_grid(E - done_g, Np)), dim3(256), 0, s,
                           J, D, u_grad, grad_out, E, Np, done_g, 0);
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

int fe_facemass_f64(const double* J, const double* R, const double* const* v, double* const* out,
                    int64_t E, int32_t Np, int32_t nf, int32_t Nfp, int32_t b,
                    int32_t layout_flags, int32_t variant, void* stream) {
    return fe_facemass_prepared_f64(J, R, nullptr, v, out, E, Np, nf, Nfp, b, layout_flags, variant, stream);
}

int fe_facemass_prepared_f64(const double* J, const double* R, const void* R_prepared, const double* const* v,
                             double* const* out, int64_t E, int32_t Np, int32_t nf, int32_t Nfp, int32_t b,
                             int32_t layout_flags, int32_t variant, void* stream) {
    int perr;   // (the J layout flag is not part of the operator)
    const void* prep = usable_prepared(R_prepared, R, kPreparedR, Np, nf, Nfp, layout_flags & ~FE_FM_J_FE, "face-mass", &perr);
    if (perr != FE_OK) return perr;
    if (E < 0) return fail(FE_EINVAL, "E must be >= 0 (got %lld)", (long long)E);
    if (Np <= 0 || nf <= 0 || Nfp <= 0 || b <= 0)
        return fail(FE_EINVAL, "face-mass: Np, nf, Nfp, b must be positive (%d %d %d %d)", Np, nf,
                    Nfp, b);
    if (layout_flags & ~7) return fail(FE_EINVAL, "face-mass: bad layout flags %d", layout_flags);
    if (variant < FE_VARIANT_AUTO || variant > FE_VARIANT_TILED)
        return fail(FE_EUNSUPPORTED, "face-mass: unknown variant %d", variant);
    if (!v || !out) return fail(FE_EINVAL, "face-mass: null pointer table");
    if (E == 0) return FE_OK;
    if (!J || !R) return fail(FE_EINVAL, "face-mass: null device pointer");
    for (int k = 0; k < b; ++k) {
        if (!v[k] || !out[k]) return fail(FE_EINVAL, "face-mass: null field pointer %d", k);
        if ((reinterpret_cast<uintptr_t>(v[k]) | reinterpret_cast<uintptr_t>(out[k])) & 7u)
            return fail(FE_EINVAL, "face-mass: field pointers must be 8-byte aligned");
    }
    if ((reinterpret_cast<uintptr_t>(J) | reinterpret_cast<uintptr_t>(R)) & 7u)
        return fail(FE_EINVAL, "face-mass: device pointers must be 8-byte aligned");
    if (E * (int64_t)Np >= (int64_t)1 << 39) return fail(FE_EINVAL, "E*Np too large");
    hipStream_t s = static_cast<hipStream_t>(stream);

    FmChoice geo{0, 16};
    const bool mfma_ok = fm_mfma_geometry(Np, nf, Nfp, &geo) && b >= 2;
    const int jfe = (layout_flags & FE_FM_J_FE) ? 1 : 0;
    // operator layouts: 0 R[f][i][j], 1 L[i][f][j], 2 R[f][j][i], 3 L[j][f][i]
    const int rifj = ((layout_flags & FE_FM_R_IFJ) ? 1 : 0) + ((layout_flags & FE_FM_R_T) ? 2 : 0);
    KernelPath path;
    {
        fe::FieldPtrs none = {};
        const fe::TiledArgs probe = tiled_args(FE_FAMILY_FACEMASS, J, R, none, 1, E, 3, Np, nf, Nfp, 0, jfe, rifj);
        // below ~10 rows the tiled kernel has too few busy lanes to beat the plain one (measured:
        // triangles p = 2, Np = 6: 1.4 vs 2.2 TFLOP/s; p = 4, Np = 15: 3.1 vs 2.6)
        const bool tiled_ok = tiled_fits(probe) && (variant == FE_VARIANT_TILED || Np >= 10);
        if (int rc = choose_path(variant, mfma_ok, tiled_ok,
                                 "face-mass (MFMA: tetrahedra p = 1..4 or triangles p = 1..5, b >= 2)", Np, &path))
            return rc;
    }
    if (path == kPathTiled) {   // groups of up to kMaxFields fields share the staged operator
        for (int k0 = 0; k0 < b; k0 += fe::kMaxFields) {
            const int nb = b - k0 < fe::kMaxFields ? b - k0 : fe::kMaxFields;
            fe::FieldPtrs P = {};
            for (int k = 0; k < nb; ++k) { P.v[k] = v[k0 + k]; P.out[k] = out[k0 + k]; }
            if (int rc = launch_tiled(tiled_args(FE_FAMILY_FACEMASS, J, R, P, nb, E, 3, Np, nf, Nfp, 0, jfe, rifj), s))
                return rc;
        }
        return FE_OK;
    }
    const int64_t jEs = jfe ? 1 : nf, jFs = jfe ? E : 1;
    const int rF = rifj == 0 ? Np * Nfp : rifj == 1 ? Nfp : rifj == 2 ? Nfp * Np : Np;
    const int rI = rifj == 0 ? Nfp : rifj == 1 ? nf * Nfp : 1;
    const int rJ = rifj == 0 || rifj == 1 ? 1 : rifj == 2 ? Np : nf * Np;
    const bool use_mfma = path == kPathMfma;
    const int64_t nTiles = use_mfma ? E / geo.tel : 0;      // full wave tiles
    const int64_t e_done = nTiles > 0 ? E : 0;   // an MFMA launch covers the remainder too
    const int max_group = use_mfma ? geo.max_group : fe::kMaxFields;
    // fields go in groups of up to max_group per launch; never leave a group of 1 for the MFMA kernel
    for (int k0 = 0; k0 < b;) {
        int nb = (b - k0 < max_group) ? b - k0 : max_group;
        if (use_mfma && b - k0 - nb == 1) nb -= 1;
        fe::FieldPtrs P;
        for (int k = 0; k < fe::kMaxFields; ++k) {
            P.v[k] = v[k0 + (k < nb ? k : 0)];
            P.out[k] = out[k0 + (k < nb ? k : 0)];
        }
        if (nTiles > 0) {
            int rc = FE_OK;
            if (nf == 3) {
                switch (Np) {   // triangles; wave tile = 16 M elements
                    case 21: rc = launch_fm<21, 6, 3, 3>(J, R, prep, P, nb, E, nTiles, jfe, rifj, s); break;
                    case 15: rc = launch_fm<15, 5, 4, 3>(J, R, prep, P, nb, E, nTiles, jfe, rifj, s); break;
                    case 10: rc = launch_fm<10, 4, 5, 3>(J, R, prep, P, nb, E, nTiles, jfe, rifj, s); break;
                    case 6: rc = launch_fm<6, 3, 6, 3>(J, R, prep, P, nb, E, nTiles, jfe, rifj, s); break;
                    default: rc = launch_fm<3, 2, 8, 3>(J, R, prep, P, nb, E, nTiles, jfe, rifj, s); break;
                }
            } else {
                switch (Np) {   // tetrahedra
                    case 56: rc = launch_fm<56, 21, 1, fe::kFmNf, true>(J, R, prep, P, nb, E, nTiles, jfe, rifj, s); break;
                    case 35: rc = launch_fm<35, 15, 1>(J, R, prep, P, nb, E, nTiles, jfe, rifj, s); break;
                    case 20: rc = launch_fm<20, 10, 1>(J, R, prep, P, nb, E, nTiles, jfe, rifj, s); break;
                    case 10: rc = launch_fm<10, 6, 2>(J, R, prep, P, nb, E, nTiles, jfe, rifj, s); break;
                    default: rc = launch_fm<4, 3, 4>(J, R, prep, P, nb, E, nTiles, jfe, rifj, s); break;
                }
            }
            if (rc != FE_OK) return rc;
        }
        if (e_done < E) {
            const dim3 grid(generic_grid(E - e_done, Np)), block(256);
#define FE_FM_CASE(NB)                                                                             \
    case NB:                                                                                       \
        hipLaunchKernelGGL(fe::facemass_generic_kernel<NB>, grid, block, 0, s, J, R, P, E, Np, nf, \
                           Nfp, jEs, jFs, rF, rI, rJ, e_done);                                     \
        break;
            switch (nb) {
                FE_FM_CASE(1) FE_FM_CASE(2) FE_FM_CASE(3) FE_FM_CASE(4)
                FE_FM_CASE(5) FE_FM_CASE(6) FE_FM_CASE(7) FE_FM_CASE(8)
            }
#undef FE_FM_CASE
        }
        k0 += nb;
    }
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

int fe_waveop3d_f64(const double* J, const double* D, const double* u_grad, double* grad_out,
                    const double* v_div, double* div_out, const double* Jface, const double* R,
                    const double* const* f, double* const* lift, int64_t E, int32_t Np, int32_t nf,
                    int32_t Nfp, int32_t b, int32_t fm_layout_flags, int32_t variant, void* stream) {
    return fe_waveop3d_prepared_f64(J, D, nullptr, u_grad, grad_out, v_div, div_out, Jface, R, nullptr, f, lift, E, Np, nf,
                                    Nfp, b, fm_layout_flags, variant, stream);
}

int fe_waveop3d_prepared_f64(const double* J, const double* D, const void* D_prepared, const double* u_grad,
                             double* grad_out, const double* v_div, double* div_out, const double* Jface,
                             const double* R, const void* R_prepared, const double* const* f, double* const* lift,
                             int64_t E, int32_t Np, int32_t nf, int32_t Nfp, int32_t b, int32_t fm_layout_flags,
                             int32_t variant, void* stream) {
    int perr;
    const void* prepD = usable_prepared(D_prepared, D, kPreparedD, Np, 0, 0, 0, "waveop", &perr);
    if (perr != FE_OK) return perr;
    const void* prepR = usable_prepared(R_prepared, R, kPreparedR, Np, nf, Nfp, fm_layout_flags & ~FE_FM_J_FE, "waveop", &perr);
    if (perr != FE_OK) return perr;
    FmChoice geo{0, 16};
    const bool fused = (variant == FE_VARIANT_AUTO || variant == FE_VARIANT_MFMA) && nf == fe::kFmNf && Np != 56 &&
                       fm_mfma_geometry(Np, nf, Nfp, &geo) && b >= 2 &&
                       b <= 4 && f && lift && E > 0 && !(fm_layout_flags & ~7);
    if (!fused) {   // three launches (argument checks included)
        if (int rc = fe_graddiv3d_prepared_f64(J, D, D_prepared, u_grad, v_div, grad_out, div_out, E, Np, variant, stream))
            return rc;
        return fe_facemass_prepared_f64(Jface, R, R_prepared, f, lift, E, Np, nf, Nfp, b, fm_layout_flags, variant, stream);
    }
    if (int rc = check_common(J, D, u_grad, grad_out, E, Np)) return rc;
    if (int rc = check_common(J, D, v_div, div_out, E, Np)) return rc;
    if (variant < FE_VARIANT_AUTO || variant > FE_VARIANT_TILED)
        return fail(FE_EUNSUPPORTED, "waveop: unknown variant %d", variant);
    if (!Jface || !R) return fail(FE_EINVAL, "waveop: null device pointer");
    uintptr_t bits = reinterpret_cast<uintptr_t>(Jface) | reinterpret_cast<uintptr_t>(R);
    for (int k = 0; k < b; ++k) {
        if (!f[k] || !lift[k]) return fail(FE_EINVAL, "waveop: null field pointer %d", k);
        bits |= reinterpret_cast<uintptr_t>(f[k]) | reinterpret_cast<uintptr_t>(lift[k]);
    }
    if (bits & 7u) return fail(FE_EINVAL, "waveop: device pointers must be 8-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const fe::GradFields Pg = grad_fields(J, &u_grad, &grad_out, 1, E, Np);
    fe::FieldPtrs Pd = {}, Pf = {};
    Pd.v[0] = v_div;  Pd.out[0] = div_out;
    for (int k = 0; k < fe::kMaxFields; ++k) {
        Pf.v[k] = f[k < b ? k : 0];
        Pf.out[k] = lift[k < b ? k : 0];
    }
    fe::WaveOpArgs a = {};
    a.J = J; a.D = D; a.Jf = Jface; a.R = R; a.E = E;
    a.prepD = prepD; a.prepR = prepR;
    a.jfe = (fm_layout_flags & FE_FM_J_FE) ? 1 : 0;
    a.rlayout = ((fm_layout_flags & FE_FM_R_IFJ) ? 1 : 0) + ((fm_layout_flags & FE_FM_R_T) ? 2 : 0);
    bool launched = false;
    int rc = FE_OK;
    switch (Np) {   // the (Np, M) geometries of the three separate launches
        case 35: rc = launch_waveop<35, 15, 1, 1, 1>(a, Pg, Pd, Pf, b, s, &launched); break;
        case 20: rc = launch_waveop<20, 10, 2, 1, 1>(a, Pg, Pd, Pf, b, s, &launched); break;
        case 10: rc = launch_waveop<10, 6, 3, 3, 2>(a, Pg, Pd, Pf, b, s, &launched); break;
        default: rc = launch_waveop<4, 3, 5, 5, 4>(a, Pg, Pd, Pf, b, s, &launched); break;
    }
    if (rc != FE_OK) return rc;
    if (!launched) {   // fewer elements than any wave tile: the generic kernels
        if (int rc2 = fe_graddiv3d_f64(J, D, u_grad, v_div, grad_out, div_out, E, Np, variant, stream)) return rc2;
        return fe_facemass_f64(Jface, R, f, lift, E, Np, nf, Nfp, b, fm_layout_flags, variant, stream);
    }
    (void)prepD; (void)prepR;
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

int fe_einsum_generic(const fe_einsum_desc* d, const void* const* operands, void* out,
                      void* stream) {
    if (!d || !operands) return fail(FE_EINVAL, "einsum: null descriptor");
    if (d->n_operands < 1 || d->n_operands > FE_MAX_EINSUM_OPERANDS || d->n_out < 0 ||
        d->n_out > FE_MAX_EINSUM_INDICES || d->n_sum < 0 || d->n_sum > FE_MAX_EINSUM_INDICES)
        return fail(FE_EINVAL, "einsum: %d operands / %d output / %d summation indices out of range",
                    d->n_operands, d->n_out, d->n_sum);
    if (d->dtype != FE_DTYPE_F64 && d->dtype != FE_DTYPE_F32)
        return fail(FE_EUNSUPPORTED, "einsum: dtype code %d not compiled (float64 / float32 only)", d->dtype);
    int64_t n_out = 1, n_sum = 1;
    for (int k = 0; k < d->n_out; ++k) {
        if (d->out_extent[k] < 0) return fail(FE_EINVAL, "einsum: negative extent");
        n_out *= d->out_extent[k];
    }
    for (int k = 0; k < d->n_sum; ++k) {
        if (d->sum_extent[k] < 0) return fail(FE_EINVAL, "einsum: negative extent");
        n_sum *= d->sum_extent[k];
    }
    if (n_out == 0) return FE_OK;
    if (!out) return fail(FE_EINVAL, "einsum: null output pointer");
    if (n_out >= ((int64_t)1 << 39)) return fail(FE_EINVAL, "einsum: output too large");
    fe_einsum_ptrs P;
    for (int p = 0; p < FE_MAX_EINSUM_OPERANDS; ++p) P.p[p] = nullptr;
    for (int p = 0; p < d->n_operands; ++p) {
        if (!operands[p] && n_sum > 0) return fail(FE_EINVAL, "einsum: null operand %d", p);
        P.p[p] = operands[p];
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const dim3 block(256);
    const bool f64 = d->dtype == FE_DTYPE_F64;
    if (d->n_sum > 0 && n_sum == 0) {   // a summation index of extent 0: every output entry is an empty sum
        FE_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)n_out * (f64 ? 8 : 4), s));
        return FE_OK;
    }

    // pointwise product of congruent contiguous operands: a vectorised stream
    bool pointwise = d->n_sum == 0;
    for (int p = 0; p < d->n_operands && pointwise; ++p) {
        int64_t dense = 1;
        for (int k = d->n_out - 1; k >= 0; --k) {
            if (d->out_extent[k] != 1 && d->op_out_stride[p][k] != dense) pointwise = false;
            dense *= d->out_extent[k];
        }
    }
    if (pointwise) {
        const int64_t want = (n_out / 2 + 255) / 256 + 1, cap = 32 * (int64_t)device_cu_count();
        const dim3 grid((unsigned)(want < cap ? want : cap));
        if (f64)
            hipLaunchKernelGGL(fe::einsum_pointwise_kernel<double>, grid, block, 0, s, P, d->n_operands,
                               static_cast<double*>(out), n_out);
        else
            hipLaunchKernelGGL(fe::einsum_pointwise_kernel<float>, grid, block, 0, s, P, d->n_operands,
                               static_cast<float*>(out), n_out);
        FE_HIP_CHECK(hipGetLastError());
        return FE_OK;
    }

    // lanes per output entry: > 1 when the fastest summation index is contiguous in an operand
    int group = 1;
    if (d->n_sum > 0 && n_sum >= 8) {
        bool contiguous = false;
        for (int p = 0; p < d->n_operands; ++p) contiguous |= d->op_sum_stride[p][d->n_sum - 1] == 1;
        if (contiguous) group = n_sum >= 32 ? 16 : 4;
    }
    if (n_out * group >= ((int64_t)1 << 39)) group = 1;
    const dim3 grid((unsigned)((n_out * group + 255) / 256));
#define FE_EINSUM_CASE(T, G) \
    hipLaunchKernelGGL((fe::einsum_generic_kernel<T, G>), grid, block, 0, s, *d, P, static_cast<T*>(out), n_out, n_sum)
    if (f64) {
        if (group == 16) FE_EINSUM_CASE(double, 16);
        else if (group == 4) FE_EINSUM_CASE(double, 4);
        else FE_EINSUM_CASE(double, 1);
    } else {
        if (group == 16) FE_EINSUM_CASE(float, 16);
        else if (group == 4) FE_EINSUM_CASE(float, 4);
        else FE_EINSUM_CASE(float, 1);
    }
#undef FE_EINSUM_CASE
    FE_HIP_CHECK(hipGetLastError());
    return FE_OK;
}

#ifdef FE_EXPERIMENTS
int fe_dbg_read_clock(unsigned long long* out2) {
    FE_HIP_CHECK(hipDeviceSynchronize());
    FE_HIP_CHECK(hipMemcpyFromSymbol(out2, HIP_SYMBOL(fe::fe_dbg_clock), 16));
    return FE_OK;
}
int fe_dbg_read_w8(unsigned long long* out, int n_waves) {
    FE_HIP_CHECK(hipDeviceSynchronize());
    FE_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(fe::fe_dbg_w8), (size_t)n_waves * 64));
    return FE_OK;
}
int fe_dbg_read_phase(unsigned long long* out, int n_waves) {
    FE_HIP_CHECK(hipDeviceSynchronize());
    FE_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(fe::fe_dbg_phase), (size_t)n_waves * 32));
    return FE_OK;
}
int fe_dbg_read_tiles(unsigned long long* out, int n_waves) {
    FE_HIP_CHECK(hipDeviceSynchronize());
    FE_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(fe::fe_dbg_tile), (size_t)n_waves * 128));
    return FE_OK;
}
int fe_dbg_read_stamps(unsigned long long* out, int n_waves) {
    FE_HIP_CHECK(hipDeviceSynchronize());
    FE_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(fe::fe_dbg_stamps), (size_t)n_waves * 32));
    return FE_OK;
}
#endif

int fe_set_tail_rounds(int32_t rounds) {
    const int before = g_tail_rounds.exchange(rounds);
    return before;
}

int fe_set_temporal_loads_mib(int32_t mib) {
    return (int)(g_temporal_input_bytes.exchange(mib < 0 ? 0 : (long long)mib << 20) >> 20);
}

int fe_set_write_through_mib(int32_t mib) {
    return (int)(g_write_through_output_bytes.exchange(mib < 0 ? 0 : (long long)mib << 20) >> 20);
}

int fe_last_launch_info(int64_t* out, int32_t n) {
    if (!out || n < 1) return fail(FE_EINVAL, "fe_last_launch_info: bad arguments");
    const LastLaunch& L = g_last_launch;
    const int64_t v[FE_LAST_LAUNCH_INFO] = {L.valid, L.dynamic_walk, L.temporal_loads, L.write_through, L.blocks, L.waves_per_block, L.kind, L.bodies,
                                            L.tiles, L.static_tiles};
    for (int k = 0; k < n && k < FE_LAST_LAUNCH_INFO; ++k) out[k] = v[k];
    return n < FE_LAST_LAUNCH_INFO ? n : FE_LAST_LAUNCH_INFO;
}

int fe_set_div_quarter_tail(int32_t on) {
    return g_div_quarter_tail.exchange(on ? 1 : 0);
}

int fe_set_grad_quarter_tail(int32_t on) {
    return g_grad_quarter_tail.exchange(on < 0 ? 0 : (on == 2 || on == 3) ? 1 : on);
}

int fe_set_grad_staggered_start(int32_t on) {
    return g_grad_staggered_start.exchange(on ? 1 : 0);
}

int fe_set_tail_min_rounds(int32_t rounds) {
    return g_tail_min_rounds_v.exchange(rounds < 2 ? 2 : rounds);
}

int64_t fe_set_div_interleave(int64_t tiles) {
    return g_div_interleave_tiles.exchange(tiles < 0 ? 0 : tiles);
}

int fe_set_phase_priority_p5(int32_t on) {
    return g_phase_priority_p5.exchange(on ? 1 : 0);
}

int fe_set_cu_limit(int32_t cus) {
    return g_cu_limit.exchange(cus < 0 ? 0 : cus);
}

int fe_stream_retired(void* stream) {
    int dev = 0;
    FE_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(FE_EINVAL, "fe_stream_retired: device %d", dev);
    TailPool& pool = g_tail[dev];
    std::lock_guard<std::mutex> guard(pool.lock);
    auto it = pool.by_stream.find(tail_stream_key(static_cast<hipStream_t>(stream)));
    if (it == pool.by_stream.end()) return 0;
    pool.spare.push_back(it->second);   // zero, like every group between launches
    pool.verified_at.erase(it->first);
    pool.by_stream.erase(it);
    return 1;
}

int fe_capture_id(void* stream, uint64_t* id) {
    if (!id) return fail(FE_EINVAL, "fe_capture_id: null result pointer");
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long cid = 0;
    FE_HIP_CHECK(hipStreamGetCaptureInfo(static_cast<hipStream_t>(stream), &st, &cid));
    *id = st == hipStreamCaptureStatusNone ? 0 : (uint64_t)cid;
    return FE_OK;
}

int fe_graph_retired(uint64_t capture_id) {
    int dev = 0;
    FE_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(FE_EINVAL, "fe_graph_retired: device %d", dev);
    TailPool& pool = g_tail[dev];
    std::lock_guard<std::mutex> guard(pool.lock);
    auto it = pool.by_capture.find((unsigned long long)capture_id);
    if (it == pool.by_capture.end()) return 0;
    const int n = (int)it->second.size();
    for (unsigned* g : it->second) pool.spare.push_back(g);   // zero, like every group between launches
    pool.captured -= n;
    pool.by_capture.erase(it);
    return n;
}

int fe_tail_stats(int64_t* out, int32_t n) {
    if (!out || n < 1) return fail(FE_EINVAL, "fe_tail_stats: bad arguments");
    int dev = 0;
    FE_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(FE_EINVAL, "fe_tail_stats: device %d", dev);
    TailPool& pool = g_tail[dev];
    std::lock_guard<std::mutex> guard(pool.lock);
    const int64_t v[FE_TAIL_STATS] = {pool.groups, (int64_t)pool.by_stream.size(), pool.captured, (int64_t)pool.spare.size(), pool.exhausted,
                                      pool.static_fallbacks, pool.grow_failures, pool.verified_after_error, pool.repaired_after_error,
                                      (int64_t)pool.by_capture.size(), kTailMaxGroups, (int64_t)g_hip_error_epoch.load()};
    for (int k = 0; k < n && k < FE_TAIL_STATS; ++k) out[k] = v[k];
    return n < FE_TAIL_STATS ? n : FE_TAIL_STATS;
}

int fe_tail_plant(void* stream, uint32_t value) {
    int dev = 0;
    FE_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(FE_EINVAL, "fe_tail_plant: device %d", dev);
    unsigned* group = nullptr;
    {
        TailPool& pool = g_tail[dev];
        std::lock_guard<std::mutex> guard(pool.lock);
        auto it = pool.by_stream.find(tail_stream_key(static_cast<hipStream_t>(stream)));
        if (it == pool.by_stream.end()) return fail(FE_EINVAL, "fe_tail_plant: the stream owns no counter group (no dynamic launch yet)");
        group = it->second;
    }
    FE_HIP_CHECK(hipDeviceSynchronize());
    FE_HIP_CHECK(hipMemcpy(group + 3 * fe::kTailStride, &value, sizeof(value), hipMemcpyHostToDevice));   // pool 3's ticket counter
    return FE_OK;
}

int fe_tail_check(int32_t repair, int64_t* dirty_words, int32_t* groups, int32_t* streams, int32_t* captured) {
    int dev = 0;
    FE_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(FE_EINVAL, "fe_tail_check: device %d", dev);
    FE_HIP_CHECK(hipDeviceSynchronize());
    TailPool& pool = g_tail[dev];
    std::lock_guard<std::mutex> guard(pool.lock);
    long long dirty = 0;
    for (unsigned* chunk : pool.chunks)
        for (int g = 0; g < kTailChunkGroups; ++g)
            if (int rc = tail_group_dirty(chunk + (size_t)g * kTailGroupWords, repair != 0, &dirty)) return rc;
    if (dirty_words) *dirty_words = dirty;
    if (groups) *groups = pool.groups;
    if (streams) *streams = (int32_t)pool.by_stream.size();
    if (captured) *captured = pool.captured;
    return FE_OK;
}

int fe_launch_f32(int32_t family, const fe_argpack* a, void* stream) {
    if (!a) return fail(FE_EINVAL, "fe_launch_f32: null argument pack");
    if (a->E < 0 || a->Np <= 0) return fail(FE_EINVAL, "fe_launch_f32: bad sizes (E=%lld Np=%d)", (long long)a->E, a->Np);
    if (a->E == 0) return FE_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ndim = a->ndim == 2 ? 2 : 3;
    // field pointers: the b-field form (v / outs) when given, else the single pair (u / out)
    const int b = a->b > 0 ? a->b : 1;
    const double* const* vin = a->v;
    double* const* vout = a->outs;
    const double* one_in[1] = {a->u};
    double* one_out[1] = {a->out};
    if (!vin || !vout) {
        if (b != 1) return fail(FE_EINVAL, "fe_launch_f32: b = %d needs the v / outs pointer arrays", b);
        vin = one_in;
        vout = one_out;
    }
    int opT = 0, jl = 0, rl = 0, nf = 0, Nfp = 0;
    switch (family) {
        case FE_FAMILY_GRAD:
        case FE_FAMILY_DIV:
        case FE_FAMILY_MATAPPLY:
            if (a->layout_flags & ~FE_OP_TRANSPOSED) return fail(FE_EINVAL, "fe_launch_f32: bad operator flags %d", a->layout_flags);
            opT = (a->layout_flags & FE_OP_TRANSPOSED) ? 1 : 0;
            break;
        case FE_FAMILY_DIVCOMP:
            if (a->layout_flags & ~(FE_OP_TRANSPOSED | FE_OP_J_ES)) return fail(FE_EINVAL, "fe_launch_f32: bad operator flags %d", a->layout_flags);
            opT = (a->layout_flags & FE_OP_TRANSPOSED) ? 1 : 0;
            jl = (a->layout_flags & FE_OP_J_ES) ? 1 : 0;
            break;
        case FE_FAMILY_FACEMASS:
            if (a->nf <= 0 || a->Nfp <= 0) return fail(FE_EINVAL, "fe_launch_f32: face-mass needs nf and Nfp");
            nf = a->nf;
            Nfp = a->Nfp;
            jl = (a->layout_flags & FE_FM_J_FE) ? 1 : 0;
            rl = ((a->layout_flags & FE_FM_R_IFJ) ? 1 : 0) + ((a->layout_flags & FE_FM_R_T) ? 2 : 0);
            break;
        default:
            return fail(FE_EUNSUPPORTED, "fe_launch_f32: family %d has no float32 kernel", family);
    }
    if (!a->D || (family != FE_FAMILY_MATAPPLY && !a->J)) return fail(FE_EINVAL, "fe_launch_f32: null device pointer");
    // grad of tetrahedra p = 4 on the matrix cores (fe_grad_f32.h): 16-byte aligned operands, E a multiple of 4 (so that
    // every row of J and every output plane starts on a 16-byte boundary) and at least one full tile; else the tiled kernel
    // div and face-mass likewise (fe_div_f32.h, fe_facemass_f32.h)
    const bool grad_lower = family == FE_FAMILY_GRAD && ndim == 3 && (a->Np == 20 || a->Np == 10 || a->Np == 4);   // p = 1 ... 3 (round 4)
    const bool div_lower = family == FE_FAMILY_DIV && ndim == 3 && (a->Np == 20 || a->Np == 10 || a->Np == 4);        // p = 1 ... 3 (round 5)
    const bool fm_lower = family == FE_FAMILY_FACEMASS && nf == 4 &&
                          ((a->Np == 20 && Nfp == 10) || (a->Np == 10 && Nfp == 6) || (a->Np == 4 && Nfp == 3));          // p = 1 ... 3 (round 5)
    const bool mfma_shape = grad_lower || div_lower || (family == FE_FAMILY_GRAD && ndim == 3 && a->Np == 35) || (family == FE_FAMILY_DIV && ndim == 3 && a->Np == 35) ||
                            (family == FE_FAMILY_FACEMASS && a->Np == 35 && nf == 4 && Nfp == 15) || fm_lower;
    if (mfma_shape && a->variant != FE_VARIANT_TILED && a->E % 4 == 0 && a->E >= 16) {
        bool aligned = ((reinterpret_cast<uintptr_t>(a->J) | reinterpret_cast<uintptr_t>(a->D)) & 15u) == 0;
        for (int k = 0; k < b; ++k)
            aligned = aligned && vin[k] && vout[k] && ((reinterpret_cast<uintptr_t>(vin[k]) | reinterpret_cast<uintptr_t>(vout[k])) & 15u) == 0;
        if (aligned && div_lower) {   // fe_div_f32.h: the kernel over the geometry (Np, M)
            auto go_np = [&](auto geom, auto kernel, const char* what, PerDeviceOnce& once) -> int {
                using G = decltype(geom);
                const int64_t nTiles = a->E / G::TEL;
                if (nTiles == 0) return 1;   // too few elements for a wave tile: the tiled kernel
                if (int rc = configured(once, kernel, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU)) return rc;
                int64_t blocks = (nTiles + G::WAVES - 1) / G::WAVES;
                const int64_t cap = (int64_t)G::BLOCKS_PER_CU * device_cu_count();
                if (blocks > cap) blocks = cap;
                for (int k = 0; k < b; ++k)
                    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), G::LDS_BYTES, s, reinterpret_cast<const float*>(a->J),
                                       reinterpret_cast<const float*>(a->D), reinterpret_cast<const float*>(vin[k]),
                                       reinterpret_cast<float*>(vout[k]), a->E, nTiles, opT);
                FE_HIP_CHECK(hipGetLastError());
                return FE_OK;
            };
            static PerDeviceOnce once20, once10, once4;
            const int rc = a->Np == 20 ? go_np(fe::DivF32GeomT<20, 1>{}, fe::div3d_mfma_f32_np_kernel<20, 1>, "div float32 Np=20 M=1", once20)
                           : a->Np == 10 ? go_np(fe::DivF32GeomT<10, 3>{}, fe::div3d_mfma_f32_np_kernel<10, 3>, "div float32 Np=10 M=3", once10)
                                         : go_np(fe::DivF32GeomT<4, 5>{}, fe::div3d_mfma_f32_np_kernel<4, 5>, "div float32 Np=4 M=5", once4);
            if (rc <= 0) return rc;
        } else if (aligned && family == FE_FAMILY_DIV) {
            // the measured alternatives (profiles/r03/float32_div_facemass.txt) stay selectable in the experiment build
#ifdef FE_EXPERIMENTS
            static const int ring = [] { const char* e = getenv("FEINSUM_F32_DIV_RING"); return e && atoi(e) == 1 ? 1 : 2; }();
            static const int small = [] { const char* e = getenv("FEINSUM_F32_SMALL"); return e ? atoi(e) : 1; }();
#else
            constexpr int ring = 2, small = 1;
#endif
            const int64_t nTiles = a->E / 16;
            auto go = [&](auto geom, auto kernel, const char* what, PerDeviceOnce& once) -> int {
                using G = decltype(geom);
                if (int rc = configured(once, kernel, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU)) return rc;
                int64_t blocks = (nTiles + G::WAVES - 1) / G::WAVES;
                const int64_t cap = (int64_t)G::BLOCKS_PER_CU * device_cu_count();
                if (blocks > cap) blocks = cap;
                for (int k = 0; k < b; ++k)
                    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), G::LDS_BYTES, s, reinterpret_cast<const float*>(a->J),
                                       reinterpret_cast<const float*>(a->D), reinterpret_cast<const float*>(vin[k]),
                                       reinterpret_cast<float*>(vout[k]), a->E, nTiles, opT);
                FE_HIP_CHECK(hipGetLastError());
                return FE_OK;
            };
            static PerDeviceOnce once2;
#ifdef FE_EXPERIMENTS
            static PerDeviceOnce once1, once3;
            static const int f32dbg = [] { const char* e = getenv("FEINSUM_F32_DBG"); return e ? atoi(e) : 0; }();   // parts of the tile work removed
            static PerDeviceOnce onced[16];
#define FE_F32DBG_CASE(D) case D: return go(fe::DivF32Geom<2>{}, fe::div3d_mfma_f32_kernel<2, true, D>, "div float32 experiment", onced[D]);
            switch (f32dbg) {
                FE_F32DBG_CASE(1) FE_F32DBG_CASE(2) FE_F32DBG_CASE(3) FE_F32DBG_CASE(4) FE_F32DBG_CASE(5) FE_F32DBG_CASE(6) FE_F32DBG_CASE(7)
                FE_F32DBG_CASE(8) FE_F32DBG_CASE(9) FE_F32DBG_CASE(12) FE_F32DBG_CASE(15)
                default: break;
            }
#undef FE_F32DBG_CASE
            if (ring == 1) return go(fe::DivF32Geom<1>{}, fe::div3d_mfma_f32_kernel<1, false>, "div float32 Np=35 one buffer", once1);
            if (!small) return go(fe::DivF32Geom<2>{}, fe::div3d_mfma_f32_kernel<2, false>, "div float32 Np=35 three row tiles", once3);
#endif
            return go(fe::DivF32Geom<2>{}, fe::div3d_mfma_f32_kernel<2, true>, "div float32 Np=35", once2);
        }
        if (aligned && fm_lower) {   // fe_facemass_f32.h: the kernel over the geometry (Np, Nfp, M)
            auto go_np = [&](auto geom, auto kernel, const char* what, PerDeviceOnce& once) -> int {
                using G = decltype(geom);
                const int64_t nTiles = a->E / G::TEL;
                if (nTiles == 0) return 1;   // too few elements for a wave tile: the tiled kernel
                if (int rc = configured(once, kernel, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU)) return rc;
                int64_t blocks = (nTiles + G::WAVES - 1) / G::WAVES;
                const int64_t cap = (int64_t)G::BLOCKS_PER_CU * device_cu_count();
                if (blocks > cap) blocks = cap;
                for (int k0 = 0; k0 < b; k0 += fe::kMaxFields) {   // groups of up to kMaxFields fields share J and the fragments
                    const int nb = b - k0 < fe::kMaxFields ? b - k0 : fe::kMaxFields;
                    fe::FieldPtrs P;
                    for (int k = 0; k < fe::kMaxFields; ++k) {
                        P.v[k] = vin[k0 + (k < nb ? k : 0)];
                        P.out[k] = vout[k0 + (k < nb ? k : 0)];
                    }
                    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), G::LDS_BYTES, s, reinterpret_cast<const float*>(a->J),
                                       reinterpret_cast<const float*>(a->D), P, nb, a->E, nTiles, jl, rl);
                }
                FE_HIP_CHECK(hipGetLastError());
                return FE_OK;
            };
            static PerDeviceOnce once20, once10, once4;
            const int rc = a->Np == 20 ? go_np(fe::FmF32GeomT<20, 10, 1>{}, fe::facemass_mfma_f32_np_kernel<20, 10, 1>, "face-mass float32 Np=20 M=1", once20)
                           : a->Np == 10 ? go_np(fe::FmF32GeomT<10, 6, 2>{}, fe::facemass_mfma_f32_np_kernel<10, 6, 2>, "face-mass float32 Np=10 M=2", once10)
                                         : go_np(fe::FmF32GeomT<4, 3, 4>{}, fe::facemass_mfma_f32_np_kernel<4, 3, 4>, "face-mass float32 Np=4 M=4", once4);
            if (rc <= 0) return rc;
        } else if (aligned && family == FE_FAMILY_FACEMASS) {
            using G = fe::FmF32Geom;
            static PerDeviceOnce once;
            auto kernel = fe::facemass_mfma_f32_kernel<true>;
            const char* what = "face-mass float32 Np=35";
            PerDeviceOnce* flag = &once;
#ifdef FE_EXPERIMENTS
            static const int small = [] { const char* e = getenv("FEINSUM_F32_SMALL"); return e ? atoi(e) : 1; }();
            static PerDeviceOnce once3;
            if (!small) { kernel = fe::facemass_mfma_f32_kernel<false>; what = "face-mass float32 Np=35 three row tiles"; flag = &once3; }
#endif
            if (int rc = configured(*flag, kernel, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU)) return rc;
            const int64_t nTiles = a->E / G::TEL;
            int64_t blocks = (nTiles + G::WAVES - 1) / G::WAVES;
            const int64_t cap = (int64_t)G::BLOCKS_PER_CU * device_cu_count();
            if (blocks > cap) blocks = cap;
            for (int k0 = 0; k0 < b; k0 += fe::kMaxFields) {   // groups of up to kMaxFields fields share J and the fragments
                const int nb = b - k0 < fe::kMaxFields ? b - k0 : fe::kMaxFields;
                fe::FieldPtrs P;
                for (int k = 0; k < fe::kMaxFields; ++k) {
                    P.v[k] = vin[k0 + (k < nb ? k : 0)];
                    P.out[k] = vout[k0 + (k < nb ? k : 0)];
                }
                hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), G::LDS_BYTES, s, reinterpret_cast<const float*>(a->J),
                                   reinterpret_cast<const float*>(a->D), P, nb, a->E, nTiles, jl, rl);
            }
            FE_HIP_CHECK(hipGetLastError());
            return FE_OK;
        }
        if (aligned) {
            // two 16-element sub-tiles per wave iteration from E = 2e5 on (round 4: 4.5 KB store bursts, two blocks per CU: 68.1 ->
            // 70.5 % of the float32 roofline at 1e6, 70.2 -> 75.5 % at 4e6; below 2e5 the three-blocks-per-CU kernel of one sub-tile
            // is faster: 15.3 against 17.7 us at 1e5; FEINSUM_F32_M=1 / 2 forces either; profiles/r04/float32_grad_two_subtiles.txt)
            static const int m_env = [] { const char* e = getenv("FEINSUM_F32_M"); return e ? atoi(e) : 0; }();
            const bool m1 = m_env == 1 || (m_env != 2 && a->E < 200000);
            // `tail_kernel`: the same kernel with a dynamic tail (behind two static rounds the tiles come by tickets; from five rounds
            // on, as for float64) or nullptr; every launch of the loop runs on `s`, so they use the stream's counters in turn
            auto launch = [&](auto geom, auto kernel, auto tail_kernel, const char* what) -> int {
                using G = decltype(geom);
                static PerDeviceOnce once, once_tail;
                if (int rc = configured(once, kernel, what, G::LDS_BYTES, 256, G::BLOCKS_PER_CU)) return rc;
                const int64_t nTiles = a->E / G::TEL;
                if (nTiles == 0) return 1;   // too few elements for a wave tile: the tiled kernel
                int64_t blocks = (nTiles + G::WAVES - 1) / G::WAVES;
                const int64_t cap = (int64_t)G::BLOCKS_PER_CU * device_cu_count();
                if (blocks > cap) blocks = cap;
                const int flags = opT | temporal_flag((9 + (int64_t)a->Np) * a->E * 4);
                int64_t t_static = nTiles;
                if constexpr (!std::is_same_v<decltype(tail_kernel), std::nullptr_t>) {
                    t_static = tail_static_tiles(nTiles, blocks, G::WAVES);
                    if (t_static < nTiles)
                        if (int rc = configured(once_tail, tail_kernel, what, G::LDS_BYTES, 256, 2)) return rc;
                }
                for (int k = 0; k < b; ++k) {
                    if constexpr (!std::is_same_v<decltype(tail_kernel), std::nullptr_t>) {
                        unsigned* tail = t_static < nTiles ? tail_slot(s) : nullptr;
                        if (tail) {
                            hipLaunchKernelGGL(tail_kernel, dim3((unsigned)blocks), dim3(256), G::LDS_BYTES, s,
                                               reinterpret_cast<const float*>(a->J), reinterpret_cast<const float*>(a->D),
                                               reinterpret_cast<const float*>(vin[k]), reinterpret_cast<float*>(vout[k]), a->E, nTiles, flags,
                                               tail, t_static);
                            continue;
                        }
                    }
                    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(256), G::LDS_BYTES, s,
                                       reinterpret_cast<const float*>(a->J), reinterpret_cast<const float*>(a->D),
                                       reinterpret_cast<const float*>(vin[k]), reinterpret_cast<float*>(vout[k]), a->E, nTiles, flags);
                }
                FE_HIP_CHECK(hipGetLastError());
                return FE_OK;
            };
            const int rc = a->Np == 20 ? launch(fe::GradF32GeomT<3, 20>{}, fe::grad3d_mfma_f32_kernel<3, 20>, fe::grad3d_mfma_f32_tail_kernel<3, 20>, "grad float32 Np=20 M=3")
                           : a->Np == 10 ? launch(fe::GradF32GeomT<5, 10>{}, fe::grad3d_mfma_f32_kernel<5, 10>, fe::grad3d_mfma_f32_tail_kernel<5, 10>, "grad float32 Np=10 M=5")
                           : a->Np == 4 ? launch(fe::GradF32GeomT<8, 4>{}, fe::grad3d_mfma_f32_kernel<8, 4>, fe::grad3d_mfma_f32_tail_kernel<8, 4>, "grad float32 Np=4 M=8")
                           : m1 ? launch(fe::GradF32GeomT<1>{}, fe::grad3d_mfma_f32_kernel<1>, nullptr, "grad float32 Np=35 M=1")
                                : launch(fe::GradF32GeomT<2>{}, fe::grad3d_mfma_f32_kernel<2>, fe::grad3d_mfma_f32_tail_kernel<2>, "grad float32 Np=35 M=2");
            if (rc <= 0) return rc;
        }
    }
    for (int k0 = 0; k0 < b; k0 += fe::kMaxFields) {   // groups of up to kMaxFields fields share the staged operator
        const int nb = b - k0 < fe::kMaxFields ? b - k0 : fe::kMaxFields;
        fe::FieldPtrs P = {};
        for (int k = 0; k < nb; ++k) {
            if (!vin[k0 + k] || !vout[k0 + k]) return fail(FE_EINVAL, "fe_launch_f32: null field pointer");
            P.v[k] = vin[k0 + k];
            P.out[k] = vout[k0 + k];
        }
        const fe::TiledArgs ta = tiled_args(family, a->J, a->D, P, family == FE_FAMILY_DIVCOMP ? 1 : nb, a->E,
                                            family == FE_FAMILY_MATAPPLY ? 1 : ndim, a->Np, nf, Nfp, opT, jl, rl);
        if (int rc = launch_tiled_t<float>(ta, s)) return rc;
        if (family == FE_FAMILY_DIVCOMP) break;
    }
    return FE_OK;
}

static int launch_family(int32_t family, const fe_argpack* a, void* stream) {
    if (family & FE_FAMILY_F32) return fe_launch_f32(family & ~FE_FAMILY_F32, a, stream);
    switch (family) {
        case FE_FAMILY_GRAD:
            if (a->ndim == 2)
                return fe_grad_f64(a->J, a->D, a->v, a->outs, a->E, 2, a->Np, a->b, a->layout_flags, a->variant, stream);
            if (a->b > 1)
                return fe_grad3d_prepared_f64(a->J, a->D, a->prepared, a->v, a->outs, a->E, a->Np, a->b, a->layout_flags,
                                              a->variant, stream);
            return fe_grad3d_prepared_f64(a->J, a->D, a->prepared, &a->u, &a->out, a->E, a->Np, 1, a->layout_flags,
                                          a->variant, stream);
        case FE_FAMILY_DIV:
            if (a->ndim == 2)
                return fe_div_f64(a->J, a->D, a->v, a->outs, a->E, 2, a->Np, a->b, a->layout_flags, a->variant, stream);
            if (a->b > 1)
                return fe_div3d_prepared_f64(a->J, a->D, a->prepared, a->v, a->outs, a->E, a->Np, a->b, a->layout_flags,
                                             a->variant, stream);
            return fe_div3d_prepared_f64(a->J, a->D, a->prepared, &a->u, &a->out, a->E, a->Np, 1, a->layout_flags,
                                         a->variant, stream);
        case FE_FAMILY_GRADPLANES:
            return fe_gradplanes3d_f64(a->j3, a->D, a->v, a->outs, a->E, a->Np, a->b, a->layout_flags,
                                       a->variant, stream);
        case FE_FAMILY_MATAPPLY:
            return fe_matapply_f64(a->J, a->D, a->v, a->outs, a->E, a->Np, a->b, a->layout_flags, a->variant,
                                   stream);
        case FE_FAMILY_GRADDIV:
            return fe_graddiv3d_prepared_f64(a->J, a->D, a->prepared, a->u, a->v_div, a->out, a->out2, a->E, a->Np,
                                             a->variant, stream);
        case FE_FAMILY_DIVCOMP:
            return fe_divcomp_f64(a->J, a->D, a->u, a->out, a->E, a->ndim == 2 ? 2 : 3, a->Np, a->layout_flags, a->variant,
                                  stream);
        case FE_FAMILY_FACEMASS:
            return fe_facemass_prepared_f64(a->J, a->D, a->prepared, a->v, a->outs, a->E, a->Np, a->nf, a->Nfp, a->b,
                                            a->layout_flags, a->variant, stream);
        default: return fail(FE_EINVAL, "unknown family %d", family);
    }
}

int fe_time_launches(int32_t family, const fe_argpack* args, int32_t n_launches, void* stream,
                     float* ms_out) {
    if (!args || !ms_out || n_launches <= 0)
        return fail(FE_EINVAL, "fe_time_launches: bad arguments");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t t0, t1;
    FE_HIP_CHECK(hipEventCreate(&t0));
    FE_HIP_CHECK(hipEventCreate(&t1));
    int rc = FE_OK;
    hipError_t e = hipEventRecord(t0, s);
    for (int i = 0; i < n_launches && rc == FE_OK && e == hipSuccess; ++i)
        rc = launch_family(family, args, stream);
    if (e == hipSuccess) e = hipEventRecord(t1, s);
    if (e == hipSuccess) e = hipEventSynchronize(t1);
    if (e == hipSuccess && rc == FE_OK) e = hipEventElapsedTime(ms_out, t0, t1);
    hipEventDestroy(t0);
    hipEventDestroy(t1);
    if (rc != FE_OK) return rc;
    if (e != hipSuccess) return fail(FE_EHIP, "fe_time_launches: %s", hipGetErrorString(e));
    return FE_OK;
}

}  // extern "C"
