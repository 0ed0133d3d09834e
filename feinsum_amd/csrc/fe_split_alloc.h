// fe_split_alloc.h -- the split allocator: device arrays backed by ALTERNATING pieces of two classes of physical memory.
//
// New functionality (the reference allocates every array separately through PyOpenCL and has no notion of placement:
// src/feinsum/measure.py:44-60,80-108).  Why it exists (profiles/r03/vmm_*.txt): on MI355X the
// physical memory falls into classes -- three "superclasses" of about a third of the memory each, which the driver
// hands out in runs of 2 ... 70 GiB -- and concurrent WRITE streams confined to one class reach 5.2 TB/s where streams
// split over two classes reach 6.8.  A DG launch writes 1 (div), 3 (grad: three planes of one array) or 4 (face-mass x 4)
// streams in lockstep; with every stream's ~10 MB write window spread over two classes all the time the launches run at
// 76-78 % of the 8 TB/s roofline instead of 67-74 % (profiles/r03/vmm_interleave_probe_valid.txt: alternating every
// 2 ... 16 MiB, every kernel, div's single stream included; coarser alternation only helps when the planes happen to be
// out of phase).
//
// What the allocator does: an array is a virtually contiguous range backed by physical handles of 4 MiB (hipMemCreate;
// hipMemMap takes no offset, so a piece is a handle of its own) that ALTERNATE between two classes.  The class of memory
// is measured, not known: two write streams, one in a reference region and one in the candidate, run at 5.1-6.1 TB/s when
// both regions are of one superclass and at 6.2-7.0 TB/s otherwise.  The probe must write more distinct memory than the
// 256 MB Infinity Cache holds (two 32 MiB streams run at 7.5-8 TB/s whatever their classes: profiles/r03/
// vmm_probe_method.txt), so pieces are obtained and classified in GROUPS of 32 consecutive handles (128 MiB, mapped side
// by side for the probe): handles created back to back come from one run of the driver's memory.  No timing scan of
// positions, no arena: memory = the footprint rounded up to 2 MiB, plus the pool of classified pieces that are currently
// free (bounded; fe_split_trim releases it).  When the pool needs a class the driver is not handing out, it skips ahead
// with unmapped "spacer" handles (held only while it searches).
//
// VIRTUAL ADDRESSES ARE NEVER RE-USED.  On this stack (ROCm 7.2, gfx950) a virtual range that was mapped once keeps
// translating to its FIRST physical handle: after hipMemUnmap(V) + hipMemMap(V, other handle) kernels writing through V
// still reach the old memory -- with a device synchronisation, a 200 ms pause or hipMemSetAccess(NONE) in between, and
// after hipMemAddressFree + hipMemAddressReserve (which hands the same range out again) alike
// (tools/vmm_remap_test.cpp, profiles/r03/vmm_remap_test.txt; it is also what made every composition of the first
// tools/vmm_interleave_probe.cpp time the same).  Every mapping here therefore gets a range that was never used before:
// reservations are kept for the life of the process (hipMemAddressReserve never returns a live reservation's range;
// a freed array gives back its memory, not its address range -- 2^47 bytes of address space are plentiful).
//
// Host code, included by feinsum_hip.hip after `fail` and FE_HIP_CHECK are defined.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "fe_common.h"

namespace fe {

// Two write streams walked in lockstep by a persistent grid: 4 KiB per wave, stream and step (four 1-KiB non-temporal
// wave stores, as a DG tile), `passes` times over `pieces` 4-KiB pieces of each stream.  The classifier's probe.
__global__ __launch_bounds__(256, 2) void split_probe_kernel(char* a, char* b, long pieces, int passes) {
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    for (int r = 0; r < passes; ++r)
        for (long p = wave; p < pieces; p += nw) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                __builtin_nontemporal_store(v2d{0.0, 0.0}, reinterpret_cast<v2d*>(a + p * 4096 + c * 1024 + lane * 16));
#pragma unroll
            for (int c = 0; c < 4; ++c)
                __builtin_nontemporal_store(v2d{0.0, 0.0}, reinterpret_cast<v2d*>(b + p * 4096 + c * 1024 + lane * 16));
        }
}

}  // namespace fe

namespace {

constexpr size_t kSplitMiB = 1ull << 20;
constexpr size_t kSplitPiece = 4 * kSplitMiB;       // one physical handle; consecutive pieces of an array alternate classes
constexpr int kSplitGroup = 32;                     // pieces created and classified together ...
constexpr size_t kSplitGroupBytes = kSplitGroup * kSplitPiece;   // ... 128 MiB: what the probe needs (see above)
constexpr size_t kSplitGran = 2 * kSplitMiB;        // rounding of array sizes (tail handle) and of addresses
constexpr int kSplitProbePasses = 2;                // the probe writes 2 streams x 128 MiB x 2 passes = 512 MiB per launch
                                                    // (4 passes read 5.5-5.8 TB/s for every pair: profiles/r03/split_alloc_check_v6_four_passes.txt)
// Same superclass: 5.1-5.5 TB/s (same 1-GiB sub-class) or 5.7-6.1 TB/s (the sibling sub-class); another superclass:
// 6.6-7.0 TB/s (profiles/r03/vmm_probe_sizes.txt; the same figures on every box and in every process, HBM clock fixed at
// 2000 MHz).  A group that straddles the end of a run reads in between and is neither handed out nor made an anchor --
// a first version that made a 6.4 TB/s group the anchor of a "new class" then read every later group as unlike it and
// gave up (profiles/r03/split_alloc_check_v7_weak_anchors.txt).  $FEINSUM_SPLIT_SAME_BELOW_GBPS /
// $FEINSUM_SPLIT_OTHER_ABOVE_GBPS override the bounds.
// A group is accepted for a class only on a CONSISTENT reading: like that class's reference (below the first bound) and
// unlike every other known reference (above the second).  A group that is half one class and half another reads ~6.0
// against both -- accepted on the first test alone it diluted its class, and the launches ran at 73-75 % instead of 77 %
// in about one process of three (profiles/r03/bench_*_v11_*.json).
constexpr double kSplitSameBelowGBps = 5950.0;
constexpr double kSplitOtherAboveGBps = 6400.0;
constexpr size_t kSplitSpacerUnit = 32 * kSplitMiB;   // see acquire()
constexpr int kSplitNoSecondClass = -1000;             // (internal: acquire() found no second class and unsplit arrays are refused)
constexpr size_t kSplitMinCollect = 128;               // pieces (512 MiB) of a run's class collected before the search skips ahead: see acquire()
constexpr size_t kSplitAnchorBytes = 256 * kSplitMiB; // a class's reference region: ONE handle (one buddy block: pure)
constexpr int kSplitMaxClasses = 3;                 // three superclasses on MI355X (a fourth "class" would be a misreading)

#define FE_SPLIT_CHECK(expr)                                                                        \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            (void)hipGetLastError();                                                                \
            return fail(FE_EHIP, "split allocator: %s failed: %s", #expr, hipGetErrorString(_e));   \
        }                                                                                           \
    } while (0)

struct SplitPiece {
    hipMemGenericAllocationHandle_t handle;
    int cls;   // superclass id in order of discovery
};

struct SplitArray {
    char* va = nullptr;
    size_t va_bytes = 0, bytes = 0, tail_bytes = 0;
    std::vector<SplitPiece> pieces;                 // full pieces in address order
    hipMemGenericAllocationHandle_t tail{};         // the last, smaller handle (not pooled); valid if tail_bytes
    double alloc_ms = 0;
};

double split_now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

class SplitPool {
  public:
    std::mutex mu;

    int ensure_ready() {
        if (ready_) return FE_OK;
        const double t0 = split_now_ms();
        FE_SPLIT_CHECK(hipGetDevice(&device_));
        prop_ = hipMemAllocationProp{};
        prop_.type = hipMemAllocationTypePinned;
        prop_.location.type = hipMemLocationTypeDevice;
        prop_.location.id = device_;
        acc_ = hipMemAccessDesc{};
        acc_.location.type = hipMemLocationTypeDevice;
        acc_.location.id = device_;
        acc_.flags = hipMemAccessFlagsProtReadWrite;
        FE_SPLIT_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        FE_SPLIT_CHECK(hipEventCreate(&e0_));
        FE_SPLIT_CHECK(hipEventCreate(&e1_));
        if (const char* cap = getenv("FEINSUM_SPLIT_POOL_GIB")) max_pooled_pieces_ = (size_t)(atof(cap) * 1024.0 * kSplitMiB / kSplitPiece);
        if (const char* thr = getenv("FEINSUM_SPLIT_SAME_BELOW_GBPS")) same_below_gbps_ = atof(thr);
        if (const char* thr = getenv("FEINSUM_SPLIT_OTHER_ABOVE_GBPS")) other_above_gbps_ = atof(thr);
        if (const char* gib = getenv("FEINSUM_SPLIT_SEARCH_GIB")) search_budget_ = (size_t)(atof(gib) * 1024.0) * kSplitMiB;
        if (const char* ms = getenv("FEINSUM_SPLIT_SEARCH_MS")) search_ms_budget_ = atof(ms);
        // several ranks of one job on this device (rehearsals; LOCAL_WORLD_SIZE ranks normally have a device each): every
        // pool may skip through its share of the free memory only
        if (const char* sh = getenv("FEINSUM_SPLIT_SHARE")) share_ = std::max(1, atoi(sh));
        if (const char* gib = getenv("FEINSUM_SPLIT_VA_GIB")) va_cap_ = (size_t)(atof(gib) * 1024.0) * kSplitMiB;
        // An array whose pieces would all come from ONE class (no second class within the search budget) is the worst placement
        // there is -- every write stream of the launch in one class, 5.2 TB/s -- and worse than what an ordinary allocator gives
        // on average (round 5, one box: grad at E = 8e6 0.694 of the roofline in such an array, 0.797 in a torch array).  The
        // allocator therefore REFUSES it (FE_EUNSUPPORTED: the caller allocates ordinarily; feinsum_amd.placement does) unless
        // $FEINSUM_SPLIT_UNSPLIT=1 asks for round 4's behaviour.  $FEINSUM_SPLIT_ONE_CLASS=1: a test hook -- the pool behaves as
        // if the device offered one class only.
        if (const char* u = getenv("FEINSUM_SPLIT_UNSPLIT")) allow_unsplit_ = atoi(u) != 0;
        if (const char* o = getenv("FEINSUM_SPLIT_ONE_CLASS")) force_one_class_ = atoi(o) != 0;
        // the anchor of class 0: ONE handle of 256 MiB (a single block of the driver's allocator, hence of one class --
        // a group of 32 small handles may straddle two runs, and an impure anchor makes every later reading ambiguous:
        // profiles/r03/split_alloc_check_v10_group_anchors_impure.txt); it stays mapped for the life of the process
        char* at = nullptr;
        if (int rc = create_anchor(&at)) return rc;
        anchors_.push_back(at);
        free_.resize(1);
        for (int i = 0; i < 16; ++i) launch_probe(at, at + kSplitGroupBytes);   // clocks up before the first measurement
        FE_SPLIT_CHECK(hipStreamSynchronize(stream_));
        ready_ = true;
        setup_ms_ += split_now_ms() - t0;
        return FE_OK;
    }

    // An array of `bytes` bytes whose 4 MiB pieces alternate between two classes of physical memory.
    int alloc(void** out, size_t bytes, int flags) {
        (void)flags;
        if (!out) return fail(FE_EINVAL, "fe_split_alloc: null result pointer");
        *out = nullptr;
        if (bytes == 0) return fail(FE_EINVAL, "fe_split_alloc: zero bytes");
        if (int rc = ensure_ready()) return rc;
        const double t0 = split_now_ms();
        SplitArray a;
        a.bytes = bytes;
        size_t n_full = bytes / kSplitPiece;
        size_t rest = bytes - n_full * kSplitPiece;
        a.tail_bytes = (rest + kSplitGran - 1) / kSplitGran * kSplitGran;
        if (n_full < 2) {   // nothing to alternate: one plain handle
            n_full = 0;
            a.tail_bytes = (bytes + kSplitGran - 1) / kSplitGran * kSplitGran;
        }
        a.va_bytes = n_full * kSplitPiece + a.tail_bytes;
        int cls[2] = {-1, -1};
        if (n_full) {
            // two classes with ceil(n / 2) free pieces each, so that either may take the even positions; the lower class
            // id takes them in even allocations and the odd positions in odd ones
            if (int rc = acquire((n_full + 1) / 2, &cls[0], &cls[1])) {
                if (rc != kSplitNoSecondClass) return rc;
                ++unsplit_refused_;
                return fail(FE_EUNSUPPORTED, "split allocator: no second class of physical memory within the search budget -- an array of one "
                                             "class would be the worst placement there is: allocate this array ordinarily (FEINSUM_SPLIT_UNSPLIT=1 "
                                             "hands it out anyway)");
            }
            if (cls[0] > cls[1]) std::swap(cls[0], cls[1]);
            if (orientation_++ & 1) std::swap(cls[0], cls[1]);
        }
        if (int rc = fresh_range(a.va_bytes, &a.va)) return rc;
        reserve_pieces_ -= std::min(reserve_pieces_, (n_full + 1) / 2);
        for (size_t q = 0; q < n_full; ++q) {
            const int c = cls[q & 1];
            SplitPiece p = free_[c].back();
            free_[c].pop_back();
            hipError_t e = hipMemMap(a.va + q * kSplitPiece, kSplitPiece, 0, p.handle, 0);
            if (e != hipSuccess) {
                free_[c].push_back(p);
                release_array(a);
                return fail(FE_EHIP, "split allocator: hipMemMap failed: %s", hipGetErrorString(e));
            }
            a.pieces.push_back(p);
        }
        if (a.tail_bytes) {
            hipError_t e = hipMemCreate(&a.tail, a.tail_bytes, &prop_, 0);
            if (e == hipSuccess) e = hipMemMap(a.va + n_full * kSplitPiece, a.tail_bytes, 0, a.tail, 0);
            if (e != hipSuccess) {
                a.tail_bytes = 0;
                release_array(a);
                return fail(FE_EHIP, "split allocator: tail handle: %s", hipGetErrorString(e));
            }
        }
        {
            hipError_t e = hipMemSetAccess(a.va, a.va_bytes, &acc_, 1);
            if (e != hipSuccess) {
                release_array(a);
                return fail(FE_EHIP, "split allocator: hipMemSetAccess failed: %s", hipGetErrorString(e));
            }
        }
        if (n_full && cls[0] == cls[1]) ++unsplit_arrays_;
        a.alloc_ms = split_now_ms() - t0;
        alloc_ms_total_ += a.alloc_ms;
        live_bytes_ += a.va_bytes;
        *out = a.va;
        live_[a.va] = std::move(a);
        return FE_OK;
    }

    bool owns(const void* ptr) const { return live_.count(static_cast<char*>(const_cast<void*>(ptr))) != 0; }

    int free_array(void* ptr) {
        auto it = live_.find(static_cast<char*>(ptr));
        if (it == live_.end()) return fail(FE_EINVAL, "fe_split_free: %p is not an array of the split allocator", ptr);
        FE_SPLIT_CHECK(hipDeviceSynchronize());   // as hipFree: no launch may still use the array
        SplitArray a = std::move(it->second);
        live_.erase(it);
        live_bytes_ -= a.va_bytes;
        return release_array(a);
    }

    int info(const void* ptr, char* buf, size_t n) {
        auto it = live_.find(static_cast<char*>(const_cast<void*>(ptr)));
        if (it == live_.end()) return fail(FE_EINVAL, "fe_split_info: %p is not an array of the split allocator", ptr);
        const SplitArray& a = it->second;
        size_t by[kSplitMaxClasses] = {};
        std::string head;
        for (size_t q = 0; q < a.pieces.size(); ++q) {
            ++by[a.pieces[q].cls];
            if (q < 16) head += (char)('0' + a.pieces[q].cls);
        }
        std::string counts = "[";
        for (int c = 0; c < kSplitMaxClasses; ++c) counts += (c ? ", " : "") + std::to_string(by[c]);
        counts += "]";
        return snprintf(buf, n, "{\"bytes\": %zu, \"mapped_bytes\": %zu, \"piece_mib\": %zu, \"pieces\": %zu, \"pieces_by_class\": %s, "
                        "\"first_pieces\": \"%s\", \"tail_bytes\": %zu, \"alloc_ms\": %.3f}",
                        a.bytes, a.va_bytes, kSplitPiece / kSplitMiB, a.pieces.size(), counts.c_str(), head.c_str(), a.tail_bytes, a.alloc_ms);
    }

    int stats(char* buf, size_t n) {
        std::string fr = "[";
        size_t pooled = 0;
        for (size_t c = 0; c < free_.size(); ++c) {
            fr += (c ? ", " : "") + std::to_string(free_[c].size());
            pooled += free_[c].size();
        }
        fr += "]";
        return snprintf(buf, n,
                        "{\"ready\": %s, \"classes\": %zu, \"free_pieces\": %s, \"pooled_bytes\": %zu, \"live_bytes\": %zu, "
                        "\"live_arrays\": %zu, \"pieces_created\": %zu, \"groups_probed\": %zu, \"probes\": %zu, \"spacer_bytes_peak\": %zu, "
                        "\"spacers_created\": %zu, \"spacer_ms\": %.1f, \"probe_ms\": %.1f, \"search_ms\": %.1f, \"search_ms_budget\": %.0f, \"release_ms\": %.1f, \"groups_discarded\": %zu, \"unsplit_arrays\": %zu, \"unsplit_refused\": %zu, \"setup_ms\": %.3f, \"alloc_ms_total\": %.3f, "
                        "\"same_class_below_gbps\": %.0f, \"walk_gave_up\": %s, \"piece_mib\": %zu, \"group_mib\": %zu, "
                        "\"address_space_reserved\": %zu, \"address_space_cap\": %zu, \"last_probes_gbps\": \"%s\"}",
                        ready_ ? "true" : "false", free_.size(), fr.c_str(), pooled * kSplitPiece, live_bytes_, live_.size(),
                        pieces_created_, groups_probed_, probes_, spacer_bytes_peak_, spacers_created_, spacer_ms_, probe_ms_, search_ms_, search_ms_budget_, release_ms_, groups_discarded_, unsplit_arrays_, unsplit_refused_, setup_ms_,
                        alloc_ms_total_, same_below_gbps_, walk_gave_up_ ? "true" : "false", kSplitPiece / kSplitMiB,
                        kSplitGroupBytes / kSplitMiB, va_reserved_, va_cap_, last_probes_.c_str());
    }

    // The caller is about to allocate arrays of `bytes` bytes in all: collect half of that of each of two classes NOW, while
    // the walk through the driver's memory is still inside the first class (arrays allocated one by one otherwise each take
    // what they need, and the class the walk has left does not come back).  Allocates nothing that stays: the pieces wait in
    // the pool (beyond its usual cap until they are handed out).
    int reserve(size_t bytes) {
        if (int rc = ensure_ready()) return rc;
        const double t0 = split_now_ms();
        reserve_pieces_ = (bytes / kSplitPiece + 1) / 2 + 1;
        int ca = -1, cb = -1;
        int rc = acquire(reserve_pieces_, &ca, &cb);
        alloc_ms_total_ += split_now_ms() - t0;
        if (rc == kSplitNoSecondClass) {   // (an announcement only: the allocations that follow are refused one by one)
            reserve_pieces_ = 0;
            rc = FE_OK;
        }
        return rc;
    }

    int trim() {   // give the free pieces back to the driver
        for (auto& list : free_) {
            for (auto& p : list) (void)hipMemRelease(p.handle);
            list.clear();
        }
        walk_gave_up_ = false;
        reserve_pieces_ = 0;
        return FE_OK;
    }

  private:
    bool ready_ = false;
    int device_ = 0;
    hipStream_t stream_{};
    hipEvent_t e0_{}, e1_{};
    hipMemAllocationProp prop_{};
    hipMemAccessDesc acc_{};
    std::vector<char*> anchors_;    // per class: a permanently mapped reference region (one 256 MiB handle) of that class
    std::vector<std::vector<SplitPiece>> free_;
    std::unordered_map<char*, SplitArray> live_;
    double same_below_gbps_ = kSplitSameBelowGBps, other_above_gbps_ = kSplitOtherAboveGBps, setup_ms_ = 0, alloc_ms_total_ = 0;
    double spacer_ms_ = 0, probe_ms_ = 0;
    double search_ms_ = 0;    // wall clock of all searches of this pool: from a search's first spacer to the end of its acquire(),
                              // the release of what was skipped included -- what search_ms_budget_ bounds IN TOTAL
    double release_ms_ = 0;   // of that: handing the skipped memory back to the driver
    int share_ = 1;           // pools (processes) searching this device at the same time ($FEINSUM_SPLIT_SHARE)
    size_t search_budget_ = 96ull << 30;   // how far a search for another class may skip ahead ($FEINSUM_SPLIT_SEARCH_GIB)
    size_t groups_discarded_ = 0;
    size_t va_cap_ = 16ull << 40;          // reserved address space this pool may reach ($FEINSUM_SPLIT_VA_GIB)
    double search_ms_budget_ = 4000.0;     // ... and how long ALL searches of this pool may take, wall clock, spacer creation, the
                                           // groups obtained and probed on the way and the release of the spacers included
                                           // ($FEINSUM_SPLIT_SEARCH_MS).  Round 4 bounded only the spacer creation of ONE search: with
                                           // four ranks searching one device at once a "2.5 s" search took 10 s, 7 of them outside
                                           // the clock (profiles/r04/rehearse4_selfspawn4.json)
    std::vector<hipMemGenericAllocationHandle_t> discarded_;   // pieces of ambiguous groups: held while a search runs
    size_t pieces_created_ = 0, groups_probed_ = 0, probes_ = 0, spacer_bytes_peak_ = 0, spacers_created_ = 0, unsplit_arrays_ = 0;
    size_t live_bytes_ = 0, va_reserved_ = 0;
    size_t max_pooled_pieces_ = (8ull << 30) / kSplitPiece;   // 8 GiB of free classified pieces are kept at most
    unsigned orientation_ = 0;
    int last_cls_ = 0;                 // class of the group created last (the driver hands out long runs of one class)
    bool walk_gave_up_ = false;        // a search for a second class ran out of budget: not repeated until fe_split_trim
    bool allow_unsplit_ = false, force_one_class_ = false;   // $FEINSUM_SPLIT_UNSPLIT, $FEINSUM_SPLIT_ONE_CLASS (ensure_ready)
    size_t unsplit_refused_ = 0;       // arrays refused because they would have been of one class
    size_t reserve_pieces_ = 0;        // pieces per class announced by reserve() and not handed out yet
    std::string last_probes_;

    void launch_probe(char* a, char* b) {
        hipLaunchKernelGGL(fe::split_probe_kernel, dim3(512), dim3(256), 0, stream_, a, b, (long)(kSplitGroupBytes / 4096),
                           kSplitProbePasses);
    }
    // GB/s of the two-stream write probe: two warm-up launches, then the median of four timed triples of launches
    // (~1.4 ms; shorter protocols read "same" up to 6.2 and "other" down to 6.2 TB/s)
    int probe_gbps(char* a, char* b, double* out) {
        const double t0 = split_now_ms();
        launch_probe(a, b);
        launch_probe(a, b);
        float ms[4];
        for (float& m : ms) {
            FE_SPLIT_CHECK(hipEventRecord(e0_, stream_));
            for (int k = 0; k < 3; ++k) launch_probe(a, b);
            FE_SPLIT_CHECK(hipEventRecord(e1_, stream_));
            FE_SPLIT_CHECK(hipEventSynchronize(e1_));
            FE_SPLIT_CHECK(hipEventElapsedTime(&m, e0_, e1_));
        }
        std::sort(ms, ms + 4);
        *out = 3.0 * 2.0 * (double)kSplitGroupBytes * kSplitProbePasses / (0.5 * (ms[1] + ms[2]) * 1e-3) * 1e-9;
        ++probes_;
        probe_ms_ += split_now_ms() - t0;
        return FE_OK;
    }

    // A virtual range that no mapping of this process has used before: a reservation of its own that is never freed
    // (hipMemAddressReserve only hands out a range twice after hipMemAddressFree).  Every array and every group gets one
    // -- and not a sub-range of one large reservation: hipMemSetAccess answered "invalid argument" for mappings of
    // different sizes inside one reservation (an array of 128 + 128 + 12 MiB pieces behind one of 6 x 128 + 34 MiB; a
    // 128 MiB handle behind 2 GiB ones in tools/vmm_probe_sizes.cpp).
    int fresh_range(size_t bytes, char** out) {
        bytes = (bytes + kSplitGran - 1) / kSplitGran * kSplitGran;
        char* base = nullptr;
        // ranges are never handed back (see the header), so a process that allocates and frees for ever grows its reserved
        // address space without bound: a documented cap ($FEINSUM_SPLIT_VA_GIB, default 16 TiB of the device's 2^47-byte
        // space -- 20 000 allocate / free cycles of the headline grad output) turns that into an error the host code answers
        // by allocating ordinarily (placement.empty falls back to torch)
        if (va_reserved_ + bytes > va_cap_)
            return fail(FE_EHIP, "split allocator: address-space cap reached (%zu GiB reserved, cap %zu GiB: FEINSUM_SPLIT_VA_GIB)",
                        va_reserved_ >> 30, va_cap_ >> 30);
        hipError_t e = hipMemAddressReserve((void**)&base, bytes, kSplitGran, nullptr, 0);
        if (e != hipSuccess) return fail(FE_EHIP, "split allocator: no address space (%zu bytes): %s", bytes, hipGetErrorString(e));
        va_reserved_ += bytes;
        *out = base;
        return FE_OK;
    }

    // kSplitGroup new physical pieces, created back to back and mapped side by side at a fresh address.
    int create_group(std::vector<hipMemGenericAllocationHandle_t>* handles, char** at) {
        char* va = nullptr;
        if (int rc = fresh_range(kSplitGroupBytes, &va)) return rc;
        handles->clear();
        hipError_t e = hipSuccess;
        for (int k = 0; k < kSplitGroup && e == hipSuccess; ++k) {
            hipMemGenericAllocationHandle_t h;
            e = hipMemCreate(&h, kSplitPiece, &prop_, 0);
            if (e != hipSuccess) break;
            handles->push_back(h);
            ++pieces_created_;
            e = hipMemMap(va + (size_t)k * kSplitPiece, kSplitPiece, 0, h, 0);
        }
        if (e == hipSuccess) e = hipMemSetAccess(va, kSplitGroupBytes, &acc_, 1);
        if (e != hipSuccess) {
            for (auto& h : *handles) (void)hipMemRelease(h);
            handles->clear();
            (void)hipGetLastError();
            return fail(FE_EHIP, "split allocator: obtaining a group of pieces failed: %s", hipGetErrorString(e));
        }
        *at = va;
        return FE_OK;
    }

    // One handle of kSplitAnchorBytes at a fresh address: a candidate reference region.
    int create_anchor(char** at, hipMemGenericAllocationHandle_t* handle_out = nullptr) {
        char* va = nullptr;
        if (int rc = fresh_range(kSplitAnchorBytes, &va)) return rc;
        hipMemGenericAllocationHandle_t h;
        FE_SPLIT_CHECK(hipMemCreate(&h, kSplitAnchorBytes, &prop_, 0));
        hipError_t e = hipMemMap(va, kSplitAnchorBytes, 0, h, 0);
        if (e == hipSuccess) e = hipMemSetAccess(va, kSplitAnchorBytes, &acc_, 1);
        if (e != hipSuccess) {
            (void)hipMemRelease(h);
            return fail(FE_EHIP, "split allocator: mapping a reference region failed: %s", hipGetErrorString(e));
        }
        if (handle_out) *handle_out = h;
        *at = va;
        return FE_OK;
    }

    // Obtain one group and put its pieces on the free list of its class.  The group is probed against the anchor of the
    // class seen last first (runs), then the others; a group unlike every anchor founds a new class and STAYS mapped as
    // its anchor (another group is then obtained).  *cls_out: the class of the pieces added.
    int grow(int* cls_out) {
        std::vector<hipMemGenericAllocationHandle_t> handles;
        char* at = nullptr;
        if (int rc = create_group(&handles, &at)) return rc;
        ++groups_probed_;
        std::vector<int> order;
        order.push_back(last_cls_);
        for (int c = 0; c < (int)anchors_.size(); ++c)
            if (c != last_cls_) order.push_back(c);
        char note[96];
        int cls = -1, slowest = 0;
        double slowest_rate = 1e30;
        for (int c : order) {
            double rate;
            if (int rc = probe_gbps(anchors_[c], at, &rate)) return rc;
            snprintf(note, sizeof note, "%d:%.0f ", c, rate);
            if (last_probes_.size() > 600) last_probes_.erase(0, 300);
            last_probes_ += note;
            if (rate < slowest_rate) { slowest_rate = rate; slowest = c; }
            if (rate < same_below_gbps_) { cls = c; break; }
        }
        if (cls >= 0) {   // like reference `cls`: it must also be unlike all the others (else: a mixture)
            for (int c = 0; c < (int)anchors_.size() && cls >= 0; ++c) {
                if (c == cls) continue;
                double rate;
                if (int rc = probe_gbps(anchors_[c], at, &rate)) return rc;
                snprintf(note, sizeof note, "%d:%.0f ", c, rate);
                last_probes_ += note;
                if (rate < other_above_gbps_) { cls = -1; slowest_rate = 0.0; }   // (0: not a candidate for a new class either)
            }
        }
        if (cls < 0 && slowest_rate >= other_above_gbps_ && (int)anchors_.size() < kSplitMaxClasses) {
            // clearly unlike every reference region: a class not seen before.  Its reference region is a fresh 256 MiB handle
            // taken here and now (same neighbourhood); it is accepted only if it, too, is unlike every known reference and
            // the group is like IT -- then the group's pieces are the first of the new class.
            char* cand = nullptr;
            hipMemGenericAllocationHandle_t ch;
            if (int rc = create_anchor(&cand, &ch)) return rc;
            bool ok = true;
            for (int c = 0; c < (int)anchors_.size() && ok; ++c) {
                double rate;
                if (int rc = probe_gbps(anchors_[c], cand, &rate)) return rc;
                snprintf(note, sizeof note, "ref%d:%.0f ", c, rate);
                last_probes_ += note;
                ok = rate >= other_above_gbps_;
            }
            if (ok) {
                double rate;
                if (int rc = probe_gbps(cand, at, &rate)) return rc;
                snprintf(note, sizeof note, "own:%.0f ", rate);
                last_probes_ += note;
                ok = rate < same_below_gbps_;
            }
            if (ok) {
                anchors_.push_back(cand);
                free_.emplace_back();
                cls = (int)anchors_.size() - 1;
                last_probes_ += "new ";
            } else {   // (the candidate's address range is simply never used again)
                FE_SPLIT_CHECK(hipStreamSynchronize(stream_));
                (void)hipMemUnmap(cand, kSplitAnchorBytes);
                (void)hipMemRelease(ch);
            }
        }
        if (cls < 0) {
            // neither like one reference region nor clearly unlike all: a group that straddles the end of a run (or a
            // reading between the bounds).  Not handed out; its memory is held until the current search ends (released
            // at once it would be handed out again)
            FE_SPLIT_CHECK(hipStreamSynchronize(stream_));
            for (int k = 0; k < kSplitGroup; ++k) {
                FE_SPLIT_CHECK(hipMemUnmap(at + (size_t)k * kSplitPiece, kSplitPiece));
                discarded_.push_back(handles[k]);
            }
            ++groups_discarded_;
            last_probes_ += "mixed ";
            *cls_out = -1;
            return FE_OK;
        }
        (void)slowest;
        FE_SPLIT_CHECK(hipStreamSynchronize(stream_));
        for (int k = 0; k < kSplitGroup; ++k) {   // one mapping per call; the group's address range is not used again
            FE_SPLIT_CHECK(hipMemUnmap(at + (size_t)k * kSplitPiece, kSplitPiece));
            free_[cls].push_back(SplitPiece{handles[k], cls});
        }
        last_cls_ = cls;
        *cls_out = cls;
        return FE_OK;
    }

    // Make sure two different classes hold `need` free pieces each; *ca / *cb: those classes.  One class for both
    // (with 2 * need pieces) when the device offers no second one within the budget.
    int acquire(size_t need, int* ca, int* cb) {
        std::vector<hipMemGenericAllocationHandle_t> spacers;
        const size_t first_spacer = 1024 * kSplitMiB, max_spacer = 8192 * kSplitMiB;
        size_t spacer_bytes = 0, next_spacer = first_spacer;
        size_t free_mem = 0, total_mem = 0;
        (void)hipMemGetInfo(&free_mem, &total_mem);
        // Searching for a second class: a superclass is about a third of the device memory and the driver hands it out in
        // runs of 2 ... 70 GiB, so the search may have to skip tens of GiB.  Skipping costs what the driver takes to hand out
        // (and clear) the skipped memory, 15-80 ms per GiB (fe_split_stats: "spacer_ms"); on fresh devices the next run began
        // 1 ... 33 GiB ahead.  Budget: $FEINSUM_SPLIT_SEARCH_GIB (default 96: a superclass is 96 GB), at most what is free minus 12 GiB; a search
        // that found nothing is not repeated until fe_split_trim.
        const size_t free_share = free_mem / (size_t)share_;
        const size_t spacer_budget = (!walk_gave_up_ && search_ms_ < search_ms_budget_ && free_share > (16ull << 30))
                                         ? std::min<size_t>(search_budget_, free_share - (12ull << 30)) : 0;
        auto pick = [&](int* a, int* b) {
            int c1 = -1, c2 = -1;   // the two fullest classes
            for (int c = 0; c < (int)free_.size(); ++c) {
                if (c1 < 0 || free_[c].size() > free_[c1].size()) { c2 = c1; c1 = c; }
                else if (c2 < 0 || free_[c].size() > free_[c2].size()) c2 = c;
            }
            if (!force_one_class_ && c1 >= 0 && c2 >= 0 && free_[c2].size() >= need) { *a = c1; *b = c2; return true; }
            return false;
        };
        const size_t collect = std::max(kSplitMinCollect, reserve_pieces_);   // of the current run's class, before skipping
        const size_t need_groups = (std::max(need, collect) + kSplitGroup - 1) / kSplitGroup;
        const size_t max_new = walk_gave_up_ ? 2 * need_groups + 2 : 8 * need_groups + 64;   // groups obtained in this call at most
        size_t made = 0, run = 0;
        int rc = FE_OK;
        // the clock of this search starts with its first spacer; half of what is left of the pool's budget is kept for giving
        // the skipped memory back (measured with four pools on one device: as long again as obtaining it)
        double t_search = -1.0;
        const double search_left = search_ms_budget_ - search_ms_;
        auto search_elapsed = [&] { return t_search < 0 ? 0.0 : split_now_ms() - t_search; };
        while (!pick(ca, cb) && made < max_new) {
            if (t_search >= 0 && search_elapsed() > 0.5 * search_left) break;   // out of time: the groups obtained while skipping count too
            // the driver hands out long runs of one class: with enough of the current run's class in the pool, skip
            // ahead with an unmapped spacer (doubling, 1 ... 16 GiB) before the next group
            // (not before the pool holds 512 MiB of the current run's class, or what fe_split_reserve announced: the arrays of a
            // workload are allocated one after the other, and a class the walk has left behind does not come back -- the
            // pipeline's four lift outputs were left unsplit because grad's output had taken all of the first class that the
            // walk had collected: profiles/r03/bench_pipeline_walk_gave_up.json)
            if (free_[last_cls_].size() >= std::max(need, collect) && run >= 3) {
                if (spacer_bytes >= spacer_budget) break;   // nowhere left to search
                if (t_search < 0) t_search = split_now_ms();
                if (spacer_bytes + next_spacer > spacer_budget) next_spacer = (spacer_budget - spacer_bytes) / kSplitGran * kSplitGran;
                if (next_spacer == 0) break;
                if (search_left < 3000.0) next_spacer = std::min<size_t>(next_spacer, 2048 * kSplitMiB);   // (a step is 15-80 ms per GiB: keep the clock's grain fine)
                // The skipped memory is taken in handles of 32 MiB, not in one large handle: the driver's buddy allocator
                // serves a request from the SMALLEST free block that fits, so 4 MiB pieces keep coming from the block it is
                // currently splitting whatever large blocks are taken elsewhere (one 1 ... 16 GiB spacer per step left the
                // pieces in one class across 34 GiB: profiles/r03/split_alloc_check_v8_*.txt); handles of a small size eat
                // that block's free buddies and then its neighbours, which is a walk through one address neighbourhood.
                // (With 256 MiB handles up to 252 MiB of smaller buddies stay behind and the next groups are islands of
                // mixed memory -- split_alloc_check_v9_*.txt; with 32 MiB at most 28 MiB: less than one group.)
                const double t_sp = split_now_ms();
                bool ok = true;
                size_t done = 0;
                for (; done < next_spacer && ok; done += kSplitSpacerUnit) {
                    if ((done & (1024 * kSplitMiB - 1)) == 0 && search_elapsed() > 0.5 * search_left) break;   // (checked every GiB)
                    hipMemGenericAllocationHandle_t sp;
                    ok = hipMemCreate(&sp, kSplitSpacerUnit, &prop_, 0) == hipSuccess;
                    if (ok) spacers.push_back(sp);
                }
                next_spacer = done;
                spacer_ms_ += split_now_ms() - t_sp;
                if (!ok) {
                    (void)hipGetLastError();
                    break;
                }
                ++spacers_created_;
                spacer_bytes += next_spacer;
                next_spacer = std::min<size_t>(next_spacer * 2, max_spacer);
            }
            const int before = last_cls_;
            int got = -1;
            rc = grow(&got);
            if (rc != FE_OK) break;
            ++made;
            if (got < 0) continue;   // an ambiguous group (discarded): the run is ending
            run = (got == before) ? run + 1 : 1;
            if (got != before) next_spacer = first_spacer;
        }
        spacer_bytes_peak_ = std::max(spacer_bytes_peak_, spacer_bytes);
        const double t_rel = split_now_ms();
        for (auto& sp : spacers) (void)hipMemRelease(sp);
        for (auto& h : discarded_) (void)hipMemRelease(h);
        discarded_.clear();
        if (t_search >= 0) {
            release_ms_ += split_now_ms() - t_rel;
            search_ms_ += split_now_ms() - t_search;
        }
        for (auto& list : free_)   // what a long search collected of the class it did not need goes back to the driver
            while (list.size() > std::max(max_pooled_pieces_ / 2, std::max(need, reserve_pieces_))) {
                (void)hipMemRelease(list.back().handle);
                list.pop_back();
            }
        if (rc != FE_OK) return rc;
        if (pick(ca, cb)) return FE_OK;
        walk_gave_up_ = true;
        if (!allow_unsplit_) {   // (what the search collected of the one class stays pooled up to half the pool's bound)
            for (auto& list : free_)
                while (list.size() > max_pooled_pieces_ / 2) {
                    (void)hipMemRelease(list.back().handle);
                    list.pop_back();
                }
            return kSplitNoSecondClass;
        }
        // $FEINSUM_SPLIT_UNSPLIT=1 -- no second class within the budget: every piece from the fullest class
        int c1 = 0;
        for (int c = 1; c < (int)free_.size(); ++c)
            if (free_[c].size() > free_[c1].size()) c1 = c;
        while (free_[c1].size() < 2 * need) {
            int got = -1;
            if (int rc2 = grow(&got)) return rc2;
            if (pick(ca, cb)) return FE_OK;
            for (int c = 0; c < (int)free_.size(); ++c)
                if (free_[c].size() > free_[c1].size()) c1 = c;
        }
        *ca = *cb = c1;
        return FE_OK;
    }

    int release_array(SplitArray& a) {
        size_t pooled = 0;
        for (auto& l : free_) pooled += l.size();
        for (size_t q = 0; q < a.pieces.size(); ++q) {
            (void)hipMemUnmap(a.va + q * kSplitPiece, kSplitPiece);   // piece by piece (one mapping per call)
            SplitPiece p = a.pieces[q];
            // (per class half of the cap: a pool full of the class the driver is handing out anyway must not push out the
            // pieces of the class that took a search to find)
            if (p.cls >= 0 && p.cls < (int)free_.size() && pooled < max_pooled_pieces_ && free_[p.cls].size() < max_pooled_pieces_ / 2) {
                free_[p.cls].push_back(p);
                ++pooled;
            } else {
                (void)hipMemRelease(p.handle);
            }
        }
        if (a.tail_bytes) {
            (void)hipMemUnmap(a.va + a.pieces.size() * kSplitPiece, a.tail_bytes);
            (void)hipMemRelease(a.tail);
        }
        a.pieces.clear();   // the address range stays reserved and is never handed out again (see the header)
        return FE_OK;
    }
};

SplitPool g_split_pools[64];

SplitPool* split_pool_of_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return &g_split_pools[dev];
}

// The device whose pool holds the array `ptr`: the current device's pool is asked first, then the others' (an array may be
// freed or asked about from a thread whose current device is another one); -1 if no pool knows it.
int split_owner_device(const void* ptr) {
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess || cur < 0 || cur >= 64) cur = 0;
    for (int k = 0; k < 64; ++k) {
        const int dev = (cur + k) % 64;
        std::lock_guard<std::mutex> lock(g_split_pools[dev].mu);
        if (g_split_pools[dev].owns(ptr)) return dev;
    }
    return -1;
}
// Makes `dev` the current device for the lifetime of the object (the pool's HIP calls act on the current device).
struct SplitDeviceScope {
    int before = -1;
    bool switched = false;
    explicit SplitDeviceScope(int dev) {
        if (hipGetDevice(&before) == hipSuccess && before != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~SplitDeviceScope() {
        if (switched) (void)hipSetDevice(before);
    }
};

}  // namespace
