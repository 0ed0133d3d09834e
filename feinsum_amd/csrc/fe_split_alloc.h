// fe_split_alloc.h -- the split allocator: device arrays whose two halves lie in DIFFERENT classes of physical memory.
//
// New functionality (the reference allocates every array separately through PyOpenCL and has no notion of placement:
// src/feinsum/measure.py:44-60,80-108).  Why it exists (profiles/r03/vmm_*.txt): on MI355X the
// physical memory falls into classes -- three "superclasses" of about a third of the memory each, in runs of 2 ... 70 GiB
// in allocation order -- and concurrent WRITE streams confined to one class reach 5.2 TB/s where streams split over two
// classes reach 6.8.  A DG launch writes 1 (div), 3 (grad: three planes of one array) or 4 (face-mass x 4) streams in
// lockstep, and it is 8-14 % faster when those streams are not all in one class.  The class of a piece of memory is
// measurable: two 128 MiB write streams, one in a reference piece and one in the candidate, run at 5.2-6.1 TB/s when both are of
// one class and at 6.7-6.9 TB/s otherwise (tools/vmm_piece_probe.cpp, tools/vmm_factor_probe.cpp).
//
// What the allocator does: an array is a virtually contiguous range (hipMemAddressReserve) backed by physical handles of
// 128 MiB (hipMemCreate; hipMemMap takes no offset, so a piece is a handle of its own), each classified by that probe
// when it is created.  The first half of the pieces is taken from one class and the second half from another, and the
// orientation alternates from one allocation to the next:
//   * an array of several planes written together (grad's [3][E][Np]) has its outer planes in different classes and the
//     cut inside the middle one -- its write windows are split 2 + 1 all the time;
//   * arrays allocated one after the other and written together (the four face-mass outputs) are cut a|b, b|a, a|b, b|a --
//     two windows in either class all the time;
//   * a single-stream array (div) is cut in the middle, which is what FE_VARIANT_MFMA_SPLIT's two write windows need.
// No timing scan of positions, no arena: memory = the footprint rounded up to 2 MiB, plus the pool of classified pieces
// that are currently free (bounded; fe_split_trim releases it).  The driver hands out physical memory in long runs of
// one class; when the pool needs the other class it skips ahead with unmapped "spacer" handles (held only while it searches).
//
// VIRTUAL ADDRESSES ARE NEVER RE-USED.  On this stack (ROCm 7.2, gfx950) a virtual range that was mapped once keeps
// translating to its FIRST physical handle: after hipMemUnmap(V) + hipMemMap(V, other handle) kernels writing through V
// still reach the old memory -- with a device synchronisation, a 200 ms pause or hipMemSetAccess(NONE) in between, and
// after hipMemAddressFree + hipMemAddressReserve (which hands the same range out again) alike
// (tools/vmm_remap_test.cpp, profiles/r03/vmm_remap_test.txt; it is also what made every composition of
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
constexpr size_t kSplitPiece = 128 * kSplitMiB;     // unit of classified physical memory (see the probe below for why not less)
constexpr size_t kSplitGran = 2 * kSplitMiB;        // rounding of array sizes (tail handle) and of addresses
constexpr int kSplitProbePasses = 2;                // the probe writes 2 streams x 128 MiB x 2 passes = 512 MiB per launch
// Two streams in pieces of ONE superclass write at 5.2-5.5 TB/s (same 1-GiB sub-class) or 5.8-6.1 TB/s (the sibling sub-class),
// in pieces of different superclasses at 6.7-6.9 TB/s; the figures were the same on every box and in every process
// (profiles/r03/vmm_*.txt; HBM clock fixed at 2000 MHz).  The probe must write more distinct memory than the 256 MB
// Infinity Cache holds: two streams of 32 MiB are absorbed by it and run at 7.5-8 TB/s whatever their classes
// (profiles/r03/vmm_probe_method.txt, and the first version of this allocator: split_alloc_check_v2_misclassified.txt) --
// hence pieces of 128 MiB.  $FEINSUM_SPLIT_SAME_BELOW_GBPS overrides the threshold.
constexpr double kSplitSameBelowGBps = 6050.0;
constexpr int kSplitMaxClasses = 4;

#define FE_SPLIT_CHECK(expr)                                                                        \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return fail(FE_EHIP, "split allocator: %s failed: %s", #expr, hipGetErrorString(_e));   \
    } while (0)

struct SplitPiece {
    hipMemGenericAllocationHandle_t handle;
    int cls;   // superclass id in order of discovery; -1: not classified
};

struct SplitArray {
    char* va = nullptr;
    size_t va_bytes = 0, bytes = 0, tail_bytes = 0;
    std::vector<SplitPiece> pieces;                 // full pieces in address order
    hipMemGenericAllocationHandle_t tail{};         // the last, smaller handle (not pooled); valid if tail_bytes
    std::string classes;                            // one character per piece: '0' + class, '?' unclassified, 't' tail
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
        if (const char* cap = getenv("FEINSUM_SPLIT_POOL_GIB")) max_pooled_pieces_ = (size_t)(atof(cap) * 1024.0 / 128.0);
        if (const char* thr = getenv("FEINSUM_SPLIT_SAME_BELOW_GBPS")) same_below_gbps_ = atof(thr);
        // the anchor of class 0: the first piece this process obtains
        SplitPiece p0{};
        char* at = nullptr;
        if (int rc = create_in_nursery(&p0, &at)) return rc;
        anchors_.push_back(at);
        free_.resize(1);
        for (int i = 0; i < 16; ++i) launch_probe(at, at);   // clocks up before the first measurement
        FE_SPLIT_CHECK(hipStreamSynchronize(stream_));
        ready_ = true;
        setup_ms_ += split_now_ms() - t0;
        return FE_OK;
    }

    // An array of `bytes` bytes: first half of its pieces from one class, second half from another.
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
        if (n_full < 2) {   // nothing to split: one plain handle
            n_full = 0;
            a.tail_bytes = (bytes + kSplitGran - 1) / kSplitGran * kSplitGran;
        }
        a.va_bytes = n_full * kSplitPiece + a.tail_bytes;
        const size_t n_first = n_full / 2, n_second = n_full - n_first;
        int ca = -1, cb = -1;
        if (n_full) {
            // two classes with n_second (>= n_first) free pieces each, so that either may come first; the lower class id
            // comes first in even allocations and second in odd ones
            if (int rc = acquire(n_second, &ca, &cb)) return rc;
            if (ca > cb) std::swap(ca, cb);
            if (orientation_++ & 1) std::swap(ca, cb);
        }
        if (int rc = fresh_range(a.va_bytes, &a.va)) return rc;
        for (size_t q = 0; q < n_full; ++q) {
            const int c = q < n_first ? ca : cb;
            SplitPiece p = free_[c].back();
            free_[c].pop_back();
            hipError_t e = hipMemMap(a.va + q * kSplitPiece, kSplitPiece, 0, p.handle, 0);
            if (e != hipSuccess) {
                free_[c].push_back(p);
                release_array(a);
                return fail(FE_EHIP, "split allocator: hipMemMap failed: %s", hipGetErrorString(e));
            }
            a.pieces.push_back(p);
            a.classes += (char)('0' + p.cls);
        }
        if (a.tail_bytes) {
            hipError_t e = hipMemCreate(&a.tail, a.tail_bytes, &prop_, 0);
            if (e == hipSuccess) e = hipMemMap(a.va + n_full * kSplitPiece, a.tail_bytes, 0, a.tail, 0);
            if (e != hipSuccess) {
                a.tail_bytes = 0;
                release_array(a);
                return fail(FE_EHIP, "split allocator: tail handle: %s", hipGetErrorString(e));
            }
            a.classes += 't';
        }
        {
            hipError_t e = hipMemSetAccess(a.va, a.va_bytes, &acc_, 1);
            if (e != hipSuccess) {
                release_array(a);
                return fail(FE_EHIP, "split allocator: hipMemSetAccess failed: %s", hipGetErrorString(e));
            }
        }
        if (n_full && ca == cb) ++unsplit_arrays_;
        a.alloc_ms = split_now_ms() - t0;
        alloc_ms_total_ += a.alloc_ms;
        live_bytes_ += a.va_bytes;
        *out = a.va;
        live_[a.va] = std::move(a);
        return FE_OK;
    }

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
        return snprintf(buf, n, "{\"bytes\": %zu, \"mapped_bytes\": %zu, \"piece_mib\": %zu, \"classes\": \"%s\", \"alloc_ms\": %.3f}",
                        a.bytes, a.va_bytes, kSplitPiece / kSplitMiB, a.classes.c_str(), a.alloc_ms);
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
                        "\"live_arrays\": %zu, \"pieces_created\": %zu, \"probes\": %zu, \"spacer_bytes_peak\": %zu, "
                        "\"spacers_created\": %zu, \"unsplit_arrays\": %zu, \"setup_ms\": %.3f, \"alloc_ms_total\": %.3f, "
                        "\"same_class_below_gbps\": %.0f, \"walk_gave_up\": %s, \"piece_mib\": %zu, \"address_space_reserved\": %zu, \"last_probes_gbps\": \"%s\"}",
                        ready_ ? "true" : "false", free_.size(), fr.c_str(), pooled * kSplitPiece, live_bytes_, live_.size(),
                        pieces_created_, probes_, spacer_bytes_peak_, spacers_created_, unsplit_arrays_, setup_ms_,
                        alloc_ms_total_, same_below_gbps_, walk_gave_up_ ? "true" : "false", kSplitPiece / kSplitMiB, va_reserved_, last_ratios_.c_str());
    }

    int trim() {   // give the free pieces back to the driver
        for (auto& list : free_) {
            for (auto& p : list) (void)hipMemRelease(p.handle);
            list.clear();
        }
        walk_gave_up_ = false;
        return FE_OK;
    }

  private:
    bool ready_ = false;
    int device_ = 0;
    hipStream_t stream_{};
    hipEvent_t e0_{}, e1_{};
    hipMemAllocationProp prop_{};
    hipMemAccessDesc acc_{};
    char* va_next_ = nullptr;       // next unused slot for a candidate piece (see the header: ranges are never re-used)
    char* va_end_ = nullptr;
    size_t va_reserved_ = 0;
    std::vector<char*> anchors_;    // per class: a permanently mapped piece of that class
    std::vector<std::vector<SplitPiece>> free_;
    std::unordered_map<char*, SplitArray> live_;
    double same_below_gbps_ = kSplitSameBelowGBps, setup_ms_ = 0, alloc_ms_total_ = 0;
    size_t pieces_created_ = 0, probes_ = 0, spacer_bytes_peak_ = 0, spacers_created_ = 0, unsplit_arrays_ = 0, live_bytes_ = 0;
    size_t max_pooled_pieces_ = 64;    // 8 GiB of free classified pieces are kept at most
    unsigned orientation_ = 0;
    int last_cls_ = 0;                 // class of the piece created last (the driver hands out long runs of one class)
    bool walk_gave_up_ = false;        // a search for a second class ran out of budget: not repeated until fe_split_trim
    std::string last_ratios_;

    void launch_probe(char* a, char* b) {
        hipLaunchKernelGGL(fe::split_probe_kernel, dim3(512), dim3(256), 0, stream_, a, b, (long)(kSplitPiece / 4096),
                           kSplitProbePasses);
    }
    // GB/s of the two-stream write probe: one warm-up launch, then the median of three timed pairs of launches
    // (profiles/r03/vmm_probe_sizes.txt: same superclass 5.1-5.9 TB/s, another one 6.2-7.0 with this short protocol)
    int probe_gbps(char* a, char* b, double* out) {
        launch_probe(a, b);
        float ms[3];
        for (float& m : ms) {
            FE_SPLIT_CHECK(hipEventRecord(e0_, stream_));
            launch_probe(a, b);
            launch_probe(a, b);
            FE_SPLIT_CHECK(hipEventRecord(e1_, stream_));
            FE_SPLIT_CHECK(hipEventSynchronize(e1_));
            FE_SPLIT_CHECK(hipEventElapsedTime(&m, e0_, e1_));
        }
        std::sort(ms, ms + 3);
        *out = 2.0 * 2.0 * (double)kSplitPiece * kSplitProbePasses / (ms[1] * 1e-3) * 1e-9;
        ++probes_;
        return FE_OK;
    }

    // A virtual range that no mapping of this process has used before: a reservation of its own that is never freed
    // (hipMemAddressReserve only hands out a range twice after hipMemAddressFree).  Arrays get one each -- and not a
    // sub-range of one large reservation: hipMemSetAccess answered "invalid argument" for mappings of different sizes
    // inside one reservation (an array of 128 + 128 + 12 MiB pieces behind one of 6 x 128 + 34 MiB; a 128 MiB handle
    // behind 2 GiB ones in tools/vmm_probe_sizes.cpp).  Candidate pieces (all of one size) share reservations of 128 slots.
    int fresh_range(size_t bytes, char** out) {
        bytes = (bytes + kSplitGran - 1) / kSplitGran * kSplitGran;
        char* base = nullptr;
        hipError_t e = hipMemAddressReserve((void**)&base, bytes, kSplitGran, nullptr, 0);
        if (e != hipSuccess) return fail(FE_EHIP, "split allocator: no address space (%zu bytes): %s", bytes, hipGetErrorString(e));
        va_reserved_ += bytes;
        *out = base;
        return FE_OK;
    }
    int fresh_piece_slot(char** out) {
        if (va_next_ == nullptr || va_next_ == va_end_) {
            if (int rc = fresh_range(128 * kSplitPiece, &va_next_)) return rc;
            va_end_ = va_next_ + 128 * kSplitPiece;
        }
        *out = va_next_;
        va_next_ += kSplitPiece;
        return FE_OK;
    }

    // A new physical piece, mapped at a fresh address.
    int create_in_nursery(SplitPiece* p, char** at) {
        char* va = nullptr;
        if (int rc = fresh_piece_slot(&va)) return rc;
        FE_SPLIT_CHECK(hipMemCreate(&p->handle, kSplitPiece, &prop_, 0));
        ++pieces_created_;
        hipError_t e = hipMemMap(va, kSplitPiece, 0, p->handle, 0);
        if (e == hipSuccess) e = hipMemSetAccess(va, kSplitPiece, &acc_, 1);
        if (e != hipSuccess) {
            (void)hipMemRelease(p->handle);
            return fail(FE_EHIP, "split allocator: mapping a new piece failed: %s", hipGetErrorString(e));
        }
        p->cls = -1;
        *at = va;
        return FE_OK;
    }

    // Create one piece and find its class: probed against the anchor of the class seen last first (the driver hands out
    // runs), then the others; a piece unlike every anchor founds a new class and STAYS in the nursery as its anchor
    // (another piece is then created for the caller).
    int create_classified(SplitPiece* out) {
        SplitPiece p{};
        char* at = nullptr;
        if (int rc = create_in_nursery(&p, &at)) return rc;
        std::vector<int> order;
        order.push_back(last_cls_);
        for (int c = 0; c < (int)anchors_.size(); ++c)
            if (c != last_cls_) order.push_back(c);
        char note[96];
        int slowest = 0;
        double slowest_rate = 1e30;
        for (int c : order) {
            double rate;
            if (int rc = probe_gbps(anchors_[c], at, &rate)) return rc;
            snprintf(note, sizeof note, "%d:%.0f ", c, rate);
            if (last_ratios_.size() > 600) last_ratios_.erase(0, 300);
            last_ratios_ += note;
            if (rate < slowest_rate) { slowest_rate = rate; slowest = c; }
            if (rate < same_below_gbps_) { p.cls = c; break; }
        }
        if (p.cls < 0 && (int)anchors_.size() < kSplitMaxClasses) {   // a new class: this piece is its anchor
            anchors_.push_back(at);
            free_.emplace_back();
            last_cls_ = (int)anchors_.size() - 1;
            last_ratios_ += "new ";
            return create_classified(out);
        }
        if (p.cls < 0) p.cls = slowest;   // more classes than anchors: the class it conflicts with most
        FE_SPLIT_CHECK(hipStreamSynchronize(stream_));
        FE_SPLIT_CHECK(hipMemUnmap(at, kSplitPiece));   // (its address range is not used again)
        last_cls_ = p.cls;
        *out = p;
        return FE_OK;
    }

    // Make sure two different classes hold `need` free pieces each; *ca / *cb: those classes.  One class for both
    // (with 2 * need pieces) when the device offers no second one within the budget.
    int acquire(size_t need, int* ca, int* cb) {
        std::vector<hipMemGenericAllocationHandle_t> spacers;
        const size_t first_spacer = 1024 * kSplitMiB, max_spacer = 16384 * kSplitMiB;
        size_t spacer_bytes = 0, next_spacer = first_spacer;
        size_t free_mem = 0, total_mem = 0;
        (void)hipMemGetInfo(&free_mem, &total_mem);
        // Searching for a second class: a superclass is about a third of the device memory and the driver hands it out in
        // one or a few long runs, so the search may have to skip ~100 GiB.  An unmapped handle costs ~0.2 ms per GiB to
        // create (profiles/r03/split_alloc_check_v3_*.txt: 18 spacers, 68 GB, inside a 38 ms call), so the budget is
        // whatever is free minus 12 GiB; a search that found nothing is not repeated until fe_split_trim.
        const size_t spacer_budget = (!walk_gave_up_ && free_mem > (16ull << 30)) ? free_mem - (12ull << 30) : 0;
        auto pick = [&](int* a, int* b) {
            int c1 = -1, c2 = -1;   // the two fullest classes
            for (int c = 0; c < (int)free_.size(); ++c) {
                if (c1 < 0 || free_[c].size() > free_[c1].size()) { c2 = c1; c1 = c; }
                else if (c2 < 0 || free_[c].size() > free_[c2].size()) c2 = c;
            }
            if (c1 >= 0 && c2 >= 0 && free_[c2].size() >= need) { *a = c1; *b = c2; return true; }
            return false;
        };
        const size_t max_new = walk_gave_up_ ? 2 * need : 4 * need + 64;   // pieces created in this call at most
        size_t made = 0, run = 0;
        int rc = FE_OK;
        while (!pick(ca, cb) && made < max_new) {
            // the driver hands out long runs of one class: with enough of the current run's class in the pool, skip
            // ahead with an unmapped spacer (doubling, 1 ... 16 GiB) before the next piece
            if (free_[last_cls_].size() >= need && run >= 4) {
                if (spacer_bytes >= spacer_budget) break;   // nowhere left to search
                if (spacer_bytes + next_spacer > spacer_budget) next_spacer = (spacer_budget - spacer_bytes) / kSplitGran * kSplitGran;
                if (next_spacer == 0) break;
                hipMemGenericAllocationHandle_t sp;
                if (hipMemCreate(&sp, next_spacer, &prop_, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    break;
                }
                spacers.push_back(sp);
                ++spacers_created_;
                spacer_bytes += next_spacer;
                next_spacer = std::min<size_t>(next_spacer * 2, max_spacer);
            }
            SplitPiece p;
            const int before = last_cls_;
            rc = create_classified(&p);
            if (rc != FE_OK) break;
            ++made;
            run = (p.cls == before) ? run + 1 : 1;
            if (p.cls != before) next_spacer = first_spacer;
            free_[p.cls].push_back(p);
        }
        spacer_bytes_peak_ = std::max(spacer_bytes_peak_, spacer_bytes);
        for (auto& sp : spacers) (void)hipMemRelease(sp);
        if (rc != FE_OK) return rc;
        if (pick(ca, cb)) return FE_OK;
        walk_gave_up_ = true;
        // no second class within the budget: both halves from the fullest class
        int c1 = 0;
        for (int c = 1; c < (int)free_.size(); ++c)
            if (free_[c].size() > free_[c1].size()) c1 = c;
        while (free_[c1].size() < 2 * need) {
            SplitPiece p;
            if (int rc2 = create_classified(&p)) return rc2;
            free_[p.cls].push_back(p);
            if (pick(ca, cb)) return FE_OK;
            for (int c = 0; c < (int)free_.size(); ++c)
                if (free_[c].size() > free_[c1].size()) c1 = c;
        }
        *ca = *cb = c1;
        return FE_OK;
    }

    int release_array(SplitArray& a) {
        for (size_t q = 0; q < a.pieces.size(); ++q) {
            (void)hipMemUnmap(a.va + q * kSplitPiece, kSplitPiece);   // piece by piece (one mapping per call)
            SplitPiece p = a.pieces[q];
            size_t pooled = 0;
            for (auto& l : free_) pooled += l.size();
            if (p.cls >= 0 && p.cls < (int)free_.size() && pooled < max_pooled_pieces_) free_[p.cls].push_back(p);
            else (void)hipMemRelease(p.handle);
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

}  // namespace
