// fe_common.h -- shared device helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fe {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define FE_AS1 __attribute__((address_space(1)))
#define FE_AS3 __attribute__((address_space(3)))

// Per-field device pointers of a batched launch (by value in the kernel arguments).
constexpr int kMaxFields = 8;
struct FieldPtrs {
    const double* v[kMaxFields];
    double* out[kMaxFields];
};

// Field k of a batched launch, k wave-uniform and only known at run time: a select chain over
// the kernel arguments (scalar ALU) instead of an indexed copy of the struct in scratch.
__device__ __forceinline__ const double* field_in(const FieldPtrs& P, int k) {
    const double* p = P.v[0];
#pragma unroll
    for (int q = 1; q < kMaxFields; ++q) p = (k == q) ? P.v[q] : p;
    return p;
}
__device__ __forceinline__ double* field_out(const FieldPtrs& P, int k) {
    double* p = P.out[0];
#pragma unroll
    for (int q = 1; q < kMaxFields; ++q) p = (k == q) ? P.out[q] : p;
    return p;
}

// 32-bit LDS byte address of a generic pointer into __shared__ memory, made
// provably wave-uniform (it is moved into M0 by the LDS-DMA helpers).
__device__ __forceinline__ unsigned lds_addr_uniform(const void* p) {
    unsigned a = (unsigned)(uintptr_t)(FE_AS3 const void*)p;
    return __builtin_amdgcn_readfirstlane(a);
}

// LDS-DMA ("global_load_lds"): each active lane copies 16 (or 4) bytes from its
// own global address `g` to LDS at  lds_base + lane*size  (destination is
// lane-linear: wave-uniform base in M0).  Issued through inline asm so that
// hipcc's waitcnt pass does not see the loads and does not drain them with a
// vmcnt(0) in front of the first ds_read of a *different* staging buffer; the
// kernels wait for them with counted `s_waitcnt vmcnt(N)` (see wait_vmcnt).
// M0 is compiler-reserved: save and restore it inside the statement.
__device__ __forceinline__ void glds16(const void* g, unsigned lds_base) {
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);   // "s" needs a provably uniform value
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(lds_base)
        : "memory");
}
__device__ __forceinline__ void glds16_nt(const void* g, unsigned lds_base) {   // non-temporal hint
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(lds_base)
        : "memory");
}
__device__ __forceinline__ void glds4(const void* g, unsigned lds_base) {
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(g), "s"(lds_base)
        : "memory");
}

// The streamed operand of a launch (u / v: every byte read once per launch) is fetched non-temporally -- unless the launch's
// inputs fit the 256 MiB Infinity Cache: then a caller who evaluates again and again on the same arrays (a time stepper; the
// reference's timing protocol, src/feinsum/measure.py:248-275) finds them there the next time, IF the loads were allowed to
// allocate.  Measured (profiles/r04/temporal_loads_ab.txt, grad, back-to-back launches): E = 1e5 24.8 -> 23.1 us, 2e5 43.1 ->
// 41.1, 5e5 96.3 -> 92.2, 7e5 132.0 -> 125.9 (inputs 235 MiB); 7.5e5 (252 MiB) level, 8e5 149.2 -> 167.3 and 1e6 184.8 ->
// 197.3 (the inputs no longer fit: the cache thrashes).  Stores stay non-temporal at every size (temporal: +9 % at 1e5,
// +21 % at 5e5).  The launcher sets kOpLoadsTemporal in a body's flag word when the launch reads at most kTemporalInputBytes.
constexpr int kOpLoadsTemporal = 16;
constexpr long long kTemporalInputBytes = 248ll << 20;
// Output stores are non-temporal (every byte written once; plain stores cost 9 - 21 %) -- except in launches that write little
// (kOpStoresWriteThrough, set by the launcher when the outputs are at most kWriteThroughOutputBytes): those store write-through
// (sc0 sc1), so that no dirty line is left in the L2s when the last wave ends and the launch's end does not wait for the write-back.
// Measured on grad (profiles/r04/store_policies_small_sizes.txt): E = 1e5 24.0 -> 23.3 us, 2e5 41.1 -> 40.6; at 5e5 write-through
// without nt costs 16 % and with nt it is level -- the write-back at the end is a fixed cost that only short launches see.
constexpr int kOpStoresWriteThrough = 32;
constexpr long long kWriteThroughOutputBytes = 176ll << 20;   // (round 5, with write-through under the dynamic walk too: -2 ... -3 % at outputs of
                                                               //  128 - 160 MiB, +1 % at 200, +9 ... 12 % from 240 on: profiles/r05/write_through_threshold_ab.txt)
// kOpPhasePriority (round 5; the eight-wave p = 5 kernels, opt-in: fe_set_phase_priority_p5): a wave's f64 VALU phases -- the
// B-fragment build of div, the Jacobian contraction of grad -- at raised issue priority, its matrix phases at priority 0.
// Measured: p = 5 div -1.0 ... -1.3 % at E >= 1e6, grad -0.8 ... +1.5 % (profiles/r05/p5_phase_priorities_ab.txt).  On the p = 4
// kernels the same was measured and removed again: grad +4 % at E = 1e5, div -2 ... -3 % (superseded by the interleaved B
// build of fe_div.h), the fused launches +-1 % (profiles/r05/phase_priorities_ab.txt).  The arbiter itself does not listen to
// s_setprio where it would matter: beside a partner's pure MFMA stream a wave's VALU instructions take 8.8 cycles each at
// either priority and the MFMA stream is not slowed (profiles/r05/mfma_arbiter_two_waves.txt).
constexpr int kOpPhasePriority = 64;
// kOpQuarterTail (round 5; the interleaved div kernel and the grad kernel under the static walk, one field): the tiles behind the
// last full round -- at most an eighth of a round of them -- are processed as QUARTER tiles of four elements, one per wave, on
// v_mfma_f64_4x4x4_4b (fe_div.h, fe_grad.h).
constexpr int kOpQuarterTail = 128;
// kOpStaggeredStart (round 5; the grad kernel under the static walk, one field, 2.5 to 4.5 rounds of tiles): the blocks on the odd
// CUs of every XCD (block index / 8 odd: blocks go round the eight XCDs) sleep kStaggerSleeps x s_sleep 16 = 5120 cycles, about
// half a SIMD's tile period, before their first instruction.  With every CU in step the older waves of all 256 CUs issue their
// first stores within one microsecond (10 - 13 TB/s requested) and the younger waves' stage 2 waits behind them; out of step the
// two halves of the chip alternate.  E = 8.2e4 ... 1.47e5: -0.5 ... -4 %; below 2.5 rounds the late half is the launch's end
// (E = 5e4: +5 %), under the dynamic walk and at E = 1e6 it costs 1 - 3 %, div gains nothing (profiles/r05/staggered_start_abl.txt).
constexpr int kOpStaggeredStart = 256;
constexpr int kStaggerSleeps = 5;
// One 16-byte write-through store per lane, in the addressing form the compiler gives its own stores (wave-uniform base in
// SGPRs + one 32-bit lane offset + immediate): with a 64-bit VGPR address per store the same instruction cost 3 - 12 %.
// (a macro: the immediate must be a constant where the statement stands -- inside an unrolled loop it is one after unrolling)
// Two hazards of gfx9 that hipcc resolves for its own instructions and cannot see inside an asm statement (LLVM
// GCNHazardRecognizer): a VALU write of the data registers of a vector-memory store of more than 8 bytes needs two wait states
// behind the store -- the trailing s_nop (without it whatever the compiler schedules next may overwrite the first data register
// while the last lanes are still being read: round 5 found the low dword of four doubles of a tile zeroed that way, in one launch
// out of a few, after a change elsewhere had moved the instructions behind the last store of a tile); and a vector-memory
// instruction that reads an SGPR written by a VALU instruction (the base comes from v_readfirstlane) needs five wait states in
// front -- FE_STORE16_WRITE_THROUGH_FIRST, for the first store of a batch.
#define FE_STORE16_WRITE_THROUGH(base_uniform, lane_offset, val, imm)                                                     \
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 sc0 sc1\n\ts_nop 1" ::"v"(lane_offset), "v"(val), "s"(base_uniform), "n"(imm))
#define FE_STORE16_WRITE_THROUGH_FIRST(base_uniform, lane_offset, val, imm)                                               \
    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 offset:%3 sc0 sc1\n\ts_nop 1" ::"v"(lane_offset), "v"(val), "s"(base_uniform), "n"(imm))
// (no memory clobber: the statement reads registers only, and nothing else touches the output)
// The chunks of one [rows][NP] output tile (CHUNKS of 16 bytes, INSTR wave-instructions), already in registers, to `op`
// (wave-uniform): non-temporal, or write-through for the short launches above.
template <int INSTR, int CHUNKS>
__device__ __forceinline__ void store_tile_held(double* op, int lane, const v2d (&held)[INSTR], bool write_through) {
    if (write_through) {
        // (the base is wave-uniform by construction; where the compiler cannot prove it, this puts it into SGPRs anyway)
        const unsigned long long a = reinterpret_cast<unsigned long long>(op);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);          // (the builtin returns a signed int:
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));    //  widen only after the cast)
        op = reinterpret_cast<double*>((unsigned long long)lo | ((unsigned long long)hi << 32));
#pragma unroll
        for (int c = 0; c < INSTR; ++c)
            if ((c + 1) * 64 <= CHUNKS || c * 64 + lane < CHUNKS) {
                if (c == 0) FE_STORE16_WRITE_THROUGH_FIRST(op, (unsigned)(lane * 16), held[c], 0);
                else FE_STORE16_WRITE_THROUGH(op, (unsigned)(lane * 16 + (c >> 2) * 4096), held[c], (c & 3) * 1024);
            }
    } else {
#pragma unroll
        for (int c = 0; c < INSTR; ++c) {
            const int qc = c * 64 + lane;
            if ((c + 1) * 64 <= CHUNKS || qc < CHUNKS) __builtin_nontemporal_store(held[c], reinterpret_cast<v2d*>(op + 2 * qc));
        }
    }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- dynamic walk: tile tickets behind two static rounds ------------------------------------------------
// A static walk (tile = first + k * stride) ends ragged: per-wave end times of the p = 4 grad kernel at E = 1e6 spread over
// 164 ... 190 us around a mean of 176 (profiles/r02/stamps_plain.csv) -- the slower XCDs, the younger wave of each SIMD, the
// waves with one tile more, the part of an array that lies in the slower class of physical memory -- and the launch lasts
// until the last wave.  Tickets for every tile from ONE or eight per-XCD counters, taken synchronously, were measured in round
// 1 and cost more than they gave (gfx950 executes global atomics at the memory side: ~88 per us and counter, against ~330
// tiles per us).  What works (profiles/r03/dynamic_walk_*.txt: grad E = 1e6 77.8 -> 80.6 % of the roofline, 69.2 -> 76.4 %
// with every array from torch):
//  * every wave walks `t_static / waves` rounds (two) statically, so that the first ticket's latency hides behind a tile;
//  * the tiles [t_static, nTiles) are dealt out by kTailPools counters: ticket t of pool p is the tile
//    t_static + kTailPools * t + (p + t) mod kTailPools, and a block's pool is (bid / 8) % kTailPools, so that every pool is
//    drained by waves of all eight XCDs (no stealing needed); 4 pools are too few (75.5 %), 8 and 16 measure alike.  The
//    residue ROTATES with t because tiles are not equally fast: with the fixed residue p the pools 3, 7, 11, 15 -- the tiles
//    = 3 mod 4 -- took 3 % longer per tile and ended 6 - 10 us behind the others, whatever the spacing of the counters
//    (per-wave time stamps: profiles/r03/dynamic_walk_stamps.txt);
//  * one returning atomic per tile, requested ONE TILE AHEAD -- at the top of the iteration that prefetches the tile before
//    -- so that its latency never shows;
//  * the wave whose ticket comes back beyond the pool stops asking and reports to the pool's second counter; the last of the
//    pool's waves to do so zeroes both: the counters are all zero between launches, whatever stream or graph replays the
//    launch.  (One report counter for the whole grid was measured first: 2048 atomics on one address take ~23 us, E = 1e5
//    went from 24 to 40 us.)
// Below four and a half rounds (E = 1e5: three) the static walk is faster (61.1 against 57.5 %) and stays.  Results do not depend on the
// walk: a tile's arithmetic is position independent (bitwise equal outputs: tests/test_gpu_parity.py).
constexpr int kTailPools = 16;
constexpr int kTailStride = 2176;                        // unsigned per pool: tickets, and 4352 bytes behind them the reports
constexpr int kTailWords = kTailPools * kTailStride;
// One returning add of 1 by lane 0.  The value comes back microseconds later, while the wave works on: it must land where
// the compiler keeps nothing and copies nothing.  Seen in the ISA: with an inline-asm atomic whose result is a C++ variable
// the register allocator copies the not yet written register in front of our counted wait; the compiler's own atomic is
// followed by vmcnt(0) at once (the merge of the one-lane branch), which drains stores and prefetches every tile; an
// accumulation register as the landing place makes the allocator spread the kernel's values over AGPRs.  So the result lands
// in a VGPR the compiler does not use: kernels with a dynamic walk are compiled with FE_TAIL_KERNEL_ATTR (amdgpu_num_vgpr(248):
// the compiler is asked to stay within v0..v247, which these kernels -- 226 to 240 registers -- do anyway), and the statements
// below name v255 (tickets) and v254 (the report) in their text and clobber lists, which also makes the kernel descriptor
// allocate all 256 registers.  The attribute is a request, not a proof: tests/test_ticket_registers.py disassembles the BUILT
// library and fails if any instruction outside these statements touches v254 / v255 in a kernel that takes tickets.
#define FE_TAIL_KERNEL_ATTR __attribute__((amdgpu_num_vgpr(248)))
template <int R>
__device__ __forceinline__ void tail_request(unsigned* counter) {
    static_assert(R == 0 || R == 1, "0: ticket (v255), 1: report (v254)");
    unsigned long long keep;
    unsigned one = 1u;
    if (R == 0)
        asm volatile(
            "s_mov_b64 %0, exec\n\t"
            "s_mov_b64 exec, 1\n\t"
            "global_atomic_add v255, %1, %2, off sc0\n\t"
            "s_mov_b64 exec, %0"
            : "=&s"(keep) : "v"(counter), "v"(one) : "memory", "v255");
    else
        asm volatile(
            "s_mov_b64 %0, exec\n\t"
            "s_mov_b64 exec, 1\n\t"
            "global_atomic_add v254, %1, %2, off sc0\n\t"
            "s_mov_b64 exec, %0"
            : "=&s"(keep) : "v"(counter), "v"(one) : "memory", "v254");
}
// wait until at most N vector-memory operations are outstanding (the ones issued after the request), then the value
template <int N, int R>
__device__ __forceinline__ unsigned tail_wait() {
    unsigned t;
    if (R == 0) asm volatile("s_waitcnt vmcnt(%1)\n\ts_nop 0\n\tv_readfirstlane_b32 %0, v255" : "=s"(t) : "n"(N) : "memory", "v255");
    else asm volatile("s_waitcnt vmcnt(%1)\n\ts_nop 0\n\tv_readfirstlane_b32 %0, v254" : "=s"(t) : "n"(N) : "memory", "v254");
    return t;
}

// ticket t of pool `pool` -> tile index, -1 beyond the last tile.  In 32-bit arithmetic on purpose (tile counts are far below
// 2^31): for the 64-bit form `x < tEnd ? x : tEnd` of wave-uniform values hipcc 7.2 emitted v_cmp_lt_i64 (VCC) followed by
// s_cselect_b32 (SCC, left over from the address addition) -- every ticket of the face-mass kernel read as "beyond the pool".
__device__ __forceinline__ int64_t tail_ticket_tile(unsigned t, int64_t t_static, int pool, int64_t n_tiles) {
    const unsigned room = (unsigned)(n_tiles - t_static);          // dynamic tiles
    const unsigned off = t * (unsigned)kTailPools + (((unsigned)pool + t) & (unsigned)(kTailPools - 1));   // (the pool's residue rotates: see above)
    const bool ok = t < (1u << 26) && off < room;
    return ok ? t_static + (int64_t)off : (int64_t)-1;
}

// the ticket counter of a pool (its report counter lies kTailStride / 2 words behind); null for a launch without counters
__device__ __forceinline__ unsigned* tail_pool_counters(unsigned* tail, int pool) {
    return tail ? tail + pool * kTailStride : nullptr;
}
__device__ __forceinline__ unsigned* tail_pool_reports(unsigned* counter) {
    return counter ? counter + kTailStride / 2 : nullptr;
}

// Orders this wave's LDS accesses for the compiler.  The hardware executes one
// wave's DS instructions in issue order, so wave-private LDS staging needs no
// s_barrier -- only a compiler fence between the writes of some lanes and the
// reads of others.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cooperative copy of an operator matrix (n doubles) from global memory into the start of the
// block's LDS, all threads, coalesced.  Caller synchronises.  (2048 waves each gathering their
// MFMA fragments of the same ~30 KB straight from L2 cost up to 18 us of prologue; staged once
// per block it is ~5 us and the whole kernel ran 9 % faster in A/B.)
template <int N, int THREADS = 256>
__device__ __forceinline__ void stage_operator(const double* __restrict__ g, double* lds) {
    // all loads are issued before the first LDS write so that their latencies overlap (a plain
    // copy loop waits for each load in turn: ~15 dependent L2 round trips).
    constexpr int kPer = (N + THREADS - 1) / THREADS;
    double tmp[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int i = threadIdx.x + k * THREADS;
        tmp[k] = (i < N) ? g[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const int i = threadIdx.x + k * THREADS;
        if (i < N) lds[i] = tmp[k];
    }
}

// The same through the LDS-DMA path: N doubles from g to the LDS byte address lds (wave-uniform),
// by the four waves of a 256-thread block.  Nothing passes through registers and the compiler
// never waits on these loads, so the caller can queue further loads behind them and wait with a
// counted vmcnt (loads return in order) before the block barrier that publishes the copy.
template <int N>
__device__ __forceinline__ void stage_operator_dma(const double* __restrict__ g, unsigned lds, int wave,
                                                   int lane) {
    constexpr int kChunks = N / 2;                       // 16-byte chunks
    constexpr int kInstr = (kChunks + 255) / 256;        // per wave
    const char* gb = reinterpret_cast<const char*>(g);
#pragma unroll
    for (int c = 0; c < kInstr; ++c) {
        const int q0 = (c * 4 + wave) * 64;              // first chunk of this wave's instruction
        if (q0 + lane < kChunks) glds16(gb + (q0 + lane) * 16, lds + q0 * 16);
    }
    if ((N & 1) && wave == 0 && lane < 2)                // odd N: the last double as two dwords
        glds4(gb + (N - 1) * 8 + lane * 4, lds + (N - 1) * 8);
}

// ---- prepared operators (fe_prepare_operator) -------------------------------------------------
// An operator matrix is constant across the launches of a time-stepping code, while every launch
// rebuilds its MFMA A fragments (stage the matrix through LDS, two block barriers, one dependent LDS
// gather per fragment: 3-9 us before a wave's first tile -- which, measured, overlaps the first
// tiles' load latency, so that prepared operators do not shorten a launch: DESIGN.md section 3e).
// A prepared operator is the same matrix written ONCE in
// fragment layout: fragment f of lane l is double  ((f >> 1) * 64 + l) * 2 + (f & 1)  of its section,
// so a wave fetches two fragments per 16-byte load, 1 KiB contiguous per wave-instruction (from L2
// after the first wave of an XCD), with no LDS staging and no barrier.
// The buffer has a fixed size (kPreparedBytes), so a kernel reading its own section can never run
// past the allocation even when handed a buffer prepared for another shape.
constexpr int kPrepHeaderBytes = 256;                 // reserved, not written: what a buffer holds is recorded on the HOST
                                                      // (feinsum_hip.hip: g_prepared -- shape, flags, source operator, device)
constexpr int kPrepGradOff = kPrepHeaderBytes;        // grad section: <= 64 fragments (Np = 35: 63)
constexpr int kPrepDivOff = kPrepGradOff + 32 * 1024; // div section: big-tile fragments, then the 4-row groups
constexpr int kPrepDivSmallOff = kPrepDivOff + 32 * 1024;
constexpr int kPrepFmOff = kPrepHeaderBytes;          // face-mass buffers hold R: big tiles, then 4-row groups
constexpr int kPreparedBytes = 96 * 1024;

// `count` fragments of this lane from a prepared section; set(f, value) receives them.  Plain loads:
// the caller makes the compiler's own wait explicit (prepared_fragments_landed) before its main loop.
template <int COUNT, class F>
__device__ __forceinline__ void load_prepared_fragments(const void* section, int lane, F set) {
    const v2d* pp = reinterpret_cast<const v2d*>(section) + lane;
#pragma unroll
    for (int p = 0; p < (COUNT + 1) / 2; ++p) {
        const v2d v = pp[p * 64];
        set(2 * p, v[0]);
        if (2 * p + 1 < COUNT) set(2 * p + 1, v[1]);
    }
}
// s_waitcnt vmcnt(0) THROUGH THE COMPILER (it models this builtin, unlike the inline-asm waits of the
// main loops): afterwards it knows the fragment loads have landed and inserts no wait of its own
// inside the loop -- a compiler-placed vmcnt(0) there would drain the prefetches and the stores of
// every iteration.  (The LDS-DMA loads of the first tiles, issued behind the fragment loads, are
// waited for too; the first iteration needs them anyway.)
__device__ __forceinline__ void prepared_fragments_landed() {
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
}
__device__ __forceinline__ void store_prepared_fragment(void* section, int f, int lane, double v) {
    reinterpret_cast<double*>(section)[((f >> 1) * 64 + lane) * 2 + (f & 1)] = v;
}

// ---- [rows][NP] tiles in LDS when NP is a multiple of 8 (p = 5: NP = 56) ---------------------------
// A row is NP * 8 = 448 bytes = 192 mod 256: rows four apart start on the same banks, so the
// column-wise accesses of the MFMA layouts (the 16 rows n of one column: B-fragment reads of the u
// tile, accumulator writes into the output transposition buffer) conflict four ways -- at p = 5
// 57 % of all LDS cycles were bank conflicts (profiles/r02/p5_pmc_before.txt).  Row n is therefore
// ROTATED by rot(n) = (n / 4) mod 4 sixteen-byte chunks: column accesses of the 32 lanes of a
// ds_read_b64 group then fall on 32 different bank pairs (24 b + 2 a + g, n = 4 a + b), rows stay
// contiguous modulo the wrap, and a row-major copy needs no padding: the LDS-DMA that brings a tile in
// simply fetches, for LDS chunk q, the global chunk tile_src_chunk(q).
template <int NP>
__device__ __forceinline__ constexpr bool tile_rotated() { return NP % 8 == 0 && NP >= 32; }
template <int NP>
__device__ __forceinline__ int tile_rot(int row) { return tile_rotated<NP>() ? ((row >> 2) & 3) : 0; }
template <int NP>
__device__ __forceinline__ int tile_index(int row, int col) {   // position (doubles) of element (row, col)
    if constexpr (tile_rotated<NP>()) {
        const int c = col + 2 * tile_rot<NP>(row);
        return row * NP + (c >= NP ? c - NP : c);
    } else {
        return row * NP + col;
    }
}
template <int NP>
__device__ __forceinline__ int tile_src_chunk(int q) {   // LDS chunk q holds this chunk of the row-major tile
    if constexpr (tile_rotated<NP>()) {
        constexpr int CPR = NP / 2;
        const int row = q / CPR, p = q - row * CPR, s = p - tile_rot<NP>(row);
        return row * CPR + (s < 0 ? s + CPR : s);
    } else {
        return q;
    }
}
template <int NP>
__device__ __forceinline__ int tile_dst_chunk(int q) {   // chunk q of the row-major tile lives in this LDS chunk
    if constexpr (tile_rotated<NP>()) {
        constexpr int CPR = NP / 2;
        const int row = q / CPR, p = q - row * CPR, s = p + tile_rot<NP>(row);
        return row * CPR + (s >= CPR ? s - CPR : s);
    } else {
        return q;
    }
}

// Tried and rejected for balancing ACROSS CUs (a few CUs finish ~10 % late): tile tickets from
// global atomic counters.  One counter retires only ~88 atomics/us (the kernels consume ~300
// tiles/us); eight per-XCD counters with the ticket taken one or two iterations ahead still
// cost +11...23 % kernel time, because gfx950 executes every global atomic at the memory side
// (the compiler emits the same instruction for workgroup and agent scope) and the ~62 500
// read-modify-writes keep their eight HBM channels busy, which gates the interleaved
// streaming traffic.  A block-local LDS ticket counter (one 512-thread block per CU) balances
// only inside a CU and was slower than two independent 256-thread blocks.  See DESIGN.md.

// Issue-arbitration balance between the two waves of a SIMD.  With two 256-thread blocks per CU
// the wave of the block dispatched first is the older one on every SIMD and wins instruction
// issue whenever both are ready: per-wave timestamps show it finishing ~13 % earlier than its
// partner on equal work, so the chip idles through a long tail.  The younger half of the grid
// therefore raises its priority on every other iteration: each partner then wins arbitration
// about half of the time.  Speed only -- nothing depends on which blocks really share a SIMD.
__device__ __forceinline__ void balance_priority(bool younger_half, int iteration) {
    if (younger_half) {
        if (iteration & 1) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
}

// The elements behind the last full wave tile (fewer than 16 M of them): the MFMA bodies never mask or clamp, so these are
// computed with the plain per-entry code of the generic kernels (item(e, i) = output entry i of element e), one entry per
// thread, by as few blocks as that takes -- before their own tiles, while their first tiles stream in.  Which blocks: the ones
// at the END OF THE OLDER HALF of the grid.  At E = 1e6 any block hides the few microseconds this takes; at E ~ 5e4 nothing
// hides it and the launch lasts as long as its slowest block -- round 3 gave all entries to block 0 (two or three entries per
// thread, one after the other: div at E = 45 000 took 22 us against 12 for E = 44 992, profiles/r04/temporal_loads_ab.txt),
// whose waves are also among those with the most tiles.  The older half starts its tiles ~3 us before the younger one (it
// wins the issue arbitration during the prologue), and within it the highest block indices are the first to have a tile
// fewer when the tiles do not fill the last round.
template <class F>
__device__ __forceinline__ void remainder_items(int64_t e_begin, int64_t E, int Np, unsigned bid, unsigned nblk,
                                                F item) {
    const int64_t n = (E - e_begin) * Np;
    if (n <= 0) return;
    const unsigned per = blockDim.x;
    const int64_t want = (n + per - 1) / per;                       // blocks at one entry per thread
    const unsigned k = want < (int64_t)nblk ? (unsigned)want : nblk;
    const unsigned half = (nblk + 1) / 2;
    const unsigned first = half >= k ? half - k : 0;
    if (bid < first || bid >= first + k) return;
    for (int64_t idx = (int64_t)(bid - first) * per + threadIdx.x; idx < n; idx += (int64_t)k * per)
        item(e_begin + idx / Np, (int)(idx % Np));
}

}  // namespace fe
