// fe_grad.h -- grad einsum  out[x,e,i] = sum_{r,j} J[x,r,e] D[r,i,j] u[e,j]
// ('xre,rij,ej->xei', reference: test/test_codegen.py:96-113; the device kernel
// this replaces is the loopy-generated one described in
// tuning/impls/xre_rij_ej_to_xei.py:26-275 / doc/compiler_writer_tutorial.rst:99-355).
//
// MFMA kernel (fp64), templated on the number of volume nodes Np = (p+1)(p+2)(p+3)/6 for
// p = 1..4 (Np = 4, 10, 20, 35: the sizes the reference's archive tunes) and on M,
// the number of 16-element sub-tiles a wave handles per iteration.  Per WAVE and per sub-tile:
//   stage 1  tmp[(r,i), e] = sum_j D[(r,i), j] * u[e, j]     on v_mfma_f64_16x16x4_f64
//            A = D zero padded to 16 RT rows x 4 KS columns, rows permuted, resident in
//            registers for the whole kernel; B = u sub-tile (4 KS x 16) read from LDS;
//            RT x KS MFMAs (Np = 35: 7 x 9 = 63).
//   stage 2  out[x,e,i] = sum_r J[x,r,e] * tmp[(r,i), e]      on VALU, lane-local.
// The row permutation of A makes stage 2 lane-local: in the f64 16x16x4 C/D layout lane
// (g = lane>>4, n = lane&15) holds rows {g + 4q} of every 16-row tile for column n, i.e. 4 RT
// "slots" s = 4*tile + q.  RT is the smallest tile count such that the TG = floor(4 RT / 3)
// whole r-triples of the four lane groups cover every i (4 TG >= Np); slot s of lane group g is
// (r, i) = (s % 3, TG g + s / 3): every lane owns all three r's of TG consecutive i's of one
// element, so the 3x3 Jacobian combine needs no cross-lane traffic.
//   Np 35: RT 7, TG 9, KS 9 (105/112 useful rows);  20: RT 4, TG 5, KS 5;  10: RT 3, TG 4, KS 3;
//   4: RT 1, TG 1, KS 1.
// Data movement: a wave tile is 16 M elements = ONE contiguous span of u (16 M Np doubles) and
// of each out plane; low orders have little data per element, so M > 1 (20: 2, 10: 3, 4: 5) keeps
// a few KB moving per LDS-DMA batch.  The u tile and the 9 x 16 M Jacobian entries come in by
// 16-byte LDS-DMA one tile ahead (8-byte aligned sources are fine: tools/align_test.hip);
// results are transposed through wave-private LDS buffers so that the global stores write
// contiguous 16-byte chunks of out[x, e0 + 16 m : e0 + 16 m + 16, :].  Waves never synchronise
// with each other after the one-time operator staging.
// Round 5 (one sub-tile per wave tile, i.e. p = 4): the A fragments are read from the staged operator INSIDE stage 1 of a
// wave's first tile (kFusedFirst: the static-walk kernels), and a ragged last round of a short launch runs as quarter tiles
// of four elements on v_mfma_f64_4x4x4_4b (kOpQuarterTail) -- both in grad3d_mfma_body.
#pragma once
#include <type_traits>

#include "fe_common.h"
#include "fe_generic.h"

namespace fe {

constexpr int grad_row_tiles(int np) {
    int t = 1;
    while (4 * ((4 * t) / 3) < np) ++t;
    return t;
}

template <int NP, int M>
struct GradGeom {
    static constexpr int TEL = 16 * M;                 // elements per wave tile
    static constexpr int RT = grad_row_tiles(NP);      // 16-row tiles of A
    static constexpr int TG = (4 * RT) / 3;            // r-triples (= i's) per lane group
    static constexpr int KS = (NP + 3) / 4;            // k-steps
    static constexpr int TILE_D = TEL * NP;            // doubles: u tile / one out plane of a tile
    static constexpr int SUB_D = 16 * NP;              // doubles per 16-element sub-tile
    static constexpr int U_CHUNKS = TILE_D / 2;        // 16-byte chunks
    static constexpr int U_INSTR = (U_CHUNKS + 63) / 64;
    static constexpr int J_ROW_CHUNKS = TEL / 2;       // 16-byte chunks per J row segment
    static constexpr int J_CHUNKS = 9 * J_ROW_CHUNKS;
    static constexpr int J_INSTR = (J_CHUNKS + 63) / 64;
    static constexpr int SUB_CHUNKS = SUB_D / 2;
    static constexpr int SUB_INSTR = (SUB_CHUNKS + 63) / 64;
    static constexpr int LOADS = U_INSTR + J_INSTR;    // vector-memory ops per (tile, field) unit: loads
                                                       // (J only with the first field of a tile) ...
    static constexpr int PLANE_STORES = M * SUB_INSTR; // ... and stores, per output plane
    static constexpr int STORES = 3 * PLANE_STORES;
    struct WaveLds {             // input side, one per wave from the start of the block's LDS
        double u[2][TILE_D];     // prefetch double buffer
        double j[2][9 * TEL];    // J[x*3+r][e0 + 0..TEL-1], double buffered (by tile)
    };
    struct WaveOut {             // output side, one per wave behind the four WaveLds
        double o[2][SUB_D];      // output transposition buffers, alternating
    };
    static constexpr int WAVES = 4;
    static constexpr int OP_D = 3 * NP * NP;           // operator doubles (staged once per block)
    // The operator is staged over the OUTPUT buffers, which nobody needs before the first
    // tile's stage 2: the input buffers are free from the first instruction, so the first two
    // tiles stream in while the prologue runs.
    static constexpr int IN_BYTES = (int)sizeof(WaveLds) * WAVES;
    static constexpr int OUT_BYTES = (int)sizeof(WaveOut) * WAVES;
    static constexpr int OP_BYTES = (OP_D * 8 + 15) / 16 * 16;
    static constexpr int LDS_BYTES = IN_BYTES + (OUT_BYTES > OP_BYTES ? OUT_BYTES : OP_BYTES);
    static_assert(4 * TG >= NP, "row permutation must cover every i");
    static_assert(LOADS + STORES <= 60, "counted vmcnt must fit the 6-bit field");
    static_assert(2 * LDS_BYTES <= 160 * 1024, "two blocks per CU");
};

// Operands of a grad-type launch.  Output plane x of field k is
//     out[k][x][e, i] = sum_r j[x][r, e] * (sum_j D[r, i, j] u[k][e, j])
// with j[x] a [3][E] array.  grad 'xre,rij,ej->xei': j[x] = J + 3 x E and out[k][x] = out_k + x E Np.
// A field may leave planes out (null pointer): the curl-type batch 're,rji,ej->ei' x 12 of
// tuning/impls/re_rji_ej_to_ei_3d_cross_product_v0.py:220-231 is six fields with two planes each
// over three [3][E] arrays, so that D u is formed once per field and not once per output.
// Every field of a launch has the same number of planes (nx).
struct GradFields {
    const double* j[3];
    const double* u[kMaxFields];
    double* out[kMaxFields][3];
};

__device__ __forceinline__ const double* grad_field_u(const GradFields& P, int k) {   // see field_in
    const double* p = P.u[0];
#pragma unroll
    for (int q = 1; q < kMaxFields; ++q) p = (k == q) ? P.u[q] : p;
    return p;
}
__device__ __forceinline__ double* grad_plane_out(const GradFields& P, int k, int x) {   // x: literal
    double* p = P.out[0][x];
#pragma unroll
    for (int q = 1; q < kMaxFields; ++q) p = (k == q) ? P.out[q][x] : p;
    return p;
}

// Plain VALU code, any Np: entry (e, i) of all three planes.  Correctness reference on the device,
// the path for shapes the MFMA kernel is not compiled for, and the remainder behind the last tile.
__device__ __forceinline__ void grad3d_item(const double* __restrict__ J, const double* __restrict__ D,
                                            const double* __restrict__ u, double* __restrict__ out, int64_t E,
                                            int Np, int64_t e, int i, int opT) {
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    const double* ue = u + e * Np;
    const int si = opT ? 1 : Np, sj = opT ? Np : 1;   // opT: D stored as [r][j][i]
    const double* d0 = D + (int64_t)0 * Np * Np + (int64_t)i * si;
    const double* d1 = D + (int64_t)1 * Np * Np + (int64_t)i * si;
    const double* d2 = D + (int64_t)2 * Np * Np + (int64_t)i * si;
#pragma unroll 5
    for (int j = 0; j < Np; ++j) {
        const double uj = ue[j];
        t0 += d0[j * sj] * uj;
        t1 += d1[j * sj] * uj;
        t2 += d2[j * sj] * uj;
    }
    for (int x = 0; x < 3; ++x)
        out[((int64_t)x * E + e) * Np + i] =
            J[(int64_t)(x * 3 + 0) * E + e] * t0 + J[(int64_t)(x * 3 + 1) * E + e] * t1 +
            J[(int64_t)(x * 3 + 2) * E + e] * t2;
}

__global__ __launch_bounds__(256) void grad3d_generic_kernel(
    const double* __restrict__ J, const double* __restrict__ D, const double* __restrict__ u,
    double* __restrict__ out, int64_t E, int Np, int64_t e_begin, int opT) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (E - e_begin) * Np) return;
    grad3d_item(J, D, u, out, E, Np, e_begin + idx / Np, (int)(idx % Np), opT);
}

// wait until at most BASE + nx * PER vector-memory operations are outstanding
template <int BASE, int PER>
__device__ __forceinline__ void wait_vmcnt_planes(int nx) {
    if (nx == 3) wait_vmcnt<BASE + 3 * PER>();
    else if (nx == 2) wait_vmcnt<BASE + 2 * PER>();
    else wait_vmcnt<BASE + PER>();
}

// `temporal` (wave-uniform; fe_common.h, kOpLoadsTemporal): plain loads, one scalar branch for the whole tile
template <int NP, int M, bool kNT = true>
__device__ __forceinline__ void grad_issue_u(const double* __restrict__ u, int64_t tile, int lane,
                                             unsigned lds_u, bool temporal = false) {
    using G = GradGeom<NP, M>;
    const char* ub = reinterpret_cast<const char*>(u) + tile * (G::TEL * NP * 8) + lane * 16;
    if (!kNT || temporal) {
#pragma unroll
        for (int c = 0; c < G::U_INSTR; ++c)
            if ((c + 1) * 64 <= G::U_CHUNKS || c * 64 + lane < G::U_CHUNKS) glds16(ub + c * 1024, lds_u + c * 1024);
    } else {
#pragma unroll
        for (int c = 0; c < G::U_INSTR; ++c)
            if ((c + 1) * 64 <= G::U_CHUNKS || c * 64 + lane < G::U_CHUNKS) glds16_nt(ub + c * 1024, lds_u + c * 1024);
    }
}

// kPlain: the planes are those of one 'xre,rij,ej->xei' (j[x] = j[0] + 3 x E, out[k][x] =
// out[k][0] + x E Np, all three wanted) -- no per-lane pointer selects, no plane tests.
template <int NP, int M, bool kPlain>
__device__ __forceinline__ void grad_issue_j(const GradFields& P, int64_t E, int64_t tile, int lane,
                                             unsigned lds_j) {
    using G = GradGeom<NP, M>;
#pragma unroll
    for (int c = 0; c < G::J_INSTR; ++c) {
        const int q = c * 64 + lane;                      // chunk -> (row x*3 + r, column chunk)
        const int row = q / G::J_ROW_CHUNKS, col = q - row * G::J_ROW_CHUNKS;
        const double* jrow;
        if (kPlain) {
            jrow = P.j[0] + (int64_t)row * E;
        } else {
            const int x = row / 3, r = row - 3 * x;
            jrow = (x == 0 ? P.j[0] : x == 1 ? P.j[1] : P.j[2]) + (int64_t)r * E;
        }
        const char* src = reinterpret_cast<const char*>(jrow + tile * G::TEL) + col * 16;
        if ((c + 1) * 64 <= G::J_CHUNKS || q < G::J_CHUNKS) glds16(src, lds_j + c * 1024);
    }
}

#ifdef FE_EXPERIMENTS
// Diagnostic build only (kDbg & 32); never read by any kernel, fetched by fe_dbg_read_*():
// shader cycles / 100 MHz ticks of wave 0's main loop, and per-wave 100 MHz timestamps
// {kernel entry, main-loop start, main-loop end, XCC_ID | HW_ID << 8}.
__device__ unsigned long long fe_dbg_clock[2];
__device__ unsigned long long fe_dbg_stamps[4096][4];
__device__ unsigned long long fe_dbg_phase[4096][4];   // prologue: operator landed, barrier passed, fragments built, second barrier passed
// per wave and per tile (the first four): 100 MHz stamps {loads landed, matrix work issued, stores issued, -} (grad);
// {loads landed, B fragments built and next loads issued, matrix work issued, stores issued} (div, kDbg & 128).  Kept in 128 bytes
// of LDS per wave behind the kernel's own (an LDS write counts on lgkmcnt, which the compiler tracks; a global store per stamp
// would count on vmcnt and break the kernels' counted waits) and copied out when the wave ends.
__device__ unsigned long long fe_dbg_tile[4096][16];
constexpr int kDbgTileLdsBytes = 4 * 128;
#define FE_TILE_STAMP(ENABLED, LDS_END, WAVE, LANE, IT, K)                                                              \
    do {                                                                                                                 \
        if ((ENABLED) && (IT) < 4) {                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                                           \
            if ((LANE) == 0) reinterpret_cast<unsigned long long*>(LDS_END)[(WAVE) * 16 + (IT) * 4 + (K)] = now_;        \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
        }                                                                                                                \
    } while (0)
#define FE_TILE_STAMPS_OUT(ENABLED, LDS_END, WAVE, LANE, GLOBAL_WAVE)                                                    \
    do {                                                                                                                 \
        if ((ENABLED) && (GLOBAL_WAVE) < 4096 && (LANE) < 16)                                                            \
            fe::fe_dbg_tile[GLOBAL_WAVE][LANE] = reinterpret_cast<unsigned long long*>(LDS_END)[(WAVE) * 16 + (LANE)];   \
    } while (0)
#else
#define FE_TILE_STAMP(ENABLED, LDS_END, WAVE, LANE, IT, K) do {} while (0)
#define FE_TILE_STAMPS_OUT(ENABLED, LDS_END, WAVE, LANE, GLOBAL_WAVE) do {} while (0)
#endif

// kDbg: experiment flags, 0 in the product build (tools/fe_check.cpp "ab" mode uses the others
// through build/libfeinsum_hip_exp.so): 1 skip MFMAs, 2 skip stores, 4 plain (temporal) stores,
// 8 skip loads, 16 plain (temporal) loads, 32 per-wave timestamps, 64 no priority balancing.
// Product: non-temporal on both sides -- every byte is touched once (A/B on MI355X: -2.5 %
// kernel time, -7 % for the data-movement skeleton).
// nb: fields per launch ('xre,rij,ej->xei' x NB sharing J and D: tuning/impls/
// batched_xre_rij_ej_to_xei.py): J is loaded once per tile and serves all nb fields;
// the wave walks (tile, field) units, field fastest.
// bid / nblk: this block's index and the number of blocks walking the tiles (blockIdx.x /
// gridDim.x for the plain kernel; the fused launches of fe_fused.h run several bodies in turn).
// kPrep: the A fragments come from a prepared operator (fe_prepare_operator; `prep` = its grad
// section) instead of being rebuilt from D: no LDS staging, no block barrier -- the waves of a block
// never meet.  D itself is still needed by the remainder code.
// kDyn (plain single-field launches): the LAST rounds of the walk are handed out by tickets -- see "dynamic tail" below;
// `tail` = the launch's ticket counters (null: static walk), `t_static` = number of statically walked tiles.
template <int NP, int M, int kDbg = 0, bool kPlain = true, bool kPrep = false, bool kDyn = false>
__device__ __forceinline__ void grad3d_mfma_body(
    const GradFields& P, const double* __restrict__ D, const void* __restrict__ prep, int nb, int nx_, int64_t E,
    int64_t nTiles, int op_flags, const unsigned bid, const unsigned nblk, unsigned* __restrict__ tail = nullptr,
    int64_t t_static = 0) {
    static_assert(!kDyn || (kPlain && !kPrep), "dynamic walk: plain launches of one field");
    const int opT = op_flags & 1;                                  // operator stored transposed
    const bool tload = (op_flags & kOpLoadsTemporal) != 0;         // the u tiles by plain loads (fe_common.h)
    const bool wthrough = kPlain && (op_flags & kOpStoresWriteThrough) != 0;   // short launches: fe_common.h
    const int nx = kPlain ? 3 : nx_;
    using G = GradGeom<NP, M>;
    using WaveLds = typename G::WaveLds;
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef FE_EXPERIMENTS
    const unsigned long long t_entry = (kDbg & 32) ? __builtin_amdgcn_s_memrealtime() : 0;
#endif
    if constexpr (kPlain && !kPrep && !kDyn && M == 1) {   // (fe_common.h: every second CU of an XCD starts half a tile period late)
        if ((op_flags & kOpStaggeredStart) && ((bid >> 3) & 1))
            for (int i = 0; i < kStaggerSleeps; ++i) __builtin_amdgcn_s_sleep(16);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds* L = reinterpret_cast<WaveLds*>(smem) + wave;
    typename G::WaveOut* LO = reinterpret_cast<typename G::WaveOut*>(smem + G::IN_BYTES) + wave;
    const int n = lane & 15, g = lane >> 4;
    // Experiment switches (the `kDbg` template constant): they exist in the experiments build only (tools/build_experiments.sh);
    // in the product every one of them is the constant `false`.
#ifdef FE_EXPERIMENTS
    constexpr bool x_no_mfma = (kDbg & 1) != 0, x_no_stores = (kDbg & 2) != 0, x_plain_stores = (kDbg & 4) != 0, x_no_loads = (kDbg & 8) != 0,
                   x_plain_loads = (kDbg & 16) != 0, x_stamps = (kDbg & 32) != 0, x_no_balance = (kDbg & 64) != 0, x_split_walk = (kDbg & 128) != 0;
#else
    static_assert(kDbg == 0, "experiment flags: experiments build only");
    constexpr bool x_no_mfma = false, x_no_stores = false, x_plain_stores = false, x_no_loads = false, x_plain_loads = false, x_stamps = false,
                   x_no_balance = false, x_split_walk = false;
#endif
    (void)x_stamps;
    constexpr bool kNT = !x_plain_loads;

    const int64_t stride = (int64_t)nblk * G::WAVES;
    int64_t tile = (int64_t)bid * G::WAVES + wave;
    // Quarter tiles (plain launches of one field on the static walk, one sub-tile per wave tile; fe_common.h: kOpQuarterTail --
    // the reason is that of fe_div.h).  R >= 1 full rounds leave r = nTiles mod (number of waves) tiles; when the launcher asks
    // for it (feinsum_hip.hip: grad_quarter_flag) the r tiles become 4 r quarter tiles of FOUR elements, one for each of the first
    // 4 r waves, behind the wave's last full tile.  Stage 1 of a quarter tile runs on v_mfma_f64_4x4x4_4b with the SAME A fragments (its four
    // blocks are the four 4-row slices of a 16-row fragment; B = the four elements' values, replicated over the blocks): lane (g, n)
    // receives row g + 4 (n >> 2) of every 16-row tile for element n & 3, i.e. slot 4 t + (n >> 2) of lane group g.  The three r of
    // an i are then on different lanes, so stage 2 goes through LDS: tmp[element][g][slot], and every lane combines whole (element, i)
    // entries -- consecutive lanes = consecutive doubles of out[x, q_e0 : q_e0 + 4, :], stored straight from registers.
    // Same products in the same order as a full tile: bitwise the same results.
    int64_t q_e0 = -1;            // first element of this wave's quarter tile, or -1
    int64_t t_full = nTiles;      // tiles walked as full tiles
    if constexpr (kPlain && !kPrep && M == 1) {
        if ((op_flags & kOpQuarterTail) && nb == 1 && !(kDyn && tail != nullptr)) {
            const int64_t r = nTiles % stride;
            if (r > 0 && 4 * r <= stride && nTiles > stride) {   // (what is possible; the launcher sets the flag by its own, narrower rule)
                t_full = nTiles - r;
                const int64_t w = (int64_t)bid * G::WAVES + wave;
                if (w < 4 * r) q_e0 = t_full * G::TEL + 4 * w;
            }
        }
    }
    const int64_t tEnd = t_full;

    double afrag[G::RT][G::KS];
    // (one sub-tile per wave tile) stage 1 of the wave's first unit runs with the fragment build: see the prologue
    constexpr bool kFusedFirst = (M == 1) && !kPrep && !kDyn && !(x_no_mfma || x_no_stores || x_plain_stores || x_no_loads || x_plain_loads || x_no_balance || x_split_walk);   // (the dynamic-walk kernels have no registers to spare for it)
    v4d acc_first[G::RT];
    bool first_ready = false;
    // experiment (kDbg & 128): the walk covers both halves of the element range at once (see fe_div.h, kDbg & 4)
    const int64_t half_tiles = (nTiles + 1) / 2;
    auto phys = [&](int64_t t) -> int64_t { return x_split_walk ? ((t & 1) ? half_tiles + (t >> 1) : (t >> 1)) : t; };
    const bool dyn = kDyn && tail != nullptr && t_static < nTiles;   // grid-uniform
    bool pre = false;   // unit 1 already requested
    // the loads of this wave's first unit -- and, with several fields, of the second field's u -- behind the operator copy / the fragment loads
    auto issue_first_units = [&]() -> int {   // returns the number of vector-memory ops that may stay in flight
        if (!(tile < tEnd) || x_no_loads) return 0;
        grad_issue_u<NP, M, kNT>(P.u[0], phys(tile), lane, lds_addr_uniform(L->u[0]), tload);
        grad_issue_j<NP, M, kPlain>(P, E, phys(tile), lane, lds_addr_uniform(L->j[0]));
        if (nb > 1) {
            grad_issue_u<NP, M, kNT>(P.u[1], phys(tile), lane, lds_addr_uniform(L->u[1]), tload);
            pre = true;
            return 1;
        }
        // (one field: the SECOND tile is requested at the top of the first step, one tile ahead like every other -- until round 5 it
        //  went out here as well, and the other block's operator, vector-memory data returning in order per CU, waited behind twice
        //  the tile data: E = 1e5 22.4 -> 21.9 us, 1.4e5 -3.7 %, 5e5 -1 %, 1.31e5 +1 %: profiles/r05/grad_second_tile_requested_later_abl.txt)
        return 3;
    };
    // The elements behind the last full tile (fe_common.h: remainder_items), entry by entry on the VALU -- with the operator read from
    // `Dsrc`: the block's LDS copy while it exists (lanes walk i, 280 bytes apart: from global memory every load instruction touches
    // 35 cache lines, ~1.7 us per wave for the 105 of an entry; from LDS they are 2-way bank conflicts), else global memory.
    auto remainder = [&](const double* Dsrc) {
        remainder_items(nTiles * G::TEL, E, NP, bid, nblk, [&](int64_t e, int i) {
            for (int k = 0; k < nb; ++k) {
                const double* uk = grad_field_u(P, k);
                if (kPlain) {
                    grad3d_item(P.j[0], Dsrc, uk, grad_plane_out(P, k, 0), E, NP, e, i, opT);
                } else {
                    double* const o[3] = {grad_plane_out(P, k, 0), grad_plane_out(P, k, 1), grad_plane_out(P, k, 2)};
#pragma unroll
                    for (int x = 0; x < 3; ++x)
                        if (o[x]) divcomp3d_item(P.j[x], Dsrc, uk, o[x], E, NP, e, i, opT, 0);
                }
            }
        });
    };
    int dbg_it = 0;   // (experiments build: units done by this wave, for the per-tile stamps)
#ifdef FE_EXPERIMENTS
    if ((kDbg & 32) && lane < 16) reinterpret_cast<unsigned long long*>(smem + G::LDS_BYTES)[wave * 16 + lane] = 0;
#endif
    // one (tile, field) unit: stage 1, stage 2 and the transposed stores, from the u tile `ut` and the J tile `jt` in LDS
    auto compute_unit = [&](int64_t tile_, int fk, const double* ut, const double* jt, auto with_acc) {
        constexpr bool kHaveAcc = decltype(with_acc)::value;   // stage 1 of this unit is in acc_first already
        FE_TILE_STAMP(x_stamps, smem + G::LDS_BYTES, wave, lane, dbg_it, 0);   // this unit's loads have landed
        double* out_x[3];
        out_x[0] = grad_plane_out(P, fk, 0);
        out_x[1] = kPlain ? out_x[0] + E * NP : grad_plane_out(P, fk, 1);
        out_x[2] = kPlain ? out_x[0] + 2 * E * NP : grad_plane_out(P, fk, 2);
        int obuf = 0;
        const int64_t e0 = phys(tile_) * G::TEL;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            // ---- stage 1 on sub-tile m
            v4d acc[G::RT];
            if constexpr (kHaveAcc) {
#pragma unroll
                for (int t = 0; t < G::RT; ++t) acc[t] = acc_first[t];
            } else {
                double bfrag[G::KS];
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    const int j = 4 * ks + g;
                    const double b = ut[(16 * m + n) * NP + (j < NP ? j : 0)];
                    bfrag[ks] = (j < NP) ? b : 0.0;
                }
#pragma unroll
                for (int t = 0; t < G::RT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
                if (x_no_mfma) {
#pragma unroll
                    for (int t = 0; t < G::RT; ++t)
                        acc[t] = v4d{bfrag[t % G::KS], bfrag[(t + 1) % G::KS], bfrag[(t + 2) % G::KS], afrag[t][0]};
                } else {
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
                        for (int t = 0; t < G::RT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[t][ks], bfrag[ks], acc[t], 0, 0, 0);
                }
            }

            FE_TILE_STAMP(x_stamps, smem + G::LDS_BYTES, wave, lane, dbg_it, 1);   // the matrix work is issued
            // ---- stage 2 + transposed store, plane by plane
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                if (!kPlain && out_x[x] == nullptr) continue;   // plane not asked for (wave-uniform)
                double* ob = LO->o[kPlain ? (m * 3 + x) & 1 : obuf];
                obuf ^= 1;
                const double j0 = jt[(x * 3 + 0) * G::TEL + 16 * m + n];
                const double j1 = jt[(x * 3 + 1) * G::TEL + 16 * m + n];
                const double j2 = jt[(x * 3 + 2) * G::TEL + 16 * m + n];
#pragma unroll
                for (int k = 0; k < G::TG; ++k) {
                    const int s = 3 * k;
                    const double t0 = acc[(s + 0) >> 2][(s + 0) & 3];
                    const double t1 = acc[(s + 1) >> 2][(s + 1) & 3];
                    const double t2 = acc[(s + 2) >> 2][(s + 2) & 3];
                    const double v = __builtin_fma(j2, t2, __builtin_fma(j1, t1, j0 * t0));   // explicit: no contraction choice left to the compiler
                    const int i = G::TG * g + k;
                    if (G::TG * 3 + k < NP || i < NP) ob[n * NP + i] = v;
                }
                wave_lds_fence();
                double* op = out_x[x] + (e0 + 16 * m) * NP;
                if constexpr (kPlain && !(x_no_mfma || x_no_stores || x_plain_stores || x_no_loads || x_plain_loads || x_no_balance || x_split_walk)) {
                    if (wthrough) {   // all values out of LDS first, then the stores back to back (as the compiler orders its own)
                        v2d vals[G::SUB_INSTR];
#pragma unroll
                        for (int c = 0; c < G::SUB_INSTR; ++c) {
                            const int q = c * 64 + lane;
                            vals[c] = ((c + 1) * 64 <= G::SUB_CHUNKS || q < G::SUB_CHUNKS) ? *reinterpret_cast<const v2d*>(ob + 2 * q) : v2d{0.0, 0.0};
                        }
                        store_tile_held<G::SUB_INSTR, G::SUB_CHUNKS>(op, lane, vals, true);
                        wave_lds_fence();
                        continue;
                    }
                }
#pragma unroll
                for (int c = 0; c < G::SUB_INSTR; ++c) {
                    const int q = c * 64 + lane;
                    if ((c + 1) * 64 <= G::SUB_CHUNKS || q < G::SUB_CHUNKS) {
                        const v2d val = *reinterpret_cast<const v2d*>(ob + 2 * q);
                        if (x_no_stores) { if (val[0] == 1.2345e-300) op[2 * q] = val[1]; }   // keep the value live
                        else if (x_plain_stores) *reinterpret_cast<v2d*>(op + 2 * q) = val;
                        else __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + 2 * q));
                    }
                }
                wave_lds_fence();
            }
        }
        FE_TILE_STAMP(x_stamps, smem + G::LDS_BYTES, wave, lane, dbg_it, 2);   // the stores are issued
        ++dbg_it;
    };

#ifdef FE_EXPERIMENTS
    unsigned long long c0 = 0, r0 = 0;
    auto write_stamps = [&](int tiles_done) {
        FE_TILE_STAMPS_OUT(kDbg & 32, smem + G::LDS_BYTES, wave, lane, bid * G::WAVES + wave);
        if (!((kDbg & 32) && lane == 0)) return;
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        const int w = bid * G::WAVES + wave;
        if (w < 4096) {
            fe_dbg_stamps[w][0] = t_entry; fe_dbg_stamps[w][1] = r0; fe_dbg_stamps[w][2] = t_end;
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID[3:0]
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
            fe_dbg_stamps[w][3] = xcc | ((unsigned long long)hw << 8) | ((unsigned long long)tiles_done << 40);
        }
        if (w == 0) { fe_dbg_clock[0] = __builtin_amdgcn_s_memtime() - c0; fe_dbg_clock[1] = t_end - r0; }
    };
#endif
    // ---- the quarter tile of this wave (see the top): its loads -- four rows of u (contiguous), nine times four doubles of J, compact
    //      as j[k * 4 + element] -- and, behind the loop, the unit itself
    constexpr int kQuarterChunks = 4 * NP / 2;                             // 16-byte chunks of four rows of u
    constexpr int kQuarterLoads = (kQuarterChunks + 63) / 64 + 1;
    auto issue_quarter_loads = [&](unsigned lds_u, unsigned lds_j) {
        const char* ub_ = reinterpret_cast<const char*>(P.u[0]) + q_e0 * (NP * 8);
#pragma unroll
        for (int c = 0; c < (kQuarterChunks + 63) / 64; ++c)
            if (c * 64 + lane < kQuarterChunks) {
                if (tload) glds16(ub_ + (c * 64 + lane) * 16, lds_u + c * 1024);
                else glds16_nt(ub_ + (c * 64 + lane) * 16, lds_u + c * 1024);
            }
        if (lane < 18) glds16(reinterpret_cast<const char*>(P.j[0]) + ((int64_t)(lane >> 1) * E + q_e0) * 8 + (lane & 1) * 16, lds_j);
    };
    // ---- the walk.  A step has a TOP (the next unit's loads and, under the dynamic walk, the ticket traffic go out; wait for this
    //      unit's loads) and a BOTTOM (stage 1, stage 2, the transposed stores; advance).  The top of a wave's first step runs in the
    //      prologue, in front of the fragment build that is fused with that unit's stage 1 (kFusedFirst).
    const bool younger_half = !x_no_balance && bid >= (nblk + 1) / 2;
    int iteration = 0;
    // (a) dynamic walk of one field (fe_common.h): static tiles first + k stride below t_static, then tickets.
    //     Vector-memory ops of an iteration in issue order: [A = ticket for the tile after next] L(next) S(cur);
    //     every wave has a static first tile (t_static >= number of waves).
    const bool use_dyn = kDyn && dyn && nb == 1;   // grid-uniform
    constexpr int NL = G::LOADS, NS = G::STORES;
    const int pool = (bid >> 3) & (kTailPools - 1);
    unsigned* const counter = tail_pool_counters(tail, pool);
    unsigned* const done = tail_pool_reports(counter);   // the pool's report counter, half a stride behind its tickets
    auto static_next = [&](int64_t t) -> int64_t { return (t < t_static && t + stride < t_static) ? t + stride : -1; };
    int64_t cur = tile, nxt = -1;   // nxt >= 0 at the start: requested by the prologue (pre)
    bool pending = false, reported = false, prev_pre = false, dfirst = true;
    int buf = 0;
    auto dyn_init = [&]() {   // behind the prologue's loads
        nxt = static_next(tile);
        if (nxt < 0) {   // one static round: the prologue's ticket, behind L(cur)
            tail_request<0>(counter);
            pending = true;
        }
    };
    auto dyn_top = [&]() {
        balance_priority(younger_half, iteration++);   // dyn
        bool extra = false;   // one more vector-memory op (ticket or report) issued in this iteration
        if (pending) {   // the next tile comes from a ticket: younger than it are L(cur) and S(previous)
            const unsigned t = dfirst ? tail_wait<0, 0>() : prev_pre ? tail_wait<NS, 0>() : tail_wait<NL + NS, 0>();
            nxt = tail_ticket_tile(t, t_static, pool, tEnd);
            pending = false;
            if (nxt < 0) {   // this wave's pool is empty: stop asking, report
                tail_request<1>(done);
                reported = true;
                extra = true;
            }
        }
        if (nxt >= 0) {
            if (static_next(nxt) < 0) {   // the tile after next is not static
                tail_request<0>(counter);
                pending = true;
                extra = true;
            }
            if (!pre) {
                grad_issue_u<NP, M, kNT>(P.u[0], nxt, lane, lds_addr_uniform(L->u[buf ^ 1]), tload);
                grad_issue_j<NP, M, kPlain>(P, E, nxt, lane, lds_addr_uniform(L->j[buf ^ 1]));
            }
        }
        // wait L(cur): younger are S(previous), the ticket / report, L(next)
        if (nxt >= 0) {
            if (dfirst) { if (extra) wait_vmcnt<NL + 1>(); else wait_vmcnt<NL>(); }
            else { if (extra) wait_vmcnt<NS + NL + 1>(); else wait_vmcnt<NS + NL>(); }
        } else {
            if (dfirst) { if (extra) wait_vmcnt<1>(); else wait_vmcnt<0>(); }
            else { if (extra) wait_vmcnt<NS + 1>(); else wait_vmcnt<NS>(); }
        }
    };
    auto dyn_bottom = [&](auto with_acc) {
        compute_unit(cur, 0, L->u[buf], L->j[buf], with_acc);
        dfirst = false;
        prev_pre = pre;
        pre = false;
        cur = nxt;
        buf ^= 1;
        if (cur >= 0 && !pending) nxt = static_next(cur);
    };
    // (b) static walk, any number of fields -- and the dynamic walk with b >= 2 fields (units (tile, field), field fastest): the
    //     ticket for the next tile is asked for at the top of the tile's first field -- in front of the next unit's loads -- and
    //     read at the top of its last field, where the wait for it (everything but the previous unit's stores) is the wait for
    //     this unit's loads as well
    int ub = 0, jbuf = 0;     // u buffer toggles per (tile, field) unit, J buffer per tile
    bool first = true;
    int fk = 0;
    const bool dynb = kDyn && dyn && nb >= 2;   // grid-uniform
    bool pendingb = false, reportedb = false;
    bool next_new_tile = true;   // (set by the top, used by the bottom)
    int64_t nt = 0;
    int nk = 0;
    auto static_top = [&]() {
        balance_priority(younger_half, iteration++);
        // Vector-memory ops in issue order: L(unit) S(previous unit) [ticket] L(next unit) | wait L(unit).
        // The stores of the previous unit and the loads of the next one are younger than this
        // unit's loads and stay in flight.
        next_new_tile = (fk + 1 == nb);
        nt = next_new_tile ? tile + stride : tile;
        nk = next_new_tile ? 0 : fk + 1;
        bool extra = false;   // a ticket or the report goes out at this top
        if constexpr (kDyn) {
            if (dynb) {
                const bool successor_static = tile < t_static && tile + stride < t_static;
                if (next_new_tile) {
                    if (pendingb) {
                        nt = tail_ticket_tile(tail_wait<G::STORES, 0>(), t_static, pool, tEnd);
                        pendingb = false;
                        if (nt < 0) {   // this wave's pool is empty: stop asking, report
                            tail_request<1>(done);
                            reportedb = true;
                            extra = true;
                            nt = tEnd;
                        }
                    } else if (!successor_static) {
                        nt = tEnd;   // (cannot happen: a tile whose successor is not static has asked at its first field)
                    }
                } else if (fk == 0 && !successor_static) {
                    tail_request<0>(counter);
                    pendingb = true;
                    extra = true;
                }
            }
        }
        if (x_no_loads) {
            wait_vmcnt<0>();
        } else if (nt < tEnd) {
            if (!pre) grad_issue_u<NP, M, kNT>(grad_field_u(P, nk), phys(nt), lane, lds_addr_uniform(L->u[ub ^ 1]), tload);
            if (next_new_tile) {
                if (!pre) grad_issue_j<NP, M, kPlain>(P, E, phys(nt), lane, lds_addr_uniform(L->j[jbuf ^ 1]));
                if (x_no_stores || first) wait_vmcnt<G::LOADS>();
                else wait_vmcnt_planes<G::LOADS, G::PLANE_STORES>(nx);
            } else if (kDyn && extra) {
                if (first) wait_vmcnt<G::U_INSTR + 1>();
                else wait_vmcnt<G::U_INSTR + G::STORES + 1>();
            } else {
                if (x_no_stores || first) wait_vmcnt<G::U_INSTR>();
                else wait_vmcnt_planes<G::U_INSTR, G::PLANE_STORES>(nx);
            }
        } else if (q_e0 >= 0) {   // (static walk, one field) the last full tile: behind its loads go the quarter tile's
            issue_quarter_loads(lds_addr_uniform(L->u[ub ^ 1]), lds_addr_uniform(L->j[jbuf ^ 1]));
            if (first) wait_vmcnt<kQuarterLoads>();
            else wait_vmcnt<kQuarterLoads + G::STORES>();
        } else {
            if (kDyn && extra) wait_vmcnt<G::STORES + 1>();
            else if (first || x_no_stores) wait_vmcnt<0>();
            else wait_vmcnt_planes<0, G::PLANE_STORES>(nx);
        }
        first = false;
        pre = false;
    };
    auto static_bottom = [&](auto with_acc) {
        compute_unit(tile, fk, L->u[ub], L->j[jbuf], with_acc);
        fk = nk;
        tile = nt;
        ub ^= 1;
        if (next_new_tile) jbuf ^= 1;
    };
    auto first_top = [&]() {   // (kFusedFirst: from the prologue)
        if constexpr (kDyn) {
            if (use_dyn) {
                dyn_init();
                dyn_top();
                return;
            }
        }
        static_top();
    };

    if constexpr (kPrep) {
        load_prepared_fragments<G::RT * G::KS>(prep, lane, [&](int f, double v) { afrag[f / G::KS][f % G::KS] = v; });
        issue_first_units();
        prepared_fragments_landed();
        remainder(D);
    } else {
        // ---- operator -> LDS (DMA), and behind it the loads of this wave's first two units
        stage_operator_dma<G::OP_D>(D, lds_addr_uniform(smem + G::IN_BYTES), wave, lane);
        const int units_issued = issue_first_units();
        switch (units_issued) {
            case 1: wait_vmcnt<G::LOADS + G::U_INSTR>(); break;
            case 3: wait_vmcnt<G::LOADS>(); break;
            default: wait_vmcnt<0>(); break;
        }
#ifdef FE_EXPERIMENTS
        if ((kDbg & 32) && lane == 0 && bid * G::WAVES + wave < 4096) fe_dbg_phase[bid * G::WAVES + wave][0] = __builtin_amdgcn_s_memrealtime();
#endif
        __syncthreads();
#ifdef FE_EXPERIMENTS
        if ((kDbg & 32) && lane == 0 && bid * G::WAVES + wave < 4096) fe_dbg_phase[bid * G::WAVES + wave][1] = __builtin_amdgcn_s_memrealtime();
#endif

        // ---- A fragments from the staged operator (addresses = row part + column part: the 63
        //      fragments of p = 4 cost one add and one LDS read each)
        const double* dl = reinterpret_cast<const double*>(smem + G::IN_BYTES);
        remainder(dl);     // (while the block's copy of the operator is there; in front of the build: few registers are live here)
        const int gp = n & 3, q = n >> 2;
        const int istride = opT ? 1 : NP, jstride = opT ? NP : 1;   // opT: D stored as [r][j][i]
        if constexpr (kFusedFirst) {
            // One sub-tile per wave tile (p = 4): the build runs INSIDE stage 1 of the wave's first unit.  The fragments of k-step
            // ks + 1 are read (LDS reads issue beside the wave's own MFMAs; its VALU instructions do not: DESIGN.md section 3h)
            // while the MFMAs of k-step ks run, so that the matrix pipe starts ~1 us earlier on the older wave of a SIMD and the
            // younger wave's build no longer crawls beside its partner's matrix phase (3 us at E = 1e5:
            // profiles/r04/grad_1e5_stamps_and_decomposition.txt).  Only the last row tile and the last k-step hold padding.
            static_assert(4 * (G::RT - 1) <= 3 * G::TG && 3 * G::TG + (4 * (G::RT - 1) - 1) / 3 < NP && 4 * G::KS - 5 < NP,
                          "padding rows in the last row tile only, padding columns in the last k-step only");
            const double* rowp[G::RT];                // row part + this lane's column g; k-step ks adds the (uniform) 4 ks jstride
            bool rowok_last = true;
#pragma unroll
            for (int t = 0; t < G::RT; ++t) {
                const int s_ = 4 * t + q;
                const int r = s_ % 3, i = G::TG * gp + s_ / 3;
                if (t == G::RT - 1) rowok_last = (s_ < 3 * G::TG) && (i < NP);
                rowp[t] = dl + r * (NP * NP) + (i < NP ? i : 0) * istride + g * jstride;
            }
            const bool jok_last = 4 * (G::KS - 1) + g < NP;
            // (column 4 (KS - 1) + g = NP of the lanes g = 3 is read from inside the block's LDS -- one double behind the row, at most
            // NP doubles behind the operator, which the output buffers cover -- and replaced by zero)
            static_assert(4 * G::KS - NP <= 1 && (G::OP_D + NP + 1) * 8 <= G::LDS_BYTES - G::IN_BYTES, "the padding column stays inside the staging area");
            const int kstep = 4 * jstride;
            auto read_step = [&](int ks) {
#pragma unroll
                for (int t = 0; t < G::RT; ++t) afrag[t][ks] = rowp[t][ks * kstep];
            };
            auto fix_step = [&](int ks) {   // (compile-time ks: the selects exist for the last row tile and the last k-step only)
#pragma unroll
                for (int t = 0; t < G::RT; ++t) {
                    if (t == G::RT - 1 && ks == G::KS - 1) afrag[t][ks] = (rowok_last && jok_last) ? afrag[t][ks] : 0.0;
                    else if (t == G::RT - 1) afrag[t][ks] = rowok_last ? afrag[t][ks] : 0.0;
                    else if (ks == G::KS - 1) afrag[t][ks] = jok_last ? afrag[t][ks] : 0.0;
                }
            };
            if (units_issued != 0) {
                read_step(0);
                first_top();   // the top of this wave's first step: ends with its first unit landed
                const double* ub0 = L->u[0] + n * NP + g;   // B operand of k-step ks: u[e = n][4 ks + g]
                double bnext = ub0[0];
#pragma unroll
                for (int t = 0; t < G::RT; ++t) acc_first[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    double bcur = bnext;
                    if (ks + 1 < G::KS) {
                        read_step(ks + 1);
                        bnext = (ks + 1 == G::KS - 1) ? ub0[jok_last ? 4 * (ks + 1) : 0] : ub0[4 * (ks + 1)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    fix_step(ks);
                    if (ks == G::KS - 1) bcur = jok_last ? bcur : 0.0;
#pragma unroll
                    for (int t = 0; t < G::RT; ++t)
                        acc_first[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[t][ks], bcur, acc_first[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                first_ready = true;
            } else {   // a wave without a tile
#pragma unroll
                for (int t = 0; t < G::RT; ++t)
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) afrag[t][ks] = 0.0;
            }
        } else {
            int joff[G::KS];
            bool jok[G::KS];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const int j = 4 * ks + g;
                jok[ks] = j < NP;
                joff[ks] = (jok[ks] ? j : 0) * jstride;
            }
#pragma unroll
            for (int t = 0; t < G::RT; ++t) {
                const int s = 4 * t + q;
                const int r = s % 3, i = G::TG * gp + s / 3;
                const bool rowok = (s < 3 * G::TG) && (i < NP);
                const double* row = dl + r * (NP * NP) + (i < NP ? i : 0) * istride;
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    const double v = row[joff[ks]];
                    afrag[t][ks] = (rowok && jok[ks]) ? v : 0.0;
                }
            }
        }
#ifdef FE_EXPERIMENTS
        if ((kDbg & 32) && lane == 0 && bid * G::WAVES + wave < 4096) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            fe_dbg_phase[bid * G::WAVES + wave][2] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        __syncthreads();   // the staging area becomes the waves' output buffers
#ifdef FE_EXPERIMENTS
        if ((kDbg & 32) && lane == 0 && bid * G::WAVES + wave < 4096) fe_dbg_phase[bid * G::WAVES + wave][3] = __builtin_amdgcn_s_memrealtime();
#endif
    }


#ifdef FE_EXPERIMENTS
    if (kDbg & 32) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    // ---- walk
    if constexpr (kDyn) {
        if (use_dyn) {
            if constexpr (kFusedFirst) {
                if (first_ready) dyn_bottom(std::true_type{});
                else dyn_init();
            } else {
                dyn_init();
            }
            while (cur >= 0) {
                dyn_top();
                dyn_bottom(std::false_type{});
            }
            // the last wave of a pool to report leaves the pool's two counters zeroed for the next launch (its own report is
            // older than its last tile's stores; nobody else touches this pool's counters any more)
            if (reported) {
                // waves of this pool: blocks b with (b / 8) % kTailPools == pool
                const unsigned pool_blocks = (nblk / (8 * kTailPools)) * 8 + (unsigned)max(0, min(8, (int)(nblk % (8 * kTailPools)) - 8 * pool));
                const unsigned before = tail_wait<NS, 1>();
                if (before + 1 == pool_blocks * G::WAVES && lane == 0) {
                    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#ifdef FE_EXPERIMENTS
            write_stamps(iteration);
#endif
            return;
        }
    }
    if constexpr (kFusedFirst) {
        if (first_ready) static_bottom(std::true_type{});
    }
    while (tile < tEnd) {
        static_top();
        static_bottom(std::false_type{});
    }
    if constexpr (kPlain && !kPrep && M == 1) {
        if (q_e0 >= 0) {
            wait_vmcnt<G::STORES>();                         // younger than the quarter tile's loads: the last full tile's stores
            const double* ut = L->u[ub];
            const double* jt = L->j[jbuf];
            const int el = n & 3;
            double bq[G::KS];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const int j = 4 * ks + g;
                const double b = ut[el * NP + (j < NP ? j : 0)];
                bq[ks] = (j < NP) ? b : 0.0;
            }
            double acc1[G::RT];
#pragma unroll
            for (int t = 0; t < G::RT; ++t) acc1[t] = 0.0;
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
                for (int t = 0; t < G::RT; ++t) acc1[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(afrag[t][ks], bq[ks], acc1[t], 0, 0, 0);
            double* tb = LO->o[0];                           // tmp[element][g][slot]: 16 x 4 RT doubles (448 of the buffer's 560)
            static_assert(16 * 4 * G::RT <= G::SUB_D, "the quarter tile's tmp fits one transposition buffer");
#pragma unroll
            for (int t = 0; t < G::RT; ++t) tb[(el * 4 + g) * (4 * G::RT) + 4 * t + (n >> 2)] = acc1[t];
            wave_lds_fence();
            double* const o0 = grad_plane_out(P, 0, 0) + q_e0 * NP;
#pragma unroll
            for (int c = 0; c < (4 * NP + 63) / 64; ++c) {
                const int p = c * 64 + lane;                 // entry (element p / NP, i = p % NP) of all three planes
                if (p < 4 * NP) {
                    const int e2 = p / NP, i = p - e2 * NP;
                    const int gi = i / G::TG, k = i - gi * G::TG;
                    const double* tp = tb + (e2 * 4 + gi) * (4 * G::RT) + 3 * k;
                    const double t0 = tp[0], t1 = tp[1], t2 = tp[2];
#pragma unroll
                    for (int x = 0; x < 3; ++x) {
                        const double j0 = jt[(x * 3 + 0) * 4 + e2], j1 = jt[(x * 3 + 1) * 4 + e2], j2 = jt[(x * 3 + 2) * 4 + e2];
                        const double v = __builtin_fma(j2, t2, __builtin_fma(j1, t1, j0 * t0));
                        __builtin_nontemporal_store(v, o0 + (int64_t)x * E * NP + p);
                    }
                }
            }
            wave_lds_fence();
        }
    }
    if constexpr (kDyn) {
        if (reportedb) {   // the last wave of a pool to report leaves the pool's counters zeroed (younger than the report: this unit's stores)
            const unsigned pool_blocks = (nblk / (8 * kTailPools)) * 8 + (unsigned)max(0, min(8, (int)(nblk % (8 * kTailPools)) - 8 * pool));
            const unsigned before = tail_wait<G::STORES, 1>();
            if (before + 1 == pool_blocks * G::WAVES && lane == 0) {
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#ifdef FE_EXPERIMENTS
    write_stamps(iteration);
#endif
}

template <int NP, int M, int kDbg = 0, bool kPlain = true, bool kPrep = false>
__global__ __launch_bounds__(256, 2) void grad3d_mfma_kernel(
    GradFields P, const double* __restrict__ D, const void* __restrict__ prep, int nb, int nx, int64_t E,
    int64_t nTiles, int opT) {
    grad3d_mfma_body<NP, M, kDbg, kPlain, kPrep>(P, D, prep, nb, nx, E, nTiles, opT, blockIdx.x, gridDim.x);
}

// the plain single-field launch with a dynamic tail (see fe_common.h)
// kBatched: the number of fields is the run-time argument; else one field, known to the compiler (the single-field kernel
// keeps the code it had before batched launches learned to walk dynamically)
template <int NP, int M = 1, int kDbg = 0, bool kBatched = false>
__global__ __launch_bounds__(256, 2) FE_TAIL_KERNEL_ATTR void grad3d_mfma_tail_kernel(
    GradFields P, const double* __restrict__ D, int nb, int64_t E, int64_t nTiles, int opT, unsigned* __restrict__ tail,
    int64_t t_static) {
    grad3d_mfma_body<NP, M, kDbg, true, false, true>(P, D, nullptr, kBatched ? nb : 1, 3, E, nTiles, opT, blockIdx.x, gridDim.x, tail,
                                                     t_static);
}

// The grad section of a prepared operator: fragment f = t * KS + ks of lane (g, n) is
// A[row slot 4 t + n / 4 of lane group n % 4][k = 4 ks + g] -- the value the prologue above builds.
// One block of 64 threads per fragment.
template <int NP, int M>
__global__ __launch_bounds__(64) void grad_prepare_kernel(const double* __restrict__ D, void* __restrict__ section,
                                                          int opT) {
    using G = GradGeom<NP, M>;
    const int f = blockIdx.x, lane = threadIdx.x;
    if (f >= G::RT * G::KS) return;
    const int t = f / G::KS, ks = f % G::KS;
    const int n = lane & 15, g = lane >> 4;
    const int gp = n & 3, q = n >> 2;
    const int s = 4 * t + q, r = s % 3, i = G::TG * gp + s / 3, j = 4 * ks + g;
    const bool ok = (s < 3 * G::TG) && (i < NP) && (j < NP);
    const int istride = opT ? 1 : NP, jstride = opT ? NP : 1;
    store_prepared_fragment(section, f, lane, ok ? D[r * (NP * NP) + i * istride + j * jstride] : 0.0);
}

}  // namespace fe
