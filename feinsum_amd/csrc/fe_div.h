// fe_div.h -- div einsum  out[e,i] = sum_{x,r,j} J[x,r,e] D[r,i,j] u[x,e,j]
// ('xre,rij,xej->ei'; the device kernel this replaces is the loopy-generated
// one of tuning/impls/xre_rij_xej_to_ei.py:26-249, target pseudo-C in
// tuning/impls/xre_rij_xej_to_ei_v6.py:41-110).
//
// Schedule = the opt_einsum-optimal one: first the cheap
// Jacobian contraction  Ju[r,e,j] = sum_x J[x,r,e] u[x,e,j]  (VALU, 3 FMAs per
// value, produced directly in MFMA B-fragment layout and kept in registers),
// then the dense one  out[i, e] = sum_{(j,r)} D'[i, (j,r)] * Ju[(j,r), e]  on the
// matrix cores, A = D' (Np x 3 Np, K padded to 3 * 4 * ceil(Np/4)):
//   rows 0 .. 16 BT - 1   BT = Np / 16 tiles on v_mfma_f64_16x16x4_f64 (64 cycles each),
//                         A fragments resident in registers;
//   the NR = Np - 16 BT   remaining rows in groups of four on v_mfma_f64_4x4x4_4b_f64: its four
//                         4x4x4 blocks are four groups of four elements sharing one
//                         (replicated) 4-row slice of D', so a group costs 16 cycles per
//                         k-step instead of a whole 16-row tile (64).  Its B operand layout
//                         (lane = 16 k + column) is the 16x16x4 one, so the same B registers
//                         feed both instructions; its 4-row A slices live in LDS (broadcast
//                         reads) instead of VGPRs.
//   Np 35: BT 2 + 1 group (rows 32-34);  20: BT 1 + 1 group;  10: 3 groups;  4: 1 group.
// K is ordered (jq, r) with j = 4 jq + g: one group of three u values (x = 0..2)
// feeds three consecutive k-steps.
// MODE 1 ("component", 'se,sij,ej->ei': out[e,i] = sum_{s,j} J[s,e] D[s,i,j] u[e,j], the einsum
// of test/test_codegen.py:34-66 and tuning/impls/re_rij_ej_to_ei.py) shares everything but the
// B fragment, which is the single multiply J[s,e] * u[e,j] from ONE u plane and three J rows
// (J stored [3][E], or [E][3] as in examples/dg_wave_div.py 'es,sij,ej->ei').
// MODE 2 ('e,ij,ej->ei': out[e,i] = J[e] sum_j D[i,j] u[e,j], tuning/impls/
// e_ij_ej_to_ei_no_prftch.py:30-38) and MODE 3 ('ij,ej->ei', tuning/impls/ij_ej_to_ei_no_prftch.py)
// are the same with ONE operator component (NC = 1): B fragment = J[e] u[e,j], or u[e,j] itself.
// MODE 5 is MODE 4 over separate geometry-factor arrays and output planes (GradFields of fe_grad.h): the rows of a
// batched 're,rij,ej->ei' that share u and D, at p = 5 (fe_gradplanes3d_f64; the cross-product batch of
// tuning/impls/re_rji_ej_to_ei_3d_cross_product_v0.py:220-231 forms D u once per field instead of once per row).
// MODE 4 is grad 'xre,rij,ej->xei' BY COMPONENTS: the ND products D_r u keep separate accumulators
// (same B fragment u for every r) and the epilogue contracts them with J[x, r, e], which is
// lane-local because every r has the same (row, element) accumulator layout.  It costs ND
// accumulator sets, so it is the grad kernel of the small 2-D operators (ND = 2, Np <= 21), where
// the row-permuted kernel of fe_grad.h (written for ND = 3) does not apply.  ND = 2 also turns
// MODE 0 into the div of triangles.
// Data movement as in fe_grad.h; a wave tile is 16 M elements.  The three u planes of a tile
// leave no LDS room for a second buffer at 8 waves/CU; instead ALL B fragments of a tile are
// computed up front, which frees the u buffer, and the next tile's loads are issued before
// this tile's MFMAs and stores.
// Arithmetic is position independent: every VALU contraction beside the MFMAs (B fragments  Ju = sum_x J u, the
// MODE 4 / 5 epilogues) is written as one multiply followed by explicit fused multiply-adds, so the compiler has no
// contraction choice to make differently for a wave's first (peeled) tile and its later ones -- round 2's plain
// `a * b + c * d` was contracted differently there, and a tile's bits depended on its place in the walk, i.e. on E
// and the CU count.  Only the elements behind the last full tile (remainder_items: scalar loop) are summed in
// another order.
#pragma once
#include "fe_grad.h"

namespace fe {

#ifdef FE_EXPERIMENTS
// Diagnostic build only: per wave of the eight-wave p = 5 kernels {shader cycles waiting for the tile, in the MFMA
// phase, in the epilogue; tiles done; HW_ID; 100 MHz stamps of loop start and end}.
__device__ unsigned long long fe_dbg_w8[4096][8];
#endif

// ALDS: the big-tile A fragments live in LDS (fragment layout, conflict-free 512-byte reads)
// instead of registers.  At Np = 56 (tetrahedra p = 5) they are 126 doubles per lane -- the whole
// register file -- while one fragment read per 64-cycle MFMA is only 6 % of the LDS bandwidth.
// The block then owns most of a CU's LDS: one block (4 waves) per CU.
// W8 (with ALDS, MODE 4): eight waves per block share the fragments; a wave's u tile and its output
// transposition buffer are the same LDS (the next tile is requested after the stores), which is
// what lets two waves per SIMD fit beside 75 KB of fragments -- the partner wave's MFMAs cover
// the exposed load.
constexpr int kDivWalkSplit = 2;   // op_flags bit of div3d_mfma_body

template <int NP, int M, int MODE = 0, int ND = 3, bool ALDS = false, bool W8 = false>
struct DivGeom {
    static constexpr int TEL = 16 * M;
    static constexpr int NPLANES = MODE == 0 ? ND : 1;  // u planes per tile
    static constexpr bool BYCOMP = MODE == 4 || MODE == 5;   // grad by components (5: separate J arrays / output planes)
    static constexpr int NC = (MODE == 0 || BYCOMP || MODE == 1) ? ND : 1;   // operator components r
    static constexpr int NJ = (MODE == 0 || BYCOMP) ? ND * ND : MODE == 1 ? ND : MODE == 2 ? 1 : 0;   // J values per element
    static constexpr int NBF = BYCOMP ? 1 : NC;         // distinct B fragments per k-quad
    static constexpr int NOUT = BYCOMP ? ND : 1;        // output planes
    static constexpr int KSJ = (NP + 3) / 4;            // j quads; k-steps = NC KSJ, ordered (jq, r)
    static constexpr int BT = NP / 16;                  // 16-row tiles
    static constexpr int NR = NP - 16 * BT;             // rows left for the 4x4x4 groups
    static constexpr int NS = (NR + 3) / 4;             // 4-row groups
    static constexpr int PLANE_D = TEL * NP;            // doubles: one u plane of a tile / the out tile
    static constexpr int SUB_D = 16 * NP;
    static constexpr int P_CHUNKS = PLANE_D / 2, P_INSTR = (P_CHUNKS + 63) / 64;
    static constexpr int J_ROW_CHUNKS = TEL / 2, J_CHUNKS = NJ * J_ROW_CHUNKS, J_INSTR = (J_CHUNKS + 63) / 64;
    static constexpr int SUB_CHUNKS = SUB_D / 2, SUB_INSTR = (SUB_CHUNKS + 63) / 64;
    static constexpr int LOADS = NPLANES * P_INSTR + J_INSTR, STORES = NOUT * M * SUB_INSTR;
    // ALDS div ("plane streaming"): the ND u planes of a tile pass through TWO plane buffers one after
    // the other and the second buffer doubles as the output transposition buffer -- with all ND planes
    // and a separate o buffer resident (86 + 29 KB for four waves at Np = 56) there is no room for
    // the fragments.
    static constexpr bool STREAM = ALDS && MODE == 0;
    static_assert(!W8 || (ALDS && (BYCOMP || MODE == 0) && M == 1), "eight-wave blocks: grad by components or div, A in LDS");
    static_assert(MODE != 5 || (W8 && ND == 3), "grad planes by components: the eight-wave p = 5 kernel");
    struct WaveLds {
        double u[STREAM ? (W8 ? 1 : 2) : NPLANES][PLANE_D];   // u[x][e0 .. e0+TEL-1][0..Np-1]  (STREAM: plane buffers; W8: ONE, which is
                                                              // also the output transposition buffer)
        double o[(STREAM || W8) ? 2 : SUB_D];      // output transposition buffer (one 16-element sub-tile)
        double j[NJ > 0 ? NJ * TEL : 2];   // J[x*3+r][e0 + 0..TEL-1]   (MODE 1: J[s][..] or J[..][s]; MODE 2: J[..])
    };
    static constexpr int WAVES = W8 ? 8 : 4;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int ASMALL_D = NC * KSJ * NS * 16; // [k-step][group][g][row] doubles, per block
    static constexpr int OP_D = NC * NP * NP;
    static constexpr int WAVE_BYTES = (int)sizeof(WaveLds) * WAVES;
    static constexpr int ABIG_D = ALDS ? BT * KSJ * NC * 64 : 0;   // [(jq, r)][t][lane] doubles, per block
    static constexpr int LDS_BYTES = (WAVE_BYTES > OP_D * 8 ? WAVE_BYTES : OP_D * 8) + (ASMALL_D + ABIG_D) * 8;
    static constexpr int BLOCKS_PER_CU = ALDS ? 1 : 2;
    static_assert(LOADS + STORES <= 60, "counted vmcnt must fit the 6-bit field");
    static_assert(BLOCKS_PER_CU * LDS_BYTES <= 160 * 1024, "blocks per CU");
    static_assert(!STREAM || M == 1, "plane streaming is written for one sub-tile per wave tile");
};

// kDbg: experiment flags (0 in the product build): 1 skip MFMAs, 2 skip stores, 8 skip loads.
// kPrep (register-resident fragments only): the A fragments come from a prepared operator (`prep` =
// the whole prepared buffer, see fe_common.h) -- the big-tile fragments by coalesced loads straight
// into registers, the 4-row groups copied as they are into their LDS table; D is still needed by the
// remainder code.
// kDyn (plain single-field div of tetrahedra, register fragments): behind two static rounds the tiles come by tickets
// (fe_common.h, dynamic walk); `tail` = the launch's counters (null: static walk), `t_static` = statically walked tiles.
// kIlv (round 5; short launches of the plain div, static walk): the B fragments are built k-quad by k-quad BETWEEN the MFMA groups
// of the same wave instead of all up front -- see the loop.
template <int NP, int M, int kDbg = 0, int MODE = 0, int ND = 3, bool ALDS = false, bool W8 = false, bool kPrep = false,
          bool kDyn = false, bool kIlv = false>
__device__ __forceinline__ void div3d_mfma_body(
    const double* __restrict__ J, const double* __restrict__ D, const void* __restrict__ prep, const FieldPtrs& P,
    int nb, int64_t E, int64_t nTiles, int op_flags, int jes, const unsigned bid, const unsigned nblk,
    const GradFields* __restrict__ Q = nullptr, unsigned* __restrict__ tail = nullptr, int64_t t_static = 0) {
    static_assert(!kPrep || (!ALDS && MODE == 0 && ND == 3), "prepared operators: plain div of tetrahedra");
    static_assert(!kDyn || ((MODE == 0 || MODE == 4) && !ALDS && !W8 && !kPrep) || ((MODE == 4 || MODE == 0) && ND == 3 && W8),
                  "dynamic walk: div (tetrahedra, triangles), grad by components (triangles), or the eight-wave kernels (p = 5)");
    // op_flags: bit 0 = operator stored transposed ([r][j][i]); bit 1 (kDivWalkSplit, plain register-fragment path
    // only) = the walk covers both halves of the element range at once, see `phys` below
    const int opT = op_flags & 1;
    const bool tload = (op_flags & kOpLoadsTemporal) != 0;   // the u planes by plain loads (fe_common.h)
    const bool phase_prio = W8 && (op_flags & kOpPhasePriority) != 0;   // the eight-wave kernels, opt-in (fe_common.h)
    using G = DivGeom<NP, M, MODE, ND, ALDS, W8>;
    using WaveLds = typename G::WaveLds;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds* L = reinterpret_cast<WaveLds*>(smem) + wave;
    const int n = lane & 15, g = lane >> 4;
    // Experiment switches (the `kDbg` template constant): experiments build only (tools/build_experiments.sh); in the product each
    // of them is the constant `false`.
#ifdef FE_EXPERIMENTS
    constexpr bool x_no_mfma = (kDbg & 1) != 0, x_no_stores = (kDbg & 2) != 0, x_reg_prefetch = (kDbg & 4) != 0, x_no_loads = (kDbg & 8) != 0,
                   x_early_request = (kDbg & 16) != 0, x_one_plane = (kDbg & 32) != 0, x_lds_ticket = (kDbg & 64) != 0, x_stamps = (kDbg & 128) != 0;
#else
    static_assert(kDbg == 0, "experiment flags: experiments build only");
    constexpr bool x_no_mfma = false, x_no_stores = false, x_reg_prefetch = false, x_no_loads = false, x_early_request = false, x_one_plane = false,
                   x_lds_ticket = false, x_stamps = false;
#endif
    (void)x_stamps; (void)x_one_plane; (void)x_lds_ticket; (void)x_early_request; (void)x_reg_prefetch; (void)x_no_mfma; (void)x_no_stores; (void)x_no_loads;

    // ---- A fragments from the LDS-staged operator.  16x16x4: lane (g, n) supplies
    //      A[row 16t + n][k = g];  4x4x4_4b group q: lane (g, n) supplies block n/4, row
    //      16 BT + 4q + n%4, k = g -- identical for the four blocks, kept once in LDS.
    constexpr int NC = G::NC;
    double abig[(G::BT > 0 && !ALDS) ? G::BT : 1][ALDS ? 1 : G::KSJ][ALDS ? 1 : NC];
    double* asmall = reinterpret_cast<double*>(smem + (G::LDS_BYTES - (G::ASMALL_D + G::ABIG_D) * 8));
    double* afr = asmall + G::ASMALL_D;   // ALDS: big-tile fragments [(jq * NC + r) * BT + t][lane]
    // The elements behind the last full tile (fe_common.h: remainder_items), entry by entry on the VALU, with the operator read from
    // `Dsrc`: the block's LDS copy while it exists (see fe_grad.h), else global memory.
    auto remainder = [&](const double* Dsrc) {
        remainder_items(nTiles * G::TEL, E, NP, bid, nblk, [&](int64_t e, int i) {
            for (int k = 0; k < nb; ++k) {
                const double* uk = field_in(P, k);
                double* ok = field_out(P, k);
                if (MODE == 0 && ND == 3) div3d_item(J, Dsrc, uk, ok, E, NP, e, i, opT);
                else if (MODE == 0) div_nd_item(J, Dsrc, uk, ok, E, NP, ND, e, i, opT);
                else if (MODE == 4) grad_nd_item(J, Dsrc, uk, ok, E, NP, ND, e, i, opT);
                else if (MODE == 5) {
#pragma unroll
                    for (int x = 0; x < 3; ++x) {
                        double* ox = grad_plane_out(*Q, k, x);
                        if (ox) divcomp3d_item(Q->j[x], Dsrc, uk, ox, E, NP, e, i, opT, 0);
                    }
                }
                else if (MODE == 1 && ND == 3) divcomp3d_item(J, Dsrc, uk, ok, E, NP, e, i, opT, jes);
                else if (MODE == 1) divcomp_nd_item(J, Dsrc, uk, ok, E, NP, ND, e, i, opT, jes);
                else matapply_item(MODE == 2 ? J : nullptr, Dsrc, uk, ok, NP, e, i, opT);
            }
        });
    };
    const unsigned lds_u = lds_addr_uniform(L->u[0]);
    const unsigned lds_j = lds_addr_uniform(L->j);
    const int64_t stride = (int64_t)nblk * G::WAVES;
    // Quarter tiles (kIlv, static walk, one field; fe_common.h: kOpQuarterTail).  A static walk of R full rounds leaves r = nTiles mod
    // (number of waves) tiles for a last, partial round: the SIMDs that get one have a whole tile more than the others -- 2.3 us of
    // matrix work at E = 1e5, 8 % of the launch for 1.7 % of its elements (profiles/r05/tiles_div_100000_interleaved.txt).  When that
    // round is at most an EIGHTH full its r tiles become 4 r quarter tiles of FOUR elements, one for each of the first 4 r waves
    // (so every SIMD of the first r blocks gets one): a quarter tile runs on v_mfma_f64_4x4x4_4b alone -- its four blocks are four
    // 4-row slices of the same 16-row A fragment the 16x16x4 instruction takes, the B operand is the four elements' values
    // replicated over the blocks -- and costs 27 x 48 instead of 3888 matrix cycles.  Measured (profiles/r05/div_quarter_tail_ab.txt):
    // E = 1e5 (r = 106) 25.3 -> 24.2 us, 1.02e5 (r = 231) -4.2 %; level at r = 356, +5 ... 9 % at r = 418 ... 452 (four quarter tiles
    // cost a third more matrix cycles than the tile, and more than half of all SIMDs then carry one or two) -- hence the eighth.
    // Bitwise the full-tile results (the same products in the same order; the 4x4x4 and 16x16x4 instructions round alike).
    int64_t q_e0 = -1;            // first element of this wave's quarter tile, or -1
    int64_t t_full = nTiles;      // tiles walked as full tiles
    if constexpr (kIlv && !kDyn) {
        if (op_flags & kOpQuarterTail) {
            const int64_t r = nTiles % stride;
            if (r > 0 && 8 * r <= stride) {
                t_full = nTiles - r;
                const int64_t w = (int64_t)bid * G::WAVES + wave;
                if (w < 4 * r) q_e0 = t_full * G::TEL + 4 * w;
            }
        }
    }
    const int64_t tEnd = t_full;
    // nb fields share J and D ('xre,rij,xej->ei' x nb: tuning/impls/batched_xre_rij_xej_to_ei.py):
    // the wave walks (tile, field) units, field fastest; J is loaded with the first field of a
    // tile and stays in LDS until the last field's B fragments are built.
    // FE_VARIANT_MFMA_SPLIT: the walk covers the two halves of the element range at the same time (even steps in the
    // first half, odd steps in the second), so that the ONE output stream of div has two write windows -- which a
    // boundary between the two classes of physical memory in the middle of the output array then splits (DESIGN.md
    // section 3d: 0.1914 against 0.1993 ms at E = 1e6; 1.3 % slower than the plain walk anywhere else)
    const int64_t half_tiles = (nTiles + 1) / 2;
    const bool split_walk = (op_flags & kDivWalkSplit) != 0;
    auto phys = [&](int64_t t) -> int64_t { return split_walk ? ((t & 1) ? half_tiles + (t >> 1) : (t >> 1)) : t; };
    auto issue_loads = [&](int64_t tile, int fk, bool with_j) {
        const int64_t e0 = phys(tile) * G::TEL;
        const char* ub = reinterpret_cast<const char*>(field_in(P, fk)) + e0 * (NP * 8);
        if (tload) {   // (fe_common.h, kOpLoadsTemporal: one scalar branch for the whole unit)
#pragma unroll
            for (int x = 0; x < (x_one_plane ? 1 : G::NPLANES); ++x) {   // (x_one_plane, experiments build: timing, wrong results)
                const char* up = ub + (int64_t)x * E * (NP * 8);
#pragma unroll
                for (int c = 0; c < G::P_INSTR; ++c)
                    if ((c + 1) * 64 <= G::P_CHUNKS || c * 64 + lane < G::P_CHUNKS)
                        glds16(up + tile_src_chunk<NP>(c * 64 + lane) * 16, lds_u + x * (G::PLANE_D * 8) + c * 1024);
            }
        } else {
#pragma unroll
            for (int x = 0; x < G::NPLANES; ++x) {
                const char* up = ub + (int64_t)x * E * (NP * 8);
#pragma unroll
                for (int c = 0; c < G::P_INSTR; ++c)
                    if ((c + 1) * 64 <= G::P_CHUNKS || c * 64 + lane < G::P_CHUNKS)
                        glds16_nt(up + tile_src_chunk<NP>(c * 64 + lane) * 16, lds_u + x * (G::PLANE_D * 8) + c * 1024);
            }
        }
        if (!with_j) return;
        const char* jb = reinterpret_cast<const char*>(J) + e0 * 8;
#pragma unroll
        for (int c = 0; c < G::J_INSTR; ++c) {
            const int q = c * 64 + lane;
            const int row = q / G::J_ROW_CHUNKS, col = q - row * G::J_ROW_CHUNKS;
            // rows of E doubles; MODE 1 with J stored [E][ND]: one contiguous span of ND TEL doubles
            const char* src;
            if constexpr (MODE == 5) {   // row = x ND + r of the x-th geometry-factor array
                const int x = row / ND, r = row - ND * x;
                src = reinterpret_cast<const char*>((x == 0 ? Q->j[0] : x == 1 ? Q->j[1] : Q->j[2]) + (int64_t)r * E + e0) +
                      col * 16;
            } else {
                src = (MODE == 1 && jes) ? jb + e0 * (8 * (ND - 1)) + q * 16 : jb + ((int64_t)row * E) * 8 + col * 16;
            }
            if ((c + 1) * 64 <= G::J_CHUNKS || q < G::J_CHUNKS) glds16(src, lds_j + c * 1024);
        }
    };
    int64_t tile = (int64_t)bid * G::WAVES + wave;
    // Waves whose own buffers the operator's staging area does not reach (it covers the first OP_D doubles of the block's LDS: waves 0
    // and 1 at p = 4) request their first tile as soon as the operator's loads are through -- 1 - 2 us before the others can, which
    // is when the first round's 27 MB start to move (profiles/r05/div_prologue_phases.txt: requested behind the prologue by all waves
    // at once, they arrive at 4.8 us in the older and 6.5 us in the younger block of a CU).  Behind the OPERATOR's loads, not in
    // front: vector-memory data return in order per CU, and the other block's operator must not queue behind tile data.
    bool first_requested = false;
    if constexpr (!kPrep) {
        double* dl = reinterpret_cast<double*>(smem);
        stage_operator<G::OP_D, G::THREADS>(D, dl);
        __syncthreads();
        if constexpr (!ALDS && !W8 && !G::STREAM) {   // (behind the barrier: the requests take their waves 0.3 - 1 us to issue, which the block need not wait for)
            if (wave * (int)sizeof(WaveLds) >= G::OP_D * 8 && tile < tEnd && !x_no_loads) {
                issue_loads(tile, 0, true);
                first_requested = true;
            }
        }
        const int istride = opT ? 1 : NP, jstride = opT ? NP : 1;   // opT: D stored [r][j][i]
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq) {
            const int j = 4 * jq + g;
            const double* col = dl + (j < NP ? j : 0) * jstride + n * istride;
#pragma unroll
            for (int r = 0; r < NC; ++r)
#pragma unroll
                for (int t = 0; t < G::BT; ++t) {
                    const double v = col[r * (NP * NP) + 16 * t * istride];
                    if (ALDS) {
                        // every wave builds the same fragments: wave w stores those of the k-quads jq = w mod 4
                        if (jq % G::WAVES == wave) afr[((jq * NC + r) * G::BT + t) * 64 + lane] = (j < NP) ? v : 0.0;
                    } else {
                        abig[t][jq][r] = (j < NP) ? v : 0.0;
                    }
                }
        }
        for (int idx = threadIdx.x; idx < G::ASMALL_D; idx += G::THREADS) {
            const int row4 = idx & 3, gg = (idx >> 2) & 3, q = (idx >> 4) % G::NS, ks = (idx >> 4) / G::NS;
            const int i = 16 * G::BT + 4 * q + row4, j = 4 * (ks / NC) + gg, r = ks % NC;
            asmall[idx] = (j < NP && i < NP) ? dl[r * (NP * NP) + i * istride + j * jstride] : 0.0;
        }
        // (while the block's copy of the operator is still there; not in the eight-wave p = 5 kernels, whose prologue has no registers
        //  to spare: the compiler took the ticket registers for it -- tests/test_ticket_registers.py)
        if constexpr (!ALDS && !W8) remainder(dl);
        __syncthreads();   // the staging area is reused as the waves' private buffers from here on
        if constexpr (ALDS || W8) remainder(D);
    } else {
        remainder(D);
    }
    const double* as_lane = asmall + g * 4 + (n & 3);
    const double* af_lane = afr + lane;
    auto a_big = [&](int t, int jq, int r) -> double {
        if constexpr (ALDS) return af_lane[((jq * NC + r) * G::BT + t) * 64];
        else return abig[t][jq][r];
    };


    if constexpr (G::STREAM) {
        // ---- plane streaming (see DivGeom): per (tile, field) unit
        //   L(p0), L(J) | L(p1) -> B += plane 0 | L(p2) -> B += plane 1 -> B += plane 2 | L(p0', J') | MFMAs | stores
        const unsigned lds_a = lds_addr_uniform(L->u[0]), lds_b = lds_addr_uniform(L->u[W8 ? 0 : 1]);
        auto issue_plane = [&](int64_t t, int fk, int x, unsigned lds) {
            const char* up = reinterpret_cast<const char*>(field_in(P, fk)) + ((int64_t)x * E + t * G::TEL) * (NP * 8);
            if (tload) {   // (fe_common.h, kOpLoadsTemporal: one scalar branch for the whole plane)
#pragma unroll
                for (int c = 0; c < G::P_INSTR; ++c)
                    if ((c + 1) * 64 <= G::P_CHUNKS || c * 64 + lane < G::P_CHUNKS)
                        glds16(up + tile_src_chunk<NP>(c * 64 + lane) * 16, lds + c * 1024);
            } else {
#pragma unroll
                for (int c = 0; c < G::P_INSTR; ++c)
                    if ((c + 1) * 64 <= G::P_CHUNKS || c * 64 + lane < G::P_CHUNKS)
                        glds16_nt(up + tile_src_chunk<NP>(c * 64 + lane) * 16, lds + c * 1024);
            }
        };
        auto issue_j = [&](int64_t t) {
            const char* jb = reinterpret_cast<const char*>(J) + t * G::TEL * 8;
#pragma unroll
            for (int c = 0; c < G::J_INSTR; ++c) {
                const int q = c * 64 + lane;
                const int row = q / G::J_ROW_CHUNKS, col = q - row * G::J_ROW_CHUNKS;
                if ((c + 1) * 64 <= G::J_CHUNKS || q < G::J_CHUNKS)
                    glds16(jb + ((int64_t)row * E) * 8 + col * 16, lds_j + c * 1024);
            }
        };
        static_assert(ND == 3, "three planes");
        bool first = true;
        int fk = 0;
        if constexpr (W8) {
            // ---- eight waves per block = two per SIMD (round 4): a wave's own VALU work and memory stalls never hide under its
            //      own MFMAs (profiles/r04/mfma_valu_overlap.txt) -- a second wave on the SIMD is what covers them.  Beside 75 KB
            //      of fragments there is room for ONE plane buffer per wave, so a unit is a serial chain
            //        L(p0), L(J) | B = J p0 | L(p1) | B += J p1 | L(p2) | B += J p2 | 210 MFMAs | out through the buffer | L(p0', J')
            //      whose three load latencies the partner wave's MFMA phase covers.
            //      Measured (profiles/r04/p5_div_eight_waves.txt): 0.402 against 0.411-0.417 ms for four waves.  Fetching the
            //      next unit's three planes into REGISTERS before the MFMA phase (84 more VGPRs, 252 in all) took the load
            //      latencies off the chain and changed nothing (0.403 ms): what binds is the matrix pipe plus the f64 VALU work
            //      of both waves (the B fragments: 126 f64 operations behind 126 LDS reads per unit), not the loads.
            (void)lds_b;
            if (tile < tEnd) { issue_plane(tile, 0, 0, lds_a); issue_j(tile); }
            // dynamic walk (fe_common.h; one field): the ticket for the next tile is asked for at the top of a tile -- behind the
            // wait for its first plane -- and read behind the wait for its second plane (every wait here is vmcnt(0): the ticket,
            // older than the second plane's loads, is back with them); the next tile's loads go out at the end of the tile
            const bool dyn8 = kDyn && tail != nullptr && t_static < nTiles && nb == 1;   // grid-uniform
            const int pool8 = (bid >> 3) & (kTailPools - 1);
            unsigned* const counter8 = tail_pool_counters(tail, pool8);
            unsigned* const done8 = tail_pool_reports(counter8);
            bool reported8 = false;
            while (tile < tEnd) {
                double* const out = field_out(P, fk);
                const bool next_new_tile = (fk + 1 == nb);
                int64_t nt = next_new_tile ? tile + stride : tile;
                const int nk = next_new_tile ? 0 : fk + 1;
                wait_vmcnt<0>();                                  // p0 and J landed (and the previous unit's stores left)
                bool asked8 = false;
                if constexpr (kDyn) {
                    if (dyn8 && !(tile < t_static && tile + stride < t_static)) {   // the next tile is not static
                        tail_request<0>(counter8);
                        asked8 = true;
                    }
                }
                if (phase_prio) __builtin_amdgcn_s_setprio(3);   // f64 VALU phase: the B fragments (fe_common.h, kOpPhasePriority)
                double jac[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) jac[k] = L->j[k * G::TEL + n];
                double bfrag[G::KSJ][3];
                auto add_plane = [&](int x) {
#pragma unroll
                    for (int jq = 0; jq < G::KSJ; ++jq) {
                        const double v = L->u[0][tile_index<NP>(n, 4 * jq + g)];
#pragma unroll
                        for (int r = 0; r < 3; ++r) bfrag[jq][r] = x == 0 ? jac[r] * v : __builtin_fma(jac[x * 3 + r], v, bfrag[jq][r]);
                    }
#pragma unroll
                    for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                        for (int r = 0; r < 3; ++r) asm volatile("" : "+v"(bfrag[jq][r]));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the plane is in registers: its buffer may be refilled
                };
                add_plane(0);
                issue_plane(tile, fk, 1, lds_a);
                wait_vmcnt<0>();
                if constexpr (kDyn) {
                    if (asked8) {   // (nothing may be outstanding: the wait above was for everything)
                        const int64_t x = tail_ticket_tile(tail_wait<0, 0>(), t_static, pool8, tEnd);
                        nt = x >= 0 ? x : tEnd;
                        if (x < 0) {   // this wave's pool is empty: stop asking, report
                            tail_request<1>(done8);
                            reported8 = true;
                        }
                    }
                }
                add_plane(1);
                issue_plane(tile, fk, 2, lds_a);
                wait_vmcnt<0>();
                add_plane(2);
                if (phase_prio) __builtin_amdgcn_s_setprio(0);   // matrix phase

                v4d acc[G::BT];
                double accs[G::NS > 0 ? G::NS : 1];
#pragma unroll
                for (int t = 0; t < G::BT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < G::NS; ++q) accs[q] = 0.0;
#pragma unroll
                for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
#pragma unroll
                        for (int t = 0; t < G::BT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_big(t, jq, r), bfrag[jq][r], acc[t], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < G::NS; ++q)
                            accs[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(as_lane[((jq * 3 + r) * G::NS + q) * 16], bfrag[jq][r],
                                                                         accs[q], 0, 0, 0);
                    }
                double* ob = L->u[0];                             // the plane buffer as the output transposition buffer
#pragma unroll
                for (int t = 0; t < G::BT; ++t)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) ob[tile_index<NP>(n, 16 * t + g + 4 * qq)] = acc[t][qq];
#pragma unroll
                for (int q = 0; q < G::NS; ++q) {
                    const int i = 16 * G::BT + 4 * q + g;
                    if (16 * G::BT + 4 * q + 3 < NP || i < NP) ob[tile_index<NP>(n, i)] = accs[q];
                }
                wave_lds_fence();
                double* op = out + tile * G::TEL * NP;
                v2d held[G::SUB_INSTR];
#pragma unroll
                for (int c = 0; c < G::SUB_INSTR; ++c) {
                    const int qc = c * 64 + lane;
                    held[c] = ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS) ? *reinterpret_cast<const v2d*>(ob + 2 * tile_dst_chunk<NP>(qc))
                                                                                  : v2d{0.0, 0.0};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // o is in registers before the buffer is refilled
                if (nt < tEnd) { issue_plane(nt, nk, 0, lds_a); issue_j(nt); }
#pragma unroll
                for (int c = 0; c < G::SUB_INSTR; ++c) {
                    const int qc = c * 64 + lane;
                    if ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS)
                        __builtin_nontemporal_store(held[c], reinterpret_cast<v2d*>(op + 2 * qc));
                }
                wave_lds_fence();
                fk = nk;
                tile = nt;
            }
            if constexpr (kDyn) {
                if (reported8) {   // the last wave of a pool to report leaves the pool's counters zeroed
                    const unsigned pool_blocks = (nblk / (8 * kTailPools)) * 8 + (unsigned)max(0, min(8, (int)(nblk % (8 * kTailPools)) - 8 * pool8));
                    const unsigned before = tail_wait<0, 1>();
                    if (before + 1 == pool_blocks * G::WAVES && lane == 0) {
                        __hip_atomic_store(counter8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(done8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            return;
        }
        // (tickets -- fe_common.h, dynamic walk -- were measured in the four-wave loop below too: 46.9 against 48.1 TFLOP/s, profiles/r03/dynamic_walk_p5.txt)
        if (tile < tEnd) { issue_plane(tile, 0, 0, lds_a); issue_j(tile); }
        while (tile < tEnd) {
            double* const out = field_out(P, fk);
            const bool next_new_tile = (fk + 1 == nb);
            const int64_t nt = next_new_tile ? tile + stride : tile;
            const int nk = next_new_tile ? 0 : fk + 1;
            issue_plane(tile, fk, 1, lds_b);                 // the previous unit's o (= buffer b) has been read out
            if (first) wait_vmcnt<G::P_INSTR>();             // p0 and J landed; younger: S(previous), L(p1)
            else wait_vmcnt<G::STORES + G::P_INSTR>();
            first = false;
            double jac[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) jac[k] = L->j[k * G::TEL + n];
            double bfrag[G::KSJ][3];
            auto add_plane = [&](int x, const double* up) {
#pragma unroll
                for (int jq = 0; jq < G::KSJ; ++jq) {
                    const int j = 4 * jq + g;
                    const double v = j < NP ? up[tile_index<NP>(n, j < NP ? j : 0)] : 0.0;
#pragma unroll
                    for (int r = 0; r < 3; ++r) bfrag[jq][r] = x == 0 ? jac[r] * v : __builtin_fma(jac[x * 3 + r], v, bfrag[jq][r]);
                }
#pragma unroll
                for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                    for (int r = 0; r < 3; ++r) asm volatile("" : "+v"(bfrag[jq][r]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the plane is in registers: its buffer may be refilled
            };
            add_plane(0, L->u[0]);
            issue_plane(tile, fk, 2, lds_a);
            wait_vmcnt<G::P_INSTR>();                         // p1 landed; younger: L(p2)
            add_plane(1, L->u[1]);
            wait_vmcnt<0>();
            add_plane(2, L->u[0]);
            if (nt < tEnd) { issue_plane(nt, nk, 0, lds_a); issue_j(nt); }

            v4d acc[G::BT];
            double accs[G::NS > 0 ? G::NS : 1];
#pragma unroll
            for (int t = 0; t < G::BT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < G::NS; ++q) accs[q] = 0.0;
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int t = 0; t < G::BT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_big(t, jq, r), bfrag[jq][r], acc[t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < G::NS; ++q)
                        accs[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(as_lane[((jq * 3 + r) * G::NS + q) * 16], bfrag[jq][r],
                                                                     accs[q], 0, 0, 0);
                }
            double* ob = L->u[1];                             // buffer b as the output transposition buffer
#pragma unroll
            for (int t = 0; t < G::BT; ++t)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) ob[tile_index<NP>(n, 16 * t + g + 4 * qq)] = acc[t][qq];
#pragma unroll
            for (int q = 0; q < G::NS; ++q) {
                const int i = 16 * G::BT + 4 * q + g;
                if (16 * G::BT + 4 * q + 3 < NP || i < NP) ob[tile_index<NP>(n, i)] = accs[q];
            }
            wave_lds_fence();
            double* op = out + tile * G::TEL * NP;
            v2d held[G::SUB_INSTR];
#pragma unroll
            for (int c = 0; c < G::SUB_INSTR; ++c) {
                const int qc = c * 64 + lane;
                held[c] = ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS) ? *reinterpret_cast<const v2d*>(ob + 2 * tile_dst_chunk<NP>(qc))
                                                                              : v2d{0.0, 0.0};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // o is in registers before buffer b is refilled
#pragma unroll
            for (int c = 0; c < G::SUB_INSTR; ++c) {
                const int qc = c * 64 + lane;
                if ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS)
                    __builtin_nontemporal_store(held[c], reinterpret_cast<v2d*>(op + 2 * qc));
            }
            wave_lds_fence();
            fk = nk;
            tile = nt;
        }
        return;
    }
    if constexpr (W8) {
        // ---- eight waves per block (see DivGeom): wait u, J -> B, J to registers -> MFMAs -> three
        //      planes through the (former u) buffer -> request the next unit
        //      Tried (experiment build, kDbg 64 / 16; in-process A/B in tools/w8_phases.py): the wave in the even
        //      hardware slot at raised priority with the block's tiles handed out by an LDS ticket counter (so that
        //      the partners of a SIMD cannot fall into step), and the next unit requested ahead of the last plane's
        //      stores -- 0.390-0.398 ms against 0.375-0.386 for this loop: neither helps.  The per-wave chain
        //      load -> MFMAs -> three planes through the one buffer is what binds (p = 5).
        constexpr bool kTicket = x_lds_ticket, kEarly = x_early_request;
        // kRegPre (round 3): the NEXT unit's u tile and J rows are fetched into REGISTERS right after this unit's B values
        // have left the buffer -- 14 + 4 doubles per lane, in flight during the 210 MFMAs -- and written into the (one)
        // tile buffer after this unit's planes have gone out through it.  The LDS-DMA flavour can only ask for the next
        // tile once that buffer is free, i.e. after the stores, and then waits at the top of the loop for the loads AND,
        // through vmcnt(0), for those stores.
        constexpr bool kRegPre = x_reg_prefetch;
        v2d nxt_u[G::P_INSTR], nxt_j[G::J_INSTR > 0 ? G::J_INSTR : 1];
        auto load_regs = [&](int64_t t, int f_, bool with_j) {
            const int64_t e0 = t * G::TEL;
            const char* ub = reinterpret_cast<const char*>(field_in(P, f_)) + e0 * (NP * 8);
#pragma unroll
            for (int c = 0; c < G::P_INSTR; ++c)
                if ((c + 1) * 64 <= G::P_CHUNKS || c * 64 + lane < G::P_CHUNKS)
                    nxt_u[c] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(ub + tile_src_chunk<NP>(c * 64 + lane) * 16));
            if (!with_j) return;
#pragma unroll
            for (int c = 0; c < G::J_INSTR; ++c) {
                const int q = c * 64 + lane;
                const int row = q / G::J_ROW_CHUNKS, col = q - row * G::J_ROW_CHUNKS;
                const char* src;
                if constexpr (MODE == 5) {
                    const int x = row / ND, r = row - ND * x;
                    src = reinterpret_cast<const char*>((x == 0 ? Q->j[0] : x == 1 ? Q->j[1] : Q->j[2]) + (int64_t)r * E + e0) + col * 16;
                } else {
                    src = reinterpret_cast<const char*>(J) + e0 * 8 + ((int64_t)row * E) * 8 + col * 16;
                }
                if ((c + 1) * 64 <= G::J_CHUNKS || q < G::J_CHUNKS) nxt_j[c] = *reinterpret_cast<const v2d*>(src);
            }
        };
        auto regs_to_lds = [&](bool with_j) {
            char* ul = reinterpret_cast<char*>(L->u[0]);
#pragma unroll
            for (int c = 0; c < G::P_INSTR; ++c)
                if ((c + 1) * 64 <= G::P_CHUNKS || c * 64 + lane < G::P_CHUNKS)
                    *reinterpret_cast<v2d*>(ul + (c * 64 + lane) * 16) = nxt_u[c];
            if (with_j) {
                char* jl = reinterpret_cast<char*>(L->j);
#pragma unroll
                for (int c = 0; c < G::J_INSTR; ++c) {
                    const int q = c * 64 + lane;
                    if ((c + 1) * 64 <= G::J_CHUNKS || q < G::J_CHUNKS) *reinterpret_cast<v2d*>(jl + q * 16) = nxt_j[c];
                }
            }
            wave_lds_fence();
        };
        unsigned* const ticket = reinterpret_cast<unsigned*>(reinterpret_cast<WaveLds*>(smem)->o);   // (o is unused here)
        if constexpr (kTicket) {
            if (threadIdx.x == 0) *ticket = G::WAVES;           // tickets 0 .. W-1 are the waves' first tiles
            __syncthreads();
            if ((__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1) == 0) __builtin_amdgcn_s_setprio(2);   // HW_ID.WAVE_ID[0]
        }
        auto next_ticket_tile = [&]() -> int64_t {   // ticket t of block b is tile b W + t mod W + (t div W) stride
            unsigned t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            t = __builtin_amdgcn_readfirstlane(t);
            return (int64_t)bid * G::WAVES + (t % G::WAVES) + (int64_t)(t / G::WAVES) * stride;
        };
        int fk = 0;
#ifdef FE_EXPERIMENTS
        unsigned long long dw = 0, dm = 0, de = 0, dn = 0;
        const unsigned long long dstart = (kDbg & 32) ? __builtin_amdgcn_s_memrealtime() : 0;
#endif
        if (tile < tEnd && !x_no_loads) {
            if constexpr (kRegPre) { load_regs(tile, 0, true); regs_to_lds(true); }
            else issue_loads(tile, 0, true);
        }
        // dynamic walk (fe_common.h; one field): the ticket for the next tile is asked for at the top of a tile and read behind
        // its last plane, where the next tile's loads are issued
        const bool dyn8 = kDyn && tail != nullptr && t_static < nTiles && nb == 1;   // grid-uniform
        const int pool8 = (bid >> 3) & (kTailPools - 1);
        unsigned* const counter8 = tail_pool_counters(tail, pool8);
        unsigned* const done8 = tail_pool_reports(counter8);
        bool reported8 = false;
        while (tile < tEnd) {
            double* const out = field_out(P, fk);
            const bool next_new_tile = (fk + 1 == nb);
            int64_t nt = next_new_tile ? (kTicket ? next_ticket_tile() : tile + stride) : tile;
            const int nk = next_new_tile ? 0 : fk + 1;
#ifdef FE_EXPERIMENTS
            const unsigned long long c0 = (kDbg & 32) ? __builtin_amdgcn_s_memtime() : 0;
#endif
            if constexpr (!kRegPre) wait_vmcnt<0>();
            if (phase_prio) __builtin_amdgcn_s_setprio(0);   // matrix phase (fe_common.h, kOpPhasePriority)
            bool asked8 = false;
            if constexpr (kDyn) {
                if (dyn8 && !(tile < t_static && tile + stride < t_static)) {   // the next tile is not static
                    tail_request<0>(counter8);
                    asked8 = true;
                }
            }
#ifdef FE_EXPERIMENTS
            const unsigned long long c1 = (kDbg & 32) ? __builtin_amdgcn_s_memtime() : 0;
#endif
            double jk[ND * ND], bf[G::KSJ];
#pragma unroll
            for (int k = 0; k < ND * ND; ++k) jk[k] = L->j[k * G::TEL + n];
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq) {
                const int j = 4 * jq + g;
                bf[jq] = j < NP ? L->u[0][tile_index<NP>(n, j < NP ? j : 0)] : 0.0;
            }
            if constexpr (kRegPre) {
                // the B values and J are in registers: ask for the next unit now (the loads land under the MFMAs)
#pragma unroll
                for (int jq = 0; jq < G::KSJ; ++jq) asm volatile("" : "+v"(bf[jq]));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (nt < tEnd && !x_no_loads) load_regs(nt, nk, next_new_tile);
            }
            v4d acc[NC][G::BT > 0 ? G::BT : 1];
            double accs[NC][G::NS > 0 ? G::NS : 1];
#pragma unroll
            for (int r = 0; r < NC; ++r) {
#pragma unroll
                for (int t = 0; t < G::BT; ++t) acc[r][t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < G::NS; ++q) accs[r][q] = 0.0;
            }
            if (x_no_mfma) {     // experiment: no MFMAs (the B values stay live)
                double sum = 0.0;
#pragma unroll
                for (int jq = 0; jq < G::KSJ; ++jq) sum += bf[jq];
#pragma unroll
                for (int r = 0; r < NC; ++r) {
#pragma unroll
                    for (int t = 0; t < G::BT; ++t) acc[r][t] = v4d{sum, sum, sum, a_big(t, 0, r)};
#pragma unroll
                    for (int q = 0; q < G::NS; ++q) accs[r][q] = sum;
                }
            } else {
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                for (int r = 0; r < NC; ++r) {
#pragma unroll
                    for (int t = 0; t < G::BT; ++t)
                        acc[r][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_big(t, jq, r), bf[jq], acc[r][t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < G::NS; ++q)
                        accs[r][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(as_lane[((jq * NC + r) * G::NS + q) * 16], bf[jq],
                                                                        accs[r][q], 0, 0, 0);
                }
            }
            double* ob = L->u[0];   // every B value is in a register by now (the MFMAs consumed them)
            bool requested = false;
            if (phase_prio) __builtin_amdgcn_s_setprio(3);   // f64 VALU phase: the Jacobian contraction of the planes
#ifdef FE_EXPERIMENTS
            const unsigned long long c2 = (kDbg & 32) ? __builtin_amdgcn_s_memtime() : 0;
#endif
#pragma unroll
            for (int x = 0; x < ND; ++x) {
                double* plane = nullptr;
                if constexpr (MODE == 5) {
                    plane = grad_plane_out(*Q, fk, x);
                    if (plane == nullptr) continue;   // plane not asked for (wave-uniform)
                }
#pragma unroll
                for (int t = 0; t < G::BT; ++t)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        double v = jk[x * ND] * acc[0][t][qq];
#pragma unroll
                        for (int r = 1; r < ND; ++r) v = __builtin_fma(jk[x * ND + r], acc[r][t][qq], v);
                        ob[tile_index<NP>(n, 16 * t + g + 4 * qq)] = v;
                    }
#pragma unroll
                for (int q = 0; q < G::NS; ++q) {
                    const int i = 16 * G::BT + 4 * q + g;
                    double v = jk[x * ND] * accs[0][q];
#pragma unroll
                    for (int r = 1; r < ND; ++r) v = __builtin_fma(jk[x * ND + r], accs[r][q], v);
                    if (16 * G::BT + 4 * q + 3 < NP || i < NP) ob[tile_index<NP>(n, i)] = v;
                }
                wave_lds_fence();
                double* op = MODE == 5 ? plane + tile * G::TEL * NP : out + ((int64_t)x * E + tile * G::TEL) * NP;
                v2d held[G::SUB_INSTR];
#pragma unroll
                for (int c = 0; c < G::SUB_INSTR; ++c) {
                    const int qc = c * 64 + lane;
                    held[c] = ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS)
                                  ? *reinterpret_cast<const v2d*>(ob + 2 * tile_dst_chunk<NP>(qc)) : v2d{0.0, 0.0};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // plane x has left the buffer
                if (!kRegPre && kEarly && x == ND - 1 && nt < tEnd && !x_no_loads) { issue_loads(nt, nk, next_new_tile); requested = true; }
#pragma unroll
                for (int c = 0; c < G::SUB_INSTR; ++c) {
                    const int qc = c * 64 + lane;
                    if ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS) {
                        if (x_no_stores) { if (held[c][0] == 1.2345e-300) op[2 * qc] = held[c][1]; }   // experiment: no stores
                        else __builtin_nontemporal_store(held[c], reinterpret_cast<v2d*>(op + 2 * qc));
                    }
                }
                wave_lds_fence();
            }
            if constexpr (kDyn) {
                if (asked8) {   // younger than the request: this tile's stores
                    const int64_t x = tail_ticket_tile(tail_wait<G::STORES, 0>(), t_static, pool8, tEnd);
                    nt = x >= 0 ? x : tEnd;
                    if (x < 0) {   // this wave's pool is empty: stop asking, report
                        tail_request<1>(done8);
                        reported8 = true;
                    }
                }
            }
            if constexpr (kRegPre) {
                if (nt < tEnd && !x_no_loads) regs_to_lds(next_new_tile);   // the planes have left the buffer: the next unit moves in
            } else if (!requested && nt < tEnd && !x_no_loads) {
                issue_loads(nt, nk, next_new_tile);   // (MODE 5: last plane not asked for)
            }
#ifdef FE_EXPERIMENTS
            const unsigned long long c3 = (kDbg & 32) ? __builtin_amdgcn_s_memtime() : 0;
            dw += c1 - c0; dm += c2 - c1; de += c3 - c2; ++dn;
#endif
            fk = nk;
            tile = nt;
        }
        if constexpr (kDyn) {
            if (reported8) {   // the last wave of a pool to report leaves the pool's counters zeroed
                const unsigned pool_blocks = (nblk / (8 * kTailPools)) * 8 + (unsigned)max(0, min(8, (int)(nblk % (8 * kTailPools)) - 8 * pool8));
                const unsigned before = tail_wait<0, 1>();
                if (before + 1 == pool_blocks * G::WAVES && lane == 0) {
                    __hip_atomic_store(counter8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(done8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
#ifdef FE_EXPERIMENTS
        if ((kDbg & 32) && lane == 0 && bid * G::WAVES + wave < 4096) {
            unsigned long long* d = fe_dbg_w8[bid * G::WAVES + wave];
            d[0] = dw; d[1] = dm; d[2] = de; d[3] = dn;
            d[4] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            d[5] = dstart; d[6] = __builtin_amdgcn_s_memrealtime(); d[7] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        }
#endif
        return;
    }
    bool first = true;
    if constexpr (kPrep) {
        // fragments from the prepared operator; the first tile streams in meanwhile (no staging area to wait for)
        const char* pb = reinterpret_cast<const char*>(prep);
        constexpr int BT1 = G::BT > 0 ? G::BT : 1;   // (no 16-row tiles below Np = 16: no fragments either)
        load_prepared_fragments<G::BT * G::KSJ * NC>(pb + kPrepDivOff, lane, [&](int f, double v) {
            abig[f % BT1][(f / BT1) / NC][(f / BT1) % NC] = v;      // f = (jq NC + r) BT + t
        });
        const double* ps = reinterpret_cast<const double*>(pb + kPrepDivSmallOff);
        constexpr int kPer = (G::ASMALL_D + G::THREADS - 1) / G::THREADS;
        double held[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int idx = threadIdx.x + k * G::THREADS;
            held[k] = idx < G::ASMALL_D ? ps[idx] : 0.0;
        }
        if (tile < tEnd && !x_no_loads) issue_loads(tile, 0, true);
        prepared_fragments_landed();
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int idx = threadIdx.x + k * G::THREADS;
            if (idx < G::ASMALL_D) asmall[idx] = held[k];
        }
        __syncthreads();   // the table of the 4-row groups is complete
    } else {
        if (!first_requested && tile < tEnd && !x_no_loads) issue_loads(tile, 0, true);
    }
    // ---- the quarter tile of this wave (see above): loads, and the unit itself
    bool q_issued = false;
    auto issue_quarter_loads = [&]() {
        q_issued = true;
        const char* ub = reinterpret_cast<const char*>(field_in(P, 0)) + q_e0 * (NP * 8);
        constexpr int QC = 4 * NP / 2;                     // 16-byte chunks of four rows (70)
#pragma unroll
        for (int x = 0; x < G::NPLANES; ++x) {
            const char* up = ub + (int64_t)x * E * (NP * 8);
#pragma unroll
            for (int c = 0; c < (QC + 63) / 64; ++c)
                if (c * 64 + lane < QC) {
                    if (tload) glds16(up + (c * 64 + lane) * 16, lds_u + x * (G::PLANE_D * 8) + c * 1024);
                    else glds16_nt(up + (c * 64 + lane) * 16, lds_u + x * (G::PLANE_D * 8) + c * 1024);
                }
        }
        // J: nine rows of four doubles, compact: j[k * 4 + element]
        if (lane < 18) glds16(reinterpret_cast<const char*>(J) + ((int64_t)(lane >> 1) * E + q_e0) * 8 + (lane & 1) * 16, lds_j);
    };
    constexpr int kQuarterLoads = G::NPLANES * ((4 * NP / 2 + 63) / 64) + 1, kQuarterStores = (4 * NP / 2 + 63) / 64;
    (void)kQuarterLoads;
    if constexpr (kIlv && !kDyn) {
        if (q_e0 >= 0 && !(tile < tEnd)) issue_quarter_loads();   // a wave without a full tile: its quarter tile comes first
    }
    const bool younger_half = bid >= (nblk + 1) / 2;
    int iteration = 0, fk = 0;
    int dbg_it = 0;   // (experiments build: units done by this wave, for the per-tile stamps of kDbg & 128)
#ifdef FE_EXPERIMENTS
    unsigned long long dbg_entry = 0;
    if (kDbg & 128) {
        dbg_entry = __builtin_amdgcn_s_memrealtime();
        if (lane < 16) reinterpret_cast<unsigned long long*>(smem + G::LDS_BYTES)[wave * 16 + lane] = 0;
    }
#endif
    // dynamic walk (plain walk): vector-memory ops of a unit in issue order [ticket or report] L(next unit) S(this unit), so the
    // counted wait at the top of a unit is that of the static walk.  One field: the ticket asked for in front of L(next) is for
    // the tile after next and is read one iteration later at the same place.  b fields (units (tile, field), field fastest):
    // the ticket for the next tile is asked for with the tile's first field and read with its last, b - 1 units later.
    const bool dyn = kDyn && tail != nullptr && t_static < nTiles && !split_walk;   // grid-uniform
    const int pool = (bid >> 3) & (kTailPools - 1);
    unsigned* const counter = tail_pool_counters(tail, pool);
    unsigned* const done = tail_pool_reports(counter);
    bool pending = false, reported = false;
    auto static_next = [&](int64_t t) -> int64_t { return (t < t_static && t + stride < t_static) ? t + stride : -1; };
    // the next unit's tile under the dynamic walk (called once per unit, in front of the next unit's loads and behind the wait for
    // this unit's: vector-memory ops in issue order [ticket or report] L(next unit) S(this unit))
    auto resolve_next_tile = [&](int64_t tile_, int fk_, bool next_new_tile_, int64_t& nt_) {
        if constexpr (kDyn) {
            if (dyn && nb == 1) {
                if (pending) {   // asked for one iteration ago, in front of this tile's loads: it is here
                    const unsigned t = tail_wait<G::STORES, 0>();
                    nt_ = tail_ticket_tile(t, t_static, pool, tEnd);
                    pending = false;
                    if (nt_ < 0) {   // this wave's pool is empty: stop asking, report
                        tail_request<1>(done);
                        reported = true;
                    }
                } else {
                    nt_ = static_next(tile_);
                }
                if (nt_ >= 0 && static_next(nt_) < 0) {   // the tile after next is not static
                    tail_request<0>(counter);
                    pending = true;
                }
                if (nt_ < 0) nt_ = tEnd;
            } else if (dyn) {
                if (fk_ == 0 && static_next(tile_) < 0) {   // first field of a tile whose successor is not static
                    tail_request<0>(counter);
                    pending = true;
                }
                if (next_new_tile_) {
                    if (pending) {   // asked for b - 1 units ago, in front of the second field's loads, which this wave has waited for
                        nt_ = tail_ticket_tile(tail_wait<G::STORES, 0>(), t_static, pool, tEnd);
                        pending = false;
                        if (nt_ < 0) {
                            tail_request<1>(done);
                            reported = true;
                        }
                    } else {
                        nt_ = static_next(tile_);
                    }
                    if (nt_ < 0) nt_ = tEnd;
                }
            }
        }
    };
    while (tile < tEnd) {
        balance_priority(younger_half, iteration++);
        const int64_t e0 = phys(tile) * G::TEL;
        double* const out = field_out(P, fk);
        const bool next_new_tile = (fk + 1 == nb);
        int64_t nt = next_new_tile ? tile + stride : tile;
        const int nk = next_new_tile ? 0 : fk + 1;
        // issue order: ... L(unit) [MFMAs(unit-1)] S(unit-1) | wait L(unit): the previous unit's stores are younger
        if (first || x_no_stores || x_no_loads) wait_vmcnt<0>();
        else wait_vmcnt<G::STORES>();
        first = false;
        FE_TILE_STAMP(x_stamps, smem + G::LDS_BYTES, wave, lane, dbg_it, 0);   // this unit's loads have landed
        if constexpr (kIlv) {
            // ---- interleaved form.  The f64 MFMAs run on the vector f64 datapath: beside its partner's matrix phase a wave's
            //      81 f64 VALU instructions of the B build only get the slots between two 64-cycle MFMAs, and the build takes
            //      2.3-4.1 us instead of 0.4-0.6 (profiles/r05/tiles_div_100000_before.txt) -- the wave is late for its own matrix
            //      phase and the pipe idles.  Here the 9 VALU instructions of k-quad jq + 1 follow the 9 MFMAs of k-quad jq in
            //      the wave's own stream (a wave's own VALU instruction costs its 4-5 cycles behind its own MFMA, no more), so a
            //      tile is ONE phase of MFMAs and VALU work and the two waves of a SIMD simply share the pipe.  The u planes are
            //      needed until the last k-quad is built: the next tile's loads go out behind the MFMAs of the last but one.
            static_assert(MODE == 0 && ND == 3 && M == 1 && !ALDS && !W8 && !kPrep && !(x_no_mfma || x_no_stores || x_reg_prefetch || x_no_loads || x_early_request || x_one_plane || x_lds_ticket), "interleaved B build: plain div of tetrahedra");
            double jac[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) jac[k] = L->j[k * G::TEL + n];
            auto read_quad = [&](int jq, double (&ux)[3]) {
                const int j = 4 * jq + g;
                const int jc = j < NP ? j : 0;
#pragma unroll
                for (int x = 0; x < 3; ++x) ux[x] = j < NP ? L->u[x][tile_index<NP>(n, jc)] : 0.0;
            };
            auto build_quad = [&](const double (&ux)[3], double (&bq)[3]) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double v = jac[r] * ux[0];
#pragma unroll
                    for (int x = 1; x < 3; ++x) v = __builtin_fma(jac[x * 3 + r], ux[x], v);   // the order of the plain form: same bits
                    bq[r] = v;
                }
            };
            v4d acc[G::BT > 0 ? G::BT : 1];
            double accs[G::NS > 0 ? G::NS : 1];
#pragma unroll
            for (int t = 0; t < G::BT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < G::NS; ++q) accs[q] = 0.0;
            double ux[2][3], bq[2][3];
            read_quad(0, ux[0]);
            read_quad(1, ux[1]);
            build_quad(ux[0], bq[0]);
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq) {
                const int cur = jq & 1, nxt = cur ^ 1;
                if (jq + 2 < G::KSJ) read_quad(jq + 2, ux[cur]);     // (ux[cur] was consumed by build_quad of this k-quad)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int t = 0; t < G::BT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_big(t, jq, r), bq[cur][r], acc[t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < G::NS; ++q)
                        accs[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(as_lane[((jq * 3 + r) * G::NS + q) * 16], bq[cur][r], accs[q], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (jq + 1 < G::KSJ) build_quad(ux[nxt], bq[nxt]);
                if (jq + 2 == G::KSJ) {   // every u value is in a register: hand the planes back to the DMA engine
#pragma unroll
                    for (int r = 0; r < 3; ++r) asm volatile("" : "+v"(bq[nxt][r]));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    resolve_next_tile(tile, fk, next_new_tile, nt);
                    if (nt < tEnd) issue_loads(nt, nk, next_new_tile);
                    else if (!kDyn && q_e0 >= 0) issue_quarter_loads();   // behind this wave's last full tile: its quarter tile
                    FE_TILE_STAMP(x_stamps, smem + G::LDS_BYTES, wave, lane, dbg_it, 1);   // the last B fragments built, the next unit's loads issued
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            FE_TILE_STAMP(x_stamps, smem + G::LDS_BYTES, wave, lane, dbg_it, 2);   // the matrix work is issued
            double* ob = L->o;
#pragma unroll
            for (int t = 0; t < G::BT; ++t)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) ob[tile_index<NP>(n, 16 * t + g + 4 * qq)] = acc[t][qq];
#pragma unroll
            for (int q = 0; q < G::NS; ++q) {
                const int i = 16 * G::BT + 4 * q + g;
                if (16 * G::BT + 4 * q + 3 < NP || i < NP) ob[tile_index<NP>(n, i)] = accs[q];
            }
            wave_lds_fence();
            double* op = out + e0 * NP;
#pragma unroll
            for (int c = 0; c < G::SUB_INSTR; ++c) {
                const int qc = c * 64 + lane;
                if ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS) {
                    const v2d val = *reinterpret_cast<const v2d*>(ob + 2 * tile_dst_chunk<NP>(qc));
                    __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + 2 * qc));
                }
            }
            wave_lds_fence();
            FE_TILE_STAMP(x_stamps, smem + G::LDS_BYTES, wave, lane, dbg_it, 3);   // the stores are issued
            ++dbg_it;
            fk = nk;
            tile = nt;
            continue;
        }

        // ---- all B fragments of the tile: Ju[(jq, r)][e = 16m + n], j = 4 jq + g
        double bfrag[M][G::KSJ][G::NBF];
        double jkeep[MODE == 4 ? M : 1][MODE == 4 ? ND * ND : 1];   // MODE 4: J for the epilogue
#pragma unroll
        for (int m = 0; m < M; ++m) {
            double jac[G::NJ > 0 ? G::NJ : 1];
#pragma unroll
            for (int k = 0; k < G::NJ; ++k)   // jac[x*ND + r]  (MODE 1: jac[s]; MODE 2: jac[0] = J[e])
                jac[k] = (MODE == 1 && jes) ? L->j[(16 * m + n) * ND + k] : L->j[k * G::TEL + 16 * m + n];
            if (G::NJ == 0) jac[0] = 1.0;
            if (MODE == 4) {
#pragma unroll
                for (int k = 0; k < ND * ND; ++k) jkeep[m][k] = jac[k];
            }
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq) {
                const int j = 4 * jq + g;
                const int jc = j < NP ? j : 0;
                double ux[G::NPLANES];
#pragma unroll
                for (int x = 0; x < G::NPLANES; ++x) ux[x] = j < NP ? L->u[x][tile_index<NP>(16 * m + n, jc)] : 0.0;
#pragma unroll
                for (int r = 0; r < G::NBF; ++r) {
                    if (MODE == 3 || MODE == 4) {
                        bfrag[m][jq][r] = ux[0];
                    } else if (MODE) {
                        bfrag[m][jq][r] = jac[r] * ux[0];
                    } else {
                        double v = jac[r] * ux[0];
#pragma unroll
                        for (int x = 1; x < ND; ++x) v = __builtin_fma(jac[x * ND + r], ux[x], v);   // explicit fma: see the note at the top
                        bfrag[m][jq][r] = v;
                    }
                }
            }
        }
        // the u / J tiles are now in registers: hand the buffers back to the DMA engine
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                for (int r = 0; r < G::NBF; ++r) asm volatile("" : "+v"(bfrag[m][jq][r]));
        if (MODE == 4) {
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int k = 0; k < ND * ND; ++k) asm volatile("" : "+v"(jkeep[m][k]));
        }
        resolve_next_tile(tile, fk, next_new_tile, nt);   // (dynamic walk: the ticket asked for earlier is read here)
        if (nt < tEnd && !x_no_loads) issue_loads(nt, nk, next_new_tile);
        FE_TILE_STAMP(x_stamps, smem + G::LDS_BYTES, wave, lane, dbg_it, 1);   // B fragments built, the next unit's loads issued
        if constexpr (x_lds_ticket && MODE == 0 && ND == 3 && M == 1) {
            // experiment (kDbg & 64): touch the tile AFTER next -- one dword per 128-byte line of its three planes and nine J rows --
            // so that its LDS-DMA loads, which can only go out one MFMA phase ahead of their use, find the lines in the L2
            const int64_t pt = nt + stride;
            if (pt < tEnd) {
                double sink = 0.0;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int line = c * 64 + lane;                 // 3 x 35 plane lines, then 9 J lines
                    const char* a = nullptr;
                    if (line < 105) a = reinterpret_cast<const char*>(field_in(P, 0)) + ((int64_t)(line / 35) * E + pt * G::TEL) * (NP * 8) + (line % 35) * 128;
                    else if (line < 114) a = reinterpret_cast<const char*>(J) + ((int64_t)(line - 105) * E + pt * G::TEL) * 8;
                    if (a) { float v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(a) : "memory"); sink += v; }
                }
                asm volatile("" :: "v"(sink));
            }
        }

        if (MODE == 4) {   // grad by components: separate accumulators per r, J contraction in the epilogue
#pragma unroll
            for (int m = 0; m < M; ++m) {
                v4d acc[NC][G::BT > 0 ? G::BT : 1];
                double accs[NC][G::NS > 0 ? G::NS : 1];
#pragma unroll
                for (int r = 0; r < NC; ++r) {
#pragma unroll
                    for (int t = 0; t < G::BT; ++t) acc[r][t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < G::NS; ++q) accs[r][q] = 0.0;
                }
#pragma unroll
                for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                    for (int r = 0; r < NC; ++r) {
#pragma unroll
                        for (int t = 0; t < G::BT; ++t)
                            acc[r][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_big(t, jq, r), bfrag[m][jq][0], acc[r][t], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < G::NS; ++q)
                            accs[r][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(as_lane[((jq * NC + r) * G::NS + q) * 16],
                                                                            bfrag[m][jq][0], accs[r][q], 0, 0, 0);
                    }
#pragma unroll
                for (int x = 0; x < ND; ++x) {
                    double* ob = L->o;
#pragma unroll
                    for (int t = 0; t < G::BT; ++t)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            double v = jkeep[m][x * ND] * acc[0][t][qq];
#pragma unroll
                            for (int r = 1; r < ND; ++r) v = __builtin_fma(jkeep[m][x * ND + r], acc[r][t][qq], v);
                            ob[tile_index<NP>(n, 16 * t + g + 4 * qq)] = v;
                        }
#pragma unroll
                    for (int q = 0; q < G::NS; ++q) {
                        const int i = 16 * G::BT + 4 * q + g;
                        double v = jkeep[m][x * ND] * accs[0][q];
#pragma unroll
                        for (int r = 1; r < ND; ++r) v = __builtin_fma(jkeep[m][x * ND + r], accs[r][q], v);
                        if (16 * G::BT + 4 * q + 3 < NP || i < NP) ob[tile_index<NP>(n, i)] = v;
                    }
                    wave_lds_fence();
                    double* op = out + ((int64_t)x * E + e0 + 16 * m) * NP;
#pragma unroll
                    for (int c = 0; c < G::SUB_INSTR; ++c) {
                        const int qc = c * 64 + lane;
                        if ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS) {
                            const v2d val = *reinterpret_cast<const v2d*>(ob + 2 * tile_dst_chunk<NP>(qc));
                            __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + 2 * qc));
                        }
                    }
                    wave_lds_fence();
                }
            }
            fk = nk;
            tile = nt;
            continue;
        }

#pragma unroll
        for (int m = 0; m < M; ++m) {
            // ---- BT x 3 KSJ big + NS x 3 KSJ small MFMAs
            v4d acc[G::BT > 0 ? G::BT : 1];
            double accs[G::NS > 0 ? G::NS : 1];
#pragma unroll
            for (int t = 0; t < G::BT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < G::NS; ++q) accs[q] = 0.0;
            if (x_no_mfma) {
                double sum = 0.0;
#pragma unroll
                for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                    for (int r = 0; r < NC; ++r) sum += bfrag[m][jq][r];
#pragma unroll
                for (int t = 0; t < G::BT; ++t) acc[t] = v4d{sum, sum, sum, a_big(t, 0, 0)};
#pragma unroll
                for (int q = 0; q < G::NS; ++q) accs[q] = sum;
            } else {
#pragma unroll
                for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                    for (int r = 0; r < NC; ++r) {
#pragma unroll
                        for (int t = 0; t < G::BT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_big(t, jq, r), bfrag[m][jq][r], acc[t], 0, 0, 0);
#pragma unroll
                        for (int q = 0; q < G::NS; ++q)
                            accs[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(as_lane[((jq * NC + r) * G::NS + q) * 16],
                                                                         bfrag[m][jq][r], accs[q], 0, 0, 0);
                    }
            }

            FE_TILE_STAMP(x_stamps, smem + G::LDS_BYTES, wave, lane, dbg_it, 2);   // the matrix work is issued
            // ---- transposed store.  16x16x4 C/D: lane (g, n) holds out[e][16t + g + 4q'];
            //      4x4x4_4b D of group q: lane (g, n) holds out[e][16 BT + 4q + g]
            double* ob = L->o;
#pragma unroll
            for (int t = 0; t < G::BT; ++t)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) ob[tile_index<NP>(n, 16 * t + g + 4 * qq)] = acc[t][qq];
#pragma unroll
            for (int q = 0; q < G::NS; ++q) {
                const int i = 16 * G::BT + 4 * q + g;
                if (16 * G::BT + 4 * q + 3 < NP || i < NP) ob[tile_index<NP>(n, i)] = accs[q];
            }
            wave_lds_fence();
            double* op = out + (e0 + 16 * m) * NP;
#pragma unroll
            for (int c = 0; c < G::SUB_INSTR; ++c) {
                const int qc = c * 64 + lane;
                if ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS) {
                    const v2d val = *reinterpret_cast<const v2d*>(ob + 2 * tile_dst_chunk<NP>(qc));
                    if (x_no_stores) { if (val[0] == 1.2345e-300) op[2 * qc] = val[1]; }
                    else __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + 2 * qc));
                }
            }
            wave_lds_fence();
        }
        FE_TILE_STAMP(x_stamps, smem + G::LDS_BYTES, wave, lane, dbg_it, 3);   // the stores are issued
        ++dbg_it;
        fk = nk;
        tile = nt;
    }
    if constexpr (kIlv && !kDyn) {
        if (q_e0 >= 0) {
            // ---- the quarter tile: elements q_e0 .. q_e0 + 3.  Lane (g, n) works for element n & 3 (the four blocks n >> 2 of the
            //      4x4x4_4b instruction see the same four elements); block b of an MFMA with the 16-row fragment a_big(t, ., .) is
            //      rows 16 t + 4 b .. + 3, so the lane receives out[16 t + 4 (n >> 2) + g][element n & 3].
            if (first) wait_vmcnt<0>();                      // (no full tile before it)
            else wait_vmcnt<G::STORES>();                    // younger than its loads: the last full tile's stores
            const int el = n & 3;
            double jac[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) jac[k] = L->j[k * 4 + el];
            auto read_quad = [&](int jq, double (&ux)[3]) {
                const int j = 4 * jq + g;
                const int jc = j < NP ? j : 0;
#pragma unroll
                for (int x = 0; x < 3; ++x) ux[x] = j < NP ? L->u[x][el * NP + jc] : 0.0;
            };
            auto build_quad = [&](const double (&ux)[3], double (&bq)[3]) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    double v = jac[r] * ux[0];
#pragma unroll
                    for (int x = 1; x < 3; ++x) v = __builtin_fma(jac[x * 3 + r], ux[x], v);
                    bq[r] = v;
                }
            };
            double acc4[G::BT > 0 ? G::BT : 1], accs[G::NS > 0 ? G::NS : 1];
#pragma unroll
            for (int t = 0; t < G::BT; ++t) acc4[t] = 0.0;
#pragma unroll
            for (int q = 0; q < G::NS; ++q) accs[q] = 0.0;
            double ux[2][3], bq[2][3];
            read_quad(0, ux[0]);
            read_quad(1, ux[1]);
            build_quad(ux[0], bq[0]);
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq) {
                const int cur = jq & 1, nxt = cur ^ 1;
                if (jq + 2 < G::KSJ) read_quad(jq + 2, ux[cur]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int t = 0; t < G::BT; ++t) acc4[t] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_big(t, jq, r), bq[cur][r], acc4[t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < G::NS; ++q)
                        accs[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(as_lane[((jq * 3 + r) * G::NS + q) * 16], bq[cur][r], accs[q], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (jq + 1 < G::KSJ) build_quad(ux[nxt], bq[nxt]);
            }
            double* ob = L->o;                               // [4][NP], compact
#pragma unroll
            for (int t = 0; t < G::BT; ++t) ob[el * NP + 16 * t + 4 * (n >> 2) + g] = acc4[t];
#pragma unroll
            for (int q = 0; q < G::NS; ++q) {
                const int i = 16 * G::BT + 4 * q + g;
                if ((n >> 2) == 0 && i < NP) ob[el * NP + i] = accs[q];
            }
            wave_lds_fence();
            double* op = field_out(P, 0) + q_e0 * NP;
            constexpr int QC = 4 * NP / 2;
#pragma unroll
            for (int c = 0; c < kQuarterStores; ++c) {
                const int qc = c * 64 + lane;
                if (qc < QC) {
                    const v2d val = *reinterpret_cast<const v2d*>(ob + 2 * qc);
                    __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + 2 * qc));
                }
            }
            wave_lds_fence();
        }
    }
#ifdef FE_EXPERIMENTS
    if (kDbg & 128) {   // stamps out: the tile stamps, and {kernel entry, -, loop end, XCC_ID | HW_ID << 8 | tiles << 40} as fe_grad.h
        const int w = bid * G::WAVES + wave;
        FE_TILE_STAMPS_OUT(true, smem + G::LDS_BYTES, wave, lane, w);
        if (lane == 0 && w < 4096) {
            fe_dbg_stamps[w][0] = dbg_entry; fe_dbg_stamps[w][1] = dbg_entry; fe_dbg_stamps[w][2] = __builtin_amdgcn_s_memrealtime();
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
            fe_dbg_stamps[w][3] = xcc | ((unsigned long long)hw << 8) | ((unsigned long long)dbg_it << 40);
        }
    }
#endif
    if constexpr (kDyn) {
        // the last wave of a pool to report leaves the pool's two counters zeroed for the next launch
        if (reported) {
            const unsigned pool_blocks = (nblk / (8 * kTailPools)) * 8 + (unsigned)max(0, min(8, (int)(nblk % (8 * kTailPools)) - 8 * pool));
            const unsigned before = tail_wait<G::STORES, 1>();
            if (before + 1 == pool_blocks * G::WAVES && lane == 0) {
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// the plain div with the B build interleaved into the matrix phase (kIlv above): short launches, static walk
template <int NP, int kDbg = 0>
__global__ __launch_bounds__(256, 2) void div3d_mfma_ilv_kernel(
    const double* __restrict__ J, const double* __restrict__ D, FieldPtrs P, int nb, int64_t E, int64_t nTiles, int opT) {
    div3d_mfma_body<NP, 1, kDbg, 0, 3, false, false, false, false, true>(J, D, nullptr, P, nb, E, nTiles, opT, 0, blockIdx.x, gridDim.x);
}

// the plain single-field div with a dynamic walk (see fe_common.h)
// (kBatched: see grad3d_mfma_tail_kernel)
template <int NP, int M, bool kBatched = false, bool kIlv = false>
__global__ __launch_bounds__(256, 2) FE_TAIL_KERNEL_ATTR void div3d_mfma_tail_kernel(
    const double* __restrict__ J, const double* __restrict__ D, FieldPtrs P, int nb, int64_t E, int64_t nTiles, int opT,
    unsigned* __restrict__ tail, int64_t t_static) {
    div3d_mfma_body<NP, M, 0, 0, 3, false, false, false, true, kIlv>(J, D, nullptr, P, kBatched ? nb : 1, E, nTiles, opT, 0, blockIdx.x,
                                                                      gridDim.x, nullptr, tail, t_static);
}

// triangles (ND = 2) with a dynamic walk: div (MODE 0) and grad by components (MODE 4), any number of fields
template <int NP, int M, int MODE>
__global__ __launch_bounds__(256, 2) FE_TAIL_KERNEL_ATTR void nd2_mfma_tail_kernel(
    const double* __restrict__ J, const double* __restrict__ D, FieldPtrs P, int nb, int64_t E, int64_t nTiles, int opT,
    unsigned* __restrict__ tail, int64_t t_static) {
    div3d_mfma_body<NP, M, 0, MODE, 2, false, false, false, true>(J, D, nullptr, P, nb, E, nTiles, opT, 0, blockIdx.x, gridDim.x, nullptr,
                                                                  tail, t_static);
}

// div in eight-wave blocks (p = 5: A in LDS, planes streamed) with a dynamic walk
template <int NP>
__global__ __launch_bounds__(512, 1) FE_TAIL_KERNEL_ATTR void div_w8_tail_kernel(
    const double* __restrict__ J, const double* __restrict__ D, FieldPtrs P, int nb, int64_t E, int64_t nTiles, int opT,
    unsigned* __restrict__ tail, int64_t t_static) {
    div3d_mfma_body<NP, 1, 0, 0, 3, true, true, false, true>(J, D, nullptr, P, nb, E, nTiles, opT, 0, blockIdx.x, gridDim.x, nullptr, tail,
                                                             t_static);
}

// grad by components in eight-wave blocks (p = 5) with a dynamic walk (nb is a run-time argument although the launcher passes
// 1: with the constant the compiler restructures the loop and needs 256 registers and scratch instead of ~200)
template <int NP>
__global__ __launch_bounds__(512, 1) FE_TAIL_KERNEL_ATTR void grad_w8_tail_kernel(
    const double* __restrict__ J, const double* __restrict__ D, FieldPtrs P, int nb, int64_t E, int64_t nTiles, int opT,
    unsigned* __restrict__ tail, int64_t t_static) {
    div3d_mfma_body<NP, 1, 0, 4, 3, true, true, false, true>(J, D, nullptr, P, nb, E, nTiles, opT, 0, blockIdx.x, gridDim.x, nullptr, tail,
                                                             t_static);
}

template <int NP, int M, int kDbg = 0, int MODE = 0, int ND = 3, bool ALDS = false, bool W8 = false, bool kPrep = false>
__global__ __launch_bounds__(W8 ? 512 : 256, W8 ? 1 : 2) void div3d_mfma_kernel(
    const double* __restrict__ J, const double* __restrict__ D, const void* __restrict__ prep, FieldPtrs P, int nb,
    int64_t E, int64_t nTiles, int opT, int jes) {
    div3d_mfma_body<NP, M, kDbg, MODE, ND, ALDS, W8, kPrep>(J, D, prep, P, nb, E, nTiles, opT, jes, blockIdx.x,
                                                            gridDim.x);
}

// grad-type planes at p = 5 (MODE 5): the fields' u pointers travel in P.v, everything else in Q
template <int NP>
__global__ __launch_bounds__(512, 1) void gradplanes_bycomp_kernel(GradFields Q, const double* __restrict__ D, FieldPtrs P,
                                                                  int nb, int64_t E, int64_t nTiles, int opT) {
    div3d_mfma_body<NP, 1, 0, 5, 3, true, true>(nullptr, D, nullptr, P, nb, E, nTiles, opT, 0, blockIdx.x, gridDim.x, &Q);
}

// The div sections of a prepared operator (plain div of tetrahedra): big-tile fragment
// f = (jq NC + r) BT + t of lane (g, n) is D'[row 16 t + n][k = (jq, r), j = 4 jq + g], and the table
// of the 4-row groups exactly as the prologue above lays it out in LDS.  Blocks 0 .. FRAGS-1 write one
// fragment each, the blocks behind them 64 table entries each.
template <int NP, int M>
__global__ __launch_bounds__(64) void div_prepare_kernel(const double* __restrict__ D, void* __restrict__ prepared,
                                                         int opT) {
    using G = DivGeom<NP, M>;
    constexpr int NC = G::NC, FRAGS = G::BT * G::KSJ * NC;
    char* pb = reinterpret_cast<char*>(prepared);
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    const int istride = opT ? 1 : NP, jstride = opT ? NP : 1;
    if ((int)blockIdx.x < FRAGS) {
        constexpr int BT1 = G::BT > 0 ? G::BT : 1;
        const int f = blockIdx.x, t = f % BT1, jq = (f / BT1) / NC, r = (f / BT1) % NC;
        const int j = 4 * jq + g, i = 16 * t + n;
        store_prepared_fragment(pb + kPrepDivOff, f, lane, j < NP ? D[r * (NP * NP) + i * istride + j * jstride] : 0.0);
        return;
    }
    const int idx = ((int)blockIdx.x - FRAGS) * 64 + lane;
    if (idx >= G::ASMALL_D) return;
    const int row4 = idx & 3, gg = (idx >> 2) & 3, q = (idx >> 4) % G::NS, ks = (idx >> 4) / G::NS;
    const int i = 16 * G::BT + 4 * q + row4, j = 4 * (ks / NC) + gg, r = ks % NC;
    reinterpret_cast<double*>(pb + kPrepDivSmallOff)[idx] =
        (j < NP && i < NP) ? D[r * (NP * NP) + i * istride + j * jstride] : 0.0;
}

}  // namespace fe
