// fe_facemass.h -- face-mass (lift) einsum, NB fields sharing J and R:
//   out_k[e,i] = sum_{f,j} J[e,f] R[f,i,j] v_k[f,e,j]
// ('ef,fij,fej->ei' x b, reference: test/test_loopy_utils.py:34-48; layout
// sibling 'ifj,fe,fej->ei': tuning/impls/ifj_fe_fej_to_ei.py:18-277, printed
// kernel doc/compiler_writer_tutorial.rst:357-493).
//
// Schedule = the opt_einsum-optimal one: Jv_k[(f,j), e] = J[e,f] v_k[f,e,j] (one
// multiply, VALU, produced in MFMA B-fragment layout), then
//   out_k[i, e] = sum_{(f,j)} R'[i, (f,j)] * Jv_k[(f,j), e]
// on the matrix cores, A = R' (Np x 4 Nfp; K = 4 Nfp = Nfp k-steps exactly) resident in
// registers: rows 0 .. 16 BT - 1 as BT = Np / 16 tiles on v_mfma_f64_16x16x4_f64 and the remaining
// rows in groups of four on v_mfma_f64_4x4x4_4b_f64 (see fe_div.h), per (wave tile, field) "unit".
// Templated on the tetrahedral orders p = 1..4: (Np, Nfp) = (4,3), (10,6), (20,10), (35,15), and
// on M: a wave tile is 16 M elements.
// Data movement: a unit's four face slabs (4 x 16 M Nfp contiguous doubles) come in by LDS-DMA
// into a 2-slot ring, two units ahead; J for a tile comes with its first unit; results leave
// through a separate LDS transposition buffer as contiguous 16-byte-per-lane stores.  Loads for
// unit m+2 are issued as soon as unit m's B fragments are in registers, i.e. BEFORE unit m's
// MFMAs and stores, so the counted vmcnt at the top of a unit never waits for stores.
#pragma once
#include "fe_generic.h"
#include "fe_grad.h"

namespace fe {

constexpr int kFmNf = 4;   // faces of a tetrahedron (the default NF); triangles: NF = 3

// ALDS: all A fragments in LDS (fragment layout) instead of registers -- tetrahedra p = 5
// (Np = 56, Nfp = 21): 105 doubles of A per lane plus the B fragments exceed the register file.
// One block per CU then (see DivGeom).
// W8 (with ALDS): eight waves per block share the fragments; the field slab of a unit and the output
// transposition buffer are the same LDS and the next unit is requested after the stores (see DivGeom).
template <int NP, int NFP, int M, int NF = kFmNf, bool ALDS = false, bool W8 = false>
struct FmGeom {
    static constexpr int TEL = 16 * M;
    static constexpr int K = NF * NFP;
    static constexpr int KS = (K + 3) / 4;              // k-steps of 4 (NF = 4: exactly NFP of them)
    static constexpr int BT = NP / 16, NR = NP - 16 * BT, NS = (NR + 3) / 4;
    static constexpr int SLAB_D = TEL * NFP;            // doubles per face slab of a unit
    static constexpr int UNIT_D = NF * SLAB_D;
    static constexpr int SUB_D = 16 * NP;
    static constexpr int SLAB_CHUNKS = SLAB_D / 2, SLAB_INSTR = (SLAB_CHUNKS + 63) / 64;
    static constexpr int J_CHUNKS = NF * TEL / 2, J_INSTR = (J_CHUNKS + 63) / 64;   // NF TEL doubles
    static constexpr int SUB_CHUNKS = SUB_D / 2, SUB_INSTR = (SUB_CHUNKS + 63) / 64;
    static constexpr int UNIT_LOADS = NF * SLAB_INSTR;   // + J_INSTR at a tile start
    static constexpr int UNIT_STORES = M * SUB_INSTR;
    static_assert(!W8 || (ALDS && M == 1 && UNIT_D >= SUB_D), "eight-wave blocks: A in LDS, o inside the slab buffer");
    struct WaveLds {
        double v[W8 ? 1 : 2][UNIT_D];   // ring of field slabs: v[slot][f][e][j]
        double o[W8 ? 2 : SUB_D];       // output transposition buffer (one 16-element sub-tile)
        double j[NF * TEL];      // J tile, [e][f] or [f][e] as in global memory
    };
    static constexpr int WAVES = W8 ? 8 : 4;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int OP_D = NF * NP * NFP;
    static constexpr int WAVE_BYTES = (int)sizeof(WaveLds) * WAVES;
    static constexpr int AFR_BIG_D = ALDS ? BT * KS * 64 : 0;      // [ks][t][lane]
    static constexpr int AFR_D = AFR_BIG_D + (ALDS ? NS * KS * 16 : 0);   // + [ks][q][g][row]: 16 distinct values per group
    static constexpr int LDS_BYTES = (WAVE_BYTES > OP_D * 8 ? WAVE_BYTES : OP_D * 8) + AFR_D * 8;
    static constexpr int BLOCKS_PER_CU = ALDS ? 1 : 2;
    static_assert((TEL * NFP) % 2 == 0, "slabs are moved in 16-byte chunks");
    static_assert(2 * UNIT_STORES + UNIT_LOADS + J_INSTR <= 60, "counted vmcnt must fit the 6-bit field");
    static_assert(BLOCKS_PER_CU * LDS_BYTES <= 160 * 1024, "blocks per CU");
};

// UNIT_LOADS x 16-byte LDS-DMA for the field slabs (+ J_INSTR for J at a tile start).
template <int NP, int NFP, int M, bool kWithJ, int NF = kFmNf, bool ALDS = false, bool W8 = false>
__device__ __forceinline__ void fm_issue_unit_loads(const double* __restrict__ J,
                                                    const double* __restrict__ vk, int64_t E,
                                                    int64_t tile, int lane, unsigned lds_v,
                                                    unsigned lds_j, int jfe) {
    using G = FmGeom<NP, NFP, M, NF, ALDS, W8>;
    const int64_t e0 = tile * G::TEL;
    const char* vb = reinterpret_cast<const char*>(vk) + e0 * (NFP * 8) + lane * 16;
    if (jfe & kOpLoadsTemporal) {   // (fe_common.h: one scalar branch for the whole unit)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const char* vf = vb + (int64_t)f * E * (NFP * 8);
#pragma unroll
            for (int c = 0; c < G::SLAB_INSTR; ++c)
                if ((c + 1) * 64 <= G::SLAB_CHUNKS || c * 64 + lane < G::SLAB_CHUNKS)
                    glds16(vf + c * 1024, lds_v + f * (G::SLAB_D * 8) + c * 1024);
        }
    } else {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const char* vf = vb + (int64_t)f * E * (NFP * 8);
#pragma unroll
            for (int c = 0; c < G::SLAB_INSTR; ++c)
                if ((c + 1) * 64 <= G::SLAB_CHUNKS || c * 64 + lane < G::SLAB_CHUNKS)
                    glds16_nt(vf + c * 1024, lds_v + f * (G::SLAB_D * 8) + c * 1024);
        }
    }
    jfe &= 1;
    if (kWithJ) {
        const char* jb = reinterpret_cast<const char*>(J);
#pragma unroll
        for (int c = 0; c < G::J_INSTR; ++c) {
            const int q = c * 64 + lane;                       // 16-byte chunk of the J tile
            const int row = q / (G::TEL / 2), col = q - row * (G::TEL / 2);   // "fe": 4 rows of TEL doubles
            const char* src = jfe ? jb + ((int64_t)row * E + e0) * 8 + col * 16
                                  : jb + e0 * (NF * 8) + q * 16;
            if ((c + 1) * 64 <= G::J_CHUNKS || q < G::J_CHUNKS) glds16(src, lds_j + c * 1024);
        }
    }
}

// bid / nblk: see grad3d_mfma_body.
// kPrep (register-resident fragments only): the A fragments come from a prepared operator (`prep` =
// the whole prepared buffer: fragments ks BT + t of the 16-row tiles, then BT KS + ks NS + q of the
// 4-row groups); no LDS staging and no block barrier.  R itself is still needed by the remainder code.
// kDyn (register fragments, NB >= 3): behind two static rounds the tiles come by tickets (fe_common.h, dynamic walk): the
// ticket for the next tile is asked for with unit 0 of a tile and read with unit NB - 2, whose prefetch is the next tile's
// first unit; `tail` = the launch's counters (null: static walk), `t_static` = statically walked tiles.
template <int NP, int NFP, int M, int NB, int NF = kFmNf, bool ALDS = false, bool W8 = false, bool kPrep = false,
          bool kDyn = false>
__device__ __forceinline__ void facemass_mfma_body(
    const double* __restrict__ J, const double* __restrict__ R, const void* __restrict__ prep, const FieldPtrs& P,
    int64_t E, int64_t nTiles, int jfe_flags, int rlayout, const unsigned bid, const unsigned nblk,
    unsigned* __restrict__ tail = nullptr, int64_t t_static = 0) {
    const int jfe = jfe_flags & 1;   // J stored [nf][E]; bit kOpLoadsTemporal: the field slabs by plain loads (fe_common.h)
    static_assert(!kPrep || !ALDS, "prepared operators: fragments in registers");
    static_assert(!kDyn || (NB >= 3 && !ALDS && !W8 && !kPrep) || (NB >= 2 && ALDS && W8),
                  "dynamic walk: three or more fields with the fragments in registers, or the eight-wave blocks (p = 5)");
    using G = FmGeom<NP, NFP, M, NF, ALDS, W8>;
    using WaveLds = typename G::WaveLds;
    static_assert(NB >= 2 && NB <= kMaxFields, "2..8 fields per launch");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    WaveLds* L = reinterpret_cast<WaveLds*>(smem) + wave;
    const int n = lane & 15, g = lane >> 4;

    // ---- per-lane K decomposition: k = 4 ks + g = NFP f + j
    int voff[G::KS];   // offset of v[f][sub-tile 0 element n][j] inside a unit slab
    int joff[G::KS];   // offset of J[sub-tile 0 element n][f] inside the J tile
    double abig[(G::BT > 0 && !ALDS) ? G::BT : 1][ALDS ? 1 : G::KS];   // 16x16x4: lane (g, n) supplies A[row 16t + n][k = g]
    double asmall[(G::NS > 0 && !ALDS) ? G::NS : 1][ALDS ? 1 : G::KS]; // 4x4x4_4b group q: block n/4, row 16 BT + 4q + n%4, k = g
    double* afr = reinterpret_cast<double*>(smem + (G::LDS_BYTES - G::AFR_D * 8));   // ALDS: big tiles [ks][t][lane] ...
    double* afs = afr + G::AFR_BIG_D;                                                //       ... small groups [ks][q][g][row]
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {   // k = 4 ks + g; padded k-steps read a valid (finite) B value against A = 0
        const int kk = 4 * ks + g, k = kk < G::K ? kk : 0;
        const int f = k / NFP, j = k - f * NFP;
        voff[ks] = f * G::SLAB_D + n * NFP + j;
        joff[ks] = jfe ? f * G::TEL + n : n * NF + f;
    }
    // The elements behind the last full tile (fe_common.h: remainder_items), entry by entry on the VALU, with the operator read from
    // `Rsrc`: the block's LDS copy while it exists (see fe_grad.h), else global memory.
    auto remainder = [&](const double* Rsrc) {
        const int64_t jEs = jfe ? 1 : NF, jFs = jfe ? E : 1;
        const int rF = rlayout == 0 ? NP * NFP : rlayout == 1 ? NFP : rlayout == 2 ? NFP * NP : NP;
        const int rI = rlayout == 0 ? NFP : rlayout == 1 ? NF * NFP : 1;
        const int rJ = rlayout == 0 || rlayout == 1 ? 1 : rlayout == 2 ? NP : NF * NP;
        remainder_items(nTiles * G::TEL, E, NP, bid, nblk, [&](int64_t e, int i) {
            facemass_item<NB>(J, Rsrc, P, E, NP, NF, NFP, jEs, jFs, rF, rI, rJ, e, i);
        });
    };
    bool first_requested = false;   // (units 0 and 1 of this wave's first tile: see the staged prologue)
    if constexpr (kPrep) {
        load_prepared_fragments<(G::BT + G::NS) * G::KS>(reinterpret_cast<const char*>(prep) + kPrepFmOff, lane,
                                                       [&](int f, double v) {
            if (f < G::BT * G::KS) abig[f % (G::BT > 0 ? G::BT : 1)][f / (G::BT > 0 ? G::BT : 1)] = v;
            else asmall[(f - G::BT * G::KS) % (G::NS > 0 ? G::NS : 1)][(f - G::BT * G::KS) / (G::NS > 0 ? G::NS : 1)] = v;
        });
    } else {
        // R goes through LDS once per block (see stage_operator)
        double* rl = reinterpret_cast<double*>(smem);
        stage_operator<G::OP_D, G::THREADS>(R, rl);
        __syncthreads();
        // Waves whose own buffers the staging area does not reach (it covers the first OP_D doubles of the block's LDS: wave 0 at
        // p = 4) request the first two units of their first tile NOW, behind the operator's loads and the barrier, instead of behind
        // the fragment build: the first round's data start to move 1 - 2 us earlier (fe_div.h; profiles/r05/div_prologue_phases.txt).
        if constexpr (!ALDS && !W8) {
            const int64_t first_ = (int64_t)bid * G::WAVES + wave;
            if (wave * (int)sizeof(WaveLds) >= G::OP_D * 8 && first_ < nTiles) {
                const unsigned lv = lds_addr_uniform(L->v[0]), lj = lds_addr_uniform(L->j);
                fm_issue_unit_loads<NP, NFP, M, true, NF, ALDS, W8>(J, P.v[0], E, first_, lane, lv, lj, jfe_flags);
                fm_issue_unit_loads<NP, NFP, M, false, NF, ALDS, W8>(J, P.v[1], E, first_, lane, lv + G::UNIT_D * 8, lj, jfe_flags);
                first_requested = true;
            }
        }
        // operator layouts: 0 R[f][i][j], 1 L[i][f][j], 2 R[f][j][i], 3 L[j][f][i] -> strides of f, i, j
        const int sF = rlayout == 0 ? NP * NFP : rlayout == 1 ? NFP : rlayout == 2 ? NFP * NP : NP;
        const int sI = rlayout == 0 ? NFP : rlayout == 1 ? NF * NFP : 1;
        const int sJ = rlayout == 0 || rlayout == 1 ? 1 : rlayout == 2 ? NP : NF * NP;
        auto ridx = [&](int f, int i, int j) { return f * sF + i * sI + j * sJ; };
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const int kk = 4 * ks + g;
            const bool kok = kk < G::K;            // NF NFP need not be a multiple of 4: padded k-steps
            const int k = kok ? kk : 0;            // read a valid (finite) B value, multiply it by A = 0
            const int f = k / NFP, j = k - f * NFP;
#pragma unroll
            for (int t = 0; t < G::BT; ++t) {
                const int i = 16 * t + n;
                const double a = rl[ridx(f, i, j)];
                if (ALDS) {   // every wave builds the same fragments: wave w stores those of k-steps ks = w mod 4
                    if (ks % G::WAVES == wave) afr[(ks * G::BT + t) * 64 + lane] = kok ? a : 0.0;
                } else {
                    abig[t][ks] = kok ? a : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < G::NS; ++q) {
                const int i3 = 16 * G::BT + 4 * q + (n & 3), i3c = i3 < NP ? i3 : 0;
                const double a3 = rl[ridx(f, i3c, j)];
                if (ALDS) {
                    if (ks % G::WAVES == wave && n < 4) afs[(ks * G::NS + q) * 16 + g * 4 + n] = (i3 < NP && kok) ? a3 : 0.0;
                } else {
                    asmall[q][ks] = (i3 < NP && kok) ? a3 : 0.0;
                }
            }
        }
        if constexpr (!ALDS && !W8) remainder(rl);   // (while the block's copy of the operator is still there; see fe_div.h)
        __syncthreads();   // the staging area is reused as the waves' private buffers from here on
    }
    if constexpr (kPrep || ALDS || W8) remainder(R);

    const double* af_lane = afr + lane;
    const double* as_lane = afs + g * 4 + (n & 3);
    auto a_big = [&](int t, int ks) -> double {
        if constexpr (ALDS) return af_lane[(ks * G::BT + t) * 64];
        else return abig[t][ks];
    };
    auto a_small = [&](int q, int ks) -> double {
        if constexpr (ALDS) return as_lane[(ks * G::NS + q) * 16];
        else return asmall[q][ks];
    };

    const unsigned lds_v0 = lds_addr_uniform(L->v[0]);
    const unsigned lds_j = lds_addr_uniform(L->j);
    const int64_t stride = (int64_t)nblk * G::WAVES, tEnd = nTiles;
    const int64_t first = (int64_t)bid * G::WAVES + wave;
    if (first >= tEnd) return;

    if constexpr (W8) {
        // ---- eight waves per block: wait the unit -> B to registers -> MFMAs -> o through the slab
        //      buffer -> stores -> request the next unit
        fm_issue_unit_loads<NP, NFP, M, true, NF, ALDS, W8>(J, P.v[0], E, first, lane, lds_v0, lds_j, jfe_flags);
        double jv8[G::KS];
        // dynamic walk (fe_common.h): the ticket for the next tile is asked for with unit 0 (behind the vmcnt(0) at its top)
        // and read behind the last unit's stores, where the next tile's first unit is requested
        const bool dyn8 = kDyn && tail != nullptr && t_static < nTiles;   // grid-uniform
        const int pool8 = (bid >> 3) & (kTailPools - 1);
        unsigned* const counter8 = tail_pool_counters(tail, pool8);
        unsigned* const done8 = tail_pool_reports(counter8);
        bool reported8 = false;
        int64_t tile = first;
        while (tile < tEnd) {
            int64_t nt = tile + stride;
            bool asked8 = false;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                wait_vmcnt<0>();
                if constexpr (kDyn) {
                    if (k == 0 && dyn8 && !(tile < t_static && tile + stride < t_static)) {   // the next tile is not static
                        tail_request<0>(counter8);
                        asked8 = true;
                    }
                }
                if (k == 0) {
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) jv8[ks] = L->j[joff[ks]];
                }
                double bf[G::KS];
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) bf[ks] = jv8[ks] * L->v[0][voff[ks]];
                v4d acc[G::BT > 0 ? G::BT : 1];
                double accs[G::NS > 0 ? G::NS : 1];
#pragma unroll
                for (int t = 0; t < G::BT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < G::NS; ++q) accs[q] = 0.0;
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
#pragma unroll
                    for (int t = 0; t < G::BT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_big(t, ks), bf[ks], acc[t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < G::NS; ++q)
                        accs[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_small(q, ks), bf[ks], accs[q], 0, 0, 0);
                }
                double* ob = L->v[0];   // the slab is in registers (the MFMAs consumed it)
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
                double* op = P.out[k] + tile * G::TEL * NP;
                v2d held[G::SUB_INSTR];
#pragma unroll
                for (int c = 0; c < G::SUB_INSTR; ++c) {
                    const int qc = c * 64 + lane;
                    held[c] = ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS)
                                  ? *reinterpret_cast<const v2d*>(ob + 2 * tile_dst_chunk<NP>(qc)) : v2d{0.0, 0.0};
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // o has left the buffer
#pragma unroll
                for (int c = 0; c < G::SUB_INSTR; ++c) {
                    const int qc = c * 64 + lane;
                    if ((c + 1) * 64 <= G::SUB_CHUNKS || qc < G::SUB_CHUNKS)
                        __builtin_nontemporal_store(held[c], reinterpret_cast<v2d*>(op + 2 * qc));
                }
                wave_lds_fence();
                // ---- the next unit: the next field of this tile, or field 0 (and J) of the next tile
                if (k + 1 < NB) {
                    fm_issue_unit_loads<NP, NFP, M, false, NF, ALDS, W8>(J, P.v[(k + 1) % NB], E, tile, lane, lds_v0, lds_j, jfe_flags);
                } else {
                    if constexpr (kDyn) {
                        if (asked8) {   // younger than the request by now: this unit's stores (every unit starts with vmcnt(0))
                            const int64_t x = tail_ticket_tile(tail_wait<G::UNIT_STORES, 0>(), t_static, pool8, tEnd);
                            nt = x >= 0 ? x : tEnd;
                            if (x < 0) {   // this wave's pool is empty: stop asking, report
                                tail_request<1>(done8);
                                reported8 = true;
                            }
                        }
                    }
                    if (nt < tEnd) fm_issue_unit_loads<NP, NFP, M, true, NF, ALDS, W8>(J, P.v[0], E, nt, lane, lds_v0, lds_j, jfe_flags);
                }
            }
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

    // prologue: units 0 and 1 of the first tile (unless requested in front of the fragment build)
    if (!first_requested) {
        fm_issue_unit_loads<NP, NFP, M, true, NF, ALDS, W8>(J, P.v[0], E, first, lane, lds_v0, lds_j, jfe_flags);
        fm_issue_unit_loads<NP, NFP, M, false, NF, ALDS, W8>(J, P.v[1], E, first, lane, lds_v0 + G::UNIT_D * 8, lds_j, jfe_flags);
    }
    if constexpr (kPrep) prepared_fragments_landed();

    int slot = 0;
    bool warm = false;   // false for the first two units of this wave
    double jv[M][G::KS];
    const bool younger_half = bid >= (nblk + 1) / 2;
    int iteration = 0;
    const bool dyn = kDyn && tail != nullptr && t_static < nTiles;   // grid-uniform
    const int pool = (bid >> 3) & (kTailPools - 1);
    unsigned* const counter = tail_pool_counters(tail, pool);
    unsigned* const done = tail_pool_reports(counter);
    bool reported = false;
    int64_t tile = first;
    while (tile < tEnd) {
        balance_priority(younger_half, iteration++);
        // the next tile: tile + stride in the static walk; with tickets: asked for at unit 0, known from unit NB - 2 on
        const bool next_static = !dyn || (tile < t_static && tile + stride < t_static);
        int64_t nt = next_static ? tile + stride : tEnd;
        bool requested = false;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            // ---- wait for this unit's loads; younger ops: S(m-2), [the ticket asked for in unit m-1,] L(m+1), S(m-1)
            const bool next_is_tile_start = (k + 1 == NB);
            const bool has_next = !next_is_tile_start || (nt < tEnd);
            if (warm && has_next) {
                if (next_is_tile_start) wait_vmcnt<2 * G::UNIT_STORES + G::UNIT_LOADS + G::J_INSTR>();
                else if (kDyn && k == 1 && requested) wait_vmcnt<2 * G::UNIT_STORES + G::UNIT_LOADS + 1>();
                else wait_vmcnt<2 * G::UNIT_STORES + G::UNIT_LOADS>();
            } else {
                wait_vmcnt<0>();
            }
            if (k >= 1) warm = true;   // units 0,1 of the first tile are cold

            if (k == 0) {
#pragma unroll
                for (int m = 0; m < M; ++m)
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks)
                        jv[m][ks] = L->j[joff[ks] + (jfe ? 16 * m : 16 * m * NF)];
            }
            const double* vs = L->v[slot];
            double bfrag[M][G::KS];
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) bfrag[m][ks] = jv[m][ks] * vs[voff[ks] + 16 * m * NFP];
            // make sure the slab (and, at k == 0, the J tile) is in registers before
            // the slot is handed back to the DMA engine
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) asm volatile("" : "+v"(bfrag[m][ks]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

            if constexpr (kDyn) {
                if (k == 0 && !next_static) {   // the next tile comes by ticket
                    tail_request<0>(counter);
                    requested = true;
                }
                if (k == NB - 2 && requested) {
                    // younger than the request: the loads and stores of units 0 .. NB - 3 (at most; this unit has waited for more)
                    constexpr int kYounger = (NB - 2) * (G::UNIT_LOADS + G::UNIT_STORES);
                    const unsigned t = tail_wait<(kYounger < 56 ? kYounger : 56), 0>();
                    const int64_t x = tail_ticket_tile(t, t_static, pool, tEnd);
                    nt = x >= 0 ? x : tEnd;
                    if (x < 0) {   // this wave's pool is empty: stop asking, report
                        tail_request<1>(done);
                        reported = true;
                    }
                }
            }
            // ---- prefetch unit m+2 into the slot just drained
            {
                const int k2 = (k + 2) % NB;   // folds after unrolling
                const int64_t tile2 = (k + 2 < NB) ? tile : nt;
                if (tile2 < tEnd) {
                    if (k2 == 0)
                        fm_issue_unit_loads<NP, NFP, M, true, NF, ALDS, W8>(J, P.v[k2], E, tile2, lane,
                                                              lds_v0 + slot * (G::UNIT_D * 8), lds_j, jfe_flags);
                    else
                        fm_issue_unit_loads<NP, NFP, M, false, NF, ALDS, W8>(J, P.v[k2], E, tile2, lane,
                                                               lds_v0 + slot * (G::UNIT_D * 8), lds_j, jfe_flags);
                }
            }

#pragma unroll
            for (int m = 0; m < M; ++m) {
                // ---- BT x KS big + NS x KS small MFMAs
                v4d acc[G::BT > 0 ? G::BT : 1];
                double accs[G::NS > 0 ? G::NS : 1];
#pragma unroll
                for (int t = 0; t < G::BT; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < G::NS; ++q) accs[q] = 0.0;
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
#pragma unroll
                    for (int t = 0; t < G::BT; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_big(t, ks), bfrag[m][ks], acc[t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < G::NS; ++q)
                        accs[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_small(q, ks), bfrag[m][ks], accs[q], 0, 0, 0);
                }

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
                double* op = P.out[k] + (tile * G::TEL + 16 * m) * NP;
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
            slot ^= 1;
        }
        tile = nt;
    }
    if constexpr (kDyn) {
        // the last wave of a pool to report leaves the pool's two counters zeroed for the next launch (younger than its
        // report: the stores of the last two units)
        if (reported) {
            const unsigned pool_blocks = (nblk / (8 * kTailPools)) * 8 + (unsigned)max(0, min(8, (int)(nblk % (8 * kTailPools)) - 8 * pool));
            const unsigned before = tail_wait<2 * G::UNIT_STORES, 1>();
            if (before + 1 == pool_blocks * G::WAVES && lane == 0) {
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// eight-wave blocks with the fragments in LDS (p = 5) with a dynamic walk
template <int NP, int NFP, int NB>
__global__ __launch_bounds__(512, 1) FE_TAIL_KERNEL_ATTR void facemass_w8_tail_kernel(
    const double* __restrict__ J, const double* __restrict__ R, FieldPtrs P, int64_t E, int64_t nTiles, int jfe, int rlayout,
    unsigned* __restrict__ tail, int64_t t_static) {
    facemass_mfma_body<NP, NFP, 1, NB, kFmNf, true, true, false, true>(J, R, nullptr, P, E, nTiles, jfe, rlayout, blockIdx.x, gridDim.x,
                                                                       tail, t_static);
}

// the launch of fields in registers with a dynamic walk (see fe_common.h)
template <int NP, int NFP, int M, int NB, int NF = kFmNf>
__global__ __launch_bounds__(256, 2) FE_TAIL_KERNEL_ATTR void facemass_mfma_tail_kernel(
    const double* __restrict__ J, const double* __restrict__ R, FieldPtrs P, int64_t E, int64_t nTiles, int jfe, int rlayout,
    unsigned* __restrict__ tail, int64_t t_static) {
    facemass_mfma_body<NP, NFP, M, NB, NF, false, false, false, true>(J, R, nullptr, P, E, nTiles, jfe, rlayout, blockIdx.x,
                                                                      gridDim.x, tail, t_static);
}

template <int NP, int NFP, int M, int NB, int NF = kFmNf, bool ALDS = false, bool W8 = false, bool kPrep = false>
__global__ __launch_bounds__(W8 ? 512 : 256, W8 ? 1 : 2) void facemass_mfma_kernel(
    const double* __restrict__ J, const double* __restrict__ R, const void* __restrict__ prep, FieldPtrs P,
    int64_t E, int64_t nTiles, int jfe, int rlayout) {
    facemass_mfma_body<NP, NFP, M, NB, NF, ALDS, W8, kPrep>(J, R, prep, P, E, nTiles, jfe, rlayout, blockIdx.x,
                                                            gridDim.x);
}

// The face-mass section of a prepared operator: one block of 64 threads per fragment (layout: see
// facemass_mfma_body, kPrep); the values are those the staging prologue builds.
template <int NP, int NFP, int M, int NF = kFmNf>
__global__ __launch_bounds__(64) void facemass_prepare_kernel(const double* __restrict__ R, void* __restrict__ prepared,
                                                              int rlayout) {
    using G = FmGeom<NP, NFP, M, NF>;
    const int fr = blockIdx.x, lane = threadIdx.x, n = lane & 15, g = lane >> 4;
    if (fr >= (G::BT + G::NS) * G::KS) return;
    const int sF = rlayout == 0 ? NP * NFP : rlayout == 1 ? NFP : rlayout == 2 ? NFP * NP : NP;
    const int sI = rlayout == 0 ? NFP : rlayout == 1 ? NF * NFP : 1;
    const int sJ = rlayout == 0 || rlayout == 1 ? 1 : rlayout == 2 ? NP : NF * NP;
    const bool big = fr < G::BT * G::KS;
    const int ks = big ? fr / (G::BT > 0 ? G::BT : 1) : (fr - G::BT * G::KS) / (G::NS > 0 ? G::NS : 1);
    const int kk = 4 * ks + g;
    const bool kok = kk < G::K;
    const int k = kok ? kk : 0, f = k / NFP, j = k - f * NFP;
    int i;
    if (big) i = 16 * (fr % (G::BT > 0 ? G::BT : 1)) + n;
    else i = 16 * G::BT + 4 * ((fr - G::BT * G::KS) % (G::NS > 0 ? G::NS : 1)) + (n & 3);
    const bool ok = kok && i < NP;
    store_prepared_fragment(reinterpret_cast<char*>(prepared) + kPrepFmOff, fr, lane,
                            ok ? R[f * sF + i * sI + j * sJ] : 0.0);
}

}  // namespace fe
