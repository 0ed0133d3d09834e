// fe_facemass_f32.h -- face-mass (lift) einsum in float32 on the matrix cores (tetrahedra p = 4: Np = 35, Nfp = 15, 4 faces),
// nb fields sharing J and R:   out_k[e,i] = sum_{f,j} J[e,f] R[f,i,j] v_k[f,e,j]      ('ef,fij,fej->ei' x b and its layout
// siblings 'ifj,fe,fej->ei' / 'jfi,fe,fej->ei': the layouts of fe_facemass.h).
//
// The float32 counterpart of fe_facemass.h's register-fragment kernel with one 16-element sub-tile per wave tile:
//   B fragments  Jv_k[(f,j), e] = J[e,f] * v_k[f,e,j]   (one multiply, VALU, produced in MFMA B layout; k = 15 f + j = 4 ks + g),
//   out_k[i, e] = sum_k R'[i, k] * Jv_k[k, e]   on v_mfma_f32_16x16x4_f32: A = R' as 48 x 60 (rows 35..47 zero; K = 60 = 15
//   k-steps exactly) resident in registers (45 floats per lane), 3 x 15 = 45 MFMAs of 32 cycles per (tile, field) unit.
// Data movement: a unit's four face slabs (4 x 960 contiguous bytes) come in by LDS-DMA into a 2-slot ring two units ahead --
// the loads of unit m + 2 are issued as soon as unit m's B fragments are in registers, before its MFMAs and stores; J for a
// tile comes with the tile's first unit (into the J buffer of that unit's slot) and is kept in registers for the tile's
// other fields; results leave through a wave-private LDS transposition buffer as 1-KiB contiguous non-temporal stores.
// Twelve waves per CU (40 KB of LDS per block of four).  The number of fields is a run-time argument (1..8): units are walked
// (tile, field), field fastest.
// 1536 B and 17 040 flops per element at four fields: HBM roofline 88.7 TFLOP/s.  Operands must be 16-byte aligned with E a
// multiple of 4; the elements behind the last full tile: remainder_items (fe_common.h).
#pragma once
#include "fe_grad_f32.h"

namespace fe {

struct FmF32Geom {
    static constexpr int NP = 35, NFP = 15, NF = 4, TEL = 16, RT = 3;
    static constexpr int K = NF * NFP, KS = K / 4;      // 60 = 15 k-steps
    static constexpr int SLAB_F = TEL * NFP;            // floats per face slab of a unit (240)
    static constexpr int UNIT_F = NF * SLAB_F;          // 960
    static constexpr int SLAB_CHUNKS = SLAB_F / 4;      // 60: one instruction per face
    static constexpr int J_CHUNKS = NF * TEL / 4;       // 16
    static constexpr int SUB_F = TEL * NP, SUB_CHUNKS = SUB_F / 4, SUB_INSTR = (SUB_CHUNKS + 63) / 64;   // 560, 140, 3
    static constexpr int UNIT_LOADS = NF, J_INSTR = 1, UNIT_STORES = SUB_INSTR;
    struct WaveIn {
        float v[2][UNIT_F];      // ring of field slabs: v[slot][f][e][j]
        float j[2][NF * TEL];    // J tile of the unit in that slot, [e][f] or [f][e] as in global memory
    };
    static constexpr int WAVES = 4;
    static constexpr int OP_F = NF * NP * NFP;          // 2100
    static constexpr int IN_BYTES = (int)sizeof(WaveIn) * WAVES;
    static constexpr int OUT_BYTES = SUB_F * 4 * WAVES;
    static constexpr int OP_BYTES = (OP_F * 4 + 15) / 16 * 16;
    static constexpr int LDS_BYTES = IN_BYTES + (OUT_BYTES > OP_BYTES ? OUT_BYTES : OP_BYTES);
    static constexpr int BLOCKS_PER_CU = 3;
    static_assert(SLAB_CHUNKS <= 64 && J_CHUNKS <= 64, "one load instruction per slab / J tile");
    static_assert(BLOCKS_PER_CU * LDS_BYTES <= 160 * 1024, "blocks per CU");
    static_assert(2 * UNIT_STORES + UNIT_LOADS + J_INSTR <= 60, "counted vmcnt must fit the 6-bit field");
};

// field k of a launch as float pointers (FieldPtrs carries them as double*: the C ABI's argument pack is untyped)
__device__ __forceinline__ const float* field_in_f32(const FieldPtrs& P, int k) { return reinterpret_cast<const float*>(field_in(P, k)); }
__device__ __forceinline__ float* field_out_f32(const FieldPtrs& P, int k) { return reinterpret_cast<float*>(field_out(P, k)); }

// SMALL: rows 32..34 on v_mfma_f32_4x4x1_16B_f32 instead of a third 16-row tile (see fe_div_f32.h).
template <bool SMALL>
__global__ __launch_bounds__(256, 3) void facemass_mfma_f32_kernel(const float* __restrict__ J, const float* __restrict__ R,
                                                                   FieldPtrs P, int nb, int64_t E, int64_t nTiles, int jfe,
                                                                   int rlayout) {
    using G = FmF32Geom;
    constexpr int NP = G::NP, NFP = G::NFP, NF = G::NF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    G::WaveIn* L = reinterpret_cast<G::WaveIn*>(smem) + wave;
    float* ob = reinterpret_cast<float*>(smem + G::IN_BYTES) + wave * G::SUB_F;
    const int n = lane & 15, g = lane >> 4;
    const unsigned bid = blockIdx.x, nblk = gridDim.x;
    const int64_t stride = (int64_t)nblk * G::WAVES, tEnd = nTiles;
    const int64_t first = (int64_t)bid * G::WAVES + wave;
    const unsigned lds_v0 = lds_addr_uniform(L->v[0]), lds_j0 = lds_addr_uniform(L->j[0]);

    auto issue_unit = [&](int64_t t, int k, int slot) {
        const int64_t e0 = t * G::TEL;
        const char* vb = reinterpret_cast<const char*>(field_in_f32(P, k) + e0 * NFP) + lane * 16;
#pragma unroll
        for (int f = 0; f < NF; ++f)
            if (lane < G::SLAB_CHUNKS)
                glds16_nt(vb + (int64_t)f * E * (NFP * 4), lds_v0 + slot * (G::UNIT_F * 4) + f * (G::SLAB_F * 4));
        if (k == 0) {   // the tile's J: "fe": 4 rows of TEL floats, else TEL x 4 contiguous floats
            const int row = lane / (G::TEL / 4), col = lane - row * (G::TEL / 4);
            const char* src = jfe ? reinterpret_cast<const char*>(J + (int64_t)row * E + e0) + col * 16
                                  : reinterpret_cast<const char*>(J + e0 * NF) + lane * 16;
            if (lane < G::J_CHUNKS) glds16(src, lds_j0 + slot * (NF * G::TEL * 4));
        }
    };
    auto advance = [&](int64_t& t, int& k) {
        if (++k == nb) { k = 0; t += stride; }
    };

    // ---- units 0 and 1 of this wave, and behind them the operator -> LDS (over the output buffers)
    int64_t tile = first, t1 = first, t2;
    int fk = 0, k1 = 0, k2;
    advance(t1, k1);
    t2 = t1, k2 = k1;
    advance(t2, k2);
    if (tile < tEnd) {
        issue_unit(tile, 0, 0);
        if (t1 < tEnd) issue_unit(t1, k1, 1);
    }
    {
        float* rl = reinterpret_cast<float*>(smem + G::IN_BYTES);
        constexpr int kPer = (G::OP_F + 255) / 256;
        float tmp[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int idx = threadIdx.x + k * 256;
            tmp[k] = idx < G::OP_F ? R[idx] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int idx = threadIdx.x + k * 256;
            if (idx < G::OP_F) rl[idx] = tmp[k];
        }
    }
    __syncthreads();

    // ---- per-lane K decomposition (k = 4 ks + g = 15 f + j) and A fragments: lane (g, n) supplies A[row 16 t + n][k = g]
    // operator layouts: 0 R[f][i][j], 1 L[i][f][j], 2 R[f][j][i], 3 L[j][f][i] -> strides of f, i, j
    const int sF = rlayout == 0 ? NP * NFP : rlayout == 1 ? NFP : rlayout == 2 ? NFP * NP : NP;
    const int sI = rlayout == 0 ? NFP : rlayout == 1 ? NF * NFP : 1;
    const int sJ = rlayout == 0 || rlayout == 1 ? 1 : rlayout == 2 ? NP : NF * NP;
    int voff[G::KS], joff[G::KS];
    float afrag[G::RT][G::KS];
    {
        const float* rl = reinterpret_cast<const float*>(smem + G::IN_BYTES);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const int k = 4 * ks + g, f = k / NFP, j = k - f * NFP;
            voff[ks] = f * G::SLAB_F + n * NFP + j;
            joff[ks] = jfe ? f * G::TEL + n : n * NF + f;
#pragma unroll
            for (int t = 0; t < G::RT; ++t) {
                const int i = (SMALL && t == 2) ? 32 + (n & 3) : 16 * t + n;
                const float a = rl[f * sF + (i < NP ? i : 0) * sI + j * sJ];
                afrag[t][ks] = i < NP ? a : 0.f;
            }
        }
    }
    {   // the elements behind the last full tile, with the operator from the block's LDS copy (see fe_grad_f32.h)
        const float* rl = reinterpret_cast<const float*>(smem + G::IN_BYTES);
        const int64_t jEs = jfe ? 1 : NF, jFs = jfe ? E : 1;
        remainder_items(nTiles * G::TEL, E, NP, bid, nblk, [&](int64_t e, int i) {
            for (int k = 0; k < nb; ++k) {
                const float* vk = field_in_f32(P, k);
                float acc = 0.f;
                for (int f = 0; f < NF; ++f) {
                    const float jf = J[e * jEs + f * jFs];
                    for (int j = 0; j < NFP; ++j)
                        acc = __builtin_fmaf(rl[f * sF + i * sI + j * sJ], jf * vk[((int64_t)f * E + e) * NFP + j], acc);
                }
                field_out_f32(P, k)[e * NP + i] = acc;
            }
        });
    }
    __syncthreads();   // the staging area becomes the waves' output buffers

    int slot = 0, done = 0, iteration = 0;
    float jv[G::KS];
    const bool younger_half = bid >= (nblk + 1) / 2;
    while (tile < tEnd) {
        if (fk == 0) balance_priority(younger_half, iteration++);
        // ---- wait for this unit's loads; younger ops: S(m-2), L(m+1), S(m-1)
        if (done >= 2 && t1 < tEnd) {
            if (k1 == 0) wait_vmcnt<2 * G::UNIT_STORES + G::UNIT_LOADS + G::J_INSTR>();
            else wait_vmcnt<2 * G::UNIT_STORES + G::UNIT_LOADS>();
        } else {
            wait_vmcnt<0>();
        }
        const float* vs = L->v[slot];
        if (fk == 0) {
            const float* js = L->j[slot];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) jv[ks] = js[joff[ks]];
        }
        float bfrag[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) bfrag[ks] = jv[ks] * vs[voff[ks]];
        // the slab (and, at a tile start, the J tile) is in registers before the slot is handed back to the DMA engine
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) asm volatile("" : "+v"(bfrag[ks]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t2 < tEnd) issue_unit(t2, k2, slot);

        v4f acc[G::RT];
#pragma unroll
        for (int t = 0; t < G::RT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
            for (int t = 0; t < G::RT; ++t) {
                if (SMALL && t == 2) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(afrag[t][ks], bfrag[ks], acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[t][ks], bfrag[ks], acc[t], 0, 0, 0);
            }

        // ---- transposed store: lane (g, n) holds out[e0 + n][16 t + 4 g + v]
#pragma unroll
        for (int t = 0; t < G::RT; ++t)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                if (SMALL && t == 2) {
                    if (v < NP - 32) {
                        float x = acc[t][v];
                        x += __shfl_xor(x, 16);
                        x += __shfl_xor(x, 32);
                        if (g == 0) ob[n * NP + 32 + v] = x;
                    }
                } else {
                    const int i = 16 * t + 4 * g + v;
                    if (16 * t + 15 < NP || i < NP) ob[n * NP + i] = acc[t][v];
                }
            }
        wave_lds_fence();
        float* op = field_out_f32(P, fk) + tile * (G::TEL * NP);
#pragma unroll
        for (int c = 0; c < G::SUB_INSTR; ++c) {
            const int q = c * 64 + lane;
            if ((c + 1) * 64 <= G::SUB_CHUNKS || q < G::SUB_CHUNKS) {
                const v4f val = *reinterpret_cast<const v4f*>(ob + 4 * q);
                __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(op + 4 * q));
            }
        }
        wave_lds_fence();
        tile = t1, fk = k1;
        t1 = t2, k1 = k2;
        advance(t2, k2);
        slot ^= 1;
        ++done;
    }
}

// ---- the lower orders p = 1 ... 3 ((Np, Nfp) = (4, 3), (10, 6), (20, 10)) on the matrix cores (round 5; the tiled VALU kernel in float
//      before).  The kernel above written over the geometry: K = 4 Nfp = 12 / 24 / 40 is a whole number of k-steps at every order;
//      BT = Np / 16 sixteen-row tiles on v_mfma_f32_16x16x4_f32, the rows behind them in groups of four on v_mfma_f32_4x4x1_16B_f32
//      (fe_div_f32.h); M sixteen-element sub-tiles per wave tile (M = 1 / 2 / 4 for p = 3 / 2 / 1, as fe_facemass.h); units (tile, field)
//      through a ring of two slots.
template <int NP_, int NFP_, int M_>
struct FmF32GeomT {
    static constexpr int NP = NP_, NFP = NFP_, NF = 4, M = M_, TEL = 16 * M;
    static constexpr int BT = NP / 16, NR = NP - 16 * BT, NS = (NR + 3) / 4;
    static constexpr int K = NF * NFP, KS = K / 4;
    static constexpr int SLAB_F = TEL * NFP, UNIT_F = NF * SLAB_F;
    static constexpr int SLAB_CHUNKS = SLAB_F / 4, SLAB_INSTR = (SLAB_CHUNKS + 63) / 64;
    static constexpr int J_CHUNKS = NF * TEL / 4, J_INSTR = (J_CHUNKS + 63) / 64;
    static constexpr int SUB_F = TEL * NP, SUB_CHUNKS = SUB_F / 4, SUB_INSTR = (SUB_CHUNKS + 63) / 64;
    static constexpr int UNIT_LOADS = NF * SLAB_INSTR, UNIT_STORES = SUB_INSTR;
    struct WaveIn {
        float v[2][UNIT_F];      // ring of field slabs: v[slot][f][e][j]
        float j[2][NF * TEL];    // J tile of the unit in that slot, [e][f] or [f][e] as in global memory
    };
    static constexpr int WAVES = 4;
    static constexpr int OP_F = NF * NP * NFP;
    static constexpr int IN_BYTES = (int)sizeof(WaveIn) * WAVES;
    static constexpr int OUT_BYTES = SUB_F * 4 * WAVES;
    static constexpr int OP_BYTES = (OP_F * 4 + 15) / 16 * 16;
    static constexpr int LDS_BYTES = IN_BYTES + (OUT_BYTES > OP_BYTES ? OUT_BYTES : OP_BYTES);
    static constexpr int BLOCKS_PER_CU = 3 * LDS_BYTES <= 160 * 1024 ? 3 : 2;
    static_assert(K % 4 == 0 && SLAB_F % 4 == 0 && SUB_F % 4 == 0, "geometry");
    static_assert(BLOCKS_PER_CU * LDS_BYTES <= 160 * 1024, "blocks per CU");
    static_assert(2 * UNIT_STORES + UNIT_LOADS + J_INSTR <= 60, "counted vmcnt must fit the 6-bit field");
};

template <int NP_, int NFP_, int M_>
__global__ __launch_bounds__(256, 2) void facemass_mfma_f32_np_kernel(const float* __restrict__ J, const float* __restrict__ R,
                                                                      FieldPtrs P, int nb, int64_t E, int64_t nTiles, int jfe,
                                                                      int rlayout) {
    using G = FmF32GeomT<NP_, NFP_, M_>;
    constexpr int NP = G::NP, NFP = G::NFP, NF = G::NF, M = G::M;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typename G::WaveIn* L = reinterpret_cast<typename G::WaveIn*>(smem) + wave;
    float* ob = reinterpret_cast<float*>(smem + G::IN_BYTES) + wave * G::SUB_F;
    const int n = lane & 15, g = lane >> 4;
    const unsigned bid = blockIdx.x, nblk = gridDim.x;
    const int64_t stride = (int64_t)nblk * G::WAVES, tEnd = nTiles;
    const int64_t first = (int64_t)bid * G::WAVES + wave;
    const unsigned lds_v0 = lds_addr_uniform(L->v[0]), lds_j0 = lds_addr_uniform(L->j[0]);

    auto issue_unit = [&](int64_t t, int k, int slot) {
        const int64_t e0 = t * G::TEL;
        const char* vb = reinterpret_cast<const char*>(field_in_f32(P, k) + e0 * NFP) + lane * 16;
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int c = 0; c < G::SLAB_INSTR; ++c)
                if ((c + 1) * 64 <= G::SLAB_CHUNKS || c * 64 + lane < G::SLAB_CHUNKS)
                    glds16_nt(vb + (int64_t)f * E * (NFP * 4) + c * 1024, lds_v0 + slot * (G::UNIT_F * 4) + f * (G::SLAB_F * 4) + c * 1024);
        if (k == 0) {   // the tile's J: "fe": 4 rows of TEL floats, else TEL x 4 contiguous floats
#pragma unroll
            for (int c = 0; c < G::J_INSTR; ++c) {
                const int q = c * 64 + lane;
                const int row = q / (G::TEL / 4), col = q - row * (G::TEL / 4);
                const char* src = jfe ? reinterpret_cast<const char*>(J + (int64_t)row * E + e0) + col * 16
                                      : reinterpret_cast<const char*>(J + e0 * NF) + q * 16;
                if ((c + 1) * 64 <= G::J_CHUNKS || q < G::J_CHUNKS) glds16(src, lds_j0 + slot * (NF * G::TEL * 4) + c * 1024);
            }
        }
    };
    auto advance = [&](int64_t& t, int& k) {
        if (++k == nb) { k = 0; t += stride; }
    };

    // ---- units 0 and 1 of this wave, and behind them the operator -> LDS (over the output buffers)
    int64_t tile = first, t1 = first, t2;
    int fk = 0, k1 = 0, k2;
    advance(t1, k1);
    t2 = t1, k2 = k1;
    advance(t2, k2);
    if (tile < tEnd) {
        issue_unit(tile, 0, 0);
        if (t1 < tEnd) issue_unit(t1, k1, 1);
    }
    {
        float* rl = reinterpret_cast<float*>(smem + G::IN_BYTES);
        for (int idx = threadIdx.x; idx < G::OP_F; idx += 256) rl[idx] = R[idx];
    }
    __syncthreads();

    // operator layouts: 0 R[f][i][j], 1 L[i][f][j], 2 R[f][j][i], 3 L[j][f][i] -> strides of f, i, j
    const int sF = rlayout == 0 ? NP * NFP : rlayout == 1 ? NFP : rlayout == 2 ? NFP * NP : NP;
    const int sI = rlayout == 0 ? NFP : rlayout == 1 ? NF * NFP : 1;
    const int sJ = rlayout == 0 || rlayout == 1 ? 1 : rlayout == 2 ? NP : NF * NP;
    int voff[G::KS], joff[G::KS];   // per-lane K decomposition: k = 4 ks + g = Nfp f + j  (offsets of sub-tile 0)
    float abig[G::BT > 0 ? G::BT : 1][G::KS], asmall[G::NS > 0 ? G::NS : 1][G::KS];
    {
        const float* rl = reinterpret_cast<const float*>(smem + G::IN_BYTES);
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const int k = 4 * ks + g, f = k / NFP, j = k - f * NFP;
            voff[ks] = f * G::SLAB_F + n * NFP + j;
            joff[ks] = jfe ? f * G::TEL + n : n * NF + f;
#pragma unroll
            for (int t = 0; t < G::BT; ++t) abig[t][ks] = rl[f * sF + (16 * t + n) * sI + j * sJ];
#pragma unroll
            for (int q = 0; q < G::NS; ++q) {
                const int i = 16 * G::BT + 4 * q + (n & 3);
                const float a = rl[f * sF + (i < NP ? i : 0) * sI + j * sJ];
                asmall[q][ks] = i < NP ? a : 0.f;
            }
        }
    }
    {   // the elements behind the last full tile, with the operator from the block's LDS copy
        const float* rl = reinterpret_cast<const float*>(smem + G::IN_BYTES);
        const int64_t jEs = jfe ? 1 : NF, jFs = jfe ? E : 1;
        remainder_items(nTiles * G::TEL, E, NP, bid, nblk, [&](int64_t e, int i) {
            for (int k = 0; k < nb; ++k) {
                const float* vk = field_in_f32(P, k);
                float acc = 0.f;
                for (int f = 0; f < NF; ++f) {
                    const float jf = J[e * jEs + f * jFs];
                    for (int j = 0; j < NFP; ++j)
                        acc = __builtin_fmaf(rl[f * sF + i * sI + j * sJ], jf * vk[((int64_t)f * E + e) * NFP + j], acc);
                }
                field_out_f32(P, k)[e * NP + i] = acc;
            }
        });
    }
    __syncthreads();   // the staging area becomes the waves' output buffers

    int slot = 0, done = 0, iteration = 0;
    float jv[M][G::KS];
    const bool younger_half = bid >= (nblk + 1) / 2;
    while (tile < tEnd) {
        if (fk == 0) balance_priority(younger_half, iteration++);
        // ---- wait for this unit's loads; younger ops: S(m-2), L(m+1), S(m-1)
        if (done >= 2 && t1 < tEnd) {
            if (k1 == 0) wait_vmcnt<2 * G::UNIT_STORES + G::UNIT_LOADS + G::J_INSTR>();
            else wait_vmcnt<2 * G::UNIT_STORES + G::UNIT_LOADS>();
        } else {
            wait_vmcnt<0>();
        }
        const float* vs = L->v[slot];
        if (fk == 0) {
            const float* js = L->j[slot];
#pragma unroll
            for (int m = 0; m < M; ++m)
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) jv[m][ks] = js[joff[ks] + (jfe ? 16 * m : 16 * m * NF)];
        }
        float bfrag[M][G::KS];
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) bfrag[m][ks] = jv[m][ks] * vs[voff[ks] + 16 * m * NFP];
        // the slabs (and, at a tile start, the J tile) are in registers before the slot is handed back to the DMA engine
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) asm volatile("" : "+v"(bfrag[m][ks]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t2 < tEnd) issue_unit(t2, k2, slot);

#pragma unroll
        for (int m = 0; m < M; ++m) {
            v4f accb[G::BT > 0 ? G::BT : 1], accq[G::NS > 0 ? G::NS : 1];
#pragma unroll
            for (int t = 0; t < G::BT; ++t) accb[t] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < G::NS; ++q) accq[q] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
#pragma unroll
                for (int t = 0; t < G::BT; ++t) accb[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(abig[t][ks], bfrag[m][ks], accb[t], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < G::NS; ++q) accq[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(asmall[q][ks], bfrag[m][ks], accq[q], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < G::BT; ++t)
#pragma unroll
                for (int v = 0; v < 4; ++v) ob[(16 * m + n) * NP + 16 * t + 4 * g + v] = accb[t][v];
#pragma unroll
            for (int q = 0; q < G::NS; ++q)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (16 * G::BT + 4 * q + v < NP) {
                        float x = accq[q][v];
                        x += __shfl_xor(x, 16);
                        x += __shfl_xor(x, 32);
                        if (g == 0) ob[(16 * m + n) * NP + 16 * G::BT + 4 * q + v] = x;
                    }
                }
        }
        wave_lds_fence();
        float* op = field_out_f32(P, fk) + tile * (G::TEL * NP);
#pragma unroll
        for (int c = 0; c < G::SUB_INSTR; ++c) {
            const int q = c * 64 + lane;
            if ((c + 1) * 64 <= G::SUB_CHUNKS || q < G::SUB_CHUNKS) {
                const v4f val = *reinterpret_cast<const v4f*>(ob + 4 * q);
                __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(op + 4 * q));
            }
        }
        wave_lds_fence();
        tile = t1, fk = k1;
        t1 = t2, k1 = k2;
        advance(t2, k2);
        slot ^= 1;
        ++done;
    }
}

}  // namespace fe
