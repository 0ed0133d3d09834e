// fe_div_f32.h -- div einsum 'xre,rij,xej->ei' in float32 on the matrix cores (tetrahedra p = 4, Np = 35).
//
// The float32 counterpart of fe_div.h's register-fragment kernel, one wave = one tile of 16 elements:
//   B fragments  Ju[(jq, r)][e] = sum_x J[x,r,e] * u[x,e,j]   on the VALU (one multiply, two explicit fused multiply-adds),
//                j = 4 jq + g, produced straight in MFMA B layout from the three u planes of the tile in LDS;
//   out[i, e] = sum_{(jq, r)} D[r, i, 4 jq + g] * Ju[(jq, r)][e]   on the matrix cores, A = D resident in registers (81 floats
//                per lane): rows 0..31 as two 16-row tiles on v_mfma_f32_16x16x4_f32 (2 x 27 MFMAs of 32 cycles), rows 32..34
//                on v_mfma_f32_4x4x1_16B_f32 (27 of 8 cycles, see SMALL below) -- float32 has no 4-row block instruction with
//                K = 4 (float64: v_mfma_f64_4x4x4_4b), and a third 16-row tile for three rows costs a third of the MFMA time.
// Data movement: the u planes and J of a tile come in by LDS-DMA into a ring of two buffers; ALL B fragments of a tile are built
// first, which frees its buffer, and tile t + 2 is requested into it before tile t's MFMAs and stores (counted vmcnt at the top
// of a tile: S(t-2), L(t+1), S(t-1) are younger than L(t)).  Two blocks per CU (73 KB of LDS each).  The output tile is
// transposed through wave-private LDS into 1-KiB contiguous non-temporal stores.  float32 C/D layout: lane
// (g = lane >> 4, n = lane & 15) holds rows 4 g + v (v = 0..3) of a 16-row tile for column (element) n.
// Measured at E = 1e6 (profiles/r03/float32_div_facemass.txt): one buffer and three blocks per CU with three row tiles 0.1465 ms,
// ring of two 0.1222 ms, ring of two with rows 32..34 on the 4x4x1 instruction 0.1094 ms = 73.0 TFLOP/s = 68.1 %.
// 596 B and 7980 flops per element: HBM roofline 107 TFLOP/s.  Operands must be 16-byte aligned with E a multiple of 4 (every
// plane and every row of J then starts on a 16-byte boundary; the launcher sends other sizes to the tiled kernel); the
// elements behind the last full tile: remainder_items (fe_common.h).
#pragma once
#include "fe_grad_f32.h"

namespace fe {

// RING: 1 = one u / J buffer per wave, three blocks per CU (the next tile is requested after this tile's B fragments);
//       2 = ring of two buffers, two blocks per CU (tile t + 2 is requested after tile t's B fragments).
template <int RING>
struct DivF32Geom {
    static constexpr int NP = 35, TEL = 16, RT = 3, KSJ = 9, KS = 3 * KSJ;
    static constexpr int PLANE_F = TEL * NP;            // floats: one u plane of a tile / the out tile (560)
    static constexpr int P_CHUNKS = PLANE_F / 4;        // 16-byte chunks (140)
    static constexpr int P_INSTR = (P_CHUNKS + 63) / 64;            // 3
    static constexpr int J_ROW_CHUNKS = TEL / 4, J_CHUNKS = 9 * J_ROW_CHUNKS;   // 36: one instruction
    static constexpr int LOADS = 3 * P_INSTR + 1, STORES = P_INSTR;
    struct Slot {
        float u[3][PLANE_F];     // u[x][e0 .. e0+15][0..34]
        float j[9 * TEL];        // J[x*3+r][e0 .. e0+15]
    };
    struct WaveIn {
        Slot s[RING];
    };
    static constexpr int WAVES = 4;
    static constexpr int OP_F = 3 * NP * NP;
    static constexpr int IN_BYTES = (int)sizeof(WaveIn) * WAVES;
    static constexpr int OUT_BYTES = PLANE_F * 4 * WAVES;           // one output transposition buffer per wave
    static constexpr int OP_BYTES = (OP_F * 4 + 15) / 16 * 16;
    static constexpr int LDS_BYTES = IN_BYTES + (OUT_BYTES > OP_BYTES ? OUT_BYTES : OP_BYTES);
    static constexpr int BLOCKS_PER_CU = RING == 1 ? 3 : 2;
    static_assert(BLOCKS_PER_CU * LDS_BYTES <= 160 * 1024, "blocks per CU");
    static_assert(2 * STORES + LOADS <= 60, "counted vmcnt must fit the 6-bit field");
};

__device__ __forceinline__ void div3d_item_f32(const float* __restrict__ J, const float* __restrict__ D,
                                               const float* __restrict__ u, float* __restrict__ out, int64_t E, int Np,
                                               int64_t e, int i, int opT) {
    const int si = opT ? 1 : Np, sj = opT ? Np : 1;
    float jac[9];
    for (int k = 0; k < 9; ++k) jac[k] = J[(int64_t)k * E + e];
    float acc = 0.f;
    for (int j = 0; j < Np; ++j) {
        float ux[3];
        for (int x = 0; x < 3; ++x) ux[x] = u[((int64_t)x * E + e) * Np + j];
        for (int r = 0; r < 3; ++r) {
            const float ju = __builtin_fmaf(jac[6 + r], ux[2], __builtin_fmaf(jac[3 + r], ux[1], jac[r] * ux[0]));
            acc = __builtin_fmaf(D[(int64_t)r * Np * Np + (int64_t)i * si + (int64_t)j * sj], ju, acc);
        }
    }
    out[e * Np + i] = acc;
}

// SMALL: rows 32..34 on v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4 x 4 x 1, 8 cycles) instead of a third 16-row tile: block
// 4 g + n / 4 of lane (g, n) is (k-slice g, element group n / 4), so the B fragment of the 16x16x4 instruction is the B operand
// as it is, lane (g, n) supplies A = D[r][32 + n % 4][4 jq + g] and receives in register v the k-slice-g part of
// out[e0 + n][32 + v]; the four parts are added across the lane groups at the end of the tile.
// kDbg (experiment build, $FEINSUM_F32_DBG): the tile work with parts removed -- 1 no MFMAs, 2 no stores, 4 no tile loads, 8 no
// fragment arithmetic (profiles/r04/float32_div_decomposition.txt)
template <int RING, bool SMALL, int kDbg = 0>
__global__ __launch_bounds__(256, RING == 1 ? 3 : 2) void div3d_mfma_f32_kernel(const float* __restrict__ J, const float* __restrict__ D,
                                                                                const float* __restrict__ u, float* __restrict__ out,
                                                                                int64_t E, int64_t nTiles, int opT) {
    using G = DivF32Geom<RING>;
    constexpr int NP = G::NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typename G::WaveIn* L = reinterpret_cast<typename G::WaveIn*>(smem) + wave;
    float* ob = reinterpret_cast<float*>(smem + G::IN_BYTES) + wave * G::PLANE_F;
    const int n = lane & 15, g = lane >> 4;
    const unsigned bid = blockIdx.x, nblk = gridDim.x;
    const int64_t stride = (int64_t)nblk * G::WAVES, tEnd = nTiles;
    int64_t tile = (int64_t)bid * G::WAVES + wave;
    const unsigned lds_s0 = lds_addr_uniform(&L->s[0]);

    auto issue_loads = [&](int64_t t, int slot) {
        if (kDbg & 4) return;
        const unsigned lds_u = lds_s0 + slot * (unsigned)sizeof(typename G::Slot), lds_j = lds_u + 3 * G::PLANE_F * 4;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const char* up = reinterpret_cast<const char*>(u + ((int64_t)x * E + t * G::TEL) * NP) + lane * 16;
#pragma unroll
            for (int c = 0; c < G::P_INSTR; ++c)
                if ((c + 1) * 64 <= G::P_CHUNKS || c * 64 + lane < G::P_CHUNKS)
                    glds16_nt(up + c * 1024, lds_u + x * (G::PLANE_F * 4) + c * 1024);
        }
        const int row = lane / G::J_ROW_CHUNKS, col = lane - row * G::J_ROW_CHUNKS;
        if (lane < G::J_CHUNKS) glds16(reinterpret_cast<const char*>(J + (int64_t)row * E + t * G::TEL) + col * 16, lds_j);
    };

    // ---- the first tile's loads (ring of two: the first two tiles'), and behind them the operator -> LDS (over the output
    //      buffers)
    if (tile < tEnd) issue_loads(tile, 0);
    if (RING == 2 && tile + stride < tEnd) issue_loads(tile + stride, 1);
    {
        float* dl = reinterpret_cast<float*>(smem + G::IN_BYTES);
        constexpr int kPer = (G::OP_F + 255) / 256;
        float tmp[kPer];
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int idx = threadIdx.x + k * 256;
            tmp[k] = idx < G::OP_F ? D[idx] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const int idx = threadIdx.x + k * 256;
            if (idx < G::OP_F) dl[idx] = tmp[k];
        }
    }
    __syncthreads();

    // ---- A fragments: lane (g, n) supplies A[row 16 t + n][k = g] of k-step (jq, r): D[r][16 t + n][4 jq + g]
    float afrag[G::RT][G::KS];
    {
        const float* dl = reinterpret_cast<const float*>(smem + G::IN_BYTES);
        const int istride = opT ? 1 : NP, jstride = opT ? NP : 1;   // opT: D stored as [r][j][i]
#pragma unroll
        for (int t = 0; t < G::RT; ++t) {
            const int i = (SMALL && t == 2) ? 32 + (n & 3) : 16 * t + n;
            const float* row = dl + (i < NP ? i : 0) * istride;
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq) {
                const int j = 4 * jq + g;
                const float* col = row + (j < NP ? j : 0) * jstride;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float val = col[r * (NP * NP)];
                    afrag[t][jq * 3 + r] = (i < NP && j < NP) ? val : 0.f;
                }
            }
        }
    }
    {   // the elements behind the last full tile, with the operator from the block's LDS copy (see fe_grad_f32.h)
        const float* dl = reinterpret_cast<const float*>(smem + G::IN_BYTES);
        remainder_items(nTiles * G::TEL, E, NP, bid, nblk, [&](int64_t e, int i) { div3d_item_f32(J, dl, u, out, E, NP, e, i, opT); });
    }
    __syncthreads();   // the staging area becomes the waves' output buffers

    const bool younger_half = bid >= (nblk + 1) / 2;
    int iteration = 0, slot = 0;
    while (tile < tEnd) {
        balance_priority(younger_half, iteration);
        // vector-memory ops in issue order -- one buffer: L(t) S(t-1) | wait L(t);  ring of two: L(t) S(t-2) L(t+1) S(t-1)
        if (RING == 1) {
            if (iteration == 0) wait_vmcnt<0>();
            else wait_vmcnt<G::STORES>();
        } else {
            if (iteration >= 2 && tile + stride < tEnd) wait_vmcnt<2 * G::STORES + G::LOADS>();
            else wait_vmcnt<0>();
        }
        ++iteration;
        const typename G::Slot* S = &L->s[slot];

        // ---- all B fragments of the tile
        float jac[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) jac[k] = S->j[k * G::TEL + n];
        float bfrag[G::KSJ][3];
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq) {
            const int j = 4 * jq + g, jc = j < NP ? j : 0;
            float ux[3];
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                const float v = S->u[x][n * NP + jc];
                ux[x] = j < NP ? v : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
                bfrag[jq][r] = (kDbg & 8) ? ux[r] : __builtin_fmaf(jac[6 + r], ux[2], __builtin_fmaf(jac[3 + r], ux[1], jac[r] * ux[0]));
        }
        // the u / J tiles are now in registers: hand the buffers back to the DMA engine
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
            for (int r = 0; r < 3; ++r) asm volatile("" : "+v"(bfrag[jq][r]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t nt = tile + stride;
        if (tile + RING * stride < tEnd) issue_loads(tile + RING * stride, slot);

        v4f acc[G::RT];
#pragma unroll
        for (int t = 0; t < G::RT; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < G::RT; ++t) {
                    if (kDbg & 1) { acc[t][(jq + r) & 3] += afrag[t][jq * 3 + r] + bfrag[jq][r]; continue; }
                    if (SMALL && t == 2) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(afrag[t][jq * 3 + r], bfrag[jq][r], acc[t], 0, 0, 0);
                    else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[t][jq * 3 + r], bfrag[jq][r], acc[t], 0, 0, 0);
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
        float* op = out + tile * (G::TEL * NP);
#pragma unroll
        for (int c = 0; c < G::P_INSTR; ++c) {
            const int q = c * 64 + lane;
            if ((c + 1) * 64 <= G::P_CHUNKS || q < G::P_CHUNKS) {
                const v4f val = *reinterpret_cast<const v4f*>(ob + 4 * q);
                if (kDbg & 2) { if (val[0] == 1.2345e-30f) op[4 * q] = val[1]; }   // keep the value live
                else __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(op + 4 * q));
            }
        }
        wave_lds_fence();
        tile = nt;
        if (RING == 2) slot ^= 1;
    }
}

// ---- the lower orders p = 1 ... 3 (Np = 4, 10, 20) on the matrix cores (round 5; they ran on the tiled VALU kernel in float before,
//      at 14-18 % of their rooflines: profiles/r04/float32_grad_orders.txt).  The same kernel as above, written over the geometry:
//      BT = Np / 16 sixteen-row tiles on v_mfma_f32_16x16x4_f32 and the NR = Np - 16 BT rows behind them in NS groups of four
//      on v_mfma_f32_4x4x1_16B_f32 (Np 20: 1 + 1 group; 10: 3 groups; 4: 1 group); M sixteen-element sub-tiles per wave tile, so
//      that a wave still moves a few KB per LDS-DMA batch (M = 1 / 3 / 5 for p = 3 / 2 / 1, as fe_div.h); ring of two tile buffers.
template <int NP_, int M_>
struct DivF32GeomT {
    static constexpr int NP = NP_, M = M_, TEL = 16 * M, BT = NP / 16, NR = NP - 16 * BT, NS = (NR + 3) / 4;
    static constexpr int KSJ = (NP + 3) / 4, KS = 3 * KSJ;
    static constexpr int PLANE_F = TEL * NP, P_CHUNKS = PLANE_F / 4, P_INSTR = (P_CHUNKS + 63) / 64;
    static constexpr int J_ROW_CHUNKS = TEL / 4, J_CHUNKS = 9 * J_ROW_CHUNKS, J_INSTR = (J_CHUNKS + 63) / 64;
    static constexpr int LOADS = 3 * P_INSTR + J_INSTR, STORES = P_INSTR;
    struct Slot {
        float u[3][PLANE_F];     // u[x][e0 .. e0+TEL-1][0..Np-1]
        float j[9 * TEL];        // J[x*3+r][e0 .. e0+TEL-1]
    };
    struct WaveIn {
        Slot s[2];
    };
    static constexpr int WAVES = 4;
    static constexpr int OP_F = 3 * NP * NP;
    static constexpr int IN_BYTES = (int)sizeof(WaveIn) * WAVES;
    static constexpr int OUT_BYTES = PLANE_F * 4 * WAVES;
    static constexpr int OP_BYTES = (OP_F * 4 + 15) / 16 * 16;
    static constexpr int LDS_BYTES = IN_BYTES + (OUT_BYTES > OP_BYTES ? OUT_BYTES : OP_BYTES);
    static constexpr int BLOCKS_PER_CU = 3 * LDS_BYTES <= 160 * 1024 ? 3 : 2;   // (Np = 20: three blocks of 40 KB; registers allow four waves per SIMD)
    static_assert(PLANE_F % 4 == 0 && BLOCKS_PER_CU * LDS_BYTES <= 160 * 1024, "geometry");
    static_assert(2 * STORES + LOADS <= 60, "counted vmcnt must fit the 6-bit field");
};

template <int NP_, int M_>
__global__ __launch_bounds__(256, 2) void div3d_mfma_f32_np_kernel(const float* __restrict__ J, const float* __restrict__ D,
                                                                   const float* __restrict__ u, float* __restrict__ out, int64_t E,
                                                                   int64_t nTiles, int opT) {
    using G = DivF32GeomT<NP_, M_>;
    constexpr int NP = G::NP, M = G::M;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typename G::WaveIn* L = reinterpret_cast<typename G::WaveIn*>(smem) + wave;
    float* ob = reinterpret_cast<float*>(smem + G::IN_BYTES) + wave * G::PLANE_F;
    const int n = lane & 15, g = lane >> 4;
    const unsigned bid = blockIdx.x, nblk = gridDim.x;
    const int64_t stride = (int64_t)nblk * G::WAVES, tEnd = nTiles;
    int64_t tile = (int64_t)bid * G::WAVES + wave;
    const unsigned lds_s0 = lds_addr_uniform(&L->s[0]);

    auto issue_loads = [&](int64_t t, int slot) {
        const unsigned lds_u = lds_s0 + slot * (unsigned)sizeof(typename G::Slot), lds_j = lds_u + 3 * G::PLANE_F * 4;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const char* up = reinterpret_cast<const char*>(u + ((int64_t)x * E + t * G::TEL) * NP) + lane * 16;
#pragma unroll
            for (int c = 0; c < G::P_INSTR; ++c)
                if ((c + 1) * 64 <= G::P_CHUNKS || c * 64 + lane < G::P_CHUNKS)
                    glds16_nt(up + c * 1024, lds_u + x * (G::PLANE_F * 4) + c * 1024);
        }
#pragma unroll
        for (int c = 0; c < G::J_INSTR; ++c) {
            const int q = c * 64 + lane;
            const int row = q / G::J_ROW_CHUNKS, col = q - row * G::J_ROW_CHUNKS;
            if ((c + 1) * 64 <= G::J_CHUNKS || q < G::J_CHUNKS)
                glds16(reinterpret_cast<const char*>(J + (int64_t)row * E + t * G::TEL) + col * 16, lds_j + c * 1024);
        }
    };

    // ---- the first two tiles' loads, and behind them the operator -> LDS (over the output buffers)
    if (tile < tEnd) issue_loads(tile, 0);
    if (tile + stride < tEnd) issue_loads(tile + stride, 1);
    {
        float* dl = reinterpret_cast<float*>(smem + G::IN_BYTES);
        for (int idx = threadIdx.x; idx < G::OP_F; idx += 256) dl[idx] = D[idx];
    }
    __syncthreads();

    // ---- A fragments.  16x16x4: lane (g, n) supplies A[row 16 t + n][k = g] of k-step (jq, r): D[r][16 t + n][4 jq + g];
    //      4x4x1 group q: D[r][16 BT + 4 q + n % 4][4 jq + g] (see SMALL above)
    float abig[G::BT > 0 ? G::BT : 1][G::KS], asmall[G::NS > 0 ? G::NS : 1][G::KS];
    {
        const float* dl = reinterpret_cast<const float*>(smem + G::IN_BYTES);
        const int istride = opT ? 1 : NP, jstride = opT ? NP : 1;   // opT: D stored as [r][j][i]
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq) {
            const int j = 4 * jq + g;
            const float* col = dl + (j < NP ? j : 0) * jstride;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int t = 0; t < G::BT; ++t) {
                    const float val = col[r * (NP * NP) + (16 * t + n) * istride];
                    abig[t][jq * 3 + r] = j < NP ? val : 0.f;
                }
#pragma unroll
                for (int q = 0; q < G::NS; ++q) {
                    const int i = 16 * G::BT + 4 * q + (n & 3);
                    const float val = col[r * (NP * NP) + (i < NP ? i : 0) * istride];
                    asmall[q][jq * 3 + r] = (i < NP && j < NP) ? val : 0.f;
                }
            }
        }
    }
    {   // the elements behind the last full tile, with the operator from the block's LDS copy
        const float* dl = reinterpret_cast<const float*>(smem + G::IN_BYTES);
        remainder_items(nTiles * G::TEL, E, NP, bid, nblk, [&](int64_t e, int i) { div3d_item_f32(J, dl, u, out, E, NP, e, i, opT); });
    }
    __syncthreads();   // the staging area becomes the waves' output buffers

    const bool younger_half = bid >= (nblk + 1) / 2;
    int iteration = 0, slot = 0;
    while (tile < tEnd) {
        balance_priority(younger_half, iteration);
        // vector-memory ops in issue order: L(t) S(t-2) L(t+1) S(t-1)
        if (iteration >= 2 && tile + stride < tEnd) wait_vmcnt<2 * G::STORES + G::LOADS>();
        else wait_vmcnt<0>();
        ++iteration;
        const typename G::Slot* S = &L->s[slot];

        // ---- all B fragments of the tile
        float bfrag[M][G::KSJ][3];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float jac[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) jac[k] = S->j[k * G::TEL + 16 * m + n];
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq) {
                const int j = 4 * jq + g, jc = j < NP ? j : 0;
                float ux[3];
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    const float v = S->u[x][(16 * m + n) * NP + jc];
                    ux[x] = j < NP ? v : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    bfrag[m][jq][r] = __builtin_fmaf(jac[6 + r], ux[2], __builtin_fmaf(jac[3 + r], ux[1], jac[r] * ux[0]));
            }
        }
        // the u / J tiles are now in registers: hand the buffers back to the DMA engine
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                for (int r = 0; r < 3; ++r) asm volatile("" : "+v"(bfrag[m][jq][r]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int64_t nt = tile + stride;
        if (tile + 2 * stride < tEnd) issue_loads(tile + 2 * stride, slot);

#pragma unroll
        for (int m = 0; m < M; ++m) {
            v4f accb[G::BT > 0 ? G::BT : 1], accq[G::NS > 0 ? G::NS : 1];
#pragma unroll
            for (int t = 0; t < G::BT; ++t) accb[t] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < G::NS; ++q) accq[q] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int t = 0; t < G::BT; ++t)
                        accb[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(abig[t][jq * 3 + r], bfrag[m][jq][r], accb[t], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < G::NS; ++q)
                        accq[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(asmall[q][jq * 3 + r], bfrag[m][jq][r], accq[q], 0, 0, 0);
                }
            // ---- into the transposition buffer: lane (g, n) holds out[e0 + 16 m + n][16 t + 4 g + v]; a 4x4x1 group holds the
            //      k-slice-g part of out[..][16 BT + 4 q + v]: the four parts are added across the lane groups
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
        float* op = out + tile * (G::TEL * NP);
#pragma unroll
        for (int c = 0; c < G::P_INSTR; ++c) {
            const int q = c * 64 + lane;
            if ((c + 1) * 64 <= G::P_CHUNKS || q < G::P_CHUNKS) {
                const v4f val = *reinterpret_cast<const v4f*>(ob + 4 * q);
                __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(op + 4 * q));
            }
        }
        wave_lds_fence();
        tile = nt;
        slot ^= 1;
    }
}

}  // namespace fe
