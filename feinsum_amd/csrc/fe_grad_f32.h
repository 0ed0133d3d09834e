// fe_grad_f32.h -- grad einsum 'xre,rij,ej->xei' in float32 on the matrix cores (tetrahedra p = 4, Np = 35).
//
// The float32 counterpart of fe_grad.h's kernel, one wave = one tile of 16 elements, same data movement (LDS-DMA loads one
// tile ahead with counted vmcnt, outputs transposed through wave-private LDS into 1-KiB contiguous non-temporal stores):
//   stage 1  tmp[(r,i), e] = sum_j D[(r,i), j] * u[e, j]   on v_mfma_f32_16x16x4_f32, A = D zero padded to 112 x 36 with
//            permuted rows, resident in registers (63 floats per lane); 7 x 9 = 63 MFMAs of 32 cycles per tile;
//   stage 2  out[x,e,i] = sum_r J[x,r,e] * tmp[(r,i), e]   on the VALU, lane-local.
// The float32 C/D layout differs from the float64 one: lane (g = lane >> 4, n = lane & 15) holds rows 4 g + v (v = 0..3)
// of every 16-row tile for column n (float64: rows g + 4 v).  Slot s = 4 tile + v of lane group g is again
// (r, i) = (s % 3, 9 g + s / 3), so row 4 g + v of tile t of A holds D[(4 t + v) % 3][9 g + (4 t + v) / 3][.]: every lane
// owns all three r of nine consecutive i of one element and the Jacobian combine needs no cross-lane traffic.
// Half the bytes of float64 per element (596 + 14 700 once), so the HBM roofline in GFLOP/s doubles: 107 TFLOP/s.
// The reference validates float32 einsums at 1e-6 (src/feinsum/measure.py:178-192).
// Operands must be 16-byte aligned with E a multiple of 4 (every row and plane then starts on a 16-byte boundary; the
// launcher sends other sizes to the tiled kernel); the elements behind the last full tile: remainder_items.
#pragma once
#include "fe_common.h"
#include "fe_grad.h"   // grad_row_tiles

namespace fe {

typedef float v4f __attribute__((ext_vector_type(4)));

// M = 16-element sub-tiles per wave iteration.  M = 1 (round 3): 2.2 KB spans per plane and tile, three blocks per CU.  M = 2
// (round 4): a wave tile is 32 elements, so that a plane leaves as one 4.5 KB burst -- the float64 kernel's span -- through an
// output buffer that holds both sub-tiles; 20 KB of LDS per wave, two blocks per CU.
// NP (round 4): the tetrahedral orders p = 1 ... 4 (Np = 4, 10, 20, 35) as in fe_grad.h -- RT row tiles such that the four
// lane groups hold TG = 4 RT / 3 whole r-triples with 4 TG >= Np; the lower orders take more sub-tiles per wave iteration
// (a wave tile of a few KB).
template <int M_ = 2, int NP_ = 35>
struct GradF32GeomT {
    static constexpr int M = M_;
    static constexpr int NP = NP_, TEL = 16 * M, RT = grad_row_tiles(NP_), TG = (4 * RT) / 3, KS = (NP_ + 3) / 4;
    static constexpr int TILE_F = TEL * NP;             // floats: u tile / one out plane of a tile (560 M)
    static constexpr int U_CHUNKS = TILE_F / 4;         // 16-byte chunks (140 M)
    static constexpr int U_INSTR = (U_CHUNKS + 63) / 64;            // 3 / 5
    static constexpr int J_ROW_CHUNKS = TEL / 4;        // 4 M
    static constexpr int J_CHUNKS = 9 * J_ROW_CHUNKS;   // 36 M
    static constexpr int J_INSTR = (J_CHUNKS + 63) / 64;            // 1 / 2
    static constexpr int LOADS = U_INSTR + J_INSTR;
    static constexpr int PLANE_STORES = U_INSTR;        // 16-byte chunks of one plane of a tile
    static constexpr int STORES = 3 * PLANE_STORES;
    struct WaveLds {
        float u[2][TILE_F];      // prefetch double buffer
        float j[2][9 * TEL];     // J[x*3+r][e0 .. e0+TEL-1], double buffered
    };
    struct WaveOut {
        float o[2][TILE_F];      // output transposition buffers (a whole plane of the tile), alternating
    };
    static constexpr int WAVES = 4;
    static constexpr int OP_F = 3 * NP * NP;            // 3675 floats
    static constexpr int IN_BYTES = (int)sizeof(WaveLds) * WAVES;
    static constexpr int OUT_BYTES = (int)sizeof(WaveOut) * WAVES;
    static constexpr int OP_BYTES = (OP_F * 4 + 15) / 16 * 16;
    static constexpr int LDS_BYTES = IN_BYTES + (OUT_BYTES > OP_BYTES ? OUT_BYTES : OP_BYTES);
    static constexpr int BLOCKS_PER_CU = (M == 1 && NP == 35) ? 3 : 2;   // M = 1: 40 KB of LDS per block of four waves; four blocks (128 VGPRs, 12 B of
                                                           // scratch) ran 6 % slower: profiles/r03/float32_grad.txt
    static_assert(BLOCKS_PER_CU * LDS_BYTES <= 160 * 1024, "blocks per CU");
    static_assert(LOADS + STORES <= 60, "counted vmcnt must fit the 6-bit field");
};
using GradF32Geom = GradF32GeomT<1>;

__device__ __forceinline__ void grad3d_item_f32(const float* __restrict__ J, const float* __restrict__ D,
                                                const float* __restrict__ u, float* __restrict__ out, int64_t E, int Np,
                                                int64_t e, int i, int opT) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    const float* ue = u + e * Np;
    const int si = opT ? 1 : Np, sj = opT ? Np : 1;
    const float* d0 = D + (int64_t)0 * Np * Np + (int64_t)i * si;
    const float* d1 = D + (int64_t)1 * Np * Np + (int64_t)i * si;
    const float* d2 = D + (int64_t)2 * Np * Np + (int64_t)i * si;
    for (int j = 0; j < Np; ++j) {
        const float uj = ue[j];
        t0 = __builtin_fmaf(d0[j * sj], uj, t0);
        t1 = __builtin_fmaf(d1[j * sj], uj, t1);
        t2 = __builtin_fmaf(d2[j * sj], uj, t2);
    }
    for (int x = 0; x < 3; ++x)
        out[((int64_t)x * E + e) * Np + i] =
            __builtin_fmaf(J[(int64_t)(x * 3 + 2) * E + e], t2,
                           __builtin_fmaf(J[(int64_t)(x * 3 + 1) * E + e], t1, J[(int64_t)(x * 3 + 0) * E + e] * t0));
}

// kDyn: behind two static rounds the tiles come by tickets (fe_common.h: dynamic walk), as in fe_grad.h
template <int M, int NP_, bool kDyn>
__device__ __forceinline__ void grad3d_mfma_f32_body(const float* __restrict__ J, const float* __restrict__ D,
                                                     const float* __restrict__ u, float* __restrict__ out, int64_t E,
                                                     int64_t nTiles, int op_flags, unsigned* __restrict__ tail = nullptr,
                                                     int64_t t_static = 0) {
    using G = GradF32GeomT<M, NP_>;
    const int opT = op_flags & 1;
    const bool tload = (op_flags & kOpLoadsTemporal) != 0;   // the launch's inputs fit the Infinity Cache (fe_common.h)
    constexpr int NP = G::NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    typename G::WaveLds* L = reinterpret_cast<typename G::WaveLds*>(smem) + wave;
    typename G::WaveOut* LO = reinterpret_cast<typename G::WaveOut*>(smem + G::IN_BYTES) + wave;
    const int n = lane & 15, g = lane >> 4;
    const unsigned bid = blockIdx.x, nblk = gridDim.x;
    const int64_t stride = (int64_t)nblk * G::WAVES, tEnd = nTiles;
    int64_t tile = (int64_t)bid * G::WAVES + wave;

    auto issue_u = [&](int64_t t, unsigned lds_u) {
        const char* ub = reinterpret_cast<const char*>(u) + t * (G::TILE_F * 4) + lane * 16;
#pragma unroll
        for (int c = 0; c < G::U_INSTR; ++c)
            if ((c + 1) * 64 <= G::U_CHUNKS || c * 64 + lane < G::U_CHUNKS) {
                if (tload) glds16(ub + c * 1024, lds_u + c * 1024);
                else glds16_nt(ub + c * 1024, lds_u + c * 1024);
            }
    };
    auto issue_j = [&](int64_t t, unsigned lds_j) {
#pragma unroll
        for (int c = 0; c < G::J_INSTR; ++c) {
            const int q = c * 64 + lane;
            const int row = q / G::J_ROW_CHUNKS, col = q - row * G::J_ROW_CHUNKS;   // chunk -> (row x*3 + r, column chunk)
            const char* src = reinterpret_cast<const char*>(J + (int64_t)row * E + t * G::TEL) + col * 16;
            if ((c + 1) * 64 <= G::J_CHUNKS || q < G::J_CHUNKS) glds16(src, lds_j + c * 1024);
        }
    };

    // ---- the loads of this wave's first two tiles (LDS-DMA), and behind them the operator -> LDS (over the output
    //      buffers, which nobody needs before the first tile's stage 2): all its loads are issued before the first LDS
    //      write, so that their latencies overlap each other and the tile loads
    bool pre = false;
    const int64_t tPre = (kDyn && tail) ? t_static : tEnd;   // the prologue prefetches a second tile only if it is a static one
    if (tile < tEnd) {
        issue_u(tile, lds_addr_uniform(L->u[0]));
        issue_j(tile, lds_addr_uniform(L->j[0]));
        if (tile + stride < tPre) {
            issue_u(tile + stride, lds_addr_uniform(L->u[1]));
            issue_j(tile + stride, lds_addr_uniform(L->j[1]));
            pre = true;
        }
    }
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

    // ---- A fragments: lane (g, n) supplies A[row n of tile t][k = 4 ks + g]; row n = 4 gp + v  ->  slot s = 4 t + v of
    //      lane group gp  ->  (r, i) = (s % 3, 9 gp + s / 3)
    float afrag[G::RT][G::KS];
    {
        const float* dl = reinterpret_cast<const float*>(smem + G::IN_BYTES);
        const int gp = n >> 2, v = n & 3;
        const int istride = opT ? 1 : NP, jstride = opT ? NP : 1;   // opT: D stored as [r][j][i]
#pragma unroll
        for (int t = 0; t < G::RT; ++t) {
            const int s = 4 * t + v;
            const int r = s % 3, i = G::TG * gp + s / 3;
            const bool rowok = (s < 3 * G::TG) && (i < NP);
            const float* row = dl + r * (NP * NP) + (i < NP ? i : 0) * istride;
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const int j = 4 * ks + g;
                const float val = row[(j < NP ? j : 0) * jstride];
                afrag[t][ks] = (rowok && j < NP) ? val : 0.f;
            }
        }
    }
    // the elements behind the last full tile: plain code, one entry per thread on a few blocks (fe_common.h: remainder_items), while the
    // block's LDS copy of the operator is still there (from global memory each of an entry's operator loads touches 35 cache lines)
    {
        const float* dl = reinterpret_cast<const float*>(smem + G::IN_BYTES);
        remainder_items(nTiles * G::TEL, E, NP, bid, nblk, [&](int64_t e, int i) { grad3d_item_f32(J, dl, u, out, E, NP, e, i, opT); });
    }
    __syncthreads();   // the staging area becomes the waves' output buffers

    // one tile: stage 1, stage 2 and the transposed stores, from the u tile `ut` and the J tile `jt` in LDS
    auto compute_tile = [&](int64_t tile_, const float* ut, const float* jt) {
        // ---- stage 1, sub-tile by sub-tile
        v4f acc[M][G::RT];
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float bfrag[G::KS];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                const int j = 4 * ks + g;
                const float b = ut[(16 * m + n) * NP + (j < NP ? j : 0)];
                bfrag[ks] = (j < NP) ? b : 0.f;
            }
#pragma unroll
            for (int t = 0; t < G::RT; ++t) acc[m][t] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
                for (int t = 0; t < G::RT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[t][ks], bfrag[ks], acc[m][t], 0, 0, 0);
        }

        // ---- stage 2 + transposed store, plane by plane (both sub-tiles of a plane leave together)
        const int64_t e0 = tile_ * G::TEL;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            float* ob = LO->o[x & 1];
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const float j0 = jt[(x * 3 + 0) * G::TEL + 16 * m + n];
                const float j1 = jt[(x * 3 + 1) * G::TEL + 16 * m + n];
                const float j2 = jt[(x * 3 + 2) * G::TEL + 16 * m + n];
#pragma unroll
                for (int k = 0; k < G::TG; ++k) {
                    const int s = 3 * k;
                    const float t0 = acc[m][(s + 0) >> 2][(s + 0) & 3];
                    const float t1 = acc[m][(s + 1) >> 2][(s + 1) & 3];
                    const float t2 = acc[m][(s + 2) >> 2][(s + 2) & 3];
                    const float val = __builtin_fmaf(j2, t2, __builtin_fmaf(j1, t1, j0 * t0));
                    const int i = G::TG * g + k;
                    if (G::TG * 3 + k < NP || i < NP) ob[(16 * m + n) * NP + i] = val;
                }
            }
            wave_lds_fence();
            float* op = out + ((int64_t)x * E + e0) * NP;
#pragma unroll
            for (int c = 0; c < G::PLANE_STORES; ++c) {
                const int q = c * 64 + lane;
                if ((c + 1) * 64 <= G::U_CHUNKS || q < G::U_CHUNKS) {
                    const v4f val = *reinterpret_cast<const v4f*>(ob + 4 * q);
                    __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(op + 4 * q));
                }
            }
            wave_lds_fence();
        }
    };

    const bool younger_half = bid >= (nblk + 1) / 2;
    if constexpr (kDyn) {
        if (tail) {
            // ---- walk with a dynamic tail: static tiles first + k stride below t_static, then tickets.  Vector-memory ops of
            //      an iteration in issue order: [ticket for the tile after next] L(next) S(cur); every wave has a static first
            //      tile (t_static >= number of waves).  The bookkeeping is fe_grad.h's.
            constexpr int NL = G::LOADS, NS = G::STORES;
            const int pool = (bid >> 3) & (kTailPools - 1);
            unsigned* const counter = tail_pool_counters(tail, pool);
            unsigned* const done = tail_pool_reports(counter);
            const unsigned pool_blocks = (nblk / (8 * kTailPools)) * 8 +
                                         (unsigned)max(0, min(8, (int)(nblk % (8 * kTailPools)) - 8 * pool));
            const unsigned pool_waves = pool_blocks * G::WAVES;
            auto static_next = [&](int64_t t) -> int64_t { return (t < t_static && t + stride < t_static) ? t + stride : -1; };
            int64_t cur = tile < tEnd ? tile : -1, nxt = cur >= 0 ? static_next(cur) : -1;   // nxt >= 0: loaded by the prologue (pre)
            bool pending = false, reported = false, prev_pre = false, first = true;
            if (cur >= 0 && nxt < 0) {   // one static round: the prologue's ticket, behind L(cur)
                tail_request<0>(counter);
                pending = true;
            }
            int buf = 0, iteration = 0;
            while (cur >= 0) {
                balance_priority(younger_half, iteration++);
                bool extra = false;   // one more vector-memory op (ticket or report) issued in this iteration
                if (pending) {   // the next tile comes from a ticket: younger than it are L(cur) and S(previous)
                    const unsigned t = first ? tail_wait<0, 0>() : prev_pre ? tail_wait<NS, 0>() : tail_wait<NL + NS, 0>();
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
                        issue_u(nxt, lds_addr_uniform(L->u[buf ^ 1]));
                        issue_j(nxt, lds_addr_uniform(L->j[buf ^ 1]));
                    }
                }
                // wait L(cur): younger are S(previous), the ticket / report, L(next)
                if (nxt >= 0) {
                    if (first) { if (extra) wait_vmcnt<NL + 1>(); else wait_vmcnt<NL>(); }
                    else { if (extra) wait_vmcnt<NS + NL + 1>(); else wait_vmcnt<NS + NL>(); }
                } else {
                    if (first) { if (extra) wait_vmcnt<1>(); else wait_vmcnt<0>(); }
                    else { if (extra) wait_vmcnt<NS + 1>(); else wait_vmcnt<NS>(); }
                }
                compute_tile(cur, L->u[buf], L->j[buf]);
                first = false;
                prev_pre = pre;
                pre = false;
                cur = nxt;
                buf ^= 1;
                if (cur >= 0 && !pending) nxt = static_next(cur);
            }
            // the last wave of a pool to report leaves the pool's two counters zeroed for the next launch
            if (reported) {
                const unsigned before = tail_wait<NS, 1>();
                if (before + 1 == pool_waves && lane == 0) {
                    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
    }

    int buf = 0;
    bool first = true;
    int iteration = 0;
    while (tile < tEnd) {
        balance_priority(younger_half, iteration++);
        // vector-memory ops in issue order: L(tile) S(previous tile) L(next tile) | wait L(tile)
        const int64_t nt = tile + stride;
        if (nt < tEnd) {
            if (!pre) {
                issue_u(nt, lds_addr_uniform(L->u[buf ^ 1]));
                issue_j(nt, lds_addr_uniform(L->j[buf ^ 1]));
            }
            if (first) wait_vmcnt<G::LOADS>();
            else wait_vmcnt<G::LOADS + G::STORES>();
        } else {
            if (first) wait_vmcnt<0>();
            else wait_vmcnt<G::STORES>();
        }
        first = false;
        pre = false;
        compute_tile(tile, L->u[buf], L->j[buf]);
        tile = nt;
        buf ^= 1;
    }
}

template <int M = 1, int NP_ = 35>
__global__ __launch_bounds__(256, (M == 1 && NP_ == 35) ? 3 : 2) void grad3d_mfma_f32_kernel(const float* __restrict__ J, const float* __restrict__ D,
                                                                             const float* __restrict__ u, float* __restrict__ out,
                                                                             int64_t E, int64_t nTiles, int op_flags) {
    grad3d_mfma_f32_body<M, NP_, false>(J, D, u, out, E, nTiles, op_flags);
}

// the same with a dynamic tail (two blocks per CU: the ticket registers v254 / v255 make the descriptor allocate 256 VGPRs)
template <int M, int NP_ = 35>
__global__ __launch_bounds__(256, 2) FE_TAIL_KERNEL_ATTR void grad3d_mfma_f32_tail_kernel(
    const float* __restrict__ J, const float* __restrict__ D, const float* __restrict__ u, float* __restrict__ out, int64_t E,
    int64_t nTiles, int op_flags, unsigned* __restrict__ tail, int64_t t_static) {
    grad3d_mfma_f32_body<M, NP_, true>(J, D, u, out, E, nTiles, op_flags, tail, t_static);
}

}  // namespace fe
