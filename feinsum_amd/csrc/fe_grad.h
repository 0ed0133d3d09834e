// fe_grad.h -- grad einsum  out[x,e,i] = sum_{r,j} J[x,r,e] D[r,i,j] u[e,j]
// ('xre,rij,ej->xei', reference: test/test_codegen.py:96-113; the device kernel
// this replaces is the loopy-generated one described in
// tuning/impls/xre_rij_ej_to_xei.py:26-275 / doc/compiler_writer_tutorial.rst:99-355).
//
// MFMA kernel (Np = 35, fp64), per WAVE and per tile of 16 elements:
//   stage 1  tmp[(r,i), e] = sum_j D[(r,i), j] * u[e, j]     on v_mfma_f64_16x16x4_f64
//            A = D as a 112 x 36 matrix (105 x 35 zero padded, rows permuted),
//            resident in registers for the whole kernel (63 doubles / lane);
//            B = u tile (36 x 16) read from LDS; 7 x 9 = 63 MFMAs per tile.
//   stage 2  out[x,e,i] = sum_r J[x,r,e] * tmp[(r,i), e]      on VALU, lane-local.
// The row permutation of A makes stage 2 lane-local: in the f64 16x16x4 C/D
// layout lane (g = lane>>4, n = lane&15) holds rows {g + 4q} of every 16-row
// tile for column n, i.e. 28 "slots" s = 4*tile + q.  Slot s of lane-group g is
// assigned (r, i) = (s % 3, 9g + s/3): every lane owns all three r's of nine
// consecutive i's of one element, so the 3x3 Jacobian combine needs no
// cross-lane traffic.  (Slot 27 and i == 35 are zero padding: 105/112 = 94 %
// useful rows, 35/36 useful k.)
// Data movement: the u tile (16 x 35 doubles = 4480 contiguous, 16-byte aligned
// bytes) and the 9 x 16 Jacobian entries come in by LDS-DMA one tile ahead;
// results are transposed through a wave-private LDS buffer so that every
// global store instruction writes 1 KiB of contiguous memory (out[x, e0:e0+16, :]
// is one contiguous 4480-byte span).  Waves never synchronise with each other.
#pragma once
#include "fe_common.h"

namespace fe {

constexpr int kNp35 = 35;
constexpr int kTE = 16;                    // elements per wave tile (MFMA N)
constexpr int kGradRowTiles = 7;           // 105 -> 112 rows
constexpr int kGradKSteps = 9;             // 35 -> 36
constexpr int kTileD35 = kTE * kNp35;      // 560 doubles
constexpr int kTileB35 = kTileD35 * 8;     // 4480 bytes

struct GradWaveLds {
    double u[2][kTileD35];     // prefetch double buffer for the u tile
    double o[2][kTileD35];     // output transposition buffers (alternate per x)
    double j[2][9 * kTE];      // J[x*3+r][e0 + 0..15], double buffered
};
static_assert(sizeof(GradWaveLds) == 20224, "LDS budget");
constexpr int kGradWavesPerBlock = 4;
constexpr int kGradLdsBytes = sizeof(GradWaveLds) * kGradWavesPerBlock;  // 80896: 2 blocks / CU

// Issues exactly 10 vector-memory instructions (5 x 16-byte + 5 x 4-byte
// LDS-DMA) for one FULL tile (the kernel only ever sees full tiles; the host
// sends the E % 16 remainder to the generic kernel), so the counted vmcnt in
// the main loop is always right.
template <bool kNT = false>
__device__ __forceinline__ void grad_issue_tile_loads(const double* __restrict__ J,
                                                      const double* __restrict__ u,
                                                      int64_t E, int64_t tile, int lane,
                                                      unsigned lds_u, unsigned lds_j) {
    const int64_t e0 = tile * kTE;
    const char* ub = reinterpret_cast<const char*>(u) + e0 * (kNp35 * 8) + lane * 16;
#pragma unroll
    for (int c = 0; c < 5; ++c)
        if (c < 4 || lane < 24) {
            if (kNT) glds16_nt(ub + c * 1024, lds_u + c * 1024);
            else glds16(ub + c * 1024, lds_u + c * 1024);
        }
    const int w = lane & 31;               // dword inside a 128-byte row
    const char* jb = reinterpret_cast<const char*>(J) + e0 * 8 + w * 4;
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int row = 2 * p + (lane >> 5);   // x*3 + r
        if (p < 4 || lane < 32) glds4(jb + (int64_t)row * E * 8, lds_j + p * 256);
    }
}
constexpr int kGradLoadsPerTile = 10;
constexpr int kGradStoresPerTile = 15;   // 3 planes x 5 x 16-byte stores

#ifdef FE_EXPERIMENTS
// Diagnostic build only (kDbg & 32); never read by any kernel, fetched by fe_dbg_read_*():
// shader cycles / 100 MHz ticks of wave 0's main loop, and per-wave 100 MHz timestamps
// {kernel entry, main-loop start, main-loop end, XCC_ID | HW_ID << 8}.
__device__ unsigned long long fe_dbg_clock[2];
__device__ unsigned long long fe_dbg_stamps[4096][4];
#endif

// kDbg: experiment flags, 0 in the product build (tools/fe_check.cpp "ab" mode uses the others
// through build/libfeinsum_hip_exp.so): 1 skip MFMAs, 2 skip stores, 4 plain (temporal) stores,
// 8 skip loads, 16 plain (temporal) loads, 32 per-wave timestamps, 64 no priority balancing.
// Product: non-temporal on both sides -- every byte is touched once (A/B on MI355X: -2.5 %
// kernel time, -7 % for the data-movement skeleton).
template <int kDbg = 0>
__global__ __launch_bounds__(256, 2) void grad3d_np35_mfma_kernel(
    const double* __restrict__ J, const double* __restrict__ D, const double* __restrict__ u,
    double* __restrict__ out, int64_t E, int64_t nTiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef FE_EXPERIMENTS
    const unsigned long long t_entry = (kDbg & 32) ? __builtin_amdgcn_s_memrealtime() : 0;
#endif
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    GradWaveLds* L = reinterpret_cast<GradWaveLds*>(smem) + wave;
    const int n = lane & 15, g = lane >> 4;

    // ---- A fragments: lane (g, n) supplies A[row 16t + n][k = 4ks + g]; D goes through LDS
    //      once per block (see stage_operator)
    double afrag[kGradRowTiles][kGradKSteps];
    {
        double* dl = reinterpret_cast<double*>(smem);
        stage_operator<3 * kNp35 * kNp35>(D, dl);
        __syncthreads();
        const int gp = n & 3, q = n >> 2;  // C/D lane group / register this row lands in
#pragma unroll
        for (int t = 0; t < kGradRowTiles; ++t) {
            const int s = 4 * t + q;
            const int r = s % 3, i = 9 * gp + s / 3;
#pragma unroll
            for (int ks = 0; ks < kGradKSteps; ++ks) {
                const int j = 4 * ks + g;
                const bool ok = (s < 27) && (i < kNp35) && (j < kNp35);
                afrag[t][ks] = ok ? dl[(r * kNp35 + i) * kNp35 + j] : 0.0;
            }
        }
        __syncthreads();   // the staging area is reused as the waves' private buffers from here on
    }

    const int64_t stride = (int64_t)gridDim.x * kGradWavesPerBlock;
    int64_t tile = (int64_t)blockIdx.x * kGradWavesPerBlock + wave;
    int buf = 0;
    bool first = true;
    if (tile < nTiles && !(kDbg & 8))
        grad_issue_tile_loads<(kDbg & 16) == 0>(J, u, E, tile, lane, lds_addr_uniform(L->u[0]), lds_addr_uniform(L->j[0]));

#ifdef FE_EXPERIMENTS
    unsigned long long c0 = 0, r0 = 0;
    if (kDbg & 32) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    const bool younger_half = !(kDbg & 64) && blockIdx.x >= (gridDim.x + 1) / 2;
    int iteration = 0;
    for (; tile < nTiles; tile += stride, buf ^= 1) {
        balance_priority(younger_half, iteration++);
        // Vector-memory ops in issue order: L(t) S(t-1) L(t+1) | wait L(t).  The
        // 15 stores of the previous tile and the 10 loads of the next one are
        // younger than this tile's loads and stay in flight.
        const int64_t nxt = tile + stride;
        if (kDbg & 8) {
            wait_vmcnt<0>();
        } else if (nxt < nTiles) {
            grad_issue_tile_loads<(kDbg & 16) == 0>(J, u, E, nxt, lane, lds_addr_uniform(L->u[buf ^ 1]),
                                  lds_addr_uniform(L->j[buf ^ 1]));
            if ((kDbg & 2) || first) wait_vmcnt<kGradLoadsPerTile>();
            else wait_vmcnt<kGradLoadsPerTile + kGradStoresPerTile>();
        } else {
            if (first || (kDbg & 2)) wait_vmcnt<0>();
            else wait_vmcnt<kGradStoresPerTile>();
        }
        first = false;

        // ---- stage 1: 63 MFMAs
        const double* ut = L->u[buf];
        double bfrag[kGradKSteps];
#pragma unroll
        for (int ks = 0; ks < kGradKSteps; ++ks) {
            const int j = 4 * ks + g;
            double b = ut[n * kNp35 + (j < kNp35 ? j : 0)];
            bfrag[ks] = (j < kNp35) ? b : 0.0;
        }
        v4d acc[kGradRowTiles];
#pragma unroll
        for (int t = 0; t < kGradRowTiles; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
        if (kDbg & 1) {
#pragma unroll
            for (int t = 0; t < kGradRowTiles; ++t)
                acc[t] = v4d{bfrag[t], bfrag[t + 1], bfrag[t + 2], afrag[t][0]};
        } else {
#pragma unroll
            for (int ks = 0; ks < kGradKSteps; ++ks)
#pragma unroll
                for (int t = 0; t < kGradRowTiles; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[t][ks], bfrag[ks], acc[t], 0, 0, 0);
        }

        // ---- stage 2 + transposed store
        const double* jt = L->j[buf];
        const int64_t e0 = tile * kTE;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            double* ob = L->o[x & 1];
            const double j0 = jt[(x * 3 + 0) * kTE + n];
            const double j1 = jt[(x * 3 + 1) * kTE + n];
            const double j2 = jt[(x * 3 + 2) * kTE + n];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int s = 3 * k;
                const double t0 = acc[(s + 0) >> 2][(s + 0) & 3];
                const double t1 = acc[(s + 1) >> 2][(s + 1) & 3];
                const double t2 = acc[(s + 2) >> 2][(s + 2) & 3];
                const double v = j0 * t0 + j1 * t1 + j2 * t2;
                if (k < 8 || g < 3) ob[n * kNp35 + 9 * g + k] = v;
            }
            wave_lds_fence();
            double* op = out + ((int64_t)x * E + e0) * kNp35;
            // 16-byte accesses at 8-byte aligned global addresses (odd E: the x = 1
            // plane) are fine on gfx950 (tools/align_test.hip).
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                if (c < 4 || lane < 24) {
                    const int q = c * 64 + lane;
                    const v2d val = *reinterpret_cast<const v2d*>(ob + 2 * q);
                    if (kDbg & 2) { if (val[0] == 1.2345e-300) op[2 * q] = val[1]; }   // keep the value live
                    else if (kDbg & 4) *reinterpret_cast<v2d*>(op + 2 * q) = val;
                    else __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + 2 * q));
                }
            }
            wave_lds_fence();
        }
    }
#ifdef FE_EXPERIMENTS
    if ((kDbg & 32) && lane == 0) {
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        const int w = blockIdx.x * kGradWavesPerBlock + wave;
        if (w < 4096) {
            fe_dbg_stamps[w][0] = t_entry; fe_dbg_stamps[w][1] = r0; fe_dbg_stamps[w][2] = t_end;
            const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID[3:0]
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
            fe_dbg_stamps[w][3] = xcc | ((unsigned long long)hw << 8);
        }
        if (w == 0) { fe_dbg_clock[0] = __builtin_amdgcn_s_memtime() - c0; fe_dbg_clock[1] = t_end - r0; }
    }
#endif
}

// Plain VALU kernel, any Np: one thread per (e, i), elements [e_begin, E).  Correctness reference on
// the device and the path for shapes the MFMA kernel is not compiled for.
__global__ __launch_bounds__(256) void grad3d_generic_kernel(
    const double* __restrict__ J, const double* __restrict__ D, const double* __restrict__ u,
    double* __restrict__ out, int64_t E, int Np, int64_t e_begin) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (E - e_begin) * Np) return;
    const int64_t e = e_begin + idx / Np;
    const int i = (int)(idx % Np);
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    const double* ue = u + e * Np;
    const double* d0 = D + (int64_t)(0 * Np + i) * Np;
    const double* d1 = D + (int64_t)(1 * Np + i) * Np;
    const double* d2 = D + (int64_t)(2 * Np + i) * Np;
    for (int j = 0; j < Np; ++j) {
        const double uj = ue[j];
        t0 += d0[j] * uj;
        t1 += d1[j] * uj;
        t2 += d2[j] * uj;
    }
    for (int x = 0; x < 3; ++x)
        out[((int64_t)x * E + e) * Np + i] =
            J[(int64_t)(x * 3 + 0) * E + e] * t0 + J[(int64_t)(x * 3 + 1) * E + e] * t1 +
            J[(int64_t)(x * 3 + 2) * E + e] * t2;
}

}  // namespace fe
