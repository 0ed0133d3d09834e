// fe_div.h -- div einsum  out[e,i] = sum_{x,r,j} J[x,r,e] D[r,i,j] u[x,e,j]
// ('xre,rij,xej->ei'; the device kernel this replaces is the loopy-generated
// one of tuning/impls/xre_rij_xej_to_ei.py:26-249, target pseudo-C in
// tuning/impls/xre_rij_xej_to_ei_v6.py:41-110).
//
// Schedule = the opt_einsum-optimal one: first the cheap
// Jacobian contraction  Ju[r,e,j] = sum_x J[x,r,e] u[x,e,j]  (VALU, 3 FMAs per
// value, produced directly in MFMA B-fragment layout), then the dense one
//   out[i, e] = sum_{(r,j)} D'[i, (r,j)] * Ju[(r,j), e]
// on v_mfma_f64_16x16x4_f64 with A = D' (35 x 105 zero padded to 48 x 108)
// resident in registers (81 doubles / lane) and 3 x 27 = 81 MFMAs per tile of
// 16 elements per wave.  K is ordered (jq, r) with j = 4 jq + g so that one
// group of three u values (x = 0..2) feeds three consecutive k-steps.
// Data movement as in fe_grad.h: LDS-DMA in (3 u planes + J), LDS transpose
// out, 1-KiB contiguous stores, waves fully independent.  The three u planes
// (13.4 KB per wave) do not leave LDS room for a second buffer at 8 waves/CU,
// so the loads of tile t+1 are issued right after tile t's last MFMA (its
// LDS reads are done by then) and overlap only tile t's epilogue; the other
// wave of the SIMD covers the rest of the latency.
#pragma once
#include "fe_grad.h"

namespace fe {

constexpr int kDivRowTiles = 3;   // 35 -> 48 rows
constexpr int kDivJq = 9;         // 35 -> 36 j's, 4 per k-step

struct DivWaveLds {
    double u[3][kTileD35];    // u[x][e0..e0+15][0..34]
    double o[kTileD35];       // output transposition buffer
    double j[9 * kTE];        // J[x*3+r][e0 + 0..15]
};
static_assert(sizeof(DivWaveLds) == 19072, "LDS budget");
constexpr int kDivWavesPerBlock = 4;
constexpr int kDivLdsBytes = sizeof(DivWaveLds) * kDivWavesPerBlock;  // 76288: 2 blocks / CU

__global__ __launch_bounds__(256, 2) void div3d_np35_mfma_kernel(
    const double* __restrict__ J, const double* __restrict__ D, const double* __restrict__ u,
    double* __restrict__ out, int64_t E, int64_t nTiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    DivWaveLds* L = reinterpret_cast<DivWaveLds*>(smem) + wave;
    const int n = lane & 15, g = lane >> 4;

    // ---- A fragments: lane (g, n) supplies A[row i = 16t + n][k = (jq, r)], j = 4 jq + g
    double afrag[kDivRowTiles][kDivJq][3];
#pragma unroll
    for (int t = 0; t < kDivRowTiles; ++t) {
        const int i = 16 * t + n;
#pragma unroll
        for (int jq = 0; jq < kDivJq; ++jq) {
            const int j = 4 * jq + g;
#pragma unroll
            for (int r = 0; r < 3; ++r)
                afrag[t][jq][r] = (i < kNp35 && j < kNp35) ? D[(r * kNp35 + i) * kNp35 + j] : 0.0;
        }
    }

    const unsigned lds_u = lds_addr_uniform(L->u[0]);
    const unsigned lds_j = lds_addr_uniform(L->j);
    const int64_t stride = (int64_t)gridDim.x * kDivWavesPerBlock;
    // 3 planes x 5 x 16-byte LDS-DMA + 5 x 4-byte LDS-DMA for J
    auto issue_loads = [&](int64_t tile) {
        const int64_t e0 = tile * kTE;
        const char* ub = reinterpret_cast<const char*>(u) + e0 * (kNp35 * 8) + lane * 16;
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const char* up = ub + (int64_t)x * E * (kNp35 * 8);
#pragma unroll
            for (int c = 0; c < 5; ++c)
                if (c < 4 || lane < 24) glds16(up + c * 1024, lds_u + x * kTileB35 + c * 1024);
        }
        const int w = lane & 31;
        const char* jb = reinterpret_cast<const char*>(J) + e0 * 8 + w * 4;
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const int row = 2 * p + (lane >> 5);
            if (p < 4 || lane < 32) glds4(jb + (int64_t)row * E * 8, lds_j + p * 256);
        }
    };
    int64_t tile = (int64_t)blockIdx.x * kDivWavesPerBlock + wave;
    bool first = true;
    if (tile < nTiles) issue_loads(tile);
    for (; tile < nTiles; tile += stride) {
        const int64_t e0 = tile * kTE;
        // issue order: L(t) S(t-1) | wait L(t): the previous tile's 5 stores stay in flight
        if (first) wait_vmcnt<0>();
        else wait_vmcnt<5>();
        first = false;

        double jac[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) jac[k] = L->j[k * kTE + n];   // jac[x*3 + r]

        v4d acc[kDivRowTiles];
#pragma unroll
        for (int t = 0; t < kDivRowTiles; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int jq = 0; jq < kDivJq; ++jq) {
            const int j = 4 * jq + g;
            const int jc = j < kNp35 ? j : 0;
            double u0 = L->u[0][n * kNp35 + jc];
            double u1 = L->u[1][n * kNp35 + jc];
            double u2 = L->u[2][n * kNp35 + jc];
            if (j >= kNp35) { u0 = 0.0; u1 = 0.0; u2 = 0.0; }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double b = jac[0 * 3 + r] * u0 + jac[1 * 3 + r] * u1 + jac[2 * 3 + r] * u2;
#pragma unroll
                for (int t = 0; t < kDivRowTiles; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[t][jq][r], b, acc[t], 0, 0, 0);
            }
        }

        // u / J tiles are fully consumed (every LDS read fed an MFMA that has
        // issued): hand the buffer back to the DMA engine for the next tile.
#pragma unroll
        for (int t = 0; t < kDivRowTiles; ++t) asm volatile("" : "+v"(acc[t]));
        if (tile + stride < nTiles) issue_loads(tile + stride);

        // ---- transposed store: lane (g, n) holds out[e0 + n][i = 16t + g + 4q]
        double* ob = L->o;
#pragma unroll
        for (int t = 0; t < kDivRowTiles; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * t + g + 4 * q;
                if (i < kNp35) ob[n * kNp35 + i] = acc[t][q];
            }
        wave_lds_fence();
        double* op = out + e0 * kNp35;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            if (c < 4 || lane < 24) {
                const int q = c * 64 + lane;
                const v2d val = *reinterpret_cast<const v2d*>(ob + 2 * q);
                *reinterpret_cast<v2d*>(op + 2 * q) = val;
            }
        }
        wave_lds_fence();
    }
}

inline bool div_mfma_supported(int Np) { return Np == kNp35; }

}  // namespace fe
