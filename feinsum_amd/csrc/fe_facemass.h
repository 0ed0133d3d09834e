// fe_facemass.h -- face-mass (lift) einsum, NB fields sharing J and R:
//   out_k[e,i] = sum_{f,j} J[e,f] R[f,i,j] v_k[f,e,j]
// ('ef,fij,fej->ei' x b, reference: test/test_loopy_utils.py:34-48; layout
// sibling 'ifj,fe,fej->ei': tuning/impls/ifj_fe_fej_to_ei.py:18-277, printed
// kernel doc/compiler_writer_tutorial.rst:357-493).
//
// Schedule = the opt_einsum-optimal one: Jv_k[(f,j), e] = J[e,f] v_k[f,e,j] (one
// multiply, VALU, produced in MFMA B-fragment layout), then
//   out_k[i, e] = sum_{(f,j)} R'[i, (f,j)] * Jv_k[(f,j), e]
// on the matrix cores, A = R' (35 x 60; K = 60 = 15 k-steps exactly) resident in
// registers (45 doubles / lane): rows 0..31 as two 16-row tiles on
// v_mfma_f64_16x16x4_f64 (2 x 15 MFMAs of 64 cycles) and rows 32..34 on
// v_mfma_f64_4x4x4_4b_f64 (15 x 16 cycles; see fe_div.h) per (16-element tile,
// field) "unit" per wave.
// Data movement: a unit's four face slabs (4 x 1920 contiguous bytes) come in
// by LDS-DMA into a 2-slot ring, two units ahead; J for a tile (512 B) comes with
// its first unit; results leave through a separate LDS transposition buffer as
// 1-KiB contiguous stores.  Loads for unit m+2 are issued as soon as unit m's B
// fragments are in registers, i.e. BEFORE unit m's MFMAs and stores, so the
// counted vmcnt at the top of a unit never waits for stores.
#pragma once
#include "fe_generic.h"
#include "fe_grad.h"

namespace fe {

constexpr int kFmNf = 4, kFmNfp = 15;
constexpr int kFmSlabD = kTE * kFmNfp;            // 240 doubles = 1920 bytes per face
constexpr int kFmUnitD = kFmNf * kFmSlabD;        // 960 doubles = 7680 bytes
constexpr int kFmKSteps = 15;                     // K = 60
constexpr int kFmBigTiles = 2;                    // rows 0..31; rows 32..34 on 4x4x4_4b

struct FmWaveLds {
    double v[2][kFmUnitD];     // ring of field slabs: v[slot][f][e][j]
    double o[kTileD35];        // output transposition buffer
    double j[kFmNf * kTE];     // J tile, [e][f] or [f][e] as in global memory
};
static_assert(sizeof(FmWaveLds) == 20352, "LDS budget");
constexpr int kFmWavesPerBlock = 4;
constexpr int kFmLdsBytes = sizeof(FmWaveLds) * kFmWavesPerBlock;  // 81408: 2 blocks / CU
constexpr int kFmStoresPerUnit = 5;

// 8 x 16-byte LDS-DMA for the field slabs (+ 2 x 4-byte for J at a tile start).
template <bool kWithJ>
__device__ __forceinline__ void fm_issue_unit_loads(const double* __restrict__ J,
                                                    const double* __restrict__ vk, int64_t E,
                                                    int64_t tile, int lane, unsigned lds_v,
                                                    unsigned lds_j, int jfe) {
    const int64_t e0 = tile * kTE;
    const char* vb = reinterpret_cast<const char*>(vk) + e0 * (kFmNfp * 8) + lane * 16;
#pragma unroll
    for (int f = 0; f < kFmNf; ++f) {
        const char* vf = vb + (int64_t)f * E * (kFmNfp * 8);
        glds16_nt(vf, lds_v + f * (kFmSlabD * 8));
        if (lane < 56) glds16_nt(vf + 1024, lds_v + f * (kFmSlabD * 8) + 1024);
    }
    if (kWithJ) {
        const char* jb = reinterpret_cast<const char*>(J);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const char* src = jfe ? jb + ((int64_t)(2 * p + (lane >> 5)) * E + e0) * 8 + (lane & 31) * 4
                                  : jb + e0 * (kFmNf * 8) + (p * 64 + lane) * 4;
            glds4(src, lds_j + p * 256);
        }
    }
}

template <int NB>
__global__ __launch_bounds__(256, 2) void facemass_np35_mfma_kernel(
    const double* __restrict__ J, const double* __restrict__ R, FieldPtrs P, int64_t E,
    int64_t nTiles, int jfe, int rifj) {
    static_assert(NB >= 2 && NB <= kMaxFields, "2..8 fields per launch");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    FmWaveLds* L = reinterpret_cast<FmWaveLds*>(smem) + wave;
    const int n = lane & 15, g = lane >> 4;

    // ---- per-lane K decomposition: k = 4 ks + g = 15 f + j
    int voff[kFmKSteps];   // offset of v[f][n][j] inside a unit slab
    int joff[kFmKSteps];   // offset of J[e0+n][f] inside the J tile
    double abig[kFmBigTiles][kFmKSteps];   // 16x16x4: lane (g, n) supplies A[row 16t + n][k = g]
    double asmall[kFmKSteps];              // 4x4x4_4b: block n/4, row 32 + n%4 (row 35 = zero), k = g
    {
        // R goes through LDS once per block (see stage_operator)
        double* rl = reinterpret_cast<double*>(smem);
        stage_operator<kFmNf * kNp35 * kFmNfp>(R, rl);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kFmKSteps; ++ks) {
            const int k = 4 * ks + g;
            const int f = k / kFmNfp, j = k - f * kFmNfp;
            voff[ks] = f * kFmSlabD + n * kFmNfp + j;
            joff[ks] = jfe ? f * kTE + n : n * kFmNf + f;
#pragma unroll
            for (int t = 0; t < kFmBigTiles; ++t) {
                const int i = 16 * t + n;
                abig[t][ks] = rl[rifj ? (i * kFmNf + f) * kFmNfp + j : (f * kNp35 + i) * kFmNfp + j];
            }
            const int i3 = 32 + (n & 3), i3c = i3 < kNp35 ? i3 : 0;
            const double a3 = rl[rifj ? (i3c * kFmNf + f) * kFmNfp + j : (f * kNp35 + i3c) * kFmNfp + j];
            asmall[ks] = (i3 < kNp35) ? a3 : 0.0;
        }
        __syncthreads();   // the staging area is reused as the waves' private buffers from here on
    }

    const unsigned lds_v0 = lds_addr_uniform(L->v[0]);
    const unsigned lds_j = lds_addr_uniform(L->j);
    const int64_t stride = (int64_t)gridDim.x * kFmWavesPerBlock;
    const int64_t first = (int64_t)blockIdx.x * kFmWavesPerBlock + wave;
    if (first >= nTiles) return;

    // prologue: units 0 and 1 of the first tile
    fm_issue_unit_loads<true>(J, P.v[0], E, first, lane, lds_v0, lds_j, jfe);
    fm_issue_unit_loads<false>(J, P.v[1], E, first, lane, lds_v0 + kFmUnitD * 8, lds_j, jfe);

    int slot = 0;
    bool warm = false;   // false for the first two units of this wave
    double jv[kFmKSteps];
    const bool younger_half = blockIdx.x >= (gridDim.x + 1) / 2;
    int iteration = 0;
    for (int64_t tile = first; tile < nTiles; tile += stride) {
        balance_priority(younger_half, iteration++);
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            // ---- wait for this unit's loads; younger ops: S(m-2), L(m+1), S(m-1)
            const bool next_is_tile_start = (k + 1 == NB);
            const bool has_next = !next_is_tile_start || (tile + stride < nTiles);
            if (warm && has_next) {
                if (next_is_tile_start) wait_vmcnt<2 * kFmStoresPerUnit + 10>();
                else wait_vmcnt<2 * kFmStoresPerUnit + 8>();
            } else {
                wait_vmcnt<0>();
            }
            if (k >= 1) warm = true;   // units 0,1 of the first tile are cold

            if (k == 0) {
#pragma unroll
                for (int ks = 0; ks < kFmKSteps; ++ks) jv[ks] = L->j[joff[ks]];
            }
            const double* vs = L->v[slot];
            double bfrag[kFmKSteps];
#pragma unroll
            for (int ks = 0; ks < kFmKSteps; ++ks) bfrag[ks] = jv[ks] * vs[voff[ks]];
            // make sure the slab (and, at k == 0, the J tile) is in registers before
            // the slot is handed back to the DMA engine
#pragma unroll
            for (int ks = 0; ks < kFmKSteps; ++ks) asm volatile("" : "+v"(bfrag[ks]));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

            // ---- prefetch unit m+2 into the slot just drained
            {
                const int k2 = (k + 2) % NB;   // folds after unrolling
                const int64_t tile2 = tile + stride * ((k + 2) / NB);
                if (tile2 < nTiles) {
                    if (k2 == 0)
                        fm_issue_unit_loads<true>(J, P.v[k2], E, tile2, lane, lds_v0 + slot * (kFmUnitD * 8), lds_j, jfe);
                    else
                        fm_issue_unit_loads<false>(J, P.v[k2], E, tile2, lane, lds_v0 + slot * (kFmUnitD * 8), lds_j, jfe);
                }
            }

            // ---- 30 + 15 MFMAs
            v4d acc[kFmBigTiles];
            double acc3 = 0.0;
#pragma unroll
            for (int t = 0; t < kFmBigTiles; ++t) acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < kFmKSteps; ++ks) {
#pragma unroll
                for (int t = 0; t < kFmBigTiles; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(abig[t][ks], bfrag[ks], acc[t], 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(asmall[ks], bfrag[ks], acc3, 0, 0, 0);
            }

            // ---- transposed store.  16x16x4 C/D: lane (g, n) holds out[e0 + n][16t + g + 4q];
            //      4x4x4_4b D: lane (g, n) holds out[e0 + n][32 + g] (g == 3 is padding)
            double* ob = L->o;
#pragma unroll
            for (int t = 0; t < kFmBigTiles; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) ob[n * kNp35 + 16 * t + g + 4 * q] = acc[t][q];
            if (g < 3) ob[n * kNp35 + 32 + g] = acc3;
            wave_lds_fence();
            double* op = P.out[k] + tile * (kTE * kNp35);
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                if (c < 4 || lane < 24) {
                    const int q = c * 64 + lane;
                    const v2d val = *reinterpret_cast<const v2d*>(ob + 2 * q);
                    __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + 2 * q));
                }
            }
            wave_lds_fence();
            slot ^= 1;
        }
    }
}

inline bool facemass_mfma_supported(int Np, int nf, int Nfp, int b) {
    return Np == kNp35 && nf == kFmNf && Nfp == kFmNfp && b >= 2;
}

}  // namespace fe
