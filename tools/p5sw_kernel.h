// p5sw_kernel.h -- EXPERIMENT (not part of the library; driver: tools/p5sw.hip).  Measured on MI355X at E = 1e6:
// 0.414 ms in its first form (the compiler hoists the J-combine out from under the MFMAs), 0.46-0.50 ms with the side
// work pinned by scheduling barriers and the next tile's fragments prefetched (512 registers and spills), 0.433 ms in
// this leaner pinned form (434 registers, no spills) -- against 0.376-0.386 ms for the shipped eight-wave kernel:
// rejected, kept for the record (p = 5).
//
// grad of tetrahedra p = 5 (Np = 56), 'xre,rij,ej->xei', ONE wave per SIMD with the operator's
// big-tile MFMA fragments resident in registers and everything else of a tile in the shadow of its MFMAs.
//
// Why another kernel: at Np = 56 the A fragments are 126 doubles per lane.  The eight-wave kernel (fe_div.h, W8)
// keeps them in LDS (75 KB), which leaves each wave ONE tile buffer: a wave requests its next tile only after its
// stores, so load latency, MFMAs and the three output planes form a serial chain per wave that the SIMD's
// second wave covers only in part (measured: compute alone 0.328 ms, data movement alone 0.340 ms, together
// 0.375-0.386 ms at E = 1e6).  Here a block is four waves (one per SIMD, 512 registers each): the fragments live
// in registers, so the LDS has room for a tile buffer that is released as soon as its B fragments are read and
// for three output-plane buffers, and ONE wave overlaps all three stages of consecutive tiles itself:
//
//   iteration t:  wait u(t), J(t) | B fragments, J -> registers | request u(t+1), J(t+1)
//                 row tile 0: 42 MFMAs   ... under them: planes of tile t-1 leave the output buffers (LDS -> HBM)
//                 row tile 1: 42 MFMAs   ... under them: J-combine of row tile 0 -> output buffers
//                 row tile 2: 42 MFMAs   ... under them: J-combine of row tile 1
//                 rows 48-55: 84 small MFMAs (4x4x4, A from an LDS table) ... J-combine of row tile 2
//                 J-combine of the small rows (6 values)
//
// MFMA order per row tile: (jq, r) with r fastest -- three independent accumulator chains, 192 cycles between
// dependent instructions.  Layouts as in fe_div.h MODE 4 (grad by components): lane (g, n) supplies
// A[16 t + n][4 jq + g] and B = u[e0 + n][4 jq + g], and holds out[e0 + n][16 t + g + 4 q].
#pragma once
#include "../feinsum_amd/csrc/fe_common.h"

namespace fe {

struct GradP5Geom {
    static constexpr int NP = 56, TEL = 16, ND = 3;
    static constexpr int KSJ = NP / 4;                  // 14 k-steps
    static constexpr int BT = NP / 16;                  // 3 row tiles of 16
    static constexpr int NS = (NP - 16 * BT) / 4;       // 2 groups of 4 rows
    static constexpr int PLANE_D = TEL * NP;            // 896 doubles
    static constexpr int CHUNKS = PLANE_D / 2, INSTR = CHUNKS / 64;   // 448 sixteen-byte chunks = 7 wave instructions
    static constexpr int J_CHUNKS = 9 * TEL / 2, J_INSTR = (J_CHUNKS + 63) / 64;   // 72 chunks = 2 instructions
    static constexpr int LOADS = INSTR + J_INSTR, STORES = 3 * INSTR;
    static constexpr int ASMALL_D = ND * KSJ * NS * 16;
    struct WaveLds {
        double u[PLANE_D];
        double o[3][PLANE_D];
        double j[2][9 * TEL];
    };
    static constexpr int WAVES = 4, THREADS = 256;
    static constexpr int OP_D = ND * NP * NP;
    static constexpr int WAVE_BYTES = (int)sizeof(WaveLds) * WAVES;
    static constexpr int LDS_BYTES = (WAVE_BYTES > OP_D * 8 ? WAVE_BYTES : OP_D * 8) + ASMALL_D * 8;
    static_assert(LDS_BYTES <= 160 * 1024, "one block per CU");
    static_assert(LOADS + STORES <= 60, "counted vmcnt");
};

template <int kDbg = 0>
__global__ __launch_bounds__(256, 1) void grad_p5_kernel(const double* __restrict__ J, const double* __restrict__ D,
                                                         const double* __restrict__ u, double* __restrict__ out,
                                                         int64_t E, int64_t nTiles, int opT) {
    using G = GradP5Geom;
    constexpr int NP = G::NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    G::WaveLds* L = reinterpret_cast<G::WaveLds*>(smem) + wave;
    double* asmall = reinterpret_cast<double*>(smem + (G::LDS_BYTES - G::ASMALL_D * 8));

    // ---- operator: staged once per block, then 126 big-tile fragments per lane into registers
    double abig[G::BT][G::KSJ][G::ND];
    {
        double* dl = reinterpret_cast<double*>(smem);
        stage_operator<G::OP_D, G::THREADS>(D, dl);
        __syncthreads();
        const int istride = opT ? 1 : NP, jstride = opT ? NP : 1;
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq)
#pragma unroll
            for (int r = 0; r < G::ND; ++r)
#pragma unroll
                for (int t = 0; t < G::BT; ++t)
                    abig[t][jq][r] = dl[r * (NP * NP) + (16 * t + n) * istride + (4 * jq + g) * jstride];
        for (int idx = threadIdx.x; idx < G::ASMALL_D; idx += G::THREADS) {
            const int row4 = idx & 3, gg = (idx >> 2) & 3, q = (idx >> 4) % G::NS, ks = (idx >> 4) / G::NS;
            const int i = 16 * G::BT + 4 * q + row4, j = 4 * (ks / G::ND) + gg, r = ks % G::ND;
            asmall[idx] = dl[r * (NP * NP) + i * istride + j * jstride];
        }
        __syncthreads();
    }
    const double* as_lane = asmall + g * 4 + (n & 3);

    const unsigned lds_u = lds_addr_uniform(L->u);
    const unsigned lds_j0 = lds_addr_uniform(L->j[0]), lds_j1 = lds_addr_uniform(L->j[1]);
    const int64_t stride = (int64_t)gridDim.x * G::WAVES;
    auto issue_loads = [&](int64_t tile, int jbuf) {
        const char* ub = reinterpret_cast<const char*>(u) + tile * (G::TEL * NP * 8);
#pragma unroll
        for (int c = 0; c < G::INSTR; ++c) glds16_nt(ub + tile_src_chunk<NP>(c * 64 + lane) * 16, lds_u + c * 1024);
        const char* jb = reinterpret_cast<const char*>(J) + tile * (G::TEL * 8);
        const unsigned lj = jbuf ? lds_j1 : lds_j0;
#pragma unroll
        for (int c = 0; c < G::J_INSTR; ++c) {
            const int q = c * 64 + lane;
            const int row = q / (G::TEL / 2), col = q - row * (G::TEL / 2);
            if ((c + 1) * 64 <= G::J_CHUNKS || q < G::J_CHUNKS) glds16(jb + ((int64_t)row * E) * 8 + col * 16, lj + c * 1024);
        }
    };
    // plane x of a finished tile: LDS -> registers (one slot) -> HBM (a later slot: the LDS latency is under MFMAs)
    v2d held[G::INSTR];
    auto drain_read = [&](int x) {
        const double* ob = L->o[x];
#pragma unroll
        for (int c = 0; c < G::INSTR; ++c) held[c] = *reinterpret_cast<const v2d*>(ob + 2 * tile_dst_chunk<NP>(c * 64 + lane));
    };
    auto drain_store = [&](int x, int64_t tile) {
        double* op = out + ((int64_t)x * E + tile * G::TEL) * NP;
#pragma unroll
        for (int c = 0; c < G::INSTR; ++c) {
            if (kDbg & 2) { if (held[c][0] == 1.2345e-300) op[2 * (c * 64 + lane)] = held[c][1]; }
            else __builtin_nontemporal_store(held[c], reinterpret_cast<v2d*>(op + 2 * (c * 64 + lane)));
        }
    };

    int64_t tile = (int64_t)blockIdx.x * G::WAVES + wave;
    if (tile >= nTiles) return;
    int64_t prev = -1;
    int jbuf = 0, iteration = 0;
    issue_loads(tile, 0);

    while (true) {
        const int64_t nt = tile + stride;
        // issue order: L(t) [iteration t-1, after its fragment reads] S(t-2) [iteration t-1] | wait L(t)
        if (iteration >= 2) wait_vmcnt<G::STORES>();
        else wait_vmcnt<0>();
        ++iteration;
        double bf[G::KSJ], jk[9];
        {
            const double* jt = L->j[jbuf];
#pragma unroll
            for (int k = 0; k < 9; ++k) jk[k] = jt[k * G::TEL + n];
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq) bf[jq] = L->u[tile_index<NP>(n, 4 * jq + g)];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq) asm volatile("" : "+v"(bf[jq]));
        if (nt < nTiles) issue_loads(nt, jbuf ^ 1);

        // one (plane, value) unit of the J-combine of an accumulator set: out[x][e0 + n][16 t + g + 4 q]
        auto combine_unit = [&](const v4d (&acc)[3], int t, int unit) {
            const int x = unit >> 2, q = unit & 3;
            const double v = jk[x * 3] * acc[0][q] + jk[x * 3 + 1] * acc[1][q] + jk[x * 3 + 2] * acc[2][q];
            L->o[x][tile_index<NP>(n, 16 * t + g + 4 * q)] = v;
        };

        v4d acc[2][3];   // ping-pong: the combine of one row tile runs under the MFMAs of the next
#pragma unroll
        for (int t = 0; t < G::BT; ++t) {
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[t & 1][r] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    acc[t & 1][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(abig[t][jq][r], bf[jq], acc[t & 1][r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (t == 0) {
                    if (prev >= 0 && jq % 4 == 1 && jq / 4 < 3) drain_read(jq / 4);
                    if (prev >= 0 && jq % 4 == 3 && jq / 4 < 3) drain_store(jq / 4, prev);
                } else if (jq >= 1 && jq <= 12) {
                    combine_unit(acc[(t - 1) & 1], t - 1, jq - 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        double accs[G::NS][3];
#pragma unroll
        for (int q = 0; q < G::NS; ++q)
#pragma unroll
            for (int r = 0; r < 3; ++r) accs[q][r] = 0.0;
        double as_c[3][G::NS], as_n[3][G::NS];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < G::NS; ++q) as_c[r][q] = as_lane[((0 * 3 + r) * G::NS + q) * 16];
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq) {
            if (jq + 1 < G::KSJ) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int q = 0; q < G::NS; ++q) as_n[r][q] = as_lane[(((jq + 1) * 3 + r) * G::NS + q) * 16];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < G::NS; ++q)
                    accs[q][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(as_c[r][q], bf[jq], accs[q][r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (jq >= 1 && jq <= 12) combine_unit(acc[(G::BT - 1) & 1], G::BT - 1, jq - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < G::NS; ++q) as_c[r][q] = as_n[r][q];
        }
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int q = 0; q < G::NS; ++q) {
                const double v = jk[x * 3] * accs[q][0] + jk[x * 3 + 1] * accs[q][1] + jk[x * 3 + 2] * accs[q][2];
                L->o[x][tile_index<NP>(n, 16 * G::BT + 4 * q + g)] = v;
            }
        wave_lds_fence();
        prev = tile;
        if (nt >= nTiles) break;
        tile = nt;
        jbuf ^= 1;
    }
#pragma unroll
    for (int x = 0; x < 3; ++x) { drain_read(x); drain_store(x, prev); }
}

}  // namespace fe
