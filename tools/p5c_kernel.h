// p5c_kernel.h -- EXPERIMENT driver kernel (tools/p5c.hip) for grad of tetrahedra p = 5 (Np = 56), 'xre,rij,ej->xei':
// ONE wave per SIMD, the operator's 16-row MFMA fragments resident in ACCUMULATION registers and read there by the MFMAs.
//
// What round 4 measured on gfx950 (tools/mfma_valu_overlap.hip, profiles/r04/mfma_valu_overlap.txt) and what follows from it:
//  * a wave's own VALU instructions never run under its own f64 MFMA (64 cycles + the side work, for f64, f32 and integer
//    instructions alike); LDS and memory instructions do.  So per tile a wave costs MFMA cycles + VALU cycles, and what a
//    schedule can hide is LDS / memory LATENCY: every LDS read is issued two MFMA groups before its consumer, the stores of a
//    finished tile trickle out one per MFMA group, and the only VALU work in the loop is the J contraction (126 f64
//    operations per tile, written as independent chains) plus the address arithmetic of 21 stores;
//  * with the fragments spilled to AGPRs by the register allocator (round 2's tools/p5sw_kernel.h) every MFMA was preceded
//    by two v_accvgpr_read -- 252 VALU instructions per tile.  Here the fragments are LOADED into AGPRs (ds_read_b64 with an
//    "a" destination) and the library is compiled with -mllvm -amdgpu-mfma-vgpr-form=1, so that the MFMA's A operand is the
//    AGPR pair itself and its accumulator a VGPR tuple: no copies (checked in the ISA: zero v_accvgpr instructions).
//
// Rows: components r = 0, 1, 2 of D, 56 rows each.  Rows 0..47 of every component are three 16-row tiles (t = 0, 1, 2: 126
// fragments, 252 AGPRs).  Rows 48..55 of components 0 and 1 are packed into ONE more 16-row tile (lane (g, n) supplies
// D[n / 8][48 + n % 8][4 jq + g]; its C rows g, g + 4 are component 0's rows 48 + g, 52 + g and its C rows g + 8, g + 12 the
// same rows of component 1), rows 48..55 of component 2 run on v_mfma_f64_4x4x4_4b (two groups of four rows, A from a 3.5 KB
// LDS table): 140 x 64 + 28 x 16 = 9408 MFMA cycles per tile as before, but 28 LDS operand reads instead of 84.
// All three components of an output row sit in the same lane, so the J contraction stays lane-local.
#pragma once
#include <type_traits>
#include "../feinsum_amd/csrc/fe_common.h"

namespace fe {

struct GradP5cGeom {
    static constexpr int NP = 56, TEL = 16, ND = 3;
    static constexpr int KSJ = NP / 4;                  // 14 k-steps
    static constexpr int BT = 3;                        // full 16-row tiles per component
    static constexpr int NS = 2;                        // 4-row groups of component 2 (rows 48..55)
    static constexpr int PLANE_D = TEL * NP;            // 896 doubles
    static constexpr int CHUNKS = PLANE_D / 2, INSTR = CHUNKS / 64;   // 448 sixteen-byte chunks = 7 wave instructions
    static constexpr int J_CHUNKS = 9 * TEL / 2, J_INSTR = (J_CHUNKS + 63) / 64;   // 72 chunks = 2 instructions
    static constexpr int LOADS = INSTR + J_INSTR, STORES = 3 * INSTR;
    static constexpr int ASMALL_D = KSJ * NS * 16;      // [jq][q][g][row4] doubles: component 2 only
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

__device__ unsigned long long p5c_clock[12];   // wave 0 of block 7: shader cycles, 100 MHz ticks, tiles; cycles per phase

template <int kDbg = 0>
__global__ __launch_bounds__(256, 1) void grad_p5c_kernel(const double* __restrict__ J, const double* __restrict__ D,
                                                          const double* __restrict__ u, double* __restrict__ out,
                                                          int64_t E, int64_t nTiles, int opT) {
    using G = GradP5cGeom;
    constexpr int NP = G::NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    G::WaveLds* L = reinterpret_cast<G::WaveLds*>(smem) + wave;
    double* asmall = reinterpret_cast<double*>(smem + (G::LDS_BYTES - G::ASMALL_D * 8));

    // ---- operator: staged once per block; 126 fragments per lane into AGPRs, 14 (the packed tile) into VGPRs
    double abig[G::BT][G::KSJ][G::ND];
    double apk[G::KSJ];
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
                for (int t = 0; t < G::BT; ++t) {   // straight from LDS into an accumulation register: the MFMAs read it there
                    const unsigned a = (unsigned)(uintptr_t)(FE_AS3 const void*)(dl + r * (NP * NP) + (16 * t + n) * istride + (4 * jq + g) * jstride);
                    asm volatile("ds_read_b64 %0, %1" : "=a"(abig[t][jq][r]) : "v"(a));
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq)
            apk[jq] = dl[(n >> 3) * (NP * NP) + (48 + (n & 7)) * istride + (4 * jq + g) * jstride];
        for (int idx = threadIdx.x; idx < G::ASMALL_D; idx += G::THREADS) {
            const int row4 = idx & 3, gg = (idx >> 2) & 3, q = (idx >> 4) % G::NS, jq = (idx >> 4) / G::NS;
            asmall[idx] = dl[2 * (NP * NP) + (48 + 4 * q + row4) * istride + (4 * jq + gg) * jstride];
        }
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq) asm volatile("" : "+v"(apk[jq]));
        __syncthreads();
    }
    const double* as_lane = asmall + g * 4 + (n & 3);

    const unsigned lds_u = lds_addr_uniform(L->u);
    const unsigned lds_j0 = lds_addr_uniform(L->j[0]), lds_j1 = lds_addr_uniform(L->j[1]);
    const int64_t stride = (int64_t)gridDim.x * G::WAVES;
    auto issue_loads = [&](int64_t tile, int jbuf) {
        if (kDbg & 8) return;
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
    // chunk k = 7 x + c of a finished tile (plane x, 1 KiB piece c): LDS -> registers at one MFMA group, registers -> HBM
    // two groups later
    // plane x of a finished tile: LDS -> registers behind one MFMA group, registers -> HBM two groups later, the seven 1 KiB
    // pieces of a plane back to back (a wave's 7 KiB span reaches the memory controller as one burst: trickled out one piece per
    // MFMA group the same stores ran at 5.3 instead of 6.2 TB/s)
    v2d held[G::INSTR];
    auto drain_read = [&](int x) {
#pragma unroll
        for (int c = 0; c < G::INSTR; ++c) held[c] = *reinterpret_cast<const v2d*>(L->o[x] + 2 * tile_dst_chunk<NP>(c * 64 + lane));
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

    double bf[G::KSJ], jk[9];
    wait_vmcnt<0>();
    {
        const double* jt = L->j[0];
#pragma unroll
        for (int k = 0; k < 9; ++k) jk[k] = jt[k * G::TEL + n];
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq) bf[jq] = L->u[tile_index<NP>(n, 4 * jq + g)];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq) asm volatile("" : "+v"(bf[jq]));
    }
    if (tile + stride < nTiles) issue_loads(tile + stride, 1);

    const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    // one tile; kDrain: a finished tile waits in the output buffers (every tile of a wave but its first -- a compile-time
    // flag, so that the loop body is straight-line code: with a run-time test per MFMA group the compiler's LDS wait counts
    // collapse to lgkmcnt(0) right behind each read)
    auto tile_body = [&](auto drain_tag) -> bool {
        constexpr bool kDrain = decltype(drain_tag)::value;
        const int64_t nt = tile + stride;
        unsigned long long ph_t = __builtin_amdgcn_s_memtime();
        auto phase_mark = [&](int k) { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph[k] += now - ph_t; ph_t = now; };
        ++iteration;

        // the J contraction of plane x of an accumulator set, four outputs at a time (four independent chains of one multiply
        // and two fused multiply-adds): out[x][e0 + n][16 t + g + 4 q], q = 0..3
        auto combine_plane = [&](const v4d (&acc)[3], int t, int x) {
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = jk[x * 3] * acc[0][q];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = __builtin_fma(jk[x * 3 + 1], acc[1][q], v[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = __builtin_fma(jk[x * 3 + 2], acc[2][q], v[q]);
#pragma unroll
            for (int q = 0; q < 4; ++q) L->o[x][tile_index<NP>(n, 16 * t + g + 4 * q)] = v[q];
        };
        // what rides behind MFMA group `grp` (0..41: the three full row tiles)
        v4d acc[2][3];   // ping-pong: the contraction of one row tile runs behind the MFMAs of the next
        auto side = [&](int grp) {
            if (kDrain) {
                if (grp == 0 || grp == 4 || grp == 8) drain_read(grp / 4);
                if (grp == 2 || grp == 6 || grp == 10) drain_store(grp / 4, prev);
            }
            if (grp == 16 || grp == 18 || grp == 20) combine_plane(acc[0], 0, (grp - 16) / 2);   // row tile 0, behind the drain
            if (grp == 30 || grp == 32 || grp == 34) combine_plane(acc[1], 1, (grp - 30) / 2);
        };

#pragma unroll
        for (int t = 0; t < G::BT; ++t) {
            phase_mark(t == 0 ? 0 : t);
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[t & 1][r] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int jq = 0; jq < G::KSJ; ++jq) {
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    if (kDbg & 1) { if (jq == 0) { asm volatile("; keep %0" :: "a"(abig[t][jq][r])); acc[t & 1][r] = v4d{bf[jq], bf[3], bf[1], bf[2]}; } }
                    else acc[t & 1][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(abig[t][jq][r], bf[jq], acc[t & 1][r], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                side(G::KSJ * t + jq);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        phase_mark(3);
        // ---- last phase: the packed tile (rows 48..55 of components 0 and 1) and the two 4-row groups of component 2; behind
        //      them the contraction of row tile 2 and, k-step by k-step, the NEXT tile's B fragments into the registers this
        //      tile no longer needs (its loads were issued a whole tile ago; younger than them: this iteration's 21 stores)
        if (kDrain) wait_vmcnt<G::STORES>();
        else wait_vmcnt<0>();
        v4d accp = {0.0, 0.0, 0.0, 0.0};
        double accs[G::NS] = {0.0, 0.0};
        double as_q[3][G::NS];
#pragma unroll
        for (int q = 0; q < G::NS; ++q) { as_q[0][q] = as_lane[(0 * G::NS + q) * 16]; as_q[1][q] = as_lane[(1 * G::NS + q) * 16]; }
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq) {
            if (jq + 2 < G::KSJ) {
#pragma unroll
                for (int q = 0; q < G::NS; ++q) as_q[(jq + 2) % 3][q] = as_lane[((jq + 2) * G::NS + q) * 16];
            }
            if (kDbg & 1) { accp[jq & 3] += apk[jq] + bf[jq]; accs[0] += as_q[jq % 3][0]; accs[1] += as_q[jq % 3][1]; }
            else {
                accp = __builtin_amdgcn_mfma_f64_16x16x4f64(apk[jq], bf[jq], accp, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < G::NS; ++q) accs[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(as_q[jq % 3][q], bf[jq], accs[q], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (jq == 2 || jq == 4 || jq == 6) combine_plane(acc[0], 2, (jq - 2) / 2);      // row tile 2 (its accumulators: acc[2 & 1])
            bf[jq] = L->u[tile_index<NP>(n, 4 * jq + g)];                                   // the next tile's fragment
            __builtin_amdgcn_sched_barrier(0);
        }
        phase_mark(4);
        {   // rows 48 + g and 52 + g: six independent chains
            double v[3][G::NS];
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int q = 0; q < G::NS; ++q) v[x][q] = jk[x * 3] * accp[q];
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int q = 0; q < G::NS; ++q) v[x][q] = __builtin_fma(jk[x * 3 + 1], accp[2 + q], v[x][q]);
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int q = 0; q < G::NS; ++q) v[x][q] = __builtin_fma(jk[x * 3 + 2], accs[q], v[x][q]);
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int q = 0; q < G::NS; ++q) L->o[x][tile_index<NP>(n, 48 + 4 * q + g)] = v[x][q];
        }
        {   // the next tile's geometry factors
            const double* jt = L->j[jbuf ^ 1];
#pragma unroll
            for (int k = 0; k < 9; ++k) jk[k] = jt[k * G::TEL + n];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int jq = 0; jq < G::KSJ; ++jq) asm volatile("" : "+v"(bf[jq]));
#pragma unroll
        for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(jk[k]));
        wave_lds_fence();
        prev = tile;
        if (nt >= nTiles) return false;
        tile = nt;
        jbuf ^= 1;
        if (tile + stride < nTiles) issue_loads(tile + stride, jbuf ^ 1);
        phase_mark(5);
        return true;
    };
    if (tile_body(std::false_type{}))
        while (tile_body(std::true_type{})) {}
#pragma unroll
    for (int x = 0; x < 3; ++x) { drain_read(x); drain_store(x, prev); }
    if (blockIdx.x == 7 && threadIdx.x == 0) {
        p5c_clock[0] = __builtin_amdgcn_s_memtime() - clk0;
        p5c_clock[1] = __builtin_amdgcn_s_memrealtime() - rt0;
        p5c_clock[2] = iteration;
        for (int k = 0; k < 6; ++k) p5c_clock[4 + k] = ph[k];
    }
}

}  // namespace fe
