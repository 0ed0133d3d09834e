// fe_tiled.h -- LDS-tiled VALU kernel for the DG families at ANY shape.
//
// The MFMA kernels are compiled for the tetrahedral orders p = 1..4.  Every other shape -- 3-D
// p = 5 (Np = 56, Nfp = 21), triangles (ndim = 2; Np = 3, 6, 10, 15, 21; three faces), odd
// sizes -- used to fall to the one-thread-per-entry generic kernels, which re-read the operator
// from L1/L2 for every output entry (~2 TFLOP/s at p = 4).  This kernel is the hand-written form
// of what the reference's transforms generate (operator and element data prefetched to local
// memory, register tiling over elements: tuning/impls/xre_rij_ej_to_xei.py:26-275,
// xre_rij_xej_to_ei_v6.py:41-110, ifj_fe_fej_to_ei.py:18-277), with run-time shapes:
//
//     out[e, i] (+ a per-family epilogue) = sum_k A_c[i, k] * B[e, k],     c < ncomp
//
// * A (the operator, all components) sits in LDS for the life of the block, transposed to
//   [c][k][i] so that the lanes of a wave -- consecutive i -- read consecutive addresses;
// * B is the element-local factor of the optimal contraction schedule, built per tile of TE
//   elements straight into LDS by all threads with coalesced global reads:
//       grad      B[e, j]       = u[e, j]                          (ncomp = ndim, K = Np)
//       div       B[e, (r, j)]  = sum_x J[x, r, e] u[x, e, j]      (K = ndim Np)
//       div comp. B[e, (s, j)]  = J[s, e] u[e, j]                  (K = 3 Np)
//       operator  B[e, j]       = J[e] u[e, j]  or  u[e, j]        (K = Np)
//       face-mass B[e, (f, j)]  = J[e, f] v[f, e, j]               (K = nf Nfp)
// * thread (i, g) owns three accumulator rows for the EB (8, or 4 when LDS is tight) elements of
//   element group g -- grad: row i of the ndim operator components; the others: output rows i,
//   i + IW, i + 2 IW (IW = ceil(Np / 3)) -- so three A values and EB broadcast B values feed 3 EB
//   FMAs;
// * grad's epilogue contracts the ndim accumulators with J[x, r, e] (staged per tile too).
// Partial last tiles are handled by guards, so there is no remainder path.
#pragma once
#include "../../include/feinsum_hip.h"
#include "fe_common.h"

namespace fe {

constexpr int kTiledRows = 3;        // accumulator rows per thread
constexpr int kTiledThreads = 512;
constexpr int kTiledMaxGroups = 64;  // element groups per tile
constexpr int64_t kTiledLdsBudget = 160 * 1024;

struct TiledArgs {
    const void* J;       // geometry factors in the family's layout, or nullptr   (element type T of the kernel:
    const void* A;       // operator in the family's layout                        double, or float for the
    FieldPtrs P;         // inputs / outputs of the nb fields                      float32 einsums of round 3)
    int64_t E;
    int family;          // FE_FAMILY_GRAD / DIV / DIVCOMP / MATAPPLY / FACEMASS
    int ndim, Np, nf, Nfp, nb;
    int opT, jlayout, rlayout;   // transposed operator; J as 'es' (div comp.) / 'fe' (face-mass); R layout 0..3
    int ncomp, K, KP, IW, EB, neg, TE, At_d;   // derived by tiled_plan()
};

// LDS bytes of a launch (operator + B tile + J tile), after filling in the derived fields.  esize: bytes per element.
inline int64_t tiled_plan(TiledArgs& a, int esize = 8) {
    a.ncomp = a.family == FE_FAMILY_GRAD ? a.ndim : 1;
    a.K = a.family == FE_FAMILY_GRAD || a.family == FE_FAMILY_MATAPPLY ? a.Np
        : a.family == FE_FAMILY_DIV ? a.ndim * a.Np
        : a.family == FE_FAMILY_DIVCOMP ? a.ndim * a.Np
                                        : a.nf * a.Nfp;
    a.KP = a.K | 1;   // odd row stride: the element groups of a wave hit different banks
    // lanes per element group: grad keeps row i of all components, the others three rows each
    a.IW = a.family == FE_FAMILY_GRAD ? a.Np : (a.Np + kTiledRows - 1) / kTiledRows;
    const int64_t jrows = a.family == FE_FAMILY_GRAD || a.family == FE_FAMILY_DIV ? (int64_t)a.ndim * a.ndim : 0;
    a.At_d = a.ncomp * a.K * a.Np + kTiledRows * a.IW;   // + slack: rows i >= Np are read, never stored
    a.At_d += a.At_d & 1;
    const int64_t op_bytes = esize * (int64_t)a.At_d;
    int64_t bytes = 0;
    for (a.EB = 8; a.EB >= 4; a.EB /= 2) {
        const int64_t group_bytes = esize * (int64_t)a.EB * (a.KP + jrows);   // B rows + J columns of one element group
        a.neg = kTiledThreads / a.IW;
        if (a.neg > kTiledMaxGroups) a.neg = kTiledMaxGroups;
        if (a.neg < 1) a.neg = 1;
        const int want = a.neg;
        // fewer element groups per tile when the operator leaves little room (e.g. div at p = 5)
        while (a.neg > 1 && op_bytes + a.neg * group_bytes > kTiledLdsBudget) --a.neg;
        bytes = op_bytes + a.neg * group_bytes;
        if (2 * a.neg >= want) break;   // at least half of the lanes busy; else try smaller groups
    }
    if (a.EB < 4) a.EB = 4;
    a.TE = a.neg * a.EB;
    return bytes;
}

// acc[c][b] = sum_k A[row c of this thread, k] B[e_b, k]: ap -> At[0][0][i] (k stride Np; the thread's
// rows are row_stride apart: K Np between grad's components, IW between output rows), bp -> the
// thread's first B row (row stride KP, all lanes of an element group read one address)
template <int NC, int EB, typename T>
__device__ __forceinline__ void tiled_gemm(const T* ap, int row_stride, const T* bp, int K, int Np, int KP,
                                           T (&acc)[kTiledRows][EB]) {
#pragma unroll
    for (int c = 0; c < kTiledRows; ++c)
#pragma unroll
        for (int b = 0; b < EB; ++b) acc[c][b] = T(0);
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
        T bv[EB], av[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) av[c] = ap[c * row_stride + k * Np];
#pragma unroll
        for (int b = 0; b < EB; ++b) bv[b] = bp[b * KP + k];
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int b = 0; b < EB; ++b) acc[c][b] += av[c] * bv[b];
    }
}

template <int EB, typename T = double>
__global__ __launch_bounds__(kTiledThreads) void tiled_apply_kernel(TiledArgs a) {
    extern __shared__ __attribute__((aligned(16))) char tsm_raw[];
    T* At = reinterpret_cast<T*>(tsm_raw);                   // [ncomp][K][Np]
    T* Bt = At + a.At_d;                                     // [TE][KP]
    T* Jt = Bt + (int64_t)a.TE * a.KP;                       // [ndim * ndim][TE]  (grad, div)
    const T* const gJ = static_cast<const T*>(a.J);
    const T* const gA = static_cast<const T*>(a.A);
    const int tid = threadIdx.x;
    const int Np = a.Np, K = a.K, KP = a.KP, TE = a.TE, nd = a.ndim;
    const int64_t E = a.E;

    // ---- operator -> LDS, transposed to [c][k][i]
    {
        int sF = 0, sI = 0, sJ = 0;   // face-mass operator strides of (f, i, j)
        if (a.family == FE_FAMILY_FACEMASS) {
            sF = a.rlayout == 0 ? Np * a.Nfp : a.rlayout == 1 ? a.Nfp : a.rlayout == 2 ? a.Nfp * Np : Np;
            sI = a.rlayout == 0 ? a.Nfp : a.rlayout == 1 ? a.nf * a.Nfp : 1;
            sJ = a.rlayout == 0 || a.rlayout == 1 ? 1 : a.rlayout == 2 ? Np : a.nf * Np;
        }
        const int total = a.ncomp * K * Np;
        for (int idx = tid; idx < total; idx += kTiledThreads) {
            const int c = idx / (K * Np), rem = idx - c * (K * Np);
            const int k = rem / Np, i = rem - k * Np;
            int64_t src;
            if (a.family == FE_FAMILY_FACEMASS) {
                const int f = k / a.Nfp, j = k - f * a.Nfp;
                src = (int64_t)f * sF + (int64_t)i * sI + (int64_t)j * sJ;
            } else {
                // component r and column j of D[r][i][j] (grad: r = c; div: k = r Np + j; operator: r = 0)
                const int r = a.family == FE_FAMILY_GRAD ? c : k / Np, j = a.family == FE_FAMILY_GRAD ? k : k % Np;
                src = a.opT ? ((int64_t)r * Np + j) * Np + i : ((int64_t)r * Np + i) * Np + j;
            }
            At[idx] = gA[src];
        }
    }

    const int gi = tid % a.IW, eg = tid / a.IW;   // first row and element group of this thread
    const bool is_grad = a.family == FE_FAMILY_GRAD;
    const bool worker = eg < a.neg;
    const int64_t nTiles = (E + TE - 1) / TE;
    for (int64_t tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
        const int64_t e0 = tile * TE;
        const int te = (int)(E - e0 < TE ? E - e0 : TE);   // elements of this tile
        __syncthreads();   // operator staged / previous tile's J and B no longer read
        // ---- geometry factors of the tile (grad epilogue, div B factor): Jt[x nd + r][e]
        if (a.family == FE_FAMILY_GRAD || a.family == FE_FAMILY_DIV)
            for (int idx = tid; idx < nd * nd * TE; idx += kTiledThreads) {
                const int row = idx / TE, e = idx - row * TE;
                Jt[idx] = e < te ? gJ[(int64_t)row * E + e0 + e] : T(0);
            }
        for (int fk = 0; fk < a.nb; ++fk) {
            const T* __restrict__ in = reinterpret_cast<const T*>(field_in(a.P, fk));
            T* __restrict__ out = reinterpret_cast<T*>(field_out(a.P, fk));
            __syncthreads();   // Jt ready; previous field's B no longer read
            // ---- B tile
            if (a.family == FE_FAMILY_FACEMASS) {
                const int per_f = TE * a.Nfp;
                for (int idx = tid; idx < a.nf * per_f; idx += kTiledThreads) {
                    const int f = idx / per_f, rem = idx - f * per_f;
                    const int e = rem / a.Nfp, j = rem - e * a.Nfp;
                    T v = T(0);
                    if (e < te) {
                        const T jf = a.jlayout ? gJ[(int64_t)f * E + e0 + e] : gJ[(e0 + e) * a.nf + f];
                        v = jf * in[((int64_t)f * E + e0 + e) * a.Nfp + j];
                    }
                    Bt[e * KP + f * a.Nfp + j] = v;
                }
            } else {
                for (int idx = tid; idx < TE * Np; idx += kTiledThreads) {
                    const int e = idx / Np, j = idx - e * Np;
                    const bool live = e < te;
                    if (a.family == FE_FAMILY_DIV) {
                        T ju[3] = {T(0), T(0), T(0)};
                        for (int x = 0; x < nd; ++x) {
                            const T ux = live ? in[((int64_t)x * E + e0 + e) * Np + j] : T(0);
                            for (int r = 0; r < nd; ++r) ju[r] += Jt[(x * nd + r) * TE + e] * ux;
                        }
                        for (int r = 0; r < nd; ++r) Bt[e * KP + r * Np + j] = ju[r];
                    } else {
                        const T ue = live ? in[(e0 + e) * Np + j] : T(0);
                        if (a.family == FE_FAMILY_DIVCOMP) {
                            for (int s = 0; s < nd; ++s) {
                                const T js = !live ? T(0) : a.jlayout ? gJ[(e0 + e) * nd + s] : gJ[(int64_t)s * E + e0 + e];
                                Bt[e * KP + s * Np + j] = js * ue;
                            }
                        } else if (a.family == FE_FAMILY_MATAPPLY) {
                            Bt[e * KP + j] = (gJ && live) ? gJ[e0 + e] * ue : ue;
                        } else {
                            Bt[e * KP + j] = ue;   // grad
                        }
                    }
                }
            }
            __syncthreads();
            // ---- out[e, i] = sum_k A_c[i, k] B[e, k]
            if (worker) {
                T acc[kTiledRows][EB];
                const T* ap = At + gi;
                const T* bp = Bt + (int64_t)eg * EB * KP;
                if (!is_grad || a.ncomp == 3) tiled_gemm<3, EB>(ap, is_grad ? K * Np : a.IW, bp, K, Np, KP, acc);
                else if (a.ncomp == 2) tiled_gemm<2, EB>(ap, K * Np, bp, K, Np, KP, acc);
                else tiled_gemm<1, EB>(ap, K * Np, bp, K, Np, KP, acc);
                // ---- epilogue
#pragma unroll
                for (int b = 0; b < EB; ++b) {
                    const int el = eg * EB + b;
                    if (el >= te) continue;
                    const int64_t e = e0 + el;
                    if (is_grad) {
                        for (int x = 0; x < nd; ++x) {
                            T v = Jt[(x * nd + 0) * TE + el] * acc[0][b];
                            if (nd > 1) v += Jt[(x * nd + 1) * TE + el] * acc[1][b];
                            if (nd > 2) v += Jt[(x * nd + 2) * TE + el] * acc[2][b];
                            out[((int64_t)x * E + e) * Np + gi] = v;
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < kTiledRows; ++m)
                            if (gi + m * a.IW < Np) out[e * Np + gi + m * a.IW] = acc[m][b];
                    }
                }
            }
        }
    }
}

}  // namespace fe
