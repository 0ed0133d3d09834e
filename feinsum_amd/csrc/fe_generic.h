// fe_generic.h -- plain VALU kernels for div and face-mass, any Np / nf / Nfp.
// One thread per output entry, operands straight from global memory (L1/L2
// serve the operator matrix).  They are the on-device restatement of the loop
// nest feinsum's generate_loopy emits (codegen/loopy.py:242-305) and the path
// for shapes the MFMA kernels are not compiled for.
#pragma once
#include "fe_common.h"

namespace fe {

// div: out[e,i] = sum_{x,r,j} J[x,r,e] D[r,i,j] u[x,e,j]
__device__ __forceinline__ void div3d_item(const double* __restrict__ J, const double* __restrict__ D,
                                           const double* __restrict__ u, double* __restrict__ out, int64_t E,
                                           int Np, int64_t e, int i, int opT) {
    double jac[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) jac[k] = J[(int64_t)k * E + e];
    const double* u0 = u + ((int64_t)0 * E + e) * Np;
    const double* u1 = u + ((int64_t)1 * E + e) * Np;
    const double* u2 = u + ((int64_t)2 * E + e) * Np;
    const int si = opT ? 1 : Np, sj = opT ? Np : 1;   // opT: D stored as [r][j][i]
    const double* d0 = D + (int64_t)0 * Np * Np + (int64_t)i * si;
    const double* d1 = D + (int64_t)1 * Np * Np + (int64_t)i * si;
    const double* d2 = D + (int64_t)2 * Np * Np + (int64_t)i * si;
    double acc = 0.0;
#pragma unroll 5
    for (int j = 0; j < Np; ++j) {
        const double a = u0[j], b = u1[j], c = u2[j];
        const double ju0 = jac[0] * a + jac[3] * b + jac[6] * c;  // r = 0: sum_x J[x,0,e] u[x,e,j]
        const double ju1 = jac[1] * a + jac[4] * b + jac[7] * c;
        const double ju2 = jac[2] * a + jac[5] * b + jac[8] * c;
        acc += d0[j * sj] * ju0 + d1[j * sj] * ju1 + d2[j * sj] * ju2;
    }
    out[e * Np + i] = acc;
}

__global__ __launch_bounds__(256) void div3d_generic_kernel(
    const double* __restrict__ J, const double* __restrict__ D, const double* __restrict__ u,
    double* __restrict__ out, int64_t E, int Np, int64_t e_begin, int opT) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (E - e_begin) * Np) return;
    div3d_item(J, D, u, out, E, Np, e_begin + idx / Np, (int)(idx % Np), opT);
}

// grad / div of ND-dimensional elements (ND = 2: triangles), entry (e, i): J [ND][ND][E], D [ND][Np][Np]
__device__ __forceinline__ void grad_nd_item(const double* __restrict__ J, const double* __restrict__ D,
                                             const double* __restrict__ u, double* __restrict__ out, int64_t E,
                                             int Np, int nd, int64_t e, int i, int opT) {
    double t[3] = {0.0, 0.0, 0.0};
    const double* ue = u + e * Np;
    const int si = opT ? 1 : Np, sj = opT ? Np : 1;
#pragma unroll 5
    for (int j = 0; j < Np; ++j)
        for (int r = 0; r < nd; ++r) t[r] += D[(int64_t)r * Np * Np + (int64_t)i * si + j * sj] * ue[j];
    for (int x = 0; x < nd; ++x) {
        double v = 0.0;
        for (int r = 0; r < nd; ++r) v += J[(int64_t)(x * nd + r) * E + e] * t[r];
        out[((int64_t)x * E + e) * Np + i] = v;
    }
}

__device__ __forceinline__ void div_nd_item(const double* __restrict__ J, const double* __restrict__ D,
                                            const double* __restrict__ u, double* __restrict__ out, int64_t E,
                                            int Np, int nd, int64_t e, int i, int opT) {
    const int si = opT ? 1 : Np, sj = opT ? Np : 1;
    double acc = 0.0;
#pragma unroll 5
    for (int j = 0; j < Np; ++j)
        for (int r = 0; r < nd; ++r) {
            double ju = 0.0;
            for (int x = 0; x < nd; ++x) ju += J[(int64_t)(x * nd + r) * E + e] * u[((int64_t)x * E + e) * Np + j];
            acc += D[(int64_t)r * Np * Np + (int64_t)i * si + j * sj] * ju;
        }
    out[e * Np + i] = acc;
}

// div component: out[e,i] = sum_{s,j} J[s,e] D[s,i,j] u[e,j]   ('se,sij,ej->ei')
__device__ __forceinline__ void divcomp3d_item(const double* __restrict__ J, const double* __restrict__ D,
                                               const double* __restrict__ u, double* __restrict__ out,
                                               int64_t E, int Np, int64_t e, int i, int opT, int jes) {
    const double j0 = jes ? J[e * 3 + 0] : J[0 * E + e];
    const double j1 = jes ? J[e * 3 + 1] : J[1 * E + e];
    const double j2 = jes ? J[e * 3 + 2] : J[2 * E + e];
    const int si = opT ? 1 : Np, sj = opT ? Np : 1;
    const double* d0 = D + (int64_t)0 * Np * Np + (int64_t)i * si;
    const double* d1 = D + (int64_t)1 * Np * Np + (int64_t)i * si;
    const double* d2 = D + (int64_t)2 * Np * Np + (int64_t)i * si;
    const double* ue = u + e * Np;
    double acc = 0.0;
#pragma unroll 5
    for (int j = 0; j < Np; ++j)
        acc += (d0[j * sj] * j0 + d1[j * sj] * j1 + d2[j * sj] * j2) * ue[j];
    out[e * Np + i] = acc;
}

// the same for ND-dimensional elements (ND = 2: triangles): J [ND][E] or [E][ND], D [ND][Np][Np]
__device__ __forceinline__ void divcomp_nd_item(const double* __restrict__ J, const double* __restrict__ D,
                                                const double* __restrict__ u, double* __restrict__ out, int64_t E,
                                                int Np, int nd, int64_t e, int i, int opT, int jes) {
    const int si = opT ? 1 : Np, sj = opT ? Np : 1;
    const double* ue = u + e * Np;
    double acc = 0.0;
    for (int s = 0; s < nd; ++s) {
        const double js = jes ? J[e * nd + s] : J[(int64_t)s * E + e];
        const double* d = D + (int64_t)s * Np * Np + (int64_t)i * si;
        double t = 0.0;
#pragma unroll 5
        for (int j = 0; j < Np; ++j) t += d[j * sj] * ue[j];
        acc += js * t;
    }
    out[e * Np + i] = acc;
}

__global__ __launch_bounds__(256) void divcomp3d_generic_kernel(
    const double* __restrict__ J, const double* __restrict__ D, const double* __restrict__ u,
    double* __restrict__ out, int64_t E, int Np, int64_t e_begin, int opT, int jes) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (E - e_begin) * Np) return;
    divcomp3d_item(J, D, u, out, E, Np, e_begin + idx / Np, (int)(idx % Np), opT, jes);
}

// element-local operator: out[e,i] = (J ? J[e] : 1) * sum_j D[i,j] u[e,j]   ('e,ij,ej->ei', 'ij,ej->ei')
__device__ __forceinline__ void matapply_item(const double* __restrict__ J, const double* __restrict__ D,
                                              const double* __restrict__ u, double* __restrict__ out, int Np,
                                              int64_t e, int i, int opT) {
    const double* d = D + (opT ? i : (int64_t)i * Np);
    const int sj = opT ? Np : 1;
    const double* ue = u + e * Np;
    double acc = 0.0;
#pragma unroll 5
    for (int j = 0; j < Np; ++j) acc += d[j * sj] * ue[j];
    out[e * Np + i] = J ? J[e] * acc : acc;
}

__global__ __launch_bounds__(256) void matapply_generic_kernel(
    const double* __restrict__ J, const double* __restrict__ D, const double* __restrict__ u,
    double* __restrict__ out, int64_t E, int Np, int64_t e_begin, int opT) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (E - e_begin) * Np) return;
    matapply_item(J, D, u, out, Np, e_begin + idx / Np, (int)(idx % Np), opT);
}

// face-mass: out_k[e,i] = sum_{f,j} J[e,f] R[f,i,j] v_k[f,e,j]
//   jEs / jFs : strides of J along e and f;  rF / rI / rJ : strides of R along f, i and j
template <int NB>
__device__ __forceinline__ void facemass_item(const double* __restrict__ J, const double* __restrict__ R,
                                              const FieldPtrs& P, int64_t E, int Np, int nf, int Nfp, int64_t jEs,
                                              int64_t jFs, int rF, int rI, int rJ, int64_t e, int i) {
    double acc[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) acc[k] = 0.0;
    for (int f = 0; f < nf; ++f) {
        const double jf = J[e * jEs + f * jFs];
        const double* rr = R + (int64_t)f * rF + (int64_t)i * rI;
        const int64_t vo = ((int64_t)f * E + e) * Nfp;
#pragma unroll 5
        for (int j = 0; j < Nfp; ++j) {
            const double w = rr[j * rJ] * jf;
#pragma unroll
            for (int k = 0; k < NB; ++k) acc[k] += w * P.v[k][vo + j];
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) P.out[k][e * Np + i] = acc[k];
}

template <int NB>
__global__ __launch_bounds__(256) void facemass_generic_kernel(
    const double* __restrict__ J, const double* __restrict__ R, FieldPtrs P, int64_t E, int Np,
    int nf, int Nfp, int64_t jEs, int64_t jFs, int rF, int rI, int rJ, int64_t e_begin) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (E - e_begin) * Np) return;
    facemass_item<NB>(J, R, P, E, Np, nf, Nfp, jEs, jFs, rF, rI, rJ, e_begin + idx / Np, (int)(idx % Np));
}

}  // namespace fe
