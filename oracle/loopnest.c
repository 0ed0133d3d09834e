/*
 * loopnest.c -- plain-C restatement of the DG-wave einsum loop nests.
 * TEST INFRASTRUCTURE ONLY (oracle + CPU baseline for bench.py); the product
 * library never links or calls this.
 *
 * "trivial" functions follow the single-statement nest that feinsum's
 * generate_loopy emits for the trivial schedule -- what the reference executes on
 * a CPU OpenCL device under the identity transform used by its own tests
 * (reference: src/feinsum/codegen/loopy.py:242-305; test/test_codegen.py:115-120):
 *     out[free] = sum_{summed} prod_k arg_k[...]
 * "hoisted" functions follow the opt_einsum-optimal 2-step schedule whose flop
 * count defines GFLOP/s (reference: src/feinsum/measure.py:278-331;
 * test/test_loopy_utils.py:231-271: 33075 -> 7980 flops per element).
 * OpenMP over elements when compiled with -fopenmp.
 */
#include <stdint.h>
#include <stdlib.h>

/* grad: out[x,e,i] = sum_{r,j} J[x,r,e] D[r,i,j] u[e,j] */
void oracle_grad3d_trivial(const double* J, const double* D, const double* u, double* out,
                           int64_t E, int Np) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e)
        for (int x = 0; x < 3; ++x)
            for (int i = 0; i < Np; ++i) {
                double acc = 0.0;
                for (int r = 0; r < 3; ++r)
                    for (int j = 0; j < Np; ++j)
                        acc += J[(int64_t)(x * 3 + r) * E + e] * D[(r * Np + i) * Np + j] * u[e * Np + j];
                out[((int64_t)x * E + e) * Np + i] = acc;
            }
}

void oracle_grad3d_hoisted(const double* J, const double* D, const double* u, double* out,
                           int64_t E, int Np) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e) {
        double tmp[3 * 64];   /* Np <= 64 */
        for (int r = 0; r < 3; ++r)
            for (int i = 0; i < Np; ++i) {
                double acc = 0.0;
                for (int j = 0; j < Np; ++j) acc += D[(r * Np + i) * Np + j] * u[e * Np + j];
                tmp[r * Np + i] = acc;
            }
        for (int x = 0; x < 3; ++x)
            for (int i = 0; i < Np; ++i) {
                double acc = 0.0;
                for (int r = 0; r < 3; ++r) acc += J[(int64_t)(x * 3 + r) * E + e] * tmp[r * Np + i];
                out[((int64_t)x * E + e) * Np + i] = acc;
            }
    }
}

/* div: out[e,i] = sum_{x,r,j} J[x,r,e] D[r,i,j] u[x,e,j] */
void oracle_div3d_trivial(const double* J, const double* D, const double* u, double* out,
                          int64_t E, int Np) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e)
        for (int i = 0; i < Np; ++i) {
            double acc = 0.0;
            for (int x = 0; x < 3; ++x)
                for (int r = 0; r < 3; ++r)
                    for (int j = 0; j < Np; ++j)
                        acc += J[(int64_t)(x * 3 + r) * E + e] * D[(r * Np + i) * Np + j] *
                               u[((int64_t)x * E + e) * Np + j];
            out[e * Np + i] = acc;
        }
}

void oracle_div3d_hoisted(const double* J, const double* D, const double* u, double* out,
                          int64_t E, int Np) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e) {
        double ju[3 * 64];
        for (int r = 0; r < 3; ++r)
            for (int j = 0; j < Np; ++j) {
                double acc = 0.0;
                for (int x = 0; x < 3; ++x)
                    acc += J[(int64_t)(x * 3 + r) * E + e] * u[((int64_t)x * E + e) * Np + j];
                ju[r * Np + j] = acc;
            }
        for (int i = 0; i < Np; ++i) {
            double acc = 0.0;
            for (int r = 0; r < 3; ++r)
                for (int j = 0; j < Np; ++j) acc += D[(r * Np + i) * Np + j] * ju[r * Np + j];
            out[e * Np + i] = acc;
        }
    }
}

/* face-mass, one field: out[e,i] = sum_{f,j} J[e,f] R[f,i,j] v[f,e,j]
 * jfe: J stored [nf][E] instead of [E][nf];  rifj: R stored [Np][nf][Nfp]. */
void oracle_facemass_trivial(const double* J, const double* R, const double* v, double* out,
                             int64_t E, int Np, int nf, int Nfp, int jfe, int rifj) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e)
        for (int i = 0; i < Np; ++i) {
            double acc = 0.0;
            for (int f = 0; f < nf; ++f)
                for (int j = 0; j < Nfp; ++j) {
                    const double jv = jfe ? J[(int64_t)f * E + e] : J[e * nf + f];
                    const double rv = rifj ? R[(i * nf + f) * Nfp + j] : R[(f * Np + i) * Nfp + j];
                    acc += jv * rv * v[((int64_t)f * E + e) * Nfp + j];
                }
            out[e * Np + i] = acc;
        }
}

void oracle_facemass_hoisted(const double* J, const double* R, const double* v, double* out,
                             int64_t E, int Np, int nf, int Nfp, int jfe, int rifj) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < E; ++e) {
        double jvv[8 * 32];   /* nf <= 8, Nfp <= 32 */
        for (int f = 0; f < nf; ++f) {
            const double jv = jfe ? J[(int64_t)f * E + e] : J[e * nf + f];
            for (int j = 0; j < Nfp; ++j) jvv[f * Nfp + j] = jv * v[((int64_t)f * E + e) * Nfp + j];
        }
        for (int i = 0; i < Np; ++i) {
            double acc = 0.0;
            for (int f = 0; f < nf; ++f)
                for (int j = 0; j < Nfp; ++j) {
                    const double rv = rifj ? R[(i * nf + f) * Nfp + j] : R[(f * Np + i) * Nfp + j];
                    acc += rv * jvv[f * Nfp + j];
                }
            out[e * Np + i] = acc;
        }
    }
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
