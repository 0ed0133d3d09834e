// Several einsums of the DG wave operator in ONE persistent launch.
//
// The reference runs div, grad and lift as three kernels separated by global barriers
// (examples/wave_3d_p4_auto.py:16-63; examples/dg_wave_div.py + dg_wave_grad.py for the pair);
// the three have no data dependence on each other -- the barriers are only launch boundaries.
// Here every block of the persistent grid runs the bodies in turn over its own tiles: a block
// that has finished its div tiles goes straight on to its grad tiles while slower blocks are
// still in div, so the ramp-down of one einsum overlaps the ramp-up of the next instead of
// idling the chip between launches.  Registers and LDS are the maximum over the bodies, not
// the sum (the operator fragments of a finished body are dead).
#pragma once

#include "fe_div.h"
#include "fe_facemass.h"
#include "fe_grad.h"

namespace fe {

// between two bodies: nothing of the finished body may still be in flight towards LDS, and
// every wave of the block must be done with its private buffers before the next body stages
// its operator over them
__device__ __forceinline__ void body_boundary() {
    wait_vmcnt<0>();
    __syncthreads();
}

template <int A, int B>
constexpr int cmax() { return A > B ? A : B; }

// div then grad, sharing J and D (BASELINE config 3).
template <int NP, int MG, int MD>
struct GradDivGeom {
    static constexpr int LDS_BYTES = cmax<GradGeom<NP, MG>::LDS_BYTES, DivGeom<NP, MD>::LDS_BYTES>();
};

// kPrep: both bodies take their A fragments from the prepared operator `prep` (fe_prepare_operator).
template <int NP, int MG, int MD, bool kPrep = false>
__global__ __launch_bounds__(256, 2) void graddiv3d_mfma_kernel(
    const double* __restrict__ J, const double* __restrict__ D, const void* __restrict__ prep, GradFields Pg,
    FieldPtrs Pd, int64_t E, int64_t nTilesG, int64_t nTilesD, int opT) {
    div3d_mfma_body<NP, MD, 0, 0, 3, false, false, kPrep>(J, D, prep, Pd, 1, E, nTilesD, opT, 0, blockIdx.x, gridDim.x);
    body_boundary();
    grad3d_mfma_body<NP, MG, 0, true, kPrep>(Pg, D, reinterpret_cast<const char*>(prep) + kPrepGradOff, 1, 3, E,
                                             nTilesG, opT, blockIdx.x, gridDim.x);
}

// div, grad and lift (face-mass x NB) of one time-step stage.
template <int NP, int NFP, int MG, int MD, int MF>
struct WaveOpGeom {
    static constexpr int LDS_BYTES =
        cmax<cmax<GradGeom<NP, MG>::LDS_BYTES, DivGeom<NP, MD>::LDS_BYTES>(), FmGeom<NP, NFP, MF>::LDS_BYTES>();
};

struct WaveOpArgs {
    const double* J;     // [3][3][E]
    const double* D;     // [3][Np][Np]
    const double* Jf;    // face-mass J
    const double* R;     // face-mass operator
    const void* prepD;   // prepared D (grad and div sections) and prepared R, or null
    const void* prepR;
    int64_t E, nTilesG, nTilesD, nTilesF;
    int jfe, rlayout;
};

template <int NP, int NFP, int MG, int MD, int MF, int NB, bool kPrep = false>
__global__ __launch_bounds__(256, 2) void waveop3d_mfma_kernel(WaveOpArgs a, GradFields Pg, FieldPtrs Pd,
                                                               FieldPtrs Pf) {
    div3d_mfma_body<NP, MD, 0, 0, 3, false, false, kPrep>(a.J, a.D, a.prepD, Pd, 1, a.E, a.nTilesD, 0, 0, blockIdx.x,
                                                          gridDim.x);
    body_boundary();
    grad3d_mfma_body<NP, MG, 0, true, kPrep>(Pg, a.D, reinterpret_cast<const char*>(a.prepD) + kPrepGradOff, 1, 3, a.E,
                                             a.nTilesG, 0, blockIdx.x, gridDim.x);
    body_boundary();
    facemass_mfma_body<NP, NFP, MF, NB, kFmNf, false, false, kPrep>(a.Jf, a.R, a.prepR, Pf, a.E, a.nTilesF, a.jfe,
                                                                    a.rlayout, blockIdx.x, gridDim.x);
}

}  // namespace fe
