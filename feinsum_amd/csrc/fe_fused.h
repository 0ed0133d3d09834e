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

// The grad section of a prepared D buffer -- formed only in the prepared instantiations (no arithmetic on a null pointer
// in the plain ones, whose `prep` argument is null).
template <bool kPrep>
__device__ __forceinline__ const void* prepared_grad_section(const void* prep) {
    if constexpr (kPrep) return reinterpret_cast<const char*>(prep) + kPrepGradOff;
    else return nullptr;
}

// between two bodies: nothing of the finished body may still be in flight towards LDS, and
// every wave of the block must be done with its private buffers before the next body stages
// its operator over them
__device__ __forceinline__ void body_boundary() {
    wait_vmcnt<0>();
    __syncthreads();
}

template <int A, int B>
constexpr int cmax() { return A > B ? A : B; }

// The div body of a fused launch.  (Its interleaved form -- fe_div.h, kIlv -- was measured here in round 5 and buys nothing beside
// a grad body on the same CU: profiles/r05/fused_interleave_ab.txt.)
template <int NP, int MD, bool kPrep, bool kDyn>
__device__ __forceinline__ void fused_div_body(const double* __restrict__ J, const double* __restrict__ D, const void* __restrict__ prep,
                                               const FieldPtrs& Pd, int64_t E, int64_t nTilesD, int op, unsigned* tail_d, int64_t static_d) {
    div3d_mfma_body<NP, MD, 0, 0, 3, false, false, kPrep, kDyn>(J, D, prep, Pd, 1, E, nTilesD, op, 0, blockIdx.x, gridDim.x, nullptr, tail_d,
                                                                static_d);
}

// div then grad, sharing J and D (BASELINE config 3).
template <int NP, int MG, int MD>
struct GradDivGeom {
    static constexpr int LDS_BYTES = cmax<GradGeom<NP, MG>::LDS_BYTES, DivGeom<NP, MD>::LDS_BYTES>();
};

// The dynamic walk of a fused launch (fe_common.h): every body has its own set of ticket counters (kTailWords apart: div,
// grad, lift) and its own number of statically walked tiles; a null `tail` is the static walk.
struct FusedTail {
    unsigned* tail;
    int64_t static_d, static_g, static_f;
};
__device__ __forceinline__ unsigned* fused_tail_set(const FusedTail& t, int set) { return t.tail ? t.tail + set * kTailWords : nullptr; }

// kPrep: both bodies take their A fragments from the prepared operator `prep` (fe_prepare_operator).
// kDyn (plain operators, p = 4): the bodies walk dynamically when `ft.tail` is given.
template <int NP, int MG, int MD, bool kPrep = false, bool kDyn = false>
__global__ __launch_bounds__(256, 2) FE_TAIL_KERNEL_ATTR void graddiv3d_mfma_kernel(
    const double* __restrict__ J, const double* __restrict__ D, const void* __restrict__ prep, GradFields Pg,
    FieldPtrs Pd, int64_t E, int64_t nTilesG, int64_t nTilesD, int opT, FusedTail ft) {
    unsigned* const tail_d = fused_tail_set(ft, 0);
    unsigned* const tail_g = fused_tail_set(ft, 1);
    // Body order (opT bits 8, 9; kFusedOrder below is what the launchers pass).  All blocks running div, then grad puts a
    // read-heavy phase (div: 76 % of its bytes are reads) in front of a write-heavy one (grad: 70 % writes); with the younger
    // half of the grid running grad FIRST each CU holds one block of either kind and the device sees the blend all the
    // time.  bit 8: the younger half swaps, bit 9: the odd blocks swap (measured: tools/fused_order_ab.py).
    // (Two straight-line sequences, not a loop over the bodies: in a loop the register allocator keeps state of one
    // body alive across the other and spills -- 408 bytes of scratch per lane and 4.7 % more HBM traffic, measured.)
    const bool swap = ((opT & 256) && blockIdx.x >= gridDim.x / 2) || ((opT & 512) && (blockIdx.x & 1));
    const int op = opT & (1 | kOpLoadsTemporal);
#ifdef FE_EXPERIMENTS
    if ((opT & 1024) && gridDim.x >= 2 && (gridDim.x & 1) == 0) {
        // ROLE SPLIT (bit 10; experiment build only -- measured in round 3 and 0.5 % slower): the older half of the grid runs div over ALL tiles,
        // the younger half grad over all tiles, both in the same tile order -- block b and block b + half lie on the same
        // XCD (blocks go round the eight XCDs), so the J tile the second of them asks for can still be in that XCD's L2
        const unsigned half = gridDim.x / 2;
        if (blockIdx.x < half)
            div3d_mfma_body<NP, MD, 0, 0, 3, false, false, kPrep>(J, D, prep, Pd, 1, E, nTilesD, op, 0, blockIdx.x, half);
        else
            grad3d_mfma_body<NP, MG, 0, true, kPrep>(Pg, D, prepared_grad_section<kPrep>(prep), 1, 3, E, nTilesG, op,
                                                     blockIdx.x - half, half);
        return;
    }
#endif
    if (!swap) {
        fused_div_body<NP, MD, kPrep, kDyn>(J, D, prep, Pd, E, nTilesD, op, tail_d, ft.static_d);
        body_boundary();
        grad3d_mfma_body<NP, MG, 0, true, kPrep, kDyn>(Pg, D, prepared_grad_section<kPrep>(prep), 1, 3, E,
                                                       nTilesG, op, blockIdx.x, gridDim.x, tail_g, ft.static_g);
    } else {
        grad3d_mfma_body<NP, MG, 0, true, kPrep, kDyn>(Pg, D, prepared_grad_section<kPrep>(prep), 1, 3, E,
                                                       nTilesG, op, blockIdx.x, gridDim.x, tail_g, ft.static_g);
        body_boundary();
        fused_div_body<NP, MD, kPrep, kDyn>(J, D, prep, Pd, E, nTilesD, op, tail_d, ft.static_d);
    }
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
    int order;           // 0: every block div, grad, lift; 3: the younger half of the grid grad, div, lift
    int load_flags;      // 0 or kOpLoadsTemporal (fe_common.h): the launch's inputs fit the Infinity Cache
};

template <int NP, int NFP, int MG, int MD, int MF, int NB, bool kPrep = false, bool kDyn = false>
__global__ __launch_bounds__(256, 2) FE_TAIL_KERNEL_ATTR void waveop3d_mfma_kernel(WaveOpArgs a, GradFields Pg, FieldPtrs Pd,
                                                                                   FieldPtrs Pf, FusedTail ft) {
    unsigned* const tail_d = fused_tail_set(ft, 0);
    unsigned* const tail_g = fused_tail_set(ft, 1);
    unsigned* const tail_f = fused_tail_set(ft, 2);
    constexpr bool kDynF = kDyn && NB >= 3;   // the lift's tickets are asked for at unit 0 and read at unit NB - 2
    // order 3: the younger half of the grid runs grad before div (see graddiv3d_mfma_kernel); the lift comes last everywhere
    const bool swap = a.order == 3 && blockIdx.x >= gridDim.x / 2;
    if (!swap) {
        fused_div_body<NP, MD, kPrep, kDyn>(a.J, a.D, a.prepD, Pd, a.E, a.nTilesD, a.load_flags, tail_d, ft.static_d);
        body_boundary();
        grad3d_mfma_body<NP, MG, 0, true, kPrep, kDyn>(Pg, a.D, prepared_grad_section<kPrep>(a.prepD), 1, 3, a.E,
                                                       a.nTilesG, a.load_flags, blockIdx.x, gridDim.x, tail_g, ft.static_g);
    } else {
        grad3d_mfma_body<NP, MG, 0, true, kPrep, kDyn>(Pg, a.D, prepared_grad_section<kPrep>(a.prepD), 1, 3, a.E,
                                                       a.nTilesG, a.load_flags, blockIdx.x, gridDim.x, tail_g, ft.static_g);
        body_boundary();
        fused_div_body<NP, MD, kPrep, kDyn>(a.J, a.D, a.prepD, Pd, a.E, a.nTilesD, a.load_flags, tail_d, ft.static_d);
    }
    body_boundary();
    facemass_mfma_body<NP, NFP, MF, NB, kFmNf, false, false, kPrep, kDynF>(a.Jf, a.R, a.prepR, Pf, a.E, a.nTilesF, a.jfe | a.load_flags,
                                                                           a.rlayout, blockIdx.x, gridDim.x, tail_f, ft.static_f);
}

}  // namespace fe
