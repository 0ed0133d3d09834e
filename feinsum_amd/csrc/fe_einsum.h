// fe_einsum.h -- generic einsum kernel: any explicit-mode einsum, one thread (or a group of
// G lanes) per output entry, loop over the flattened summation space.
// It is the device restatement of the single-instruction loop nest feinsum's
// generate_loopy emits for the trivial schedule (codegen/loopy.py:242-305):
//   out[o...] = sum_{s...} prod_p operand_p[o..., s...]
// and exists so that every BatchedEinsum the builders accept can be validated
// and timed through the same boundary; the DG families never take this path.
#pragma once
#include "../../include/feinsum_hip.h"
#include "fe_common.h"

namespace fe {

// G lanes share one output entry: lane l of the group takes summation points l, l + G, ... (last
// summation index fastest) and the partial sums are combined with wave shuffles.  The host
// picks G > 1 when the fastest summation index is contiguous in an operand, so that a group reads
// consecutive addresses (one thread per output walks that operand with a stride of a whole row
// per lane: 64 cache lines per load instruction).  G = 1 is the plain restatement.
template <typename T, int G>
__global__ __launch_bounds__(256) void einsum_generic_kernel(fe_einsum_desc d, fe_einsum_ptrs ops,
                                                             T* __restrict__ out,
                                                             int64_t n_out_entries, int64_t n_sum_points) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t idx = tid / G;
    const int part = (int)(tid % G);
    const bool live = idx < n_out_entries;   // whole groups are live or dead; dead lanes still shuffle
    // output multi-index (last index fastest) -> base offset of every operand
    int64_t base[FE_MAX_EINSUM_OPERANDS];
    for (int p = 0; p < d.n_operands; ++p) base[p] = 0;
    int64_t rem = live ? idx : 0;
    for (int k = d.n_out - 1; k >= 0; --k) {
        const int64_t ok = rem % d.out_extent[k];
        rem /= d.out_extent[k];
        for (int p = 0; p < d.n_operands; ++p) base[p] += ok * d.op_out_stride[p][k];
    }
    T acc = T(0);
    int64_t sidx[FE_MAX_EINSUM_INDICES];
    for (int k = 0; k < d.n_sum; ++k) sidx[k] = 0;
    auto carry = [&]() {   // odometer, last summation index fastest
        for (int k = d.n_sum - 1; k > 0; --k)
            while (sidx[k] >= d.sum_extent[k]) {
                sidx[k] -= d.sum_extent[k];
                ++sidx[k - 1];
            }
    };
    if (d.n_sum > 0 && n_sum_points > 0) {   // (an empty summation space has zero extents: carry() would not end)
        sidx[d.n_sum - 1] = part;
        carry();
    }
    for (int64_t s = part; live && s < n_sum_points; s += G) {
        T prod = T(1);
        for (int p = 0; p < d.n_operands; ++p) {
            int64_t off = base[p];
            for (int k = 0; k < d.n_sum; ++k) off += sidx[k] * d.op_sum_stride[p][k];
            prod *= static_cast<const T*>(ops.p[p])[off];
        }
        acc += prod;
        if (d.n_sum > 0) {
            sidx[d.n_sum - 1] += G;
            carry();
        }
    }
#pragma unroll
    for (int w = G / 2; w >= 1; w /= 2) acc += __shfl_xor(acc, w, 64);
    if (live && part == 0) out[idx] = acc;
}

// Pointwise product of operands that all have the output's own contiguous layout
// ('ij,ij->ij', 'ijk,ijk->ijk': tuning/impls/ij_ij_to_ij.py, ijk_ijk_to_ijk.py): a pure
// stream, two values per lane and iteration.
template <typename T>
__global__ __launch_bounds__(256) void einsum_pointwise_kernel(fe_einsum_ptrs ops, int n_operands,
                                                               T* __restrict__ out, int64_t n) {
    typedef T v2 __attribute__((ext_vector_type(2)));
    const int64_t pairs = n / 2;
    const bool aligned = [&] {
        uintptr_t bits = reinterpret_cast<uintptr_t>(out);
        for (int p = 0; p < n_operands; ++p) bits |= reinterpret_cast<uintptr_t>(ops.p[p]);
        return (bits & (2 * sizeof(T) - 1)) == 0;
    }();
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (aligned) {
        for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < pairs; q += stride) {
            v2 acc = __builtin_nontemporal_load(static_cast<const v2*>(ops.p[0]) + q);
            for (int p = 1; p < n_operands; ++p) acc *= __builtin_nontemporal_load(static_cast<const v2*>(ops.p[p]) + q);
            __builtin_nontemporal_store(acc, reinterpret_cast<v2*>(out) + q);
        }
    }
    for (int64_t i = (aligned ? 2 * pairs : 0) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        T acc = static_cast<const T*>(ops.p[0])[i];
        for (int p = 1; p < n_operands; ++p) acc *= static_cast<const T*>(ops.p[p])[i];
        out[i] = acc;
    }
}

}  // namespace fe
