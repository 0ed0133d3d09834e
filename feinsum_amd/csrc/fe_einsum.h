// fe_einsum.h -- generic einsum kernel: any explicit-mode einsum, one thread
// per output entry, sequential loop over the flattened summation space.
// It is the device restatement of the single-instruction loop nest feinsum's
// generate_loopy emits for the trivial schedule (codegen/loopy.py:242-305):
//   out[o...] = sum_{s...} prod_p operand_p[o..., s...]
// and exists so that every BatchedEinsum the builders accept can be validated
// and timed through the same boundary; the DG families never take this path.
#pragma once
#include "../../include/feinsum_hip.h"
#include "fe_common.h"

namespace fe {

template <typename T>
__global__ __launch_bounds__(256) void einsum_generic_kernel(fe_einsum_desc d, fe_einsum_ptrs ops,
                                                             T* __restrict__ out,
                                                             int64_t n_out_entries, int64_t n_sum_points) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_out_entries) return;
    // output multi-index (last index fastest) -> base offset of every operand
    int64_t base[FE_MAX_EINSUM_OPERANDS];
    for (int p = 0; p < d.n_operands; ++p) base[p] = 0;
    int64_t rem = idx;
    for (int k = d.n_out - 1; k >= 0; --k) {
        const int64_t ok = rem % d.out_extent[k];
        rem /= d.out_extent[k];
        for (int p = 0; p < d.n_operands; ++p) base[p] += ok * d.op_out_stride[p][k];
    }
    T acc = T(0);
    int64_t sidx[FE_MAX_EINSUM_INDICES];
    for (int k = 0; k < d.n_sum; ++k) sidx[k] = 0;
    for (int64_t s = 0; s < n_sum_points; ++s) {
        T prod = T(1);
        for (int p = 0; p < d.n_operands; ++p) {
            int64_t off = base[p];
            for (int k = 0; k < d.n_sum; ++k) off += sidx[k] * d.op_sum_stride[p][k];
            prod *= static_cast<const T*>(ops.p[p])[off];
        }
        acc += prod;
        for (int k = d.n_sum - 1; k >= 0; --k) {   // odometer, last summation index fastest
            if (++sidx[k] < d.sum_extent[k]) break;
            sidx[k] = 0;
        }
    }
    out[idx] = acc;
}

}  // namespace fe
