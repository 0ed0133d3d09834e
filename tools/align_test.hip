// align_test.hip -- do global_load_lds_dwordx4 / global_store_dwordx4 accept 8-byte
// (not 16-byte) aligned GLOBAL addresses on gfx950?  (decides how odd-E planes are handled)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../feinsum_amd/csrc/fe_common.h"
__global__ void k(const double* src, double* dst, int n16) {
    __shared__ __attribute__((aligned(16))) double lds[128];
    const int lane = threadIdx.x;
    fe::glds16(reinterpret_cast<const char*>(src) + lane * 16, fe::lds_addr_uniform(lds));
    fe::wait_vmcnt<0>();
    fe::wave_lds_fence();
    fe::v2d v = *reinterpret_cast<fe::v2d*>(&lds[2 * lane]);
    *reinterpret_cast<fe::v2d*>(reinterpret_cast<char*>(dst) + lane * 16) = v;
}
int main() {
    std::vector<double> h(256), o(256, -1);
    for (int i = 0; i < 256; ++i) h[i] = i + 0.5;
    double *ds, *dd;
    hipMalloc(&ds, 4096); hipMalloc(&dd, 4096);
    hipMemcpy(ds, h.data(), 2048, hipMemcpyHostToDevice);
    for (int off = 0; off < 2; ++off) {
        hipMemset(dd, 0, 4096);
        k<<<1, 64>>>(ds + off, dd + off, 64);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(o.data(), dd, 2048, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = off; i < off + 128; ++i) bad += o[i] != h[i];
        printf("offset %d doubles (%s-aligned): %s, %d mismatches\n", off, off ? "8B" : "16B", hipGetErrorString(e), bad);
    }
    return 0;
}
