// p5sw.hip -- stand-alone driver of tools/p5sw_kernel.h (grad p = 5, one wave per SIMD): check against a
// plain kernel, time.     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/p5sw.hip -o build/p5sw && build/p5sw [E]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <string>
#include "p5c_kernel.h"
#include "../include/feinsum_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill(double* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (x & 0xFFFFFF) * (1.0 / 16777216.0);
    }
}
__global__ void ref_kernel(const double* J, const double* D, const double* u, double* out, long E, long e_begin, long e_end) {
    const long idx = blockIdx.x * 256L + threadIdx.x;
    const long e = e_begin + idx / 56; const int i = idx % 56;
    if (e >= e_end) return;
    double t[3] = {0, 0, 0};
    for (int r = 0; r < 3; ++r) for (int j = 0; j < 56; ++j) t[r] += D[(r * 56 + i) * 56 + j] * u[e * 56 + j];
    for (int x = 0; x < 3; ++x)
        out[((long)x * E + e) * 56 + i] = J[(x * 3 + 0) * E + e] * t[0] + J[(x * 3 + 1) * E + e] * t[1] + J[(x * 3 + 2) * E + e] * t[2];
}

int main(int argc, char** argv) {
    const long E = argc > 1 ? atol(argv[1]) : 1000000;
    const int variant = argc > 2 ? atoi(argv[2]) : 0;
    using G = fe::GradP5cGeom;
    const long nTiles = E / 16;
    double *J, *D, *u, *out, *ref;
    CK(hipMalloc(&J, 9 * E * 8)); CK(hipMalloc(&D, 3 * 56 * 56 * 8)); CK(hipMalloc(&u, E * 56 * 8));
    const bool split = argc > 3 && std::string(argv[3]) == "split";
    if (split) { void* p = nullptr; if (fe_split_alloc(&p, 3 * E * 56 * 8, 0) != 0) { printf("fe_split_alloc: %s\n", fe_last_error()); return 1; } out = (double*)p; }
    else CK(hipMalloc(&out, 3 * E * 56 * 8));
    CK(hipMalloc(&ref, 3 * E * 56 * 8));
    fill<<<1024, 256>>>(J, 9 * E, 1); fill<<<64, 256>>>(D, 3 * 56 * 56, 2); fill<<<1024, 256>>>(u, E * 56, 3);
    CK(hipMemset(out, 0, 3 * E * 56 * 8));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fe::grad_p5c_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fe::grad_p5c_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(fe::grad_p5c_kernel<0>)));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fe::grad_p5c_kernel<0>, 256, G::LDS_BYTES));
    printf("# grad_p5c_kernel: %d VGPRs (arch + acc), scratch %zu B, LDS %d B, blocks/CU %d\n", fa.numRegs, (size_t)fa.localSizeBytes, G::LDS_BYTES, occ);
    const int blocks = (int)std::min<long>(256, (nTiles + 3) / 4);
#define P5C_CASE(V) case V: CK_ATTR(V); hipLaunchKernelGGL(fe::grad_p5c_kernel<V>, dim3(blocks), dim3(256), G::LDS_BYTES, 0, J, D, u, out, E, nTiles, 0); break;
#define CK_ATTR(V) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fe::grad_p5c_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES)
    auto go = [&]() {
        switch (variant) { P5C_CASE(1) P5C_CASE(2) P5C_CASE(3) P5C_CASE(8) P5C_CASE(9) P5C_CASE(10) P5C_CASE(11) default: hipLaunchKernelGGL(fe::grad_p5c_kernel<0>, dim3(blocks), dim3(256), G::LDS_BYTES, 0, J, D, u, out, E, nTiles, 0); }
    };
    go();
    CK(hipDeviceSynchronize());
    // check the first, a middle and the last 2048 full-tile elements
    const long covered = nTiles * 16;
    double worst = 0;
    for (long e0 : {0L, std::max(0L, covered / 2 - 1024), std::max(0L, covered - 2048)}) {
        const long e1 = std::min(covered, e0 + 2048);
        ref_kernel<<<(unsigned)(((e1 - e0) * 56 + 255) / 256), 256>>>(J, D, u, ref, E, e0, e1);
        CK(hipDeviceSynchronize());
        std::vector<double> a((e1 - e0) * 56), b((e1 - e0) * 56);
        for (int x = 0; x < 3; ++x) {
            CK(hipMemcpy(a.data(), out + ((long)x * E + e0) * 56, a.size() * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), ref + ((long)x * E + e0) * 56, b.size() * 8, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < a.size(); ++k) worst = std::max(worst, std::fabs(a[k] - b[k]) / (1.0 + std::fabs(b[k])));
        }
    }
    printf("max relative difference against the plain kernel (3 x 2048 elements): %.3g\n", worst);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        {   // the shipped eight-wave kernel (A fragments in LDS, two waves per SIMD) on the same arrays
            for (int i = 0; i < 10; ++i) if (fe_grad3d_f64(J, D, u, out, E, 56, 0, nullptr) != 0) { printf("fe_grad3d_f64: %s\n", fe_last_error()); return 1; }
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) fe_grad3d_f64(J, D, u, out, E, 56, 0, nullptr);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float msw; CK(hipEventElapsedTime(&msw, e0, e1)); msw /= 20;
            printf("library kernel (eight waves per block, fragments in LDS), %s output array: %.4f ms  %.1f TFLOP/s\n", split ? "split-allocator" : "hipMalloc", msw,
                   (2.0 * 3 * 56 * 56 + 18.0 * 56) * E / msw * 1e-9);
        }
        for (int i = 0; i < 10; ++i) go();
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) go();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
        const double flops = (2.0 * 3 * 56 * 56 + 18.0 * 56) * E;
        unsigned long long clk[12];
        CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(fe::p5c_clock), sizeof clk));
        printf("[variant %d: loop of one wave: %llu tiles, %.0f cycles per tile at %.0f MHz] ", variant, clk[2], (double)clk[0] / clk[2], (double)clk[0] / clk[1] * 100.0);
        printf("phases/tile: rt0 %.0f | rt1 %.0f | rt2 %.0f | small %.0f | operands %.0f || ", (double)clk[5] / clk[2], (double)clk[6] / clk[2], (double)clk[7] / clk[2], (double)clk[8] / clk[2], (double)clk[9] / clk[2]);
        printf("E = %ld: %.4f ms  %.1f TFLOP/s (%.1f %% of 78.6)  %.0f GB/s\n", E, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 78.6 * 100,
               (9.0 + 56 + 168) * 8 * E / ms * 1e-6);
    }
    return 0;
}
