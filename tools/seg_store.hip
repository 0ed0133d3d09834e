// seg_store.hip -- (1) what v_permlane16_swap_b32 does on gfx950; (2) write bandwidth of the p = 5 grad output
// pattern when a wave stores its [16][56] tile of each of three planes as 7 instructions of 16 rows x 64 B
// (accumulator layout paired up with lane swaps, no LDS) against 7 instructions of 1 KB contiguous (LDS-transposed).
//   hipcc --offload-arch=gfx950 -O3 tools/seg_store.hip -o build/seg_store && build/seg_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ void swap_probe(unsigned* out) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(1000 + l, 2000 + l, false, false);
    out[l] = r[0];
    out[64 + l] = r[1];
}

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(double* out, long E, long nTiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, n = lane & 15;
    const long stride = (long)gridDim.x * 8;
    for (long tile = (long)blockIdx.x * 8 + wave; tile < nTiles; tile += stride) {
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            double* op = out + ((long)x * E + tile * 16) * 56;
#pragma unroll
            for (int m = 0; m < 7; ++m) {
                const v2d val = {(double)tile + m, (double)lane};
                if (MODE == 0) {          // 1 KB contiguous per instruction
                    __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + 2 * (m * 64 + lane)));
                } else if (MODE == 1) {   // 16 rows x 64 B
                    const int col = 8 * m + ((g & 1) ? 4 + (g - 1) : g);
                    __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + n * 56 + col));
                } else if (m < 6) {       // 8 rows x 128 B
                    const int row = (lane & 7) + 8 * (m & 1), col = 16 * (m >> 1) + 2 * (lane >> 3);
                    __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + row * 56 + col));
                } else {                  // the last 8 columns: 16 rows x 64 B
                    __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(op + n * 56 + 48 + 2 * g));
                }
            }
        }
    }
}

int main() {
    unsigned* d;
    CK(hipMalloc(&d, 128 * 4));
    swap_probe<<<1, 64>>>(d);
    unsigned h[128];
    CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("permlane16_swap(a = 1000 + lane, b = 2000 + lane):\n r[0]:");
    for (int l = 0; l < 64; l += 8) printf(" %u", h[l]);
    printf("\n r[1]:");
    for (int l = 0; l < 64; l += 8) printf(" %u", h[64 + l]);
    printf("\n");
    const long E = 1000000, nTiles = E / 16;
    double* out;
    CK(hipMalloc(&out, 3 * E * 56 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            auto go = [&]() {
                if (mode == 0) store_kernel<0><<<256, 512>>>(out, E, nTiles);
                else if (mode == 1) store_kernel<1><<<256, 512>>>(out, E, nTiles);
                else store_kernel<2><<<256, 512>>>(out, E, nTiles);
            };
            for (int i = 0; i < 5; ++i) go();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < 20; ++i) go();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 20;
            printf("%s: %.4f ms  %.0f GB/s\n", mode == 0 ? "1 KB contiguous per instruction" : mode == 1 ? "16 rows x 64 B per instruction  " : "8 rows x 128 B (+ 1 of 7: 64 B)", ms,
                   3.0 * E * 56 * 8 / ms * 1e-6);
        }
    return 0;
}
