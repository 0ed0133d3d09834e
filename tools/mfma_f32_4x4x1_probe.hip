// mfma_f32_4x4x1_probe.hip -- which lanes of A and B meet in which (lane, register) of D for v_mfma_f32_4x4x1_16B_f32
// (16 blocks of 4 x 4 x 1): A is one-hot at lane a, B holds lane + 1, so every non-zero D entry names its B lane.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_f32_4x4x1_probe.hip -o build/mfma_f32_4x4x1_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(int a_lane, float* d) {
    const int l = threadIdx.x;
    const float a = l == a_lane ? 1.f : 0.f, b = (float)(l + 1);
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) d[l * 4 + v] = c[v];
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4);
    float h[256];
    const int lanes[] = {0, 1, 2, 3, 4, 5, 17, 38, 63};
    for (int a : lanes) {
        probe<<<1, 64>>>(a, d);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("A one-hot at lane %2d (block %2d, row %d):", a, a / 4, a % 4);
        for (int l = 0; l < 64; ++l)
            for (int v = 0; v < 4; ++v)
                if (h[l * 4 + v] != 0.f) printf("  D[lane %d][reg %d] <- B lane %d", l, v, (int)h[l * 4 + v] - 1);
        printf("\n");
    }
    return 0;
}
