// wpattern.hip -- which store pattern reaches hipMemset-class write bandwidth on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2);} } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

// P1: grid-stride, 16 B per lane per step (the fill-kernel shape)
__global__ void w_stride(v2d* __restrict__ o, size_t n) {
    const v2d v = {1.0, 2.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) o[i] = v;
}
// P3/P4/P5: grad output pattern: tile = 16 elements x 35 doubles = 280 chunks of 16 B, three planes.
// MODE 0 cyclic tile->wave (tile = it*nwaves + wave), 1 blocked (wave owns a contiguous run of tiles)
template <int MODE>
__global__ void w_tiles(v2d* __restrict__ o, size_t nTiles, size_t planeChunks) {
    const v2d v = {1.0, 2.0};
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t per = (nTiles + nw - 1) / nw;
    for (size_t it = 0; it < per; ++it) {
        const size_t tile = MODE ? wave * per + it : it * nw + wave;
        if (tile >= nTiles) break;
        for (int x = 0; x < 3; ++x) {
            v2d* p = o + x * planeChunks + tile * 280;
#pragma unroll
            for (int c = 0; c < 5; ++c) if (c < 4 || lane < 24) p[c * 64 + lane] = v;
        }
    }
}
// read u-tile (280 chunks) + J (9 x 128 B) per tile and write 3 planes: the full grad traffic, no compute
template <int MODE>
__global__ void rw_tiles(const v2d* __restrict__ u, const v2d* __restrict__ J, v2d* __restrict__ o, size_t nTiles,
                         size_t planeChunks, size_t jRowChunks) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t per = (nTiles + nw - 1) / nw;
    for (size_t it = 0; it < per; ++it) {
        const size_t tile = MODE ? wave * per + it : it * nw + wave;
        if (tile >= nTiles) break;
        v2d r[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) r[c] = (c < 4 || lane < 24) ? u[tile * 280 + c * 64 + lane] : v2d{0, 0};
        v2d jv = (lane < 72 - 64 + 64) ? J[(size_t)(lane >> 3) * jRowChunks + tile * 8 + (lane & 7)] : v2d{0, 0};
        v2d jv2 = (lane < 8) ? J[(size_t)8 * jRowChunks + tile * 8 + lane] : v2d{0, 0};
        r[0] += jv + jv2;
        for (int x = 0; x < 3; ++x) {
            v2d* p = o + x * planeChunks + tile * 280;
#pragma unroll
            for (int c = 0; c < 5; ++c) if (c < 4 || lane < 24) p[c * 64 + lane] = r[c];
        }
    }
}
template <typename F> static float time_ms(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
    const size_t E = 1000000, nTiles = E / 16, planeChunks = E * 35 / 2, n = 3 * planeChunks;   // 840 MB
    v2d *o, *u, *J;
    CK(hipMalloc(&o, n * 16)); CK(hipMalloc(&u, planeChunks * 16)); CK(hipMalloc(&J, 9 * E * 8));
    CK(hipMemset(o, 0, n * 16)); CK(hipMemset(u, 0, planeChunks * 16)); CK(hipMemset(J, 0, 9 * E * 8));
    const double gb = n * 16 / 1e6, gbrw = (n * 16 + planeChunks * 16 + 9.0 * E * 8) / 1e6;
    for (int thr : {256, 512, 1024}) for (int bpc : {1, 2, 4}) {
        if (thr * bpc > 2048) continue;
        const int grid = 256 * bpc;
        float a = time_ms([&] { w_stride<<<grid, thr>>>(o, n); }, 5);
        float b = time_ms([&] { w_tiles<0><<<grid, thr>>>(o, nTiles, planeChunks); }, 5);
        float c = time_ms([&] { w_tiles<1><<<grid, thr>>>(o, nTiles, planeChunks); }, 5);
        float d = time_ms([&] { rw_tiles<0><<<grid, thr>>>(u, J, o, nTiles, planeChunks, E / 2); }, 5);
        float e = time_ms([&] { rw_tiles<1><<<grid, thr>>>(u, J, o, nTiles, planeChunks, E / 2); }, 5);
        printf("thr=%4d blocks/CU=%d (waves/CU=%2d): stride %.0f | tiles cyclic %.0f blocked %.0f GB/s (write only) || grad traffic cyclic %.0f (%.3f ms) blocked %.0f (%.3f ms) GB/s\n",
               thr, bpc, thr * bpc / 64, gb / a, gb / b, gb / c, gbrw / d, d, gbrw / e, e);
    }
    float m = time_ms([&] { CK(hipMemsetAsync(o, 0, n * 16)); }, 5);
    printf("hipMemsetAsync %.0f GB/s\n", gb / m);
    return 0;
}
