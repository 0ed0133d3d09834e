// microbench.hip -- gfx950 calibration for the DG einsum kernels:
//   (1) v_mfma_f64_16x16x4_f64 operand / result lane layout check (exact integers),
//   (2) fp64 MFMA issue rate (16x16x4 and 4x4x4_4b) and v_fma_f64 rate,
//   (3) HBM streaming bandwidth: read-only, write-only, copy, and the grad
//       einsum's 352 B read : 840 B write mix.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o build/microbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                            \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));      \
            exit(2);                                                     \
        }                                                                \
    } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// ---- (1) layout: one wave, C = A(16x4) * B(4x16)
__global__ void layout_kernel(const double* A, const double* B, double* C) {
    const int l = threadIdx.x;
    const double a = A[(l & 15) * 4 + (l >> 4)];   // A[row = l&15][k = l>>4]
    const double b = B[(l >> 4) * 16 + (l & 15)];  // B[k = l>>4][col = l&15]
    v4d acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int q = 0; q < 4; ++q) C[((l >> 4) + 4 * q) * 16 + (l & 15)] = acc[q];  // row = (l>>4)+4q
}

// ---- (2) rates (inline asm so hipcc cannot shuffle accumulators through AGPRs)
template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma16_rate(double* out, int iters) {
    v4d acc[NACC];
    const double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
    for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma4_rate(double* out, int iters) {
    double acc[NACC];
    const double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) out[threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void fma_rate(double* out, int iters) {
    double acc[NACC];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) out[threadIdx.x] = s;
}

// 4x4x4_4b lane-layout discovery: wave (la, lb) feeds one-hot A (lane la) and one-hot B (lane lb)
// and records which result lane is non-zero.
__global__ void layout4_kernel(int* res) {
    const int l = threadIdx.x, la = blockIdx.x >> 6, lb = blockIdx.x & 63;
    const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    if (l == 0) res[blockIdx.x] = -1;
    __syncthreads();
    if (d != 0.0) res[blockIdx.x] = l;
}

// ---- (3) bandwidth
__global__ __launch_bounds__(256) void bw_read(const v2d* __restrict__ in, double* out, size_t n) {
    v2d s = {0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += in[i];
    if (s[0] + s[1] == 12345.678) out[0] = s[0];
}
__global__ __launch_bounds__(256) void bw_write(v2d* __restrict__ o, size_t n) {
    const v2d v = {1.0, 2.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = v;
}
__global__ __launch_bounds__(256) void bw_copy(const v2d* __restrict__ in, v2d* __restrict__ o, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) o[i] = in[i];
}
// grad-like mix: per "element" read 22 x 16 B, write 52.5 x 16 B (approximated as 3 writes per
// read-chunk group: read n16, write 2.386*n16): each thread reads 1 chunk and writes W chunks.
__global__ __launch_bounds__(256) void bw_mix(const v2d* __restrict__ in, v2d* __restrict__ o, size_t n_in,
                                              size_t n_out) {
    // n_out = n_in * 105 / 44 exactly (caller guarantees); thread i handles in chunk i
    // and out chunks {i, i + n_in, i + 2 n_in (if < n_out)}
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_in; i += (size_t)gridDim.x * 256) {
        const v2d v = in[i];
        o[i] = v;
        o[i + n_in] = v;
        if (i + 2 * n_in < n_out) o[i + 2 * n_in] = v;
    }
}

// store-path variants: MODE 0 plain, 1 nontemporal; SEG = contiguous bytes written per wave per step
template <int MODE, int SEG16>
__global__ __launch_bounds__(256) void bw_write_seg(v2d* __restrict__ o, size_t n) {
    // each wave owns SEG16 consecutive 16-B chunks per step (SEG16 multiple of 64)
    const v2d v = {1.0, 2.0};
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    for (size_t base = wave * SEG16; base + SEG16 <= n; base += nw * SEG16)
#pragma unroll
        for (int k = 0; k < SEG16 / 64; ++k) {
            v2d* p = o + base + k * 64 + lane;
            if (MODE == 1) __builtin_nontemporal_store(v, p); else *p = v;
        }
}
template <int MODE>
__global__ __launch_bounds__(256) void bw_mix2(const v2d* __restrict__ in, v2d* __restrict__ o, size_t n_in,
                                               size_t n_out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_in; i += (size_t)gridDim.x * 256) {
        const v2d v = MODE ? __builtin_nontemporal_load(in + i) : in[i];
        if (MODE) {
            __builtin_nontemporal_store(v, o + i);
            __builtin_nontemporal_store(v, o + i + n_in);
            if (i + 2 * n_in < n_out) __builtin_nontemporal_store(v, o + i + 2 * n_in);
        } else {
            o[i] = v; o[i + n_in] = v;
            if (i + 2 * n_in < n_out) o[i + 2 * n_in] = v;
        }
    }
}

template <typename F>
static float time_ms(F f, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s arch %s CUs %d clock %.0f MHz\n", p.name, p.gcnArchName, p.multiProcessorCount,
           p.clockRate * 1e-3);
    const int cus = p.multiProcessorCount;

    {  // layout
        std::vector<double> A(64), B(64), C(256, -1), R(256, 0);
        for (int i = 0; i < 64; ++i) { A[i] = (i * 7) % 11 + 1; B[i] = (i * 5) % 13 + 2; }
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) for (int k = 0; k < 4; ++k)
            R[m * 16 + n] += A[m * 4 + k] * B[k * 16 + n];
        double *dA, *dB, *dC;
        CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 2048));
        CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
        layout_kernel<<<1, 64>>>(dA, dB, dC);
        CK(hipMemcpy(C.data(), dC, 2048, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += C[i] != R[i];
        printf("mfma_f64_16x16x4 layout (A[l&15][l>>4], B[l>>4][l&15], D row=(l>>4)+4q col=l&15): %s (%d mismatches)\n",
               bad ? "MISMATCH" : "OK", bad);
    }

    {
        int* dres; std::vector<int> res(4096);
        CK(hipMalloc(&dres, 4096 * 4));
        layout4_kernel<<<4096, 64>>>(dres);
        CK(hipMemcpy(res.data(), dres, 4096 * 4, hipMemcpyDeviceToHost));
        printf("mfma_f64_4x4x4_4b one-hot map: A-lane: {B-lane->D-lane}\n");
        for (int la = 0; la < 64; ++la) {
            printf("  A%2d:", la);
            for (int lb = 0; lb < 64; ++lb) if (res[la * 64 + lb] >= 0) printf(" B%d->D%d", lb, res[la * 64 + lb]);
            printf("\n");
        }
    }
    double* dout;
    CK(hipMalloc(&dout, 4096));
    const int iters = 4000;
    auto rate = [&](const char* nm, auto kern, int nacc, double flop_per_inst, int wg_per_cu, int threads) {
        const int grid = cus * wg_per_cu;
        float ms = time_ms([&] { kern<<<grid, threads>>>(dout, iters); }, 3);
        const double waves = (double)grid * threads / 64;
        const double inst = waves * iters * nacc;
        const double tf = inst * flop_per_inst / (ms * 1e-3) * 1e-12;
        const double cyc = (ms * 1e-3) * 2.4e9 / ((double)iters * nacc * (waves / (cus * 4.0)));
        printf("%-34s wg/cu=%d thr=%d: %.3f ms  %.1f TFLOP/s  ~%.1f cyc/inst/SIMD @2.4GHz\n", nm, wg_per_cu, threads, ms, tf, cyc);
    };
    rate("mfma_f64_16x16x4 x1 acc", mfma16_rate<1>, 1, 2048, 1, 256);
    rate("mfma_f64_16x16x4 x4 acc", mfma16_rate<4>, 4, 2048, 1, 256);
    rate("mfma_f64_16x16x4 x7 acc", mfma16_rate<7>, 7, 2048, 1, 256);
    rate("mfma_f64_16x16x4 x7 acc 2w/SIMD", mfma16_rate<7>, 7, 2048, 2, 256);
    rate("mfma_f64_16x16x4 x2 acc 2w/SIMD", mfma16_rate<2>, 2, 2048, 2, 256);
    rate("mfma_f64_4x4x4_4b x1 acc", mfma4_rate<1>, 1, 512, 1, 256);
    rate("mfma_f64_4x4x4_4b x8 acc", mfma4_rate<8>, 8, 512, 1, 256);
    rate("mfma_f64_4x4x4_4b x8 acc 2w/SIMD", mfma4_rate<8>, 8, 512, 2, 256);
    rate("v_fma_f64 x8 acc", fma_rate<8>, 8, 128, 1, 256);
    rate("v_fma_f64 x8 acc 2w/SIMD", fma_rate<8>, 8, 128, 2, 256);
    rate("v_fma_f64 x8 acc 4w/SIMD", fma_rate<8>, 8, 128, 4, 256);

    const size_t n_in = (size_t)44 * 1000000;       // 16-B chunks: 704 MB
    const size_t n_out = (size_t)105 * 1000000;     // 1.68 GB
    v2d *bi, *bo;
    CK(hipMalloc(&bi, n_out * 16));
    CK(hipMalloc(&bo, n_out * 16));
    CK(hipMemset(bi, 0, n_out * 16));
    CK(hipMemset(bo, 0, n_out * 16));
    for (int wgpc : {4, 8, 16}) {
        const int grid = cus * wgpc;
        float r = time_ms([&] { bw_read<<<grid, 256>>>(bi, dout, n_out); }, 5);
        float w = time_ms([&] { bw_write<<<grid, 256>>>(bo, n_out); }, 5);
        float c = time_ms([&] { bw_copy<<<grid, 256>>>(bi, bo, n_out); }, 5);
        float m = time_ms([&] { bw_mix<<<grid, 256>>>(bi, bo, n_in, n_out); }, 5);
        printf("HBM wg/cu=%2d: read %.0f GB/s | write %.0f GB/s | copy %.0f GB/s (r+w) | grad-mix (352R:840W) %.0f GB/s\n",
               wgpc, n_out * 16 / (r * 1e6), n_out * 16 / (w * 1e6), 2.0 * n_out * 16 / (c * 1e6),
               (double)(n_in + n_out) * 16 / (m * 1e6));
    }
    for (int wgpc : {2, 4, 8}) {
        const int grid = cus * wgpc;
        float w0 = time_ms([&] { bw_write_seg<0, 64><<<grid, 256>>>(bo, n_out); }, 5);
        float w1 = time_ms([&] { bw_write_seg<1, 64><<<grid, 256>>>(bo, n_out); }, 5);
        float w2 = time_ms([&] { bw_write_seg<0, 256><<<grid, 256>>>(bo, n_out); }, 5);
        float w3 = time_ms([&] { bw_write_seg<1, 256><<<grid, 256>>>(bo, n_out); }, 5);
        float w4 = time_ms([&] { bw_write_seg<0, 1024><<<grid, 256>>>(bo, n_out); }, 5);
        float w5 = time_ms([&] { bw_write_seg<1, 1024><<<grid, 256>>>(bo, n_out); }, 5);
        float m0 = time_ms([&] { bw_mix2<0><<<grid, 256>>>(bi, bo, n_in, n_out); }, 5);
        float m1 = time_ms([&] { bw_mix2<1><<<grid, 256>>>(bi, bo, n_in, n_out); }, 5);
        const double gb = n_out * 16 / 1e6, gm = (double)(n_in + n_out) * 16 / 1e6;
        printf("WRITE wg/cu=%d: seg1K plain %.0f nt %.0f | seg4K plain %.0f nt %.0f | seg16K plain %.0f nt %.0f GB/s || mix plain %.0f nt %.0f GB/s\n",
               wgpc, gb / w0, gb / w1, gb / w2, gb / w3, gb / w4, gb / w5, gm / m0, gm / m1);
    }
    {
        float ms = time_ms([&] { CK(hipMemsetAsync(bo, 0, n_out * 16)); }, 5);
        printf("hipMemsetAsync: %.0f GB/s\n", n_out * 16 / (ms * 1e6));
        float mc = time_ms([&] { CK(hipMemcpyAsync(bo, bi, n_out * 16, hipMemcpyDeviceToDevice)); }, 5);
        printf("hipMemcpyAsync D2D: %.0f GB/s (r+w)\n", 2.0 * n_out * 16 / (mc * 1e6));
    }
    return 0;
}
