// mfma_valu_overlap.hip -- does other work of a SIMD run under an f64 MFMA?  (gfx950 probe, round 4)
// One wave issues, per iteration, ONE v_mfma_f64_16x16x4_f64 (three accumulators in rotation: no dependent stall) followed by
// N independent instructions of one kind -- f64 FMAs, f32 FMAs, integer adds, LDS reads -- and counts shader cycles per
// iteration.  If the side work hides under the MFMA's 64 cycles the count stays at ~64 until N x cost exceeds it; if the
// MFMA occupies the unit the side work needs, the count is 64 + N x cost from N = 1 on.  Run with one wave per SIMD and with
// two (the second wave only does side work): can ANOTHER wave's side work overlap?
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_valu_overlap.hip -o build/mfma_valu_overlap && build/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v8s __attribute__((ext_vector_type(8)));
// CONTROL (round 5): the same harness on the bf16 MFMAs for which /opt/skills/guides/MI355X_MICROARCH.md (rows 'vector-instruction
// ISSUE cost' and 'single-issue instructions HIDDEN per v_mfma_f32_32x32x16_bf16 gap') documents that a wave's own VALU
// instructions DO hide: issue costs summing to <= 24 cycles in a 32-cycle gap.  If this harness reproduces that, its finding for
// the 64-cycle f64 / f32 MFMAs (nothing of the wave's own VALU work hides) is a property of those instructions, not of the harness.
#if defined(MFMA_BF16_32)
#define ACC_T v16f
#define ACC_ZERO {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define MFMA_ASM "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
#define AB_T v8s
#define AB_INIT(x) v8s{0x3f80, 0x3f00, 0x3e80, 0x3f80, 0x3f00, 0x3e80, 0x3f80, (short)(0x3f00 + ((int)(x) & 1))}
#define MFMA_NAME "v_mfma_f32_32x32x16_bf16 (control: 8 passes = 32 cycles)"
#elif defined(MFMA_BF16_16)
#define ACC_T v4f
#define ACC_ZERO {0, 0, 0, 0}
#define MFMA_ASM "v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0"
#define AB_T v8s
#define AB_INIT(x) v8s{0x3f80, 0x3f00, 0x3e80, 0x3f80, 0x3f00, 0x3e80, 0x3f80, (short)(0x3f00 + ((int)(x) & 1))}
#define MFMA_NAME "v_mfma_f32_16x16x32_bf16 (control: 4 passes = 16 cycles)"
#elif defined(MFMA_F32)   // the same probe with v_mfma_f32_32x32x2_f32 (16 passes = 64 cycles too)
#define ACC_T v16f
#define ACC_ZERO {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define MFMA_ASM "v_mfma_f32_32x32x2_f32 %0, %1, %2, %0"
#define AB_T float
#define AB_INIT(x) (float)(x)
#define MFMA_NAME "v_mfma_f32_32x32x2_f32"
#else
#define ACC_T v4d
#define ACC_ZERO {0, 0, 0, 0}
#define MFMA_ASM "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0"
#define AB_T double
#define AB_INIT(x) (double)(x)
#define MFMA_NAME "v_mfma_f64_16x16x4_f64"
#endif

__device__ unsigned long long result[16];

// KIND 0: f64 FMA, 1: f32 FMA, 2: u32 add, 3: ds_read_b64, 4: f64 FMA as ONE dependent chain
// second_wave_mode: 0 idle, 1 side work, 2 side work at s_setprio 3; bit 4: the MFMA waves issue NO side work (pure MFMA stream)
template <int N, int KIND, bool MFMA>
__global__ __launch_bounds__(512) void probe(double* sink, int iters, int second_wave_mode) {
    __shared__ double lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 1e-3;
    __syncthreads();
    const bool mfma_wave = __builtin_amdgcn_readfirstlane(wave) < 4;   // waves 4..7 (when launched): side work only (wave-uniform: scalar branches)
    ACC_T acc0 = ACC_ZERO, acc1 = acc0, acc2 = acc0;
    double a = 1.0 + lane * 1e-3, b = 2.0 - lane * 1e-3;
    AB_T ma = AB_INIT(a), mb = AB_INIT(b);
    double f[16];
    float g[16];
    unsigned h[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { f[k] = k + lane; g[k] = k - lane; h[k] = k * lane; }
    const unsigned ldsaddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)(lds + lane);
    auto side = [&]() {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (KIND == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[k % 16]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(g[k % 16]) : "v"((float)1.5f), "v"((float)0.5f));
            if (KIND == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(h[k % 16]) : "v"(lane));
            if (KIND == 3) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f[k % 16]) : "v"(ldsaddr), "n"((k % 16) * 512));
            if (KIND == 4) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[0]) : "v"(a), "v"(b));
            if (KIND == 5) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(ldsaddr), "v"(f[k % 16]), "n"((k % 16) * 512) : "memory");
        }
        if (KIND == 3 || KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const bool pure = (second_wave_mode & 16) != 0;
    if (mfma_wave) {
        if (pure) {
            for (int it = 0; it < iters; ++it) {
                asm volatile(MFMA_ASM : "+v"(acc0) : "v"(ma), "v"(mb));
                asm volatile(MFMA_ASM : "+v"(acc1) : "v"(ma), "v"(mb));
                asm volatile(MFMA_ASM : "+v"(acc2) : "v"(ma), "v"(mb));
            }
        } else {
            for (int it = 0; it < iters; ++it) {   // three accumulators in turn: an MFMA never waits for its predecessor's result
                if (MFMA) asm volatile(MFMA_ASM : "+v"(acc0) : "v"(ma), "v"(mb));
                side();
                if (MFMA) asm volatile(MFMA_ASM : "+v"(acc1) : "v"(ma), "v"(mb));
                side();
                if (MFMA) asm volatile(MFMA_ASM : "+v"(acc2) : "v"(ma), "v"(mb));
                side();
            }
        }
    } else if (second_wave_mode & 3) {
        if ((second_wave_mode & 3) == 2) __builtin_amdgcn_s_setprio(3);
        for (int it = 0; it < iters; ++it) { side(); side(); side(); }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = acc0[0] + acc1[1] + acc2[2];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += f[k] + g[k] + h[k];
    if (s == 1.2345e-300) sink[0] = s;
    if (blockIdx.x == 3 && threadIdx.x == 0) result[0] = t1 - t0;
    if (blockIdx.x == 3 && threadIdx.x == 256) result[1] = t1 - t0;
}

template <int N, int KIND>
int run(const char* what, double* sink) {
    const int iters = 4000;
    unsigned long long r[2];
    double base[3];
    for (int mode = 0; mode < 3; ++mode) {   // 0: four waves (one per SIMD) with MFMA; 1: the same without MFMA; 2: eight waves, the second four side work only
        if (mode == 0) hipLaunchKernelGGL((probe<N, KIND, true>), dim3(256), dim3(256), 0, 0, sink, iters, 0);
        if (mode == 1) hipLaunchKernelGGL((probe<N, KIND, false>), dim3(256), dim3(256), 0, 0, sink, iters, 0);
        if (mode == 2) hipLaunchKernelGGL((probe<0, KIND, true>), dim3(256), dim3(256), 0, 0, sink, iters, 0);
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(r, HIP_SYMBOL(result), sizeof r));
        base[mode] = (double)r[0] / iters / 3.0;
    }
    printf("%-28s N = %2d: MFMA + side %6.1f cycles per MFMA | side alone %6.1f | MFMA alone %6.1f | sum %6.1f, max %6.1f\n", what, N,
           base[0], base[1], base[2], base[1] + base[2], base[1] > base[2] ? base[1] : base[2]);
    return 0;
}

// two waves per SIMD: waves 0..3 issue MFMAs only, waves 4..7 the side work only
template <int N, int KIND>
int run2(const char* what, double* sink) {
    const int iters = 4000;
    unsigned long long r[2];
    hipLaunchKernelGGL((probe<N, KIND, true>), dim3(256), dim3(512), 0, 0, sink, iters, 0);   // second waves idle
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((probe<N, KIND, true>), dim3(256), dim3(512), 0, 0, sink, iters, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpyFromSymbol(r, HIP_SYMBOL(result), sizeof r));
    printf("%-28s N = %2d: wave A (MFMA + side) %6.1f cycles per MFMA, wave B on the same SIMD (side only) %6.1f\n", what, N, (double)r[0] / iters / 3.0, (double)r[1] / iters / 3.0);
    return 0;
}

// two waves per SIMD: wave A = a PURE MFMA stream, wave B = N side instructions per A-MFMA-slot, at priority 0 and 3: how the
// arbiter shares the SIMD between one wave's matrix phase and its partner's VALU phase (both loops run `iters` iterations:
// the one that finishes first leaves the other alone, so read the LARGER figure as "beside the partner" only when both are close)
template <int N, int KIND>
int run3(const char* what, double* sink) {
    const int iters = 4000;
    unsigned long long r[2];
    for (int mode : {17, 18}) {
        hipLaunchKernelGGL((probe<N, KIND, true>), dim3(256), dim3(512), 0, 0, sink, iters, mode);
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(r, HIP_SYMBOL(result), sizeof r));
        printf("%-28s N = %2d, B at priority %d: wave A (pure MFMA) %6.1f cycles per MFMA, wave B (side only) %6.1f cycles per group of N = %5.1f per instruction\n", what, N,
               mode == 18 ? 3 : 0, (double)r[0] / iters / 3.0, (double)r[1] / iters / 3.0, (double)r[1] / iters / 3.0 / N);
    }
    return 0;
}

int main() {
    double* sink;
    CK(hipMalloc(&sink, 64));
    printf("# " MFMA_NAME "\n# one wave per SIMD: per MFMA (three independent accumulators in turn): the MFMA + N side instructions behind it (shader cycles, s_memtime)\n");
    run<0, 0>("nothing", sink);
    run<2, 0>("v_fma_f64 (independent)", sink); run<4, 0>("v_fma_f64 (independent)", sink); run<8, 0>("v_fma_f64 (independent)", sink); run<16, 0>("v_fma_f64 (independent)", sink);
    run<4, 4>("v_fma_f64 (one chain)", sink); run<8, 4>("v_fma_f64 (one chain)", sink);
    run<1, 1>("v_fma_f32", sink); run<2, 1>("v_fma_f32", sink); run<3, 1>("v_fma_f32", sink); run<4, 1>("v_fma_f32", sink); run<5, 1>("v_fma_f32", sink);
    run<6, 1>("v_fma_f32", sink); run<8, 1>("v_fma_f32", sink); run<16, 1>("v_fma_f32", sink);
    run<1, 2>("v_add_u32", sink); run<2, 2>("v_add_u32", sink); run<3, 2>("v_add_u32", sink); run<4, 2>("v_add_u32", sink); run<5, 2>("v_add_u32", sink);
    run<6, 2>("v_add_u32", sink); run<8, 2>("v_add_u32", sink); run<16, 2>("v_add_u32", sink);
    run<4, 5>("ds_write_b64 + wait", sink); run<8, 5>("ds_write_b64 + wait", sink);
    run<2, 3>("ds_read_b64 + wait", sink); run<4, 3>("ds_read_b64 + wait", sink); run<8, 3>("ds_read_b64 + wait", sink);
    printf("# two waves per SIMD: wave A = MFMA + N side instructions, wave B = N side instructions only\n");
    run2<8, 0>("v_fma_f64 (independent)", sink); run2<16, 0>("v_fma_f64 (independent)", sink);
    run2<16, 1>("v_fma_f32", sink); run2<16, 2>("v_add_u32", sink); run2<8, 3>("ds_read_b64 + wait", sink);
    printf("# two waves per SIMD: wave A = pure MFMA stream, wave B = side work only, at priority 0 / 3\n");
    run3<4, 0>("v_fma_f64 (independent)", sink); run3<16, 0>("v_fma_f64 (independent)", sink);
    run3<16, 1>("v_fma_f32", sink); run3<16, 2>("v_add_u32", sink); run3<8, 5>("ds_write_b64 + wait", sink); run3<8, 3>("ds_read_b64 + wait", sink);
    return 0;
}
