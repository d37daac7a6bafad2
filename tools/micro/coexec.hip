// Do vector instructions of one wave issue while another wave's MFMA occupies the matrix pipe of the same SIMD?
// Each SIMD gets 2 waves (workgroup of 512 threads = 8 waves on one CU).  Modes: 0 = every wave runs MFMAs,
// 1 = every wave runs v_fma_f32, 2 = waves 0-3 MFMA, waves 4-7 v_fma (co-execution if time ~ max, not sum).
// Variants: f32-input v_mfma_f32_16x16x4_f32 (32 cycles) and v_mfma_f32_16x16x16_bf16 (8 cycles on gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: f32 16x16x4, 1: bf16 16x16x16, 2: bf16 16x16x32
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);   // waves 0-3 and 4-7 land on SIMDs 0-3 each
    const bool do_valu = mode == 1 || (mode == 2 && wave >= 4);
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1}, acc2 = {2, 2, 2, 2}, acc3 = {3, 3, 3, 3};
    float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.0f, v3 = 3.0f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    const float a = 1.0001f, b = 0.5f;
    if (do_mfma) {
        for (int i = 0; i < iters; ++i) {
            if constexpr (KIND == 0) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);
            } else if constexpr (KIND == 1) {
                const s16x4 A = {0x3f80, 0x3f80, 0x3f80, 0x3f80}, B = {0x3f00, 0x3f00, 0x3f00, 0x3f00};
                acc0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A, B, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A, B, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A, B, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A, B, acc3, 0, 0, 0);
            } else {
                bf16x8 A, B;
                for (int t = 0; t < 8; ++t) { A[t] = (__bf16)1.0f; B[t] = (__bf16)0.5f; }
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, acc3, 0, 0, 0);
            }
        }
    }
    if (do_valu) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0 = __builtin_fmaf(v0, a, b); v1 = __builtin_fmaf(v1, a, b); v2 = __builtin_fmaf(v2, a, b); v3 = __builtin_fmaf(v3, a, b);
                v4 = __builtin_fmaf(v4, a, b); v5 = __builtin_fmaf(v5, a, b); v6 = __builtin_fmaf(v6, a, b); v7 = __builtin_fmaf(v7, a, b);
            }
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int KIND> void run(const char *name, float *d)
{
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, 100, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d, iters, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // per wave: mode 0: 4 MFMA / iter; mode 1: 32 fma / iter
        printf("%s mode %d (%s): %.3f ms  -> %.1f cycles per iteration per SIMD at 2.4 GHz (2 waves per SIMD)\n", name, mode,
               mode == 0 ? "all waves MFMA" : mode == 1 ? "all waves v_fma" : "waves 0-3 MFMA, waves 4-7 v_fma", ms,
               ms * 1e-3 * 2.4e9 / iters);
    }
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 512 * sizeof(float));
    run<0>("f32  16x16x4 ", d);
    run<1>("bf16 16x16x16", d);
    run<2>("bf16 16x16x32", d);
    return 0;
}
