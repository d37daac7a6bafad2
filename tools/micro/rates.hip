// rates.hip -- issue rates on gfx950 that decide how the Glow conv blocks should be written:
//   v_fma_f32 (SGPR x VGPR), v_pk_fma_f32, v_mfma_f32_4x4x1_16B_f32, v_mfma_f32_16x16x4_f32,
// each as a stream of independent instructions, 1 / 2 / 4 waves per SIMD, every CU busy.
// hipcc --offload-arch=gfx950 -O3 tools/micro/rates.hip -o tools/micro/rates && tools/micro/rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, float s, int iters)
{
    const int lane = threadIdx.x;
    float r = 0.0f;
    if (MODE == 0) {                       // 32 independent v_fma_f32 with an SGPR operand
        float a[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) a[i] = lane * 0.001f + i;
        const float x = lane * 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(s), "v"(x));
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) r += a[i];
    } else if (MODE == 1) {                // 16 independent v_pk_fma_f32 (= 32 fmas per lane)
        f2 a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = f2{lane * 0.001f + i, lane * 0.002f + i};
        const f2 x = {lane * 0.5f, lane * 0.25f};
        const f2 w = {s, s};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(w), "v"(x));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) r += a[i].x + a[i].y;
    } else if (MODE == 2) {                // 8 independent v_mfma_f32_4x4x1_16B_f32
        f4 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = f4{0, 0, 0, 0};
        const float x = lane * 0.5f, w = s + lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, a[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    } else if (MODE == 3) {                // 8 independent v_mfma_f32_16x16x4_f32
        f4 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = f4{0, 0, 0, 0};
        const float x = lane * 0.5f, w = s + lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    } else {                               // MODE 4: 4x4x1 MFMAs with one ds_read_b32 + 2 v_fma between them
        __shared__ float sh[4096];
        for (int i = lane; i < 4096; i += blockDim.x) sh[i] = i;
        __syncthreads();
        f4 a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = f4{0, 0, 0, 0};
        float w = s + lane, x = 0.0f;
        int p = lane;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if ((i & 1) == 0) x = sh[(p + 64 * i + rep) & 4095];
                    a[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(w, x, a[i], 0, 0, 0);
                }
            p += 7;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1] + a[i][2] + a[i][3];
    }
    if (r == 12345.678f) out[0] = r;
}

template <int MODE>
static void run(const char *name, double macs_per_iter_per_wave, float *d)
{
    for (int block : {256, 512, 1024}) {
        const int iters = 4000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, d, 1.5f, 10);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, d, 1.5f, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double waves = 256.0 * block / 64;
        const double macs = waves * iters * macs_per_iter_per_wave;
        printf("%-28s %d waves/SIMD: %8.3f ms  %7.1f TMAC/s  (%.1f TFLOP/s)\n", name, block / 256, ms,
               macs / ms * 1e-9, 2 * macs / ms * 1e-9);
    }
}

int main()
{
    float *d;
    hipMalloc(&d, 4096);
    run<0>("v_fmac_f32 sgpr", 32.0 * 64, d);
    run<1>("v_pk_fma_f32", 32.0 * 64, d);
    run<2>("v_mfma_f32_4x4x1_16B_f32", 32.0 * 256, d);
    run<3>("v_mfma_f32_16x16x4_f32", 8.0 * 1024, d);
    run<4>("4x4x1 + ds_read every 2nd", 32.0 * 256, d);
    return 0;
}
