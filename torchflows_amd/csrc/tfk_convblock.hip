// tfk_convblock.hip -- one block of the Glow ConvNet conditioner in one launch:
//   conv3x3(padding 1) -> ReLU -> MaxPool2d(2) -> BatchNorm2d (inference: per-channel scale + shift)
// (reference multiscale/conditioning/classic.py: ConvNetBlock.forward), for the tiny channel counts
// of that network (4 -> 8 -> 8 -> 4).  The library route runs a Winograd convolution, a ReLU, a
// pooling and a normalisation kernel and moves the full-resolution activation through HBM three
// times (270 us + 55 us + 40 us + 50 us per block at 8192 x 4 x 32 x 32); here one lane computes one
// POOLED pixel for all output channels -- the 2x2 window of convolution outputs never leaves
// registers -- so HBM sees the input once and the pooled output once.
//   FMAs per pooled pixel: 4 * 9 * CIN * COUT; weights are LDS broadcasts, [ci][co][12] (9 taps + pad)
//   read as three ds_read_b128 per 36 FMAs.
// Layout: NCHW, H and W even.  Accumulation order: bias, then input channels in order, taps row-major
// (fp32 FMA chain; the reference's CPU convolution sums in a different order: a few ulp apart).
#include "tfk_common.h"

namespace tfk {

// PX pooled pixels per lane along W (2 when the pooled width is even: the weights and the two shared
// patch columns are reused).
template <int CIN, int COUT, int PX>
__global__ __launch_bounds__(kBlock) void k_conv3x3_relu_pool_affine(
    const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
    const float *__restrict__ scale, const float *__restrict__ shift, float *__restrict__ out,
    long long N, int H, int W)
{
    __shared__ __attribute__((aligned(16))) float ws[CIN * COUT * 12];
    __shared__ float bs[3 * COUT];
    for (int i = threadIdx.x; i < CIN * COUT * 12; i += kBlock) {
        const int tap = i % 12, co = (i / 12) % COUT, ci = i / (12 * COUT);
        ws[i] = tap < 9 ? w[(co * CIN + ci) * 9 + tap] : 0.0f;       // weight (COUT, CIN, 3, 3)
    }
    for (int i = threadIdx.x; i < COUT; i += kBlock) {
        bs[i] = bias[i];
        bs[COUT + i] = scale[i];
        bs[2 * COUT + i] = shift[i];
    }
    __syncthreads();
    const int PH = H >> 1, PW = W >> 1, PWX = PW / PX;
    const long long total = N * (long long)PH * PWX;
    const long long plane = (long long)H * W;
    constexpr int OW = 2 * PX, PC = OW + 2;              // conv outputs / patch columns per lane
    for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kBlock) {
        const int pw = (int)(idx % PWX) * PX;
        const int ph = (int)((idx / PWX) % PH);
        const long long n = idx / ((long long)PWX * PH);
        float acc[COUT][2][OW];
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                for (int ox = 0; ox < OW; ++ox) acc[co][oy][ox] = bs[co];
        const int ih0 = 2 * ph - 1, iw0 = 2 * pw - 1;
#pragma unroll 1
        for (int ci = 0; ci < CIN; ++ci) {   // rolled: one patch live at a time keeps the lane under 128 VGPRs
            const float *src = x + (n * CIN + ci) * plane;
            float p[4][PC];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ih = ih0 + r;
                const bool rok = (ih >= 0) && (ih < H);
#pragma unroll
                for (int c = 0; c < PC; ++c) {
                    const int iw = iw0 + c;
                    p[r][c] = (rok && iw >= 0 && iw < W) ? src[(long long)ih * W + iw] : 0.0f;
                }
            }
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                const float4 w0 = *reinterpret_cast<const float4 *>(ws + (ci * COUT + co) * 12);
                const float4 w1 = *reinterpret_cast<const float4 *>(ws + (ci * COUT + co) * 12 + 4);
                const float4 w2 = *reinterpret_cast<const float4 *>(ws + (ci * COUT + co) * 12 + 8);
                const float k[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x};
#pragma unroll
                for (int oy = 0; oy < 2; ++oy)
#pragma unroll
                    for (int ox = 0; ox < OW; ++ox) {
                        float a = acc[co][oy][ox];
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) a = fmaf(k[3 * ky + kx], p[oy + ky][ox + kx], a);
                        acc[co][oy][ox] = a;
                    }
            }
        }
        float *dst = out + (n * COUT) * (long long)PH * PW + (long long)ph * PW + pw;
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int px = 0; px < PX; ++px) {
                float v = fmaxf(fmaxf(acc[co][0][2 * px], acc[co][0][2 * px + 1]),
                                fmaxf(acc[co][1][2 * px], acc[co][1][2 * px + 1]));
                v = fmaxf(v, 0.0f);                                  // relu and max commute
                dst[(long long)co * PH * PW + px] = fmaf(v, bs[COUT + co], bs[2 * COUT + co]);
            }
    }
}

// ConvModifier (classic.py:8-42) with a 1x1 kernel: a channel-mixing GEMM whose result sits in the middle
// of a larger frame (padding beyond kernel - 1 only adds positions that equal the bias).  One lane per
// output pixel, all COUT channels; x may be a row-strided view (x_stride floats between images).
template <int COUT, int PX>
__global__ __launch_bounds__(kBlock) void k_conv1x1_frame(
    const float *__restrict__ x, long long x_stride, const float *__restrict__ weight,
    const float *__restrict__ bias, float *__restrict__ out, long long N, int C, int H, int W, int HT, int WT,
    int top, int left)
{
    // PX = 4: four consecutive output pixels per lane, float4 stores (WT % 4 == 0, out 16-byte aligned)
    const int WQ = WT / PX;
    const long long total = N * (long long)HT * WQ;
    const long long plane_in = (long long)H * W, plane_out = (long long)HT * WT;
    for (long long idx = (long long)blockIdx.x * kBlock + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kBlock) {
        const int j = (int)(idx % WQ) * PX;
        const int i = (int)((idx / WQ) % HT);
        const long long n = idx / ((long long)WQ * HT);
        float acc[COUT][PX];
#pragma unroll
        for (int co = 0; co < COUT; ++co)
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[co][p] = 0.0f;
        const int ii = i - top, jj = j - left;
        if (ii >= 0 && ii < H && jj + PX > 0 && jj < W) {
            const float *src = x + n * x_stride + (long long)ii * W;
            bool ok[PX];
#pragma unroll
            for (int p = 0; p < PX; ++p) ok[p] = (jj + p >= 0) && (jj + p < W);
            for (int c = 0; c < C; ++c) {
                float v[PX];
#pragma unroll
                for (int p = 0; p < PX; ++p) v[p] = ok[p] ? src[c * plane_in + jj + p] : 0.0f;
#pragma unroll
                for (int co = 0; co < COUT; ++co) {
                    const float wv = weight[co * C + c];
#pragma unroll
                    for (int p = 0; p < PX; ++p) acc[co][p] = fmaf(wv, v[p], acc[co][p]);
                }
            }
        }
        float *dst = out + n * COUT * plane_out + (long long)i * WT + j;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            const float b = bias[co];
            if (PX == 4) {
                *reinterpret_cast<float4 *>(dst + co * plane_out) =
                    make_float4(acc[co][0] + b, acc[co][1] + b, acc[co][2] + b, acc[co][3] + b);
            } else {
#pragma unroll
                for (int p = 0; p < PX; ++p) dst[co * plane_out + p] = acc[co][p] + b;
            }
        }
    }
}

// Bounded conditioner output (transforms.py:107-113): out = lo + (hi - lo) * sigmoid(h), the
// reference's three roundings (sigmoid, multiply, add) kept; in may alias h.
template <int V>
__global__ __launch_bounds__(kBlock) void k_bounded_sigmoid(const float *in, float *h, long long n, float lo, float range)
{
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        if (V == 4) {
            float4 v = reinterpret_cast<const float4 *>(in)[i];
            v.x = 1.0f / (1.0f + expf(-v.x)) * range + lo;
            v.y = 1.0f / (1.0f + expf(-v.y)) * range + lo;
            v.z = 1.0f / (1.0f + expf(-v.z)) * range + lo;
            v.w = 1.0f / (1.0f + expf(-v.w)) * range + lo;
            reinterpret_cast<float4 *>(h)[i] = v;
        } else {
            h[i] = 1.0f / (1.0f + expf(-in[i])) * range + lo;
        }
    }
}

// Reverse mode of the bounded output: d/dh [lo + range * sigmoid(h)] = range * s (1 - s), s recovered from the OUTPUT
// ((out - lo) / range: within 1e-7 of the sigmoid the forward pass rounded), so nothing but the output is kept.
__global__ __launch_bounds__(kBlock) void k_bounded_sigmoid_bwd(const float *__restrict__ out, const float *__restrict__ g,
                                                                float *__restrict__ g_in, long long n, float lo, float range)
{
    const float inv = 1.0f / range;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kBlock) {
        const float s = (out[i] - lo) * inv;
        g_in[i] = g[i] * (range * (s * (1.0f - s)));
    }
}

}  // namespace tfk

using namespace tfk;

extern "C" {

int tfk_conv3x3_block_supported(int32_t c_in, int32_t c_out)
{
    return ((c_in == 4 || c_in == 8) && (c_out == 4 || c_out == 8)) ? 1 : 0;
}

int tfk_conv3x3_relu_pool_affine(const float *x, const float *weight, const float *bias, const float *scale,
                                 const float *shift, float *out, int64_t N, int32_t c_in, int32_t c_out,
                                 int32_t H, int32_t W, void *stream)
{
    const char *fn = "tfk_conv3x3_relu_pool_affine";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (!tfk_conv3x3_block_supported(c_in, c_out))
        return fail(TFK_EINVAL, "%s: channels %d -> %d (kernels exist for 4 / 8 -> 4 / 8)", fn, c_in, c_out);
    if (H < 2 || W < 2 || (H & 1) || (W & 1)) return fail(TFK_EINVAL, "%s: H = %d, W = %d must be even and >= 2", fn, H, W);
    if (N == 0) return TFK_OK;
    if (!x || !weight || !bias || !scale || !shift || !out) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // one pooled pixel per lane: 122 VGPRs = 4 waves / SIMD.  Two per lane (PX = 2: shared weights and patch
    // columns, 160 VGPRs) measured 5-25 % slower on the Glow shapes -- occupancy matters more here.
    const int64_t total = N * (int64_t)(H / 2) * (W / 2);
    const int grid = grid_for(total, kBlock);
#define TFK_CB(CI, CO)                                                                                       \
    hipLaunchKernelGGL((k_conv3x3_relu_pool_affine<CI, CO, 1>), dim3(grid), dim3(kBlock), 0, s, x, weight,   \
                       bias, scale, shift, out, (long long)N, H, W)
    if (c_in == 4 && c_out == 8) TFK_CB(4, 8);
    else if (c_in == 8 && c_out == 8) TFK_CB(8, 8);
    else if (c_in == 8 && c_out == 4) TFK_CB(8, 4);
    else if (c_in == 4 && c_out == 4) TFK_CB(4, 4);
#undef TFK_CB
    return check_launch(fn);
}

int tfk_conv1x1_frame(const float *x, int64_t x_stride, const float *weight, const float *bias, float *out,
                      int64_t N, int32_t c_in, int32_t c_out, int32_t H, int32_t W, int32_t H_out, int32_t W_out,
                      void *stream)
{
    const char *fn = "tfk_conv1x1_frame";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (c_out != 1 && c_out != 4) return fail(TFK_EINVAL, "%s: c_out = %d (kernels exist for 1 and 4)", fn, c_out);
    if (c_in < 1 || H < 1 || W < 1 || H_out < H || W_out < W || ((H_out - H) & 1) || ((W_out - W) & 1))
        return fail(TFK_EINVAL, "%s: need c_in >= 1 and an even, non-negative frame (%dx%d -> %dx%d)", fn, H, W,
                    H_out, W_out);
    if (x_stride < (int64_t)c_in * H * W) return fail(TFK_EINVAL, "%s: x_stride %lld < c_in*H*W", fn, (long long)x_stride);
    if (N == 0) return TFK_OK;
    if (!x || !weight || !bias || !out) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int px = ((W_out % 4) == 0 && aligned16(out)) ? 4 : 1;
    const int grid = grid_for(N * (int64_t)H_out * (W_out / px), kBlock);
    const int top = (H_out - H) / 2, left = (W_out - W) / 2;
#define TFK_CF(CO, PX)                                                                                          \
    hipLaunchKernelGGL((k_conv1x1_frame<CO, PX>), dim3(grid), dim3(kBlock), 0, s, x, (long long)x_stride, weight, \
                       bias, out, (long long)N, c_in, H, W, H_out, W_out, top, left)
    if (c_out == 4 && px == 4) TFK_CF(4, 4);
    else if (c_out == 4) TFK_CF(4, 1);
    else if (px == 4) TFK_CF(1, 4);
    else TFK_CF(1, 1);
#undef TFK_CF
    return check_launch(fn);
}

int tfk_bounded_sigmoid(const float *in, float *h, int64_t n, float lo, float hi, void *stream)
{
    const char *fn = "tfk_bounded_sigmoid";
    if (n < 0) return fail(TFK_EINVAL, "%s: n = %lld < 0", fn, (long long)n);
    if (!(lo < hi)) return fail(TFK_EINVAL, "%s: need lo < hi", fn);
    if (n == 0) return TFK_OK;
    if (!h || !in) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((n % 4) == 0 && aligned16(h) && aligned16(in))
        hipLaunchKernelGGL((k_bounded_sigmoid<4>), dim3(grid_for(n / 4, kBlock)), dim3(kBlock), 0, s, in, h,
                           (long long)(n / 4), lo, hi - lo);
    else
        hipLaunchKernelGGL((k_bounded_sigmoid<1>), dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, in, h,
                           (long long)n, lo, hi - lo);
    return check_launch(fn);
}

int tfk_bounded_sigmoid_bwd(const float *out, const float *g, float *g_in, int64_t n, float lo, float hi, void *stream)
{
    const char *fn = "tfk_bounded_sigmoid_bwd";
    if (n < 0) return fail(TFK_EINVAL, "%s: n = %lld < 0", fn, (long long)n);
    if (!(lo < hi)) return fail(TFK_EINVAL, "%s: need lo < hi", fn);
    if (n == 0) return TFK_OK;
    if (!out || !g || !g_in) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipLaunchKernelGGL(k_bounded_sigmoid_bwd, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       out, g, g_in, (long long)n, lo, hi - lo);
    return check_launch(fn);
}

}  // extern "C"
