// tfk_rqs.hip -- rational-quadratic spline coupling (Neural Spline Flow layer).
//
// Replaces CouplingBijection.forward/inverse (layers_base.py:145-163) around
// MonotonicSpline.forward/inverse (spline/base.py:53-72) and
// RationalQuadratic.rqs_forward_1d / rqs_inverse_1d (rational_quadratic.py:45-200).
//
// Data movement (the kernel is priced against HBM: 4*(D + T*P + D) + 8 bytes/row):
//   * h is (N, T, P = 3K-1) contiguous, i.e. one flat stream of P-float parameter
//     records, 92 B each for K = 8 -- not 16-byte aligned per record.  A workgroup
//     takes a tile of R = 256/T rows = up to 256 records (23.5 KB for K = 8), pulls it
//     from HBM with fully coalesced 16-byte loads (1 KiB per wave-instruction) and
//     parks it in LDS; each lane then reads its own record with ds_read_b32 at a
//     P-dword stride, which is bank-conflict-free for odd P (23, 11, 47).
//   * one lane = one spline element: 2 softmaxes (K exps each), knot cumsum, bin
//     search by compare/select (no indexed registers), only the two derivatives
//     delta_k, delta_k+1 that the selected bin needs (fetched from LDS by index).
//   * per-row log-det: __shfl_xor inside the T-lane group when T is a power of two
//     <= 64 (T = 32 for D = 64), LDS otherwise.  No atomics, deterministic.
//   * out-of-box elements (strict (-B, B), spline/base.py:29-33) pass through with
//     zero log-det, branch per lane; no host sync (the reference's torch.any / assert
//     syncs have no counterpart).
// Arithmetic keeps the reference's fp32 op order, including ATen's CPU softmax
// (e * (1/sum)); built with -ffp-contract=off.
#include <cmath>

#include "tfk_common.h"
#include "tfk_spline.h"

namespace tfk {

// TILE = threads per workgroup = spline elements (parameter records) per LDS tile.
// TILE = 64: every wavefront is its own workgroup with its own 5.9 KB tile -- no cross-wave
// barrier, up to 7 independent waves per SIMD whose load / evaluate phases interleave freely.

// Dynamic LDS layout: [TILE * P + 4 floats of records | TILE floats of log-dets |
//                      D bytes of target mask (only tgt_idx && !inplace)]
template <int KT, bool INVERSE, int TILE>
__global__ __launch_bounds__(TILE) void k_rqs_coupling(
    const float *x, const float *__restrict__ h, float *z, float *logdet, long long N, int D,
    const int *__restrict__ tgt_idx, int T, int T_shift, int Krt, RqsConst C, int accumulate,
    int inplace, int h_vec_ok)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = KT > 0 ? KT : Krt;
    const int P = 3 * K - 1;
    float *rec = lds;
    float *ld_s = lds + TILE * P + 4;
    unsigned char *is_tgt = reinterpret_cast<unsigned char *>(ld_s + TILE);
    const int tid = threadIdx.x;
    const bool use_mask = (tgt_idx != nullptr) && !inplace;
    if (use_mask) {
        for (int e = tid; e < D; e += TILE) is_tgt[e] = 0;
        __syncthreads();
        for (int t = tid; t < T; t += TILE) is_tgt[tgt_idx[t]] = 1;
    }

    const int R = T <= TILE ? TILE / T : 1;          // rows per tile
    const int chunks = T <= TILE ? 1 : (T + TILE - 1) / TILE;
    const long long n_tiles = (N + R - 1) / R;
    const bool shfl_reduce = (T_shift >= 0) && (T <= kWave);

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row0 = tile * R;
        const int rows = (int)((N - row0) < (long long)R ? (N - row0) : (long long)R);
        float ld_thread = 0.0f;    // T > TILE: this thread's share of the (single) row

        for (int ch = 0; ch < chunks; ++ch) {
            const int cbase = ch * TILE;
            const int E = (chunks == 1) ? rows * T : ((T - cbase) < TILE ? (T - cbase) : TILE);
            const long long hoff = (row0 * (long long)T + cbase) * P;    // floats
            const int nfl = E * P;

            __syncthreads();   // previous tile's readers are done with rec / ld_s
            if (h_vec_ok && (hoff & 3) == 0) {
                const float4 *src = reinterpret_cast<const float4 *>(h + hoff);
                float4 *dst = reinterpret_cast<float4 *>(rec);
                const int nv = nfl >> 2;
                for (int i = tid; i < nv; i += TILE) dst[i] = nt_load4(src + i);
                for (int i = (nv << 2) + tid; i < nfl; i += TILE) rec[i] = nt_load(h + hoff + i);
            } else {
                for (int i = tid; i < nfl; i += TILE) rec[i] = nt_load(h + hoff + i);
            }
            __syncthreads();

            float ld = 0.0f;
            long long row = row0;
            int t = cbase + tid;
            if (chunks == 1) {
                const int r = T_shift >= 0 ? (tid >> T_shift) : (tid / T);
                t = tid - r * T;
                row = row0 + r;
            }
            if (tid < E) {
                const int idx = tgt_idx ? tgt_idx[t] : D - T + t;
                const float v = x[row * D + idx];
                float o = v;                                       // spline/base.py:54-55
                if (v > C.minimum && v < C.maximum)                // strict, base.py:29-33
                    rqs_eval<KT, INVERSE>(rec + tid * P, K, v, C, o, ld);
                z[row * D + idx] = o;
            }

            if (chunks > 1) {
                ld_thread += ld;
            } else if (shfl_reduce) {
                const float sum = group_sum(ld, T);                 // sum_except_batch, base.py:59
                if (tid < E && t == 0)
                    logdet[row] = accumulate ? logdet[row] + sum : sum;
            } else {
                ld_s[tid] = ld;
                __syncthreads();
                if (tid < rows) {
                    float sum = 0.0f;
                    for (int j = 0; j < T; ++j) sum += ld_s[tid * T + j];
                    logdet[row0 + tid] = accumulate ? logdet[row0 + tid] + sum : sum;
                }
            }
        }

        if (chunks > 1) {
            __syncthreads();
            ld_s[tid] = ld_thread;
            __syncthreads();
            for (int o = TILE / 2; o > 0; o >>= 1) {
                if (tid < o) ld_s[tid] += ld_s[tid + o];
                __syncthreads();
            }
            if (tid == 0) logdet[row0] = accumulate ? logdet[row0] + ld_s[0] : ld_s[0];
        }

        if (!inplace) {                                            // clone, layers_base.py:146
            const int total = rows * D;
            for (int e = tid; e < total; e += TILE) {
                const int r = e / D;
                const int c = e - r * D;
                const bool tgt = tgt_idx ? (is_tgt[c] != 0) : (c >= D - T);
                if (!tgt) z[(row0 + r) * D + c] = x[(row0 + r) * D + c];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// LDS-DMA pipeline (the fast path: compile-time K, T a power of two <= 64, 16-byte
// aligned tiles).  Per tile of 256 records (23 KiB for K = 8):
//   1. barrier: the tile's DMA has landed (its s_waitcnt vmcnt(0) is the only thing that
//      orders an LDS-DMA against ds_reads);
//   2. every lane copies its own record LDS -> registers (P ds_read_b32 at a P-dword stride,
//      conflict-free for odd P);
//   3. barrier: the LDS tile is free again -> global_load_lds_dwordx4 for the NEXT tile is
//      issued right away (no VGPR staging, 1 KiB per wave-instruction, six in flight per
//      wave) together with the prefetch of the next tile's inputs x;
//   4. ~500 VALU ops per element run out of registers while that DMA is in flight.
// A single 23 KiB buffer per workgroup keeps 6 workgroups = 6 waves per SIMD resident.
// Dynamic LDS: (256*P + 8) floats [+ D bytes of target mask].
// ---------------------------------------------------------------------------
template <int KT, bool INVERSE>
__global__ __launch_bounds__(kBlock) void k_rqs_coupling_dma(
    const float *x, const float *__restrict__ h, float *z, float *logdet, long long N, int D,
    const int *__restrict__ tgt_idx, int T, int T_shift, RqsConst C, int accumulate, int inplace)
{
    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int P = 3 * KT - 1;
    constexpr int BUF = kBlock * P + 8;                 // floats (16-byte multiple)
    unsigned char *is_tgt = reinterpret_cast<unsigned char *>(lds + BUF);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool use_mask = (tgt_idx != nullptr) && !inplace;
    if (use_mask) {
        for (int e = tid; e < D; e += kBlock) is_tgt[e] = 0;
        __syncthreads();
        for (int t = tid; t < T; t += kBlock) is_tgt[tgt_idx[t]] = 1;
    }
    const int R = kBlock >> T_shift;                    // rows per tile
    const long long n_tiles = (N + R - 1) / R;
    const int r = tid >> T_shift, t = tid & (T - 1);
    const int idx = tgt_idx ? tgt_idx[t] : D - T + t;

    auto issue = [&](long long tile) {                  // this wave's share of a tile's records
        const long long row0 = tile * R;
        const int rows = (int)((N - row0) < (long long)R ? (N - row0) : (long long)R);
        const int nbytes = rows * T * P * 4;            // multiple of 16 (host checks T*P % 4 == 0)
        const char *src = reinterpret_cast<const char *>(h + row0 * (long long)T * P);
        for (int c = w; c * 1024 < nbytes; c += kBlock / 64) {
            const int off = c * 1024 + lane * 16;
            if (off < nbytes)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + off),
                                                 (lptr_t)(reinterpret_cast<char *>(lds) + c * 1024),
                                                 16, 0, kDmaNonTemporal);           // h is read once
        }
    };

    long long tile = blockIdx.x;
    float v_cur = 0.0f;
    if (tile < n_tiles) {
        issue(tile);
        const long long row = tile * R + r;
        if (row < N) v_cur = x[row * D + idx];
    }
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();                                 // (1) this tile's DMA has landed
        float p[P];
#pragma unroll
        for (int j = 0; j < P; ++j) p[j] = lds[tid * P + j];   // (2)
        __syncthreads();                                 // (3) everyone holds its record
        const long long nxt = tile + gridDim.x;
        float v_nxt = 0.0f;
        if (nxt < n_tiles) {
            issue(nxt);
            const long long rown = nxt * R + r;
            if (rown < N) v_nxt = x[rown * D + idx];
        }
        const long long row0 = tile * R;
        const long long row = row0 + r;
        const bool live = row < N;
        float ld = 0.0f;
        if (live) {                                      // (4)
            const float v = v_cur;
            float o = v;                                       // spline/base.py:54-55
            if (v > C.minimum && v < C.maximum)                // strict, base.py:29-33
                rqs_eval<KT, INVERSE, true, float[P]>(p, KT, v, C, o, ld);
            z[row * D + idx] = o;
        }
        const float sum = group_sum(ld, T);                     // sum_except_batch, base.py:59
        if (live && t == 0) logdet[row] = accumulate ? logdet[row] + sum : sum;
        if (!inplace) {                                         // clone, layers_base.py:146
            const int rows = (int)((N - row0) < (long long)R ? (N - row0) : (long long)R);
            const int total = rows * D;
            for (int e = tid; e < total; e += kBlock) {
                const int rr = e / D;
                const int c = e - rr * D;
                const bool tgt = tgt_idx ? (is_tgt[c] != 0) : (c >= D - T);
                if (!tgt) z[(row0 + rr) * D + c] = x[(row0 + rr) * D + c];
            }
        }
        v_cur = v_nxt;
    }
}

template <bool INVERSE>
static int rqs_coupling(const float *x, const float *h, float *z, float *logdet, int64_t N,
                        int32_t D, const int32_t *tgt_idx, int32_t T, int32_t K, float boundary,
                        int32_t accumulate, void *stream, const char *fn)
{
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D <= 0) return fail(TFK_EINVAL, "%s: D = %d must be positive", fn, D);
    if (T <= 0 || T > D) return fail(TFK_EINVAL, "%s: T = %d must be in [1, D = %d]", fn, T, D);
    if (K < 2 || K > 32) return fail(TFK_EINVAL, "%s: n_bins K = %d must be in [2, 32]", fn, K);
    if (!(boundary > 0.0f)) return fail(TFK_EINVAL, "%s: boundary must be positive", fn);
    if (N == 0) return TFK_OK;
    if (!x || !h || !z || !logdet) return fail(TFK_EINVAL, "%s: null pointer", fn);
    const bool inplace = (x == z);
    const int P = 3 * K - 1;
    // one wavefront per workgroup unless a row needs more than 64 lanes' worth of shuffles
    const int tile = 256;   // (64 = one wavefront per workgroup measured slower: 293 vs 229 us at C3)
    size_t lds = ((size_t)tile * P + 4 + tile) * sizeof(float);
    if (tgt_idx && !inplace) lds += (size_t)D;
    if (lds > 160 * 1024) return fail(TFK_EINVAL, "%s: LDS tile of %zu bytes exceeds 160 KiB", fn, lds);
    if (lds > 64 * 1024) {
        // large K (run-time-K kernel only): opt in to more than 64 KiB of dynamic LDS
        hipError_t e = hipFuncSetAttribute(tile == 64 ? reinterpret_cast<const void *>(&k_rqs_coupling<0, INVERSE, 64>)
                                                      : reinterpret_cast<const void *>(&k_rqs_coupling<0, INVERSE, 256>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS: %s", fn, lds, hipGetErrorString(e));
        }
    }

    RqsConst C;
    C.minimum = -boundary;
    C.maximum = boundary;
    C.span = (float)((double)boundary - (double)(-boundary));
    C.scale = (float)(1.0 - 1e-3 * (double)K);
    C.c = (float)std::log(std::expm1(1.0 - 1e-5));

    int T_shift = -1;
    if ((T & (T - 1)) == 0) {
        T_shift = 0;
        while ((1 << T_shift) < T) ++T_shift;
    }
    const int R = T <= tile ? tile / T : 1;
    const int64_t n_tiles = (N + R - 1) / R;
    // LDS-bound residency: 160 KiB / tile; cap the grid there and stride the rest
    int per_cu = (int)((160 * 1024) / lds);
    const int max_blocks = 2048 / tile > 28 ? 28 : 2048 / tile;     // 32 waves per CU, ~7 per SIMD by VGPRs
    if (per_cu < 1) per_cu = 1;
    if (per_cu > max_blocks) per_cu = max_blocks;
    int64_t grid = n_tiles < (int64_t)cu_count() * per_cu ? n_tiles : (int64_t)cu_count() * per_cu;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int h_vec_ok = aligned16(h) ? 1 : 0;

    // fast path: LDS-DMA double buffering
    if ((K == 8 || K == 4) && T_shift >= 0 && T <= 64 && ((T * P) & 3) == 0 && h_vec_ok) {
        size_t lds2 = ((size_t)kBlock * P + 8) * sizeof(float) + ((tgt_idx && !inplace) ? (size_t)D : 0);
        if (lds2 <= 64 * 1024) {
            const int Rr = kBlock >> T_shift;
            const int64_t tiles = (N + Rr - 1) / Rr;
            int pc = (int)((160 * 1024) / lds2);
            if (pc > 8) pc = 8;
            const int64_t g = tiles < (int64_t)cu_count() * pc ? tiles : (int64_t)cu_count() * pc;
            if (K == 8)
                hipLaunchKernelGGL((k_rqs_coupling_dma<8, INVERSE>), dim3((unsigned)g), dim3(kBlock), lds2, s,
                                   x, h, z, logdet, (long long)N, D, tgt_idx, T, T_shift, C, accumulate, inplace ? 1 : 0);
            else
                hipLaunchKernelGGL((k_rqs_coupling_dma<4, INVERSE>), dim3((unsigned)g), dim3(kBlock), lds2, s,
                                   x, h, z, logdet, (long long)N, D, tgt_idx, T, T_shift, C, accumulate, inplace ? 1 : 0);
            return check_launch(fn);
        }
    }

#define TFK_RQS_LAUNCH(KT_, TILE_)                                                                  \
    hipLaunchKernelGGL((k_rqs_coupling<KT_, INVERSE, TILE_>), dim3((unsigned)grid), dim3(TILE_), lds, s, \
                       x, h, z, logdet, (long long)N, D, tgt_idx, T, T_shift, K, C, accumulate,    \
                       inplace ? 1 : 0, h_vec_ok)
    if (tile == 64) {
        if (K == 8) TFK_RQS_LAUNCH(8, 64);
        else if (K == 4) TFK_RQS_LAUNCH(4, 64);
        else TFK_RQS_LAUNCH(0, 64);
    } else {
        if (K == 8) TFK_RQS_LAUNCH(8, 256);
        else if (K == 4) TFK_RQS_LAUNCH(4, 256);
        else TFK_RQS_LAUNCH(0, 256);
    }
#undef TFK_RQS_LAUNCH
    return check_launch(fn);
}

}  // namespace tfk

extern "C" {

int tfk_rqs_coupling_fwd(const float *x, const float *h, float *z, float *logdet, int64_t N,
                         int32_t D, const int32_t *tgt_idx, int32_t T, int32_t K, float boundary,
                         int32_t accumulate, void *stream)
{
    return tfk::rqs_coupling<false>(x, h, z, logdet, N, D, tgt_idx, T, K, boundary, accumulate,
                                    stream, "tfk_rqs_coupling_fwd");
}

int tfk_rqs_coupling_inv(const float *z, const float *h, float *x, float *logdet, int64_t N,
                         int32_t D, const int32_t *tgt_idx, int32_t T, int32_t K, float boundary,
                         int32_t accumulate, void *stream)
{
    return tfk::rqs_coupling<true>(z, h, x, logdet, N, D, tgt_idx, T, K, boundary, accumulate,
                                   stream, "tfk_rqs_coupling_inv");
}

}  // extern "C"
