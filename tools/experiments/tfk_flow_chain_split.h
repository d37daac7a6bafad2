// tfk_flow_chain_split.h -- affine coupling chains at D = 256 with every row SPLIT OVER TWO WAVES.
//
// EXPERIMENT, NOT BUILT INTO libtfk (round 3; kept for the record).  Correct (the D = 256 parity tests
// passed with it dispatched from launch_chain_k's streamed branch), but slower than the kernel it was to replace, same box,
// RealNVP-256, 2^19 rows: 733 us per launch against 622.  Ablations of THIS kernel: 691 us without the barrier, 681 without
// the operand DMA, 630 without both -- i.e. even with no synchronisation at all the 16-wave form only ties the 8-wave kernel
// WITH its barrier and DMA (543 us without them): twice the waves do not buy back the duplicated tanh / pending-map work,
// the LDS round trip of the partial pre-activations and a barrier that now sits in the middle of every coupling.
//
// k_flow_chain<32, 512, .., STREAM> (tfk_flow_chain.h) keeps 64 row elements per lane: 256 VGPRs, 2 waves per SIMD, and
// nothing to cover the one barrier per coupling that the streamed operands need (rocprofv3, profiles/r02/rnvp256:
// SQ_WAIT_ANY 39 % of the wave cycles, matrix pipe 44 %).  Here a PAIR of waves owns 16 rows: wave h of the pair holds, of
// every lane's 32 elements per plane, the 16 with local index 16 h .. 16 h + 15 -- 32 row registers, ~100 VGPRs, 16 waves
// per CU.  Both waves are busy in every coupling (unlike a split by plane, where the wave that holds the source plane
// would wait for the other's transform):
//   GEMM 1   each wave contracts ITS 64 source elements (16 of the 32 MFMAs), the two partial pre-activations (16 hidden x
//            16 rows, 1 KB per wave) are exchanged through the LDS;
//   GEMM 2   each wave computes the 8 tiles of ITS 16 targets per lane (24 of the 48 MFMAs) and transforms them.
// The operand blocks are the ones fused.py packs for the 256-wide kernel, unchanged: the k-steps of GEMM 1 and the tiles of
// GEMM 2 are lane-local, so "half of every lane" is "half of the k-step groups / half of the tiles".
// One barrier per coupling does three jobs: it publishes the partials, it publishes the operand block of the NEXT
// coupling (requested a whole coupling earlier: three blocks in the LDS, each wave waits for its own share before it
// arrives), and it frees the block of the previous coupling for the request that follows.
#pragma once
#include "tfk_flow_chain.h"

namespace tfk {

constexpr int kSplitBufs = 3;

// one coupling, this wave's half.  src / tgt: the wave's 16 elements per lane of the source / target plane.
template <int STEPS2, int KIND, bool FAST>
__device__ __forceinline__ void couple_split(const float *prm, float *xchg_mine, const float *xchg_other, int lane, int q,
                                             int h, const float (&src)[16], float (&tgt)[16], float &ld2, float &umin,
                                             const float *next_block, float *next_buf, int wave, int nwaves)
{
    constexpr int EPL = 32, HALF = 128, T2 = 16;
    constexpr int NA2 = (T2 * STEPS2 + 3) & ~3;
    constexpr int LB = chain_block_floats<32, STEPS2, KIND>();
    const cf32x4 *A1 = reinterpret_cast<const cf32x4 *>(prm);
    const float *b1 = prm + EPL * 64;
    const cf32x4 *A2 = reinterpret_cast<const cf32x4 *>(b1 + 16);
    const float *b2 = b1 + 16 + NA2 * 64;
    const float *pre = b2 + T2 * 16;

    // GEMM 1, this wave's k-steps (the bias rides with half 0)
    cf32x4 acc = h == 0 ? *reinterpret_cast<const cf32x4 *>(b1 + 4 * q) : cf32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const cf32x4 w = A1[(4 * h + g) * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k], src[4 * g + k], acc, 0, 0, 0);
    }
    *reinterpret_cast<cf32x4 *>(xchg_mine + 4 * lane) = acc;
    // this wave's targets take their pending elementwise layers now (one fma), beside the exchange
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const cf32x4 s = *reinterpret_cast<const cf32x4 *>(pre + EPL * q + 16 * h + 4 * i);
        const cf32x4 t = *reinterpret_cast<const cf32x4 *>(pre + HALF + EPL * q + 16 * h + 4 * i);
#pragma unroll
        for (int k = 0; k < 4; ++k) tgt[4 * i + k] = fmaf(s[k], tgt[4 * i + k], t[k]);
    }
    // my share of the NEXT coupling's block has landed; everyone's partial and share are visible behind the barrier,
    // and everyone has left the previous coupling's block
#if defined(TFK_SPLIT_NOBARRIER)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif defined(TFK_SPLIT_NOVMWAIT)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#ifdef TFK_SPLIT_NODMA
    if (false) {
#else
    if (next_block) {
#endif                                        // the request after next, into the block just freed
        typedef __attribute__((address_space(1))) const void *gptr_t;
        typedef __attribute__((address_space(3))) void *lptr_t;
        const char *srcb = reinterpret_cast<const char *>(next_block);
        for (int c = wave; c * 1024 < LB * 4; c += nwaves) {
            const int off = c * 1024 + lane * 16;
            if (off < LB * 4)
                __builtin_amdgcn_global_load_lds((gptr_t)(srcb + off), (lptr_t)(reinterpret_cast<char *>(next_buf) + c * 1024), 16, 0, 0);
        }
    }
    const cf32x4 other = *reinterpret_cast<const cf32x4 *>(xchg_other + 4 * lane);
    float hid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)                              // tanh, transforms.py:293-304 (both waves of the pair: 8 instructions)
        hid[r] = fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc[r] + other[r]) + 1.0f), 1.0f);

    // GEMM 2, this wave's 8 tiles, in groups whose A-operands fill whole ds_read_b128s
    constexpr int GT = (STEPS2 == 4) ? 1 : ((STEPS2 == 2) ? 2 : 4);
    constexpr int GREG = (GT * STEPS2 + 3) & ~3;
#pragma unroll
    for (int t0 = 0; t0 < 8; t0 += GT) {
        float a2[GREG];
#pragma unroll
        for (int g = 0; g < GREG / 4; ++g) {
            const cf32x4 w = A2[(((8 * h + t0) * STEPS2) / 4 + g) * 64 + lane];
            a2[4 * g] = w[0]; a2[4 * g + 1] = w[1]; a2[4 * g + 2] = w[2]; a2[4 * g + 3] = w[3];
        }
#pragma unroll
        for (int tt = 0; tt < GT; ++tt) {
            const int t = 8 * h + t0 + tt;                   // tile of the 256-wide layout; local elements 2 (t0 + tt), + 1
            cf32x4 o = *reinterpret_cast<const cf32x4 *>(b2 + (t * 4 + q) * 4);
#pragma unroll
            for (int k = 0; k < STEPS2; ++k)
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tt * STEPS2 + k], hid[k], o, 0, 0, 0);
            const int e0 = 2 * (t0 + tt);
            if constexpr (FAST && KIND == 1) {
                ld2 += log2_scales<true>(o[0], o[2], 0.0f, 0.0f, umin);
#pragma unroll
                for (int i = 0; i < 2; ++i) tgt[e0 + i] = (tgt[e0 + i] - o[2 * i + 1]) * __builtin_amdgcn_exp2f(-o[2 * i]);
            } else {
                const float al[2] = {__builtin_amdgcn_exp2f(o[0]) + kAffMinScale, __builtin_amdgcn_exp2f(o[2]) + kAffMinScale};
                ld2 += log2_scales<FAST>(o[0], o[2], al[0], al[1], umin);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if constexpr (KIND == 0) tgt[e0 + i] = fmaf(al[i], tgt[e0 + i], o[2 * i + 1]);
                    else tgt[e0 + i] = (tgt[e0 + i] - o[2 * i + 1]) * __builtin_amdgcn_rcpf(al[i]);
                }
            }
        }
    }
}

template <int STEPS2, int KIND, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_flow_chain_split(
    const float *__restrict__ x, float *z, float *logdet, const float *__restrict__ gauss_loc,
    const float *__restrict__ gauss_log_scale, float *logprob, long long N, const float *__restrict__ params,
    ChainProg prog, int flags)
{
    static_assert(KIND < 2 && STEPS2 >= 1 && STEPS2 <= 4, "affine couplings, fp32 operands");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int EPL = 32, D = 256, HALF = 128, NW = BLOCK / 64;
    constexpr int LB = chain_block_floats<32, STEPS2, KIND>();
    const int accumulate = flags & 1;
    const bool reverse_out = (flags & 2) != 0;
    const bool base_of_input = (flags & 4) != 0;
    float *ew_s = lds + kSplitBufs * LB;                             // closing TFK_OP_EW_FMA: s[D] | t[D] | const
    float *base_s = ew_s + 2 * D + 4;                                // loc[D] | 1 / scale[D] | const
    float *xchg = base_s + 2 * D + 4;                                // [2 parities][NW waves][64 lanes][4]
    float *pairbuf = xchg + 2 * NW * 256;                            // [NW / 2 pairs][16 rows][2]: half 1's log-det / squares
    if (prog.ew_offset >= 0)
        for (int i = threadIdx.x; i < 2 * D + 4; i += BLOCK) ew_s[i] = params[prog.ew_offset + i];
    if (logprob) {
        for (int e = threadIdx.x; e < D; e += BLOCK) {
            base_s[e] = gauss_loc[e];
            base_s[D + e] = expf(-gauss_log_scale[e]);
        }
        if (threadIdx.x < 64) {
            float c = 0.0f;
            for (int e = threadIdx.x; e < D; e += 64) c += gauss_log_scale[e] + kHalfLog2Pi;
            c = group_sum(c, 64);
            if (threadIdx.x == 0) base_s[2 * D] = c;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int h = wave & 1, pair = wave >> 1;
    const int n_c = prog.n_c > 0 ? prog.n_c : 1;
    // the first two blocks of the ring
    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    auto request = [&](int l, int buf) {
        const char *srcb = reinterpret_cast<const char *>(params + prog.offset[l]);
        for (int c = wave; c * 1024 < LB * 4; c += NW) {
            const int off = c * 1024 + lane * 16;
            if (off < LB * 4)
                __builtin_amdgcn_global_load_lds((gptr_t)(srcb + off), (lptr_t)(reinterpret_cast<char *>(lds + buf * LB) + c * 1024), 16, 0, 0);
        }
    };
    request(0, 0);
    request(1 % n_c, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float base_const = logprob ? base_s[2 * D] : 0.0f;
    long long step = 0;                                              // couplings done by this workgroup: ring position
    constexpr int rows_per_block = (NW / 2) * 16;
    const long long stride = (long long)gridDim.x * rows_per_block;
    for (long long blk0 = (long long)blockIdx.x * rows_per_block; blk0 < N; blk0 += stride) {
        const long long row = blk0 + pair * 16 + j;
        const long long rr = row < N ? row : N - 1;                  // tail: compute a valid row, store nothing
        float a[16], b[16];
        auto load_rows = [&]() {
            const float4 *pa = reinterpret_cast<const float4 *>(x + rr * D + EPL * q + 16 * h);
            const float4 *pb = reinterpret_cast<const float4 *>(x + rr * D + HALF + EPL * q + 16 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 va = pa[i], vb = pb[i];
                a[4 * i] = va.x; a[4 * i + 1] = va.y; a[4 * i + 2] = va.z; a[4 * i + 3] = va.w;
                b[4 * i] = vb.x; b[4 * i + 1] = vb.y; b[4 * i + 2] = vb.z; b[4 * i + 3] = vb.w;
            }
        };
        load_rows();
        float ld = (q == 0 && h == 0 && logdet && accumulate) ? logdet[rr] : 0.0f;
        float sq = 0.0f;
        auto base_terms = [&]() {                                    // gaussian.py:46-54, this wave's elements
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const cf32x4 la = *reinterpret_cast<const cf32x4 *>(base_s + EPL * q + 16 * h + 4 * i);
                const cf32x4 lb = *reinterpret_cast<const cf32x4 *>(base_s + HALF + EPL * q + 16 * h + 4 * i);
                const cf32x4 ia = *reinterpret_cast<const cf32x4 *>(base_s + D + EPL * q + 16 * h + 4 * i);
                const cf32x4 ib = *reinterpret_cast<const cf32x4 *>(base_s + D + HALF + EPL * q + 16 * h + 4 * i);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float ta = (a[4 * i + k] - la[k]) * ia[k];
                    const float tb = (b[4 * i + k] - lb[k]) * ib[k];
                    sq = fmaf(ta, ta, sq);
                    sq = fmaf(tb, tb, sq);
                }
            }
        };
        if (logprob && base_of_input) base_terms();                  // Flow.sample (flows.py:699-707)

        float ld2 = 0.0f, umin = 0.0f;
        auto chain = [&](auto fast_tag) {
            constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll 1
            for (int l = 0; l < prog.n_c; ++l, ++step) {
                const int buf = (int)(step % kSplitBufs), nbuf = (int)((step + 2) % kSplitBufs);
                const int par = (int)(step & 1);
                int l2 = l + 2;                                      // the coupling two steps ahead (wrapping into the next rows' chain)
                l2 = l2 >= prog.n_c ? l2 - prog.n_c : l2;
                l2 = l2 >= prog.n_c ? l2 - prog.n_c : l2;
                float *mine = xchg + (par * NW + wave) * 256;
                const float *other = xchg + (par * NW + (wave ^ 1)) * 256;
                if (((prog.first_src + l) & 1) == 0)
                    couple_split<STEPS2, KIND, FAST>(lds + buf * LB, mine, other, lane, q, h, a, b, ld2, umin,
                                                     params + prog.offset[l2], lds + nbuf * LB, wave, NW);
                else
                    couple_split<STEPS2, KIND, FAST>(lds + buf * LB, mine, other, lane, q, h, b, a, ld2, umin,
                                                     params + prog.offset[l2], lds + nbuf * LB, wave, NW);
            }
        };
        constexpr bool kShortcut = TFK_LOG_SHORTCUT != 0;
        if constexpr (kShortcut) {
            chain(std::true_type{});
            if (__syncthreads_or(umin < kLogShortcutMin)) {          // scales near the 1e-10 floor: with logarithms
                load_rows();
                ld2 = 0.0f;
                chain(std::false_type{});
            }
        } else {
            chain(std::false_type{});
        }
        if constexpr (KIND == 0) ld = fmaf(ld2, __int_as_float(0x3f317218), ld);           // ln 2
        else ld = fmaf(ld2, -__int_as_float(0x3f317218), ld);

        if (prog.ew_offset >= 0) {                                   // what is still pending, one fma per element
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const cf32x4 sa = *reinterpret_cast<const cf32x4 *>(ew_s + EPL * q + 16 * h + 4 * i);
                const cf32x4 sb = *reinterpret_cast<const cf32x4 *>(ew_s + HALF + EPL * q + 16 * h + 4 * i);
                const cf32x4 ta = *reinterpret_cast<const cf32x4 *>(ew_s + D + EPL * q + 16 * h + 4 * i);
                const cf32x4 tb = *reinterpret_cast<const cf32x4 *>(ew_s + D + HALF + EPL * q + 16 * h + 4 * i);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    a[4 * i + k] = fmaf(sa[k], a[4 * i + k], ta[k]);
                    b[4 * i + k] = fmaf(sb[k], b[4 * i + k], tb[k]);
                }
            }
            if (q == 0 && h == 0) ld = ld + ew_s[2 * D];
        }
        if (logprob && !base_of_input) base_terms();
        ld += __shfl_xor(ld, 16, kWave);
        ld += __shfl_xor(ld, 32, kWave);
        if (logprob) {
            sq += __shfl_xor(sq, 16, kWave);
            sq += __shfl_xor(sq, 32, kWave);
        }
        if (h == 1 && q == 0) {
            pairbuf[(pair * 16 + j) * 2] = ld;
            pairbuf[(pair * 16 + j) * 2 + 1] = sq;
        }
        if (row < N && z) {
            if (!reverse_out) {
                float4 *qa = reinterpret_cast<float4 *>(z + row * D + EPL * q + 16 * h);
                float4 *qb = reinterpret_cast<float4 *>(z + row * D + HALF + EPL * q + 16 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qa[i] = make_float4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
                    qb[i] = make_float4(b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]);
                }
            } else {                                                 // a reversal after the program, folded into the store
                float4 *qa = reinterpret_cast<float4 *>(z + row * D + D - EPL * (q + 1) + 16 * (1 - h));
                float4 *qb = reinterpret_cast<float4 *>(z + row * D + HALF - EPL * (q + 1) + 16 * (1 - h));
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qa[i] = make_float4(a[15 - 4 * i], a[14 - 4 * i], a[13 - 4 * i], a[12 - 4 * i]);
                    qb[i] = make_float4(b[15 - 4 * i], b[14 - 4 * i], b[13 - 4 * i], b[12 - 4 * i]);
                }
            }
        }
        __syncthreads();                                             // half 1's sums are in the LDS
        if (h == 0 && q == 0 && row < N) {
            ld += pairbuf[(pair * 16 + j) * 2];
            sq += pairbuf[(pair * 16 + j) * 2 + 1];
            if (logdet) logdet[row] = ld;
            if (logprob) logprob[row] = (fmaf(-0.5f, sq, -base_const)) + ld;              // flows.py:648
        }
        // (pairbuf is rewritten after the next rows' chain, i.e. behind at least one more barrier)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (blocks requested for steps that never came)
}

template <int STEPS2, int KIND>
static int launch_chain_split(const float *x, float *z, float *logdet, const float *loc, const float *log_scale,
                              float *logprob, int64_t N, const float *params, const ChainProg &prog, int flags,
                              hipStream_t s, const char *fn)
{
    constexpr int BLOCK = 1024, NW = BLOCK / 64, D = 256;
    constexpr size_t lds = ((size_t)kSplitBufs * chain_block_floats<32, STEPS2, KIND>() + 2 * (2 * D + 4) + 2 * NW * 256 +
                            (NW / 2) * 32) * sizeof(float);
    static_assert(lds <= 160 * 1024, "three operand blocks + the exchange buffers fit the LDS");
    auto kern = &k_flow_chain_split<STEPS2, KIND, BLOCK>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", fn, lds);
        }
        attr_done = true;
    }
    constexpr int rows_per_block = (NW / 2) * 16;
    const int64_t want = (N + rows_per_block - 1) / rows_per_block;
    const int64_t cap = (int64_t)cu_count() * TFK_CHAIN_OVERSUB;
    const int grid = (int)(want < cap ? want : cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, s, x, z, logdet, loc, log_scale, logprob, (long long)N, params,
                       prog, flags);
    return check_launch(fn);
}

}  // namespace tfk
