// tfk_common.h -- shared host/device helpers of libtfk (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tfk.h"

namespace tfk {

constexpr int kWave = 64;         // CDNA4 wavefront
constexpr int kBlock = 256;       // 4 waves, one per SIMD
// compute units of the device the calling thread is on, asked once per process (256 on MI355X; the same
// 256 when no device answers, so that the size queries stay usable without a GPU)
int cu_count();
// Grid of the compute-bound flow-program kernels, in resident sets of workgroups.  With exactly
// one resident set every CU gets the same share and the launch ends with the slowest CU; measured
// on MI355X (RealNVP D=64, 2^20 rows): 1 set 557 us, 1.5 sets 543, 2 sets 531, 4 sets 511,
// 8 sets 504, 16 sets 511 (each workgroup re-stages its parameter block from L2).
constexpr int kGridOversubscribe = 4;
inline int max_grid() { return cu_count() * 8; }  // memory-bound kernels: <= 8 resident blocks per CU, grid-stride the rest

// affine.py:19-23 -- python doubles rounded once to fp32, as ATen does when a python
// scalar meets an fp32 tensor: m = 1e-10, c0 = log(1 - 1e-10) = -1.000000082790371e-10
constexpr float kAffMinScale = 1e-10f;
constexpr float kAffC0 = -1.000000082790371e-10f;

// rational_quadratic.py:36-38
constexpr float kRqsMinBin = 1e-3f;
constexpr float kRqsMinDelta = 1e-5f;

// gaussian.py:26,53: 0.5 * log(2 * pi)
constexpr float kHalfLog2Pi = 0.9189385332046727f;

// ---- lean math: same values as the ocml routines on the ranges used here ----------------
// (measured on gfx950: ocml expf = 12 VALU ops, logf = 11, log1pf = 121 (!), IEEE '/' = 9)

// n / d as q = n*r, q += r*(n - d*q) with r = v_rcp_f32(d): the quotient-correction step of the
// IEEE division expansion without its range scaling and reciprocal refinement.  Within half an
// ulp plus ~1e-7 ulp of n/d, i.e. the correctly rounded quotient except on near-ties, for
// operands whose quotient and reciprocal are normal numbers (true everywhere it is used: the
// divisors are scale factors, bin widths and softmax sums).  4 ops.
__device__ __forceinline__ float div_fast(float n, float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = n * r;
    return fmaf(fmaf(-d, q, n), r, q);
}

// v / 1000 (the reference divides its spline logits by 1000, rational_quadratic.py:76-77)
__device__ __forceinline__ float div_1000(float v) {
    const float r = 1.0f / 1000.0f;                  // compile-time constant
    const float q = v * r;
    return fmaf(fmaf(-1000.0f, q, v), r, q);
}

// expf for arguments that cannot overflow (x <= ~88): ocml's own argument reduction
// (x*log2(e) split hi/lo, v_exp_f32, ldexp) without its overflow / underflow selects --
// bit-identical to expf on that range, underflows to denormals / 0 through ldexp.  9 ops.
__device__ __forceinline__ float exp_noovf(float x) {
    const float L2E_HI = __int_as_float(0x3fb8aa3b);    // 1.44269502
    const float L2E_LO = __int_as_float(0x32a5705f);    // 1.92596303e-08
    const float t = x * L2E_HI;
    const float n = __builtin_rintf(t);
    const float e = fmaf(L2E_LO, x, fmaf(x, L2E_HI, -t));
    const float f = (t - n) + e;
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// exp(x) in 5 ops for results that are ADDED to something of order one or larger right away (the
// softmax numerators after the max-subtraction, alpha = exp(.) + 1e-10): exp2(t) * 2^e with the product
// t = x * log2(e) compensated to first order, 2^e = 1 + e ln 2 (e ~ 1e-7 |t|: the dropped e^2 term is below
// 1e-13).  Within 1 ulp of exp_noovf (v_exp_f32 sees the whole t instead of its reduced fraction; both are
// 1-ulp evaluations); results below the normal range flush to zero instead of going through ldexp.
__device__ __forceinline__ float exp_lean(float x) {
    const float L2E_HI = __int_as_float(0x3fb8aa3b);
    const float LN2 = __int_as_float(0x3f317218);
    const float t = x * L2E_HI;
    const float e = fmaf(x, L2E_HI, -t);              // what rounding t lost (log2(e)'s own low part, 1.9e-8 |x|
    const float p = __builtin_amdgcn_exp2f(t);        // relative, is left out: < 0.3 ulp for |x| < 3)
    return p * fmaf(LN2, e, 1.0f);                    // (inf stays inf, 0 stays 0)
}

// log(x) as v_log_f32 * ln 2, 2 ops: the hi/lo split of ln 2 in log_normal only removes the rounding of this
// one multiplication (<= 0.5 ulp of the result) while v_log_f32 itself is a 1-ulp evaluation; inf / NaN / 0
// propagate through the multiplication.  For the datapath-bound flow programs.
__device__ __forceinline__ float log_lean(float x) {
    return __builtin_amdgcn_logf(x) * __int_as_float(0x3f317218);
}

// logf for positive NORMAL arguments (and +inf / NaN, passed through): ocml's own sequence
// v_log_f32 -> * ln2 split hi/lo -> add, without its denormal pre-scaling -- bit-identical to
// logf on that domain.  Every log on this path takes a scale factor alpha >= 1e-10, a spline
// slope, or 1 + exp(.) >= 1.  7 ops instead of 13.
__device__ __forceinline__ float log_normal(float x) {
    const float LN2_HI = __int_as_float(0x3f317217);    // 0.693147123
    const float LN2_LO = __int_as_float(0x3377d1cf);    // 5.77e-08
    const float r = __builtin_amdgcn_logf(x);           // log2(x), v_log_f32
    const float hi = r * LN2_HI;
    const float y = hi + fmaf(r, LN2_LO, fmaf(r, LN2_HI, -hi));
    return __builtin_fabsf(r) < __builtin_inff() ? y : r;
}

// constrain_scale, affine.py:33-34: exp(c0 + u / 2) + m   (u / 2 == u * 0.5 exactly).
// exp_noovf == expf bit for bit wherever expf is finite, and overflows / underflows to the same
// inf / 0 through ldexp.
__device__ __forceinline__ float aff_alpha(float u) {
    return exp_noovf(u * 0.5f + kAffC0) + kAffMinScale;
}
// the same in the VALU-issue-bound flow programs: 3 ops fewer per element, within 1 ulp of the above
__device__ __forceinline__ float aff_alpha_lean(float u) {
    return exp_lean(u * 0.5f + kAffC0) + kAffMinScale;
}

// log1p(y) for y >= 0: log(u) + (y - (u - 1)) / u with u = fl(1 + y) -- the second term gives
// back what rounding 1 + y lost, so the result is within ~1 ulp like log1pf, at 1/7 of its cost.
__device__ __forceinline__ float log1p_pos(float y) {
    const float u = 1.0f + y;
    const float lost = y - (u - 1.0f);
    return log_normal(u) + lost * __builtin_amdgcn_rcpf(u);
}

// sum over the G (power of two, <= 64) consecutive lanes that share a row
// Non-temporal access for streams that are touched once (the conditioner's output h: 2 T or 23 T floats per row, written
// by a GEMM and read by exactly one transform kernel; the transformed rows of a 2^20-row batch: 268 MB, gone from every
// cache before the next kernel asks for them): the lines bypass the L2's retention and leave it to the data that is
// reused.  Measured at 2^20 rows (profiles/r02, bench.py --no-fused): tfk_affine_coupling_fwd 121.2 -> 105.7 us
// (4.50 -> 5.16 TB/s), tfk_rqs_coupling_fwd 213.1 -> 157.9 us (3.95 -> 5.32 TB/s).
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4 *p)
{
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float nt_load(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void nt_store4(float4 *p, const float4 &v)
{
    __builtin_nontemporal_store(nt_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<nt_f4 *>(p));
}
constexpr int kDmaNonTemporal = 2;      // aux operand of global_load_lds: the NT bit of the gfx940+ cache policy

// Sum of the log-probabilities a launch produced, in fp64 and in a fixed order, without further launches (feeds the
// all-reduce of SURVEY.md 8(e)): every lane hands in the fp64 sum of the rows it wrote; the workgroup's total goes to
// ws[1 + blockIdx.x]; the LAST workgroup to finish (ticket from the counter in ws[0]) adds the partials in index order,
// writes out[0] and resets the counter for the next launch.
// Visibility across the 8 XCDs' L2s WITHOUT release / acquire fences (an agent-scope release writes the whole L2 back:
// +24 us on a 242 us launch when 3 072 workgroups each do it, measured): the partial is an agent-scope ATOMIC store
// (written through), its completion is awaited (vmcnt) before the agent-scope ticket increment, and the last workgroup
// reads the partials with agent-scope atomic loads.
// `scratch`: >= BLOCK / 64 + 2 doubles of LDS nobody else touches any more.
template <int BLOCK>
__device__ __forceinline__ void finish_sum_f64(double acc, double *scratch, double *ws, double *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = BLOCK / 64;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, kWave);
    __syncthreads();                                             // the LDS is free from here on
    if (lane == 0) scratch[wave] = acc;
    __syncthreads();
    unsigned long long *counter = reinterpret_cast<unsigned long long *>(ws);
    double *partials = ws + 1;
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < NW; ++w) t += scratch[w];
        __hip_atomic_store(partials + blockIdx.x, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the partial is at the coherence point
        const unsigned long long ticket =
            __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        scratch[NW] = (ticket == (unsigned long long)gridDim.x - 1ull) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (scratch[NW] == 0.0) return;
    double a = 0.0;                                              // (last workgroup: every partial has been written)
    for (int i = threadIdx.x; i < (int)gridDim.x; i += BLOCK)
        a += __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, kWave);
    __syncthreads();
    if (lane == 0) scratch[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < NW; ++w) t += scratch[w];
        out[0] = t;
        __hip_atomic_store(counter, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ float group_sum(float v, int G) {
    for (int o = G >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// host: error reporting (thread-local text behind tfk_last_error)
int fail(int code, const char *fmt, ...);
int check_launch(const char *what);

inline int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

inline int grid_for(int64_t units_of_work, int per_block) {
    int64_t b = (units_of_work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_grid()) b = max_grid();
    return (int)b;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace tfk
