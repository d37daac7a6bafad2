// tfk_common.h -- shared host/device helpers of libtfk (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tfk.h"

namespace tfk {

constexpr int kWave = 64;         // CDNA4 wavefront
constexpr int kBlock = 256;       // 4 waves, one per SIMD
constexpr int kCUs = 256;         // MI355X
constexpr int kMaxGrid = kCUs * 8;  // memory-bound kernels: <= 8 resident blocks per CU, grid-stride the rest

// affine.py:19-23 -- python doubles rounded once to fp32, as ATen does when a python
// scalar meets an fp32 tensor: m = 1e-10, c0 = log(1 - 1e-10) = -1.000000082790371e-10
constexpr float kAffMinScale = 1e-10f;
constexpr float kAffC0 = -1.000000082790371e-10f;

// rational_quadratic.py:36-38
constexpr float kRqsMinBin = 1e-3f;
constexpr float kRqsMinDelta = 1e-5f;

// gaussian.py:26,53: 0.5 * log(2 * pi)
constexpr float kHalfLog2Pi = 0.9189385332046727f;

// constrain_scale, affine.py:33-34: exp(c0 + u / 2) + m   (u / 2 == u * 0.5 exactly)
__device__ __forceinline__ float aff_alpha(float u) {
    return expf(u * 0.5f + kAffC0) + kAffMinScale;
}

// sum over the G (power of two, <= 64) consecutive lanes that share a row
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
    if (b > kMaxGrid) b = kMaxGrid;
    return (int)b;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace tfk
