// HBM streaming microbenchmark (dev tool): read-only, write-only and copy bandwidth on this GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ in, float* out, long long n4) {
    float acc = 0.f;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ in, float4* __restrict__ out, long long n4) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) out[i] = in[i];
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ out, long long n4) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) out[i] = make_float4(1, 2, 3, 4);
}
// tiles: each block reads a contiguous tile of `tile_bytes` then jumps by gridDim tiles (the RQS access pattern)
__global__ __launch_bounds__(256) void k_read_tiles(const float4* __restrict__ in, float* out, long long n_tiles, int tile_v4) {
    float acc = 0.f;
    for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const float4* src = in + t * tile_v4;
        for (int i = threadIdx.x; i < tile_v4; i += 256) { float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const long long bytes = 2LL << 30; const long long n4 = bytes / 16;
    float4 *a, *b; float* o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double gb, auto&& launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, gb / (ms * 1e-3));
    };
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "read  U=1 grid=%d", g); timeit(nm, bytes / 1e9, [&] { k_read<1><<<g, 256>>>(a, o, n4); });
        snprintf(nm, 64, "read  U=4 grid=%d", g); timeit(nm, bytes / 1e9, [&] { k_read<4><<<g, 256>>>(a, o, n4); });
        snprintf(nm, 64, "read  U=8 grid=%d", g); timeit(nm, bytes / 1e9, [&] { k_read<8><<<g, 256>>>(a, o, n4); });
        snprintf(nm, 64, "copy      grid=%d", g); timeit(nm, 2 * bytes / 1e9, [&] { k_copy<<<g, 256>>>(a, b, n4); });
        snprintf(nm, 64, "write     grid=%d", g); timeit(nm, bytes / 1e9, [&] { k_write<<<g, 256>>>(b, n4); });
    }
    for (int tile_v4 : {1472, 4096}) for (int g : {768, 1536, 3072}) {
        char nm[64]; snprintf(nm, 64, "read tiles %dB grid=%d", tile_v4 * 16, g);
        long long nt = n4 / tile_v4;
        timeit(nm, nt * tile_v4 * 16 / 1e9, [&] { k_read_tiles<<<g, 256>>>(a, o, nt, tile_v4); });
    }
    return 0;
}
