// Scratch microbenchmark: what a pure write stream reaches on MI355X, in the two shapes the
// noise-shaped PWM bank could use for its duty bytes (duty[tick][channel], 1 B per channel-tick):
//   row4 : lane owns 4 adjacent channels, one 4-byte store per lane per tick (a wave writes 256
//          contiguous bytes of one row per instruction) -- the shape pwm_bank_kernel uses
//   row16: 4 ticks are transposed among 4 adjacent lanes first, one 16-byte store per lane per 4
//          ticks (a wave writes 256 contiguous bytes in each of 4 rows per instruction)
//   flat : plain streaming fill, 16 B per lane, consecutive addresses (upper bound)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void w_row4(uint32_t *out, uint32_t n4 /* n/4 */, uint32_t nticks)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n4) return;
    uint32_t v = g;
    uint32_t *p = out + g;
    for (uint32_t t = 0; t < nticks; t++) {
        v = v * 1664525u + 1013904223u;
        if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        p += n4;
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void w_row16(uint32_t *out, uint32_t n4, uint32_t nticks)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= n4) return;
    uint32_t v = g;
    // lane 4q+j writes row t+j, dwords 4q..4q+3 of its wave's 64-dword span
    const uint32_t lane = threadIdx.x & 63, j = lane & 3, q = lane >> 2;
    const uint32_t base = (g & ~63u) + 4 * q;
    u32x4 *p = reinterpret_cast<u32x4 *>(out + (size_t)j * n4 + base);
    for (uint32_t t = 0; t < nticks; t += 4) {
        u32x4 x;
        v = v * 1664525u + 1013904223u; x.x = v;
        v = v * 1664525u + 1013904223u; x.y = v;
        v = v * 1664525u + 1013904223u; x.z = v;
        v = v * 1664525u + 1013904223u; x.w = v;
        if (NT) __builtin_nontemporal_store(x, p); else *p = x;
        p += n4;            // 4 rows of n4 dwords = n4 u32x4
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void w_flat(u32x4 *out, size_t nvec)
{
    const size_t stride = (size_t)gridDim.x * 256u;
    for (size_t i = blockIdx.x * 256u + threadIdx.x; i < nvec; i += stride) {
        u32x4 x = (uint32_t)i;
        if (NT) __builtin_nontemporal_store(x, out + i); else out[i] = x;
    }
}

// per-lane contiguous chunks of K x 16 B (a wave covers K KiB), grid-stride over chunks
template <int K, int BS>
__global__ __launch_bounds__(BS) void w_chunk(u32x4 *out, size_t nvec)
{
    const size_t stride = (size_t)gridDim.x * BS * K;
    for (size_t i = ((size_t)blockIdx.x * BS + threadIdx.x) * K; i < nvec; i += stride) {
#pragma unroll
        for (int k = 0; k < K; k++) { u32x4 x = (uint32_t)(i + k); out[i + k] = x; }
    }
}
// wave-contiguous: K consecutive 1-KiB wave stores per trip (lane stride 16 B, instruction stride 1 KiB)
template <int K, int BS>
__global__ __launch_bounds__(BS) void w_wavechunk(u32x4 *out, size_t nvec)
{
    const size_t stride = (size_t)gridDim.x * BS * K;
    for (size_t i = (size_t)blockIdx.x * BS * K + threadIdx.x; i < nvec; i += stride) {
#pragma unroll
        for (int k = 0; k < K; k++) { u32x4 x = (uint32_t)(i + k); out[i + (size_t)k * BS] = x; }
    }
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3 * reps; i++) f();          // sustained rate: short bursts are partly absorbed by the Infinity Cache
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

int main() {
    const uint32_t n = 1u << 20, nticks = 1024, n4 = n / 4;
    const size_t bytes = (size_t)n * nticks;
    uint32_t *out; (void)hipMalloc(&out, bytes);
    float a = timeit([&] { hipLaunchKernelGGL((w_row4<false>), dim3(n4 / 256), dim3(256), 0, 0, out, n4, nticks); }, 30);
    float b = timeit([&] { hipLaunchKernelGGL((w_row4<true>), dim3(n4 / 256), dim3(256), 0, 0, out, n4, nticks); }, 30);
    float c = timeit([&] { hipLaunchKernelGGL((w_row16<false>), dim3(n4 / 256), dim3(256), 0, 0, out, n4, nticks); }, 30);
    float d = timeit([&] { hipLaunchKernelGGL((w_row16<true>), dim3(n4 / 256), dim3(256), 0, 0, out, n4, nticks); }, 30);
    printf("1 Mi channels x 1024 ticks (1 GiB): row4 %.0f  row4-nt %.0f  row16 %.0f  row16-nt %.0f GB/s\n",
           bytes / a / 1e6, bytes / b / 1e6, bytes / c / 1e6, bytes / d / 1e6);
    for (int gx : {1024, 2048, 4096, 8192}) {
        float e = timeit([&] { hipLaunchKernelGGL((w_flat<false>), dim3(gx), dim3(256), 0, 0, (u32x4 *)out, bytes / 16); }, 30);
        float f = timeit([&] { hipLaunchKernelGGL((w_flat<true>), dim3(gx), dim3(256), 0, 0, (u32x4 *)out, bytes / 16); }, 30);
        printf("flat fill 1 GiB, grid %5d: plain %.0f  nt %.0f GB/s\n", gx, bytes / e / 1e6, bytes / f / 1e6);
    }
    for (int gx : {256, 512, 1024, 2048, 4096}) {
        float e = timeit([&] { hipLaunchKernelGGL((w_chunk<4, 256>), dim3(gx), dim3(256), 0, 0, (u32x4 *)out, bytes / 16); }, 30);
        float f = timeit([&] { hipLaunchKernelGGL((w_wavechunk<4, 256>), dim3(gx), dim3(256), 0, 0, (u32x4 *)out, bytes / 16); }, 30);
        float g = timeit([&] { hipLaunchKernelGGL((w_wavechunk<8, 256>), dim3(gx), dim3(256), 0, 0, (u32x4 *)out, bytes / 16); }, 30);
        float h = timeit([&] { hipLaunchKernelGGL((w_wavechunk<4, 1024>), dim3(gx), dim3(1024), 0, 0, (u32x4 *)out, bytes / 16); }, 30);
        float k1 = timeit([&] { hipLaunchKernelGGL((w_wavechunk<1, 1024>), dim3(gx), dim3(1024), 0, 0, (u32x4 *)out, bytes / 16); }, 30);
        printf("grid %5d: lane-chunk64B %.0f  wave4x256thr %.0f  wave8x256thr %.0f  wave4x1024thr %.0f  flat1024thr %.0f GB/s\n", gx,
               bytes / e / 1e6, bytes / f / 1e6, bytes / g / 1e6, bytes / h / 1e6, bytes / k1 / 1e6);
    }
    float m = timeit([&] { (void)hipMemsetAsync(out, 0, bytes, 0); }, 30);
    printf("hipMemsetAsync 1 GiB: %.0f GB/s\n", bytes / m / 1e6);
    return 0;
}
