// Scratch microbenchmark: tick-mode saw with lazily materialised state: read inc + state0,
// phase = state0 + T*inc, no store.  8 B per voice-tick.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, int BS>
__global__ __launch_bounds__(BS) void tick_ro(const u32x4 *__restrict__ inc, const u32x4 *__restrict__ st0, int32_t *bus, uint32_t ngroups, uint32_t T)
{
    int32_t acc = 0;
    const uint32_t stride = gridDim.x * BS;
    for (uint32_t g = blockIdx.x * BS + threadIdx.x; g < ngroups; g += stride) {
        u32x4 a, b;
        if (NT) { a = __builtin_nontemporal_load(&inc[g]); b = __builtin_nontemporal_load(&st0[g]); }
        else    { a = inc[g]; b = st0[g]; }
        const u32x4 p = b + T * a;
        acc += (a.x ? (int32_t)p.x >> 4 : 0) + (a.y ? (int32_t)p.y >> 4 : 0) + (a.z ? (int32_t)p.z >> 4 : 0) + (a.w ? (int32_t)p.w >> 4 : 0);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ int32_t w[16];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { int32_t t = 0; for (int i = 0; i < BS / 64; i++) t += w[i]; atomicAdd(bus, t); }
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

int main() {
    const uint32_t n = 1u << 26, ng = n / 4;
    uint32_t *inc, *s0; int32_t *bus;
    (void)hipMalloc(&inc, (size_t)n * 4); (void)hipMalloc(&s0, (size_t)n * 4); (void)hipMalloc(&bus, 256);
    std::vector<uint32_t> h(n);
    for (uint32_t i = 0; i < n; i++) h[i] = i * 2654435761u | 1;
    (void)hipMemcpy(inc, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(s0, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    const double bytes = 8.0 * n;
    for (int gx : {256, 384, 512, 768, 1024, 1536, 2048, 4096}) {
        float a = timeit([&] { hipLaunchKernelGGL((tick_ro<true, 256>), dim3(gx), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)s0, bus, ng, 777u); }, 20);
        float b = timeit([&] { hipLaunchKernelGGL((tick_ro<false, 256>), dim3(gx), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)s0, bus, ng, 777u); }, 20);
        float c = timeit([&] { hipLaunchKernelGGL((tick_ro<true, 512>), dim3(gx), dim3(512), 0, 0, (const u32x4 *)inc, (const u32x4 *)s0, bus, ng, 777u); }, 20);
        float d = timeit([&] { hipLaunchKernelGGL((tick_ro<true, 1024>), dim3(gx), dim3(1024), 0, 0, (const u32x4 *)inc, (const u32x4 *)s0, bus, ng, 777u); }, 20);
        printf("grid %5d: nt256 %.1f plain256 %.1f nt512 %.1f nt1024 %.1f GB/s  (nt256: %.4f ms, %.0f Gvoice-ticks/s)\n", gx, bytes / a / 1e6, bytes / b / 1e6, bytes / c / 1e6, bytes / d / 1e6, a, n / a / 1e6);
    }
    return 0;
}
