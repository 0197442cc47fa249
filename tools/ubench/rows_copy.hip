// Scratch microbenchmark: the memory side of the cproc kernel alone -- tick-major rows in[t][n] -> out[t][n], one lane per
// VW consecutive instances, 8 rows requested ahead of the 8 rows being written (software-pipelined like cproc_kernel).
// 1 Mi instances x 256 ticks = 1 GiB read + 1 GiB written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int VW>
__global__ __launch_bounds__(1024) void rows(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n, uint32_t nticks)
{
    typedef uint32_t vec __attribute__((ext_vector_type(VW)));
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;     // group of VW instances
    const uint32_t ng = n / VW;
    if (g >= ng) return;
    const vec *src = reinterpret_cast<const vec *>(in) + g;
    vec *dst = reinterpret_cast<vec *>(out) + g;
    vec v[8], w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = src[(size_t)i * ng];
    vec acc = 0;
    for (uint32_t t0 = 0; t0 < nticks; t0 += 8) {
        const bool more = t0 + 8 < nticks;
        if (more) {
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = src[(size_t)(t0 + 8 + i) * ng];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) { acc += v[i]; dst[(size_t)(t0 + i) * ng] = acc; }   // a running sum: one dependent add per tick
        if (more) {
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = w[i];
        }
    }
}

template <int VW> void run(const char *name, const uint32_t *in, uint32_t *out, uint32_t n, uint32_t nt, uint32_t bs = 256)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t ng = n / VW;
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(rows<VW>, dim3((ng + bs - 1) / bs), dim3(bs), 0, 0, in, out, n, nt);
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(rows<VW>, dim3((ng + bs - 1) / bs), dim3(bs), 0, 0, in, out, n, nt);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-40s %8.1f us  %6.2f TB/s (read + write)\n", name, ms * 1e3, 2.0 * n * nt * 4 / (ms * 1e-3) / 1e12);
}

int main()
{
    const uint32_t n = 1u << 20, nt = 256;
    uint32_t *in, *out;
    (void)hipMalloc(&in, (size_t)n * nt * 4); (void)hipMalloc(&out, (size_t)n * nt * 4);
    (void)hipMemset(in, 1, (size_t)n * nt * 4);
    run<1>("4 bytes per lane per row", in, out, n, nt);
    run<2>("8 bytes per lane per row", in, out, n, nt);
    run<4>("16 bytes per lane per row", in, out, n, nt);
    run<1>("4 bytes per lane per row (again)", in, out, n, nt);
    run<1>("4 bytes per lane, 512-thread workgroups", in, out, n, nt, 512);
    run<1>("4 bytes per lane, 1024-thread workgroups", in, out, n, nt, 1024);
    run<1>("4 bytes per lane, 64-thread workgroups", in, out, n, nt, 64);
    return 0;
}
