// Scratch microbenchmark: does the read stream of the tick step run faster over ONE interleaved array
// ({inc x4, state x4} per lane: 32 contiguous bytes, 2 KB per wave) than over the two arrays of the SoA layout?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void sweep(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b, uint32_t nrows, uint32_t *out)
{
    uint32_t acc = 0;
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        u32x4 x, y;
        if constexpr (MODE == 0) {            // two arrays
            const uint32_t r = row * 1024u + threadIdx.x;
            x = __builtin_nontemporal_load(a + r); y = __builtin_nontemporal_load(b + r);
        } else if constexpr (MODE == 1) {     // one array, lane reads 32 contiguous bytes
            const uint32_t r = (row * 1024u + threadIdx.x) * 2u;
            x = __builtin_nontemporal_load(a + r); y = __builtin_nontemporal_load(a + r + 1);
        } else {                              // one array, the wave reads two consecutive 1 KB segments
            const uint32_t r = row * 2048u + (threadIdx.x >> 6) * 128u + (threadIdx.x & 63);
            x = __builtin_nontemporal_load(a + r); y = __builtin_nontemporal_load(a + r + 64);
        }
        acc += x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE> void run(const char *name, const u32x4 *a, const u32x4 *b, uint32_t nrows, uint32_t *out, size_t bytes)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(sweep<MODE>, dim3(256), dim3(1024), 0, 0, a, b, nrows, out);
    (void)hipDeviceSynchronize();
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 50; i++) hipLaunchKernelGGL(sweep<MODE>, dim3(256), dim3(1024), 0, 0, a, b, nrows, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 50; if (ms < best) best = ms;
    }
    printf("%-56s %7.2f us  %5.2f TB/s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t n = (size_t)1 << 26;           // voices
    uint32_t *a, *out; (void)hipMalloc(&a, n * 8); (void)hipMalloc(&out, 4);
    (void)hipMemset(a, 1, n * 8);
    const u32x4 *A = (const u32x4 *)a, *B = (const u32x4 *)(a + n);
    const uint32_t nrows = (uint32_t)(n / 4 / 1024);
    for (int k = 0; k < 2; k++) {
        run<0>("two arrays (SoA: inc[], state0[])", A, B, nrows, out, n * 8);
        run<1>("one array, 32 contiguous bytes per lane", A, B, nrows, out, n * 8);
        run<2>("one array, two consecutive 1 KB segments per wave", A, B, nrows, out, n * 8);
    }
    return 0;
}
