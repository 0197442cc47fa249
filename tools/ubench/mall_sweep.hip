// Scratch microbenchmark: does the 256 MB Infinity Cache keep part of a bank between two launches?
// A read-only sum over two uint32 arrays (the tick kernel's access pattern: 16-byte loads, grid-stride rows of
// 1024 lanes), (a) always front to back, (b) alternating front-to-back / back-to-front, each with temporal and with
// non-temporal loads, for banks of 32 MB .. 1 GB.  GB/s per launch over back-to-back launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(1024) void sweep(const u32x4 *__restrict__ a, const u32x4 *__restrict__ b,
                                              uint32_t nrows, uint32_t rev, uint32_t *out)
{
    uint32_t acc = 0;
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const uint32_t r = (rev ? nrows - 1 - row : row) * 1024u + threadIdx.x;
        u32x4 x, y;
        if constexpr (NT) { x = __builtin_nontemporal_load(a + r); y = __builtin_nontemporal_load(b + r); }
        else { x = a[r]; y = b[r]; }
        acc += x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int lg = 22; lg <= 27; lg++) {                       // voices; bytes = 8 << lg
        const size_t n = (size_t)1 << lg;
        uint32_t *a, *b;
        (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4);
        (void)hipMemset(a, 1, n * 4); (void)hipMemset(b, 2, n * 4);
        const uint32_t nrows = (uint32_t)(n / 4 / 1024);
        for (int nt = 0; nt < 2; nt++)
            for (int alt = 0; alt < 2; alt++)
                for (uint32_t grid : {256u, 512u, 1024u}) {
                    auto launch = [&](uint32_t rev) {
                        if (nt) hipLaunchKernelGGL(sweep<true>, dim3(grid), dim3(1024), 0, 0, (const u32x4 *)a, (const u32x4 *)b, nrows, rev, out);
                        else    hipLaunchKernelGGL(sweep<false>, dim3(grid), dim3(1024), 0, 0, (const u32x4 *)a, (const u32x4 *)b, nrows, rev, out);
                    };
                    for (int i = 0; i < 10; i++) launch(alt ? (i & 1) : 0);
                    (void)hipDeviceSynchronize();
                    (void)hipEventRecord(e0);
                    const int reps = 40;
                    for (int i = 0; i < reps; i++) launch(alt ? (i & 1) : 0);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
                    printf("%4zu MB  %-12s %-12s grid %4u  %8.1f us  %7.2f TB/s\n", (n * 8) >> 20, nt ? "non-temporal" : "temporal",
                           alt ? "alternating" : "same-dir", grid, ms * 1e3, n * 8.0 / (ms * 1e-3) / 1e12);
                }
        (void)hipFree(a); (void)hipFree(b);
    }
    return 0;
}
