// LDS instruction rates on gfx950 at the poly bank's occupancy (16 waves per CU) and at full occupancy:
// cycles the CU's LDS needs per wave64 instruction for the instruction kinds the bus mix-down can be built from.
//   hipcc --offload-arch=gfx950 -O3 -o lds_rates tools/ubench/lds_rates.hip && ./lds_rates
// Each kernel: every wave issues ITER x 8 LDS instructions on its own column (the poly kernel's access shape:
// address = (row * 65 + lane) * 4, conflict-free), nothing else in the loop but the loop counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum Kind { ADD32 = 0, ADD32_RTN, WRITE32, WRITE64, ADD64, READ32, READ128, ADD32_VALU_MIX };
static const char *names[] = {"ds_add_u32 (no return)", "ds_add_rtn_u32", "ds_write_b32", "ds_write_b64", "ds_add_u64 (no return)",
                              "ds_read_b32", "ds_read_b128", "ds_add_u32 + 7 v_fma between two of them"};

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters)
{
    __shared__ uint32_t M[128 * 65 + 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i < 128 * 65; i += 256) M[i] = 0;
    __syncthreads();
    uint32_t acc = tid;
    float f = (float)tid;
    uint32_t *m = &M[lane];
    unsigned long long *m64 = reinterpret_cast<unsigned long long *>(&M[0]) + lane;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int row = (j * 16 + (it & 15));
            if (KIND == ADD32) atomicAdd(m + 65 * row, acc);
            if (KIND == ADD32_RTN) acc += atomicAdd(m + 65 * row, acc);
            if (KIND == WRITE32) { *(volatile uint32_t *)(m + 65 * row) = acc; }
            if (KIND == WRITE64) { *(volatile unsigned long long *)(m64 + 33 * (row & 63)) = acc; }
            if (KIND == ADD64) atomicAdd(m64 + 33 * (row & 63), (unsigned long long)acc);
            if (KIND == READ32) acc += *(volatile uint32_t *)(m + 65 * row);
            if (KIND == READ128) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 v = *(volatile u32x4 *)(&M[(((row & 31) * 64 + lane) * 4) & 0x1FFCu]);
                acc += v.x + v.w;
            }
            if (KIND == ADD32_VALU_MIX) {
                atomicAdd(m + 65 * row, acc);
#pragma unroll
                for (int q = 0; q < 7; q++) f = __fmaf_rn(f, 1.0001f, 0.5f);
            }
        }
    }
    if (acc == 0xFFFFFFFFu || f == 1.2345f) out[tid] = acc;       // keep the loop
}

template <int KIND>
void run(uint32_t *d, int wg_per_cu)
{
    const int iters = 4096;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    // per CU: wg_per_cu * 4 waves, each iters * 8 instructions
    const double inst_per_cu = (double)wg_per_cu * 4 * iters * 8;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-42s %2d waves/CU: %7.3f ms  -> %5.2f cycles of the CU per wave64 instruction\n", names[KIND], wg_per_cu * 4, ms, cycles / inst_per_cu);
}

int main()
{
    uint32_t *d;
    (void)hipMalloc(&d, 4096);
    for (int wg : {4, 8}) {
        run<ADD32>(d, wg); run<ADD32_RTN>(d, wg); run<WRITE32>(d, wg); run<WRITE64>(d, wg); run<ADD64>(d, wg);
        run<READ32>(d, wg); run<READ128>(d, wg); run<ADD32_VALU_MIX>(d, wg);
    }
    return 0;
}
