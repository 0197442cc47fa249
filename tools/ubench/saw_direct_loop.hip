// Scratch microbenchmark: the direct form's inner loop (16 frames x 4 voices per lane per row) with everything in
// registers -- what does one frame of 4 voices cost IN SITU, and do other instruction choices change it?
//   V0: as the compiler writes it (v_ashrrev x4, v_add3 x2, v_add x4 per frame)
//   V1: the accumulate forced into plain v_add_u32 (no v_add3)
//   V2: 8 voices per lane (two rows interleaved)
//   V3: V0 with the accumulators split in two halves (even / odd voices) to shorten the dependence chains
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int V>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t rows, uint32_t seed)
{
    constexpr int TC = 16;
    int32_t acc[TC], acc2[TC];
#pragma unroll
    for (int t = 0; t < TC; t++) { acc[t] = 0; acc2[t] = 0; }
    uint32_t vi[8], vs[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { vi[q] = seed * (threadIdx.x + 1 + q); vs[q] = seed ^ (threadIdx.x * 7919u + q); }
    for (uint32_t r = 0; r < rows; r++) {
#pragma unroll
        for (int q = 0; q < 8; q++) vs[q] += r * vi[q];            // a new row's phases (stands for the loads)
#pragma unroll
        for (int t = 0; t < TC; t++) {
            if constexpr (V == 0) {
                acc[t] += ((int32_t)vs[0] >> 4) + ((int32_t)vs[1] >> 4);
                acc[t] += ((int32_t)vs[2] >> 4) + ((int32_t)vs[3] >> 4);
#pragma unroll
                for (int q = 0; q < 4; q++) vs[q] += vi[q];
            } else if constexpr (V == 1) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int32_t p = (int32_t)vs[q] >> 4;
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc[t]) : "v"(p));
                    vs[q] += vi[q];
                }
            } else if constexpr (V == 2) {
                acc[t] += ((int32_t)vs[0] >> 4) + ((int32_t)vs[1] >> 4);
                acc[t] += ((int32_t)vs[2] >> 4) + ((int32_t)vs[3] >> 4);
                acc2[t] += ((int32_t)vs[4] >> 4) + ((int32_t)vs[5] >> 4);
                acc2[t] += ((int32_t)vs[6] >> 4) + ((int32_t)vs[7] >> 4);
#pragma unroll
                for (int q = 0; q < 8; q++) vs[q] += vi[q];
            } else {
                acc[t] += ((int32_t)vs[0] >> 4) + ((int32_t)vs[2] >> 4);
                acc2[t] += ((int32_t)vs[1] >> 4) + ((int32_t)vs[3] >> 4);
#pragma unroll
                for (int q = 0; q < 4; q++) vs[q] += vi[q];
            }
        }
    }
    int32_t s = 0;
#pragma unroll
    for (int t = 0; t < TC; t++) s += acc[t] + acc2[t];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)s;
}

template <int V>
void run(const char *name, uint32_t *out, double voices_per_lane)
{
    const uint32_t rows = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(2048), dim3(256), 0, 0, out, rows, 12345u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<V>, dim3(2048), dim3(256), 0, 0, out, rows, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // 2048 x 4 waves over 1024 SIMDs = 8 waves per SIMD; each does rows x 16 frames
    const double per = ms * 1e-3 / (8.0 * rows * 16.0) * 2.4e9;            // cycles per frame of `voices_per_lane` voices
    printf("%-62s %7.3f ms  %5.1f cycles per frame and wave = %5.2f per 4 voices\n", name, ms, per, per * 4.0 / voices_per_lane);
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 2048 * 256 * 4);
    run<0>("V0 production loop (ashr x4, add3 x2, add x4)", out, 4);
    run<1>("V1 accumulate with v_add_u32 only", out, 4);
    run<2>("V2 8 voices per lane", out, 8);
    run<3>("V3 two accumulator sets (even / odd voices)", out, 4);
    return 0;
}
