// Scratch microbenchmark: carry-out PDM with the wave's carry mask written by SCALAR stores
// (s_store_dwordx2) instead of v_writelane + LDS transpose.  Checks results against a CPU loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ __launch_bounds__(1024)
void pdm_sstore(const uint32_t *__restrict__ setpoint, uint32_t *__restrict__ accu,
                unsigned long long *__restrict__ bits64, uint32_t words64_per_tick, uint32_t nticks)
{
    const uint32_t ch = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t sp = setpoint[ch];
    uint32_t a = accu[ch];
    // wave-uniform byte offset of this wave's word in row 0
    const uint32_t wave_global = __builtin_amdgcn_readfirstlane(ch >> 6);
    uint32_t off = wave_global * 8u;
    const uint32_t pitch = words64_per_tick * 8u;
    for (uint32_t t = 0; t < nticks; t += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            unsigned long long m;
            asm volatile("v_add_co_u32_e64 %0, %1, %0, %2" : "+v"(a), "=s"(m) : "v"(sp));
            asm volatile("s_store_dwordx2 %0, %1, %2" :: "s"(m), "s"(bits64), "s"(off) : "memory");
            off += pitch;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
    accu[ch] = a;
}

int main() {
    const uint32_t n = 1u << 20, nt = 4096;
    uint32_t *sp, *ac; unsigned long long *bits;
    (void)hipMalloc(&sp, n * 4); (void)hipMalloc(&ac, n * 4); (void)hipMalloc(&bits, (size_t)nt * n / 8);
    std::vector<uint32_t> h(n);
    for (uint32_t i = 0; i < n; i++) h[i] = 0x40000000u + (i * 2654435761u) % 0x80000001u;
    (void)hipMemcpy(sp, h.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemset(ac, 0, n * 4);
    (void)hipMemset(bits, 0, (size_t)nt * n / 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(pdm_sstore, dim3(n / 1024), dim3(1024), 0, 0, sp, ac, bits, n / 64, nt);
    (void)hipDeviceSynchronize();
    // verify first 3 waves' words for all ticks against CPU
    std::vector<unsigned long long> got((size_t)nt * (n / 64));
    (void)hipMemcpy(got.data(), bits, got.size() * 8, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (uint32_t w : {0u, 1u, 17u, n / 64 - 1}) {
        std::vector<uint32_t> acc(64, 0);
        for (uint32_t t = 0; t < nt; t++) {
            unsigned long long m = 0;
            for (int l = 0; l < 64; l++) { uint32_t s = h[w * 64 + l]; uint32_t a1 = acc[l] + s; if (a1 < acc[l]) m |= 1ull << l; acc[l] = a1; }
            if (got[(size_t)t * (n / 64) + w] != m) bad++;
        }
    }
    printf("mismatches: %zu\n", bad);
    (void)hipMemset(ac, 0, n * 4);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(pdm_sstore, dim3(n / 1024), dim3(1024), 0, 0, sp, ac, bits, n / 64, nt);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("sstore: %.4f ms  %.1f G ch-ticks/s  out %.1f GB/s\n", ms, (double)n * nt / ms / 1e6, (double)n * nt / 8 / ms / 1e6);
    return 0;
}
