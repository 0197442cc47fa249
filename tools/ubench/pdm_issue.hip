// Scratch microbenchmark: issue cost (cycles per 64-channel tick per wave slot) of the instruction sequences a
// tick-major carry-out PDM tick can be built from on gfx950.  Timing only (no result check): every variant runs
// the same loop shape at full occupancy (2 x 1024 threads per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int V>
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t iters, uint32_t sp_in)
{
    uint32_t a = threadIdx.x * 2654435761u, b = a ^ 0x5555u, x = sp_in + threadIdx.x, y = x * 3u;
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    __shared__ unsigned long long S[1024];
    unsigned long long m64 = 0;
    const uint32_t ldsaddr = (threadIdx.x & 63) == 0 ? (threadIdx.x >> 6) * 512u : 0xFFFFF000u;
    if (sp_in == 0) S[threadIdx.x] = 1;
    asm volatile("s_mov_b64 s[64:65], exec\n\ts_mov_b64 vcc, exec" ::: "s64", "s65", "vcc");
    for (uint32_t i = 0; i < iters; i++) {
        if constexpr (V == 0) {        // v1: add_co(vcc) + s_nop 1 + 2 writelane(vcc)
            REP64(asm volatile("v_add_co_u32_e32 %0, vcc, %3, %0\n\ts_nop 1\n\tv_writelane_b32 %1, vcc_lo, 5\n\tv_writelane_b32 %2, vcc_hi, 5"
                               : "+v"(a), "+v"(w0), "+v"(w1) : "v"(x) : "vcc");)
        } else if constexpr (V == 1) { // add_co e32 only
            REP64(asm volatile("v_add_co_u32_e32 %0, vcc, %1, %0" : "+v"(a) : "v"(x) : "vcc");)
        } else if constexpr (V == 2) { // add_co e64 -> s[64:65] only
            REP64(asm volatile("v_add_co_u32_e64 %0, s[64:65], %0, %1" : "+v"(a) : "v"(x) : "s64", "s65");)
        } else if constexpr (V == 3) { // 2 writelane from vcc
            REP64(asm volatile("v_writelane_b32 %0, vcc_lo, 5\n\tv_writelane_b32 %1, vcc_hi, 5" : "+v"(w0), "+v"(w1));)
        } else if constexpr (V == 4) { // 2 writelane from s64/s65
            REP64(asm volatile("v_writelane_b32 %0, s64, 5\n\tv_writelane_b32 %1, s65, 5" : "+v"(w0), "+v"(w1));)
        } else if constexpr (V == 5) { // two channel sets, two ticks: 4 add_co e64 + 8 writelane (= 4 set-ticks)
            REP8(REP8(asm volatile(
                "v_add_co_u32_e64 %0, s[64:65], %0, %6\n\tv_add_co_u32_e64 %1, s[66:67], %1, %7\n\t"
                "v_add_co_u32_e64 %0, s[68:69], %0, %6\n\tv_add_co_u32_e64 %1, s[70:71], %1, %7\n\t"
                "v_writelane_b32 %2, s64, 5\n\tv_writelane_b32 %3, s65, 5\n\tv_writelane_b32 %4, s66, 5\n\tv_writelane_b32 %5, s67, 5\n\t"
                "v_writelane_b32 %2, s68, 6\n\tv_writelane_b32 %3, s69, 6\n\tv_writelane_b32 %4, s70, 6\n\tv_writelane_b32 %5, s71, 6"
                : "+v"(a), "+v"(b), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(x), "v"(y)
                : "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71");))       // 64 blocks x 4 set-ticks
        } else if constexpr (V == 6) { // add_co(vcc) + s_nop 1 + scalar copy of the mask (no writelane)
            REP64(asm volatile("v_add_co_u32_e32 %0, vcc, %1, %0\n\ts_nop 1\n\ts_mov_b64 s[64:65], vcc" : "+v"(a) : "v"(x) : "vcc", "s64", "s65");)
        } else if constexpr (V == 7) { // add_co + s_nop 1 only
            REP64(asm volatile("v_add_co_u32_e32 %0, vcc, %1, %0\n\ts_nop 1" : "+v"(a) : "v"(x) : "vcc");)
        } else if constexpr (V == 8) { // v1 WITHOUT the nop (wrong results; timing only)
            REP64(asm volatile("v_add_co_u32_e32 %0, vcc, %3, %0\n\tv_writelane_b32 %1, vcc_lo, 5\n\tv_writelane_b32 %2, vcc_hi, 5"
                               : "+v"(a), "+v"(w0), "+v"(w1) : "v"(x) : "vcc");)
        } else if constexpr (V == 9) { // channel-stream step: add_co e64 + addc e64 (own carry into own word)
            REP64(asm volatile("v_add_co_u32_e64 %0, s[64:65], %0, %2\n\tv_add_u32 %3, %3, %2\n\tv_add_u32 %3, %3, %2\n\tv_addc_co_u32_e64 %1, vcc, %1, %1, s[64:65]"
                               : "+v"(a), "+v"(w0), "+v"(x), "+v"(b) :: "vcc", "s64", "s65");)
        } else if constexpr (V == 10) { // v1 with a useful VALU op + s_nop 0 in the hazard slot
            REP64(asm volatile("v_add_co_u32_e32 %0, vcc, %3, %0\n\tv_add_u32 %4, %4, %3\n\ts_nop 0\n\tv_writelane_b32 %1, vcc_lo, 5\n\tv_writelane_b32 %2, vcc_hi, 5"
                               : "+v"(a), "+v"(w0), "+v"(w1), "+v"(x), "+v"(b) :: "vcc");)
        } else if constexpr (V == 11) { // two sets through VCC alternately, hazard slots filled by the other set's writelanes
            REP64(asm volatile(
                "v_add_co_u32_e32 %0, vcc, %6, %0\n\ts_nop 1\n\ts_mov_b64 s[64:65], vcc\n\t"
                "v_add_co_u32_e32 %1, vcc, %7, %1\n\tv_writelane_b32 %2, s64, 5\n\tv_writelane_b32 %3, s65, 5\n\t"
                "v_writelane_b32 %4, vcc_lo, 5\n\tv_writelane_b32 %5, vcc_hi, 5"
                : "+v"(a), "+v"(b), "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(x), "v"(y) : "vcc", "s64", "s65");)   // 2 set-ticks per block
        } else if constexpr (V == 13) { // mask broadcast by v_mov_b64, stored to LDS by lane 0 (other lanes: out-of-range address, dropped)
            REP64(asm volatile("v_add_co_u32_e32 %0, vcc, %1, %0\n\ts_nop 1\n\tv_mov_b64 %2, vcc\n\tds_write_b64 %3, %2 offset:40"
                               : "+v"(a), "+v"(x), "=&v"(m64) : "v"(ldsaddr) : "vcc", "memory");)
        } else if constexpr (V == 15) { // v_mov_b64 only
            REP64(asm volatile("v_mov_b64 %0, vcc" : "=v"(m64));)
        } else if constexpr (V == 16) { // software-pipelined: add_co e64 (tick t) + one-lane-EXEC v_mov_b64 of tick t-1's mask
            asm volatile("s_mov_b64 s[68:69], 1" ::: "s68", "s69");
            REP8(REP8(asm volatile(
                "v_add_co_u32_e64 %0, s[64:65], %0, %2\n\t"
                "s_mov_b64 exec, s[68:69]\n\t"
                "v_mov_b64 %1, s[66:67]\n\t"
                "s_lshl_b64 s[68:69], s[68:69], 1\n\t"
                "s_mov_b64 exec, -1\n\t"
                "v_add_co_u32_e64 %0, s[66:67], %0, %2\n\t"
                "s_mov_b64 exec, s[68:69]\n\t"
                "v_mov_b64 %1, s[64:65]\n\t"
                "s_lshl_b64 s[68:69], s[68:69], 1\n\t"
                "s_mov_b64 exec, -1"
                : "+v"(a), "+v"(m64) : "v"(x) : "s64", "s65", "s66", "s67", "s68", "s69", "scc");))   // 2 ticks per block
        } else if constexpr (V == 17) { // the same with the EXEC switches left out (full-EXEC v_mov_b64; timing only)
            REP8(REP8(asm volatile(
                "v_add_co_u32_e64 %0, s[64:65], %0, %2\n\t"
                "v_mov_b64 %1, s[66:67]\n\t"
                "v_add_co_u32_e64 %0, s[66:67], %0, %2\n\t"
                "v_mov_b64 %1, s[64:65]"
                : "+v"(a), "+v"(m64) : "v"(x) : "s64", "s65", "s66", "s67");))
        } else if constexpr (V == 18) { // pipelined, via VCC (e32 add) and a scalar copy: add_co e32, s_mov mask, one-lane v_mov_b64
            asm volatile("s_mov_b64 s[68:69], 1" ::: "s68", "s69");
            REP64(asm volatile(
                "v_add_co_u32_e32 %0, vcc, %2, %0\n\t"
                "s_mov_b64 exec, s[68:69]\n\t"
                "v_mov_b64 %1, s[66:67]\n\t"
                "s_lshl_b64 s[68:69], s[68:69], 1\n\t"
                "s_mov_b64 exec, -1\n\t"
                "s_mov_b64 s[66:67], vcc"
                : "+v"(a), "+v"(m64) : "v"(x) : "vcc", "s66", "s67", "s68", "s69", "scc");)
        } else if constexpr (V == 12) { // plain VALU add for reference
            REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(x));)
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a + b + w0 + w1 + w2 + w3 + x + (uint32_t)m64 + (uint32_t)S[threadIdx.x ^ 1];
}

template <int V>
void run(const char *name, double ticks_per_rep, uint32_t *out)
{
    const uint32_t iters = 64;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(512), dim3(1024), 0, 0, out, iters, 12345u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<V>, dim3(512), dim3(1024), 0, 0, out, iters, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // 512 WGs x 16 waves over 256 CUs x 4 SIMDs = 8 waves per SIMD, all resident
    const double per = ms * 1e-3 / (8.0 * iters * 64.0 * ticks_per_rep);
    printf("%-70s %8.3f ms  %6.2f ns = %5.2f cycles @2.4 GHz per 64-channel tick (set-tick)\n", name, ms, per * 1e9, per * 2.4e9);
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 512 * 1024 * 4);
    run<12>("v_add_u32 (reference: one full-rate VALU op)", 1, out);
    run<1>("v_add_co_u32_e32 -> vcc", 1, out);
    run<2>("v_add_co_u32_e64 -> s[64:65]", 1, out);
    run<7>("v_add_co_u32_e32 + s_nop 1", 1, out);
    run<3>("2 x v_writelane from vcc", 1, out);
    run<4>("2 x v_writelane from s64/s65", 1, out);
    run<0>("v1 tick: add_co(vcc) + s_nop 1 + 2 writelane(vcc)", 1, out);
    run<8>("v1 tick without the s_nop (wrong bits, timing only)", 1, out);
    run<10>("v1 tick with v_add_u32 + s_nop 0 in the hazard slot", 1, out);
    run<6>("add_co(vcc) + s_nop 1 + s_mov_b64 (mask kept on the scalar side)", 1, out);
    run<5>("two sets x two ticks: 4 add_co e64 + 8 writelane(sgpr)", 4.0 / 64.0 * 64.0, out);
    run<11>("two sets via vcc: add, nop, s_mov, add, 4 writelane", 2, out);
    run<15>("v_mov_b64 v[..], vcc", 1, out);
    run<13>("add_co(vcc) + s_nop 1 + v_mov_b64 + ds_write_b64 (lane 0 only, offset = tick)", 1, out);
    run<9>("stream step: add_co e64 + 2 v_add + addc e64", 1, out);
    run<16>("pipelined: add_co e64 + one-lane-EXEC v_mov_b64 (2 s_mov exec + s_lshl per tick)", 2, out);
    run<17>("the same without the EXEC switches (timing only)", 2, out);
    run<18>("pipelined via vcc: add_co e32, one-lane v_mov_b64 of the previous mask, s_mov copy", 1, out);
    return 0;
}
