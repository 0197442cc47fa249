// Scratch microbenchmark: issue rate of the integer vector instructions the bank kernels are built from, on gfx950,
// at full occupancy (8 waves per SIMD), independent operands (4 chains).  Prints cycles per wave64 instruction at an
// assumed 2.4 GHz; a full-rate op on a 16-lane SIMD would be 4.0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(X) X X X X X X X X
#define REP16(X) REP8(X) REP8(X)

#define OP4(fmt) asm volatile(fmt(0, 4) "\n\t" fmt(1, 5) "\n\t" fmt(2, 6) "\n\t" fmt(3, 7) \
    : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc", "s64", "s65", "s66", "s67");

#define F_ADD(d, s)    "v_add_u32 %" #d ", %" #d ", %" #s
#define F_ASHR(d, s)   "v_ashrrev_i32 %" #d ", 4, %" #s
#define F_ADD3(d, s)   "v_add3_u32 %" #d ", %" #d ", %" #s ", %" #s
#define F_LSHLADD(d, s) "v_lshl_add_u32 %" #d ", %" #s ", 3, %" #d
#define F_MULLO(d, s)  "v_mul_lo_u32 %" #d ", %" #d ", %" #s
#define F_MUL24(d, s)  "v_mul_u32_u24 %" #d ", %" #d ", %" #s
#define F_MAD24(d, s)  "v_mad_u32_u24 %" #d ", %" #d ", %" #s ", %" #s
#define F_ADDCO(d, s)  "v_add_co_u32_e32 %" #d ", vcc, %" #s ", %" #d
#define F_ADDC(d, s)   "v_addc_co_u32_e32 %" #d ", vcc, %" #s ", %" #d ", vcc"
#define F_BFE(d, s)    "v_bfe_i32 %" #d ", %" #s ", 4, 28"
#define F_SAD8(d, s)   "v_sad_u8 %" #d ", %" #s ", 0, %" #d
#define F_AND(d, s)    "v_and_b32 %" #d ", %" #d ", %" #s
#define F_XOR3(d, s)   "v_xad_u32 %" #d ", %" #d ", %" #s ", %" #s
#define F_PERM(d, s)   "v_perm_b32 %" #d ", %" #d ", %" #s ", %" #s
#define F_CNDMASK(d, s) "v_cndmask_b32 %" #d ", %" #d ", %" #s ", vcc"
#define F_PKADD16(d, s) "v_pk_add_u16 %" #d ", %" #d ", %" #s
#define F_DOT4(d, s)   "v_dot4_u32_u8 %" #d ", %" #s ", %" #s ", %" #d
#define F_ALIGNBIT(d, s) "v_alignbit_b32 %" #d ", %" #d ", %" #s ", 4"
#define F_MOV(d, s)    "v_mov_b32 %" #d ", %" #s
#define F_FMA(d, s)    "v_fma_f32 %" #d ", %" #d ", %" #s ", %" #s
#define F_CNDMASK64(d, s) "v_cndmask_b32_e64 %" #d ", %" #d ", %" #s ", s[64:65]"
#define F_CMP(d, s)    "v_cmp_ne_u32_e32 vcc, %" #d ", %" #s
#define F_CMP64(d, s)  "v_cmp_ne_u32_e64 s[66:67], %" #d ", %" #s
#define F_CMPCND(d, s) "v_cmp_ne_u32_e32 vcc, 0, %" #s "\n\tv_cndmask_b32_e32 %" #d ", 0, %" #d ", vcc"
#define F_MASKARITH(d, s) "v_sub_u32 %" #d ", 0, %" #s "\n\tv_or_b32 %" #d ", %" #d ", %" #s "\n\tv_ashrrev_i32 %" #d ", 31, %" #d "\n\tv_and_b32 %" #d ", %" #d ", %" #s
#define F_MED3(d, s)   "v_med3_u32 %" #d ", %" #d ", %" #s ", %" #s
#define F_MIN(d, s)    "v_min_u32 %" #d ", %" #d ", %" #s
#define F_LSHLREV(d, s) "v_lshlrev_b32 %" #d ", 3, %" #s
#define F_SUBREV(d, s) "v_subrev_u32 %" #d ", %" #s ", %" #d
#define F_XOR(d, s)    "v_xor_b32 %" #d ", %" #d ", %" #s
#define F_ADDNC64(d, s) "v_lshl_add_u64 v[20:21], v[20:21], 0, v[22:23]"

template <int V>
__global__ __launch_bounds__(1024) void k(uint32_t *out, uint32_t iters, uint32_t seed)
{
    uint32_t r0 = threadIdx.x * seed, r1 = r0 ^ 0x1234u, r2 = r0 + 77u, r3 = r0 * 3u;
    uint32_t x0 = seed + threadIdx.x, x1 = x0 * 5u, x2 = x0 ^ 0xABCDu, x3 = x0 + 9u;
    unsigned long long q0 = r0, q1 = r1, q2 = r2, q3 = r3, y0 = x0, y1 = x1, y2 = x2, y3 = x3;
    asm volatile("s_mov_b64 s[64:65], 0x5555" ::: "s64", "s65");
    for (uint32_t i = 0; i < iters; i++) {
        if constexpr (V == 0)  { REP16(OP4(F_ADD)) }
        if constexpr (V == 1)  { REP16(OP4(F_ASHR)) }
        if constexpr (V == 2)  { REP16(OP4(F_ADD3)) }
        if constexpr (V == 3)  { REP16(OP4(F_LSHLADD)) }
        if constexpr (V == 4)  { REP16(OP4(F_MULLO)) }
        if constexpr (V == 5)  { REP16(OP4(F_MUL24)) }
        if constexpr (V == 6)  { REP16(OP4(F_MAD24)) }
        if constexpr (V == 7)  { REP16(OP4(F_ADDCO)) }
        if constexpr (V == 8)  { REP16(OP4(F_ADDC)) }
        if constexpr (V == 9)  { REP16(OP4(F_BFE)) }
        if constexpr (V == 10) { REP16(OP4(F_SAD8)) }
        if constexpr (V == 11) { REP16(OP4(F_AND)) }
        if constexpr (V == 12) { REP16(OP4(F_XOR3)) }
        if constexpr (V == 13) { REP16(OP4(F_PERM)) }
        if constexpr (V == 14) { REP16(OP4(F_CNDMASK)) }
        if constexpr (V == 15) { REP16(OP4(F_PKADD16)) }
        if constexpr (V == 16) { REP16(OP4(F_DOT4)) }
        if constexpr (V == 17) { REP16(OP4(F_ALIGNBIT)) }
        if constexpr (V == 18) { REP16(OP4(F_MOV)) }
        if constexpr (V == 19) { REP16(OP4(F_FMA)) }
        if constexpr (V == 20) { REP16(OP4(F_CNDMASK64)) }
        if constexpr (V == 21) { REP16(OP4(F_CMP)) }
        if constexpr (V == 22) { REP16(OP4(F_CMP64)) }
        if constexpr (V == 23) { REP16(OP4(F_CMPCND)) }
        if constexpr (V == 24) { REP16(OP4(F_MASKARITH)) }
        if constexpr (V == 25) { REP16(OP4(F_MED3)) }
        if constexpr (V == 26) { REP16(OP4(F_MIN)) }
        if constexpr (V == 27) { REP16(OP4(F_LSHLREV)) }
        if constexpr (V == 28) { REP16(OP4(F_SUBREV)) }
        if constexpr (V == 29) { REP16(OP4(F_XOR)) }
        if constexpr (V == 30) {       // 64-bit add: {counter, phase} += {0, inc} -- carry into the counter in one instruction
            REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n\tv_lshl_add_u64 %1, %1, 0, %5\n\t"
                               "v_lshl_add_u64 %2, %2, 0, %6\n\tv_lshl_add_u64 %3, %3, 0, %7"
                               : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(y0), "v"(y1), "v"(y2), "v"(y3));)
        }
        if constexpr (V == 31) {       // the same through the multiplier: D64 = a32 * 1 + c64
            REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, 1, %0\n\tv_mad_u64_u32 %1, vcc, %5, 1, %1\n\t"
                               "v_mad_u64_u32 %2, vcc, %6, 1, %2\n\tv_mad_u64_u32 %3, vcc, %7, 1, %3"
                               : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3) : "vcc");)
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = r0 + r1 + r2 + r3 + (uint32_t)((q0 + q1 + q2 + q3) >> 16);
}

template <int V>
void run(const char *name, uint32_t *out)
{
    const uint32_t iters = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(512), dim3(1024), 0, 0, out, iters, 12345u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(k<V>, dim3(512), dim3(1024), 0, 0, out, iters, 12345u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double per = ms * 1e-3 / (8.0 * iters * 64.0);          // 8 waves per SIMD, 64 instructions per iteration
    printf("%-28s %8.3f ms  %5.2f cycles @2.4 GHz per wave64 instruction  (%5.1f T lane-ops/s chip-wide)\n", name, ms,
           per * 2.4e9, 64.0 / per * 1024 / 1e12);
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 512 * 1024 * 4);
    run<0>("v_add_u32", out); run<1>("v_ashrrev_i32", out); run<2>("v_add3_u32", out); run<3>("v_lshl_add_u32", out);
    run<4>("v_mul_lo_u32", out); run<5>("v_mul_u32_u24", out); run<6>("v_mad_u32_u24", out);
    run<7>("v_add_co_u32 (vcc)", out); run<8>("v_addc_co_u32 (vcc)", out); run<9>("v_bfe_i32", out);
    run<10>("v_sad_u8", out); run<11>("v_and_b32", out); run<12>("v_xad_u32", out); run<13>("v_perm_b32", out);
    run<14>("v_cndmask_b32", out); run<15>("v_pk_add_u16", out); run<16>("v_dot4_u32_u8", out);
    run<17>("v_alignbit_b32", out); run<18>("v_mov_b32", out); run<19>("v_fma_f32", out);
    run<20>("v_cndmask_b32_e64 (sgpr mask)", out); run<21>("v_cmp_ne_u32 -> vcc", out); run<22>("v_cmp_ne_u32_e64 -> sgpr", out);
    run<23>("v_cmp + v_cndmask (2 instr)", out); run<24>("sub,or,ashr,and (4 instr)", out); run<25>("v_med3_u32", out);
    run<26>("v_min_u32", out); run<27>("v_lshlrev_b32", out); run<28>("v_subrev_u32", out); run<29>("v_xor_b32", out);
    run<30>("v_lshl_add_u64", out); run<31>("v_mad_u64_u32", out);
    return 0;
}
