// pdm_bank.hip -- N-channel first-order carry-out pulse-density modulator bank
// for gfx950 (MI355X).
//
// Replaces the PDM ISR body of stm32f103/mod_pdm.c:214-264, which on the ARM
// is two instructions per channel ("adds" accu += setpoint+dither; "rrx"
// shifts the carry flag into a shift register).  Per channel and tick:
//     sp   = setpoint + dither          (mod 2^32; dither shared by the bank)
//     accu = accu + sp                  (mod 2^32)
//     out  = carry out of that add      (the pulse)
//
// Mapping: one lane per channel, setpoint/accu in registers for the whole run
// of ticks (HBM: 8 B read + 4 B written per channel per launch).  The carry of
// a wave's 64 channels IS the compare mask of the add (an SGPR pair): the
// 64-wide analogue of the reference's rrx shift register.  Lane t of the wave
// keeps the mask of tick t (v_writelane), so after 64 ticks a wave holds a
// 64 x 64 bit tile; tiles of the workgroup's 16 waves are transposed through
// LDS and leave as full 128-byte rows of the tick-major pulse matrix
// bits[tick][channel/32] (channel c -> bit c&31 of word c>>5).
#include "smx_common.h"

namespace {

// One PDM tick for the 64 channels of a wave, tick index T in 0..63:
//   v_add_co_u32   accu += x, carry-outs of all 64 lanes -> VCC   (the "adds")
//   v_writelane x2 lane T of (wlo, whi) := VCC                    (the "rrx")
// gfx950 needs 2 wait states between a VALU write of an SGPR/VCC and a VALU
// read of it; hipcc does not see inside asm, so the s_nop is explicit.  The
// lane select must be an inline constant: VCC already uses the instruction's
// one constant-bus slot.
template <int T>
__device__ __forceinline__ void pdm_tick(uint32_t &a, uint32_t x, uint32_t &wlo, uint32_t &whi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_add_co_u32_e32 %0, vcc, %3, %0\n\t"
        "s_nop 1\n\t"
        "v_writelane_b32 %1, vcc_lo, %4\n\t"
        "v_writelane_b32 %2, vcc_hi, %4"
        : "+v"(a), "+v"(wlo), "+v"(whi)
        : "v"(x), "n"(T)
        : "vcc");
#else
    (void)a; (void)x; (void)wlo; (void)whi;
#endif
}

// 64 ticks, unrolled at compile time so that every lane select is a constant.
template <int T, bool DITHER>
struct PdmTicks {
    static __device__ __forceinline__ void run(uint32_t &a, uint32_t sp, const uint32_t *d,
                                               uint32_t &wlo, uint32_t &whi)
    {
        const uint32_t x = DITHER ? sp + d[T] : sp;
        pdm_tick<T>(a, x, wlo, whi);
        PdmTicks<T + 1, DITHER>::run(a, sp, d, wlo, whi);
    }
};
template <bool DITHER>
struct PdmTicks<64, DITHER> {
    static __device__ __forceinline__ void run(uint32_t &, uint32_t, const uint32_t *, uint32_t &,
                                               uint32_t &) {}
};

// Ticks START .. START+CNT-1 of a tile (compile-time lane selects), and a ragged tile of nt < 64
// ticks as the binary decomposition of nt: 32 | 16 | 8 | 4 | 2 | 1 ticks, each block starting
// where the larger ones ended (63 instantiations, 192 tick bodies in total).
template <int START, int CNT, bool DITHER>
struct PdmRange {
    static __device__ __forceinline__ void run(uint32_t &a, uint32_t sp, const uint32_t *d,
                                               uint32_t &wlo, uint32_t &whi)
    {
        const uint32_t x = DITHER ? sp + d[START] : sp;
        pdm_tick<START>(a, x, wlo, whi);
        PdmRange<START + 1, CNT - 1, DITHER>::run(a, sp, d, wlo, whi);
    }
};
template <int START, bool DITHER>
struct PdmRange<START, 0, DITHER> {
    static __device__ __forceinline__ void run(uint32_t &, uint32_t, const uint32_t *, uint32_t &,
                                               uint32_t &) {}
};
template <int BIT, int START, bool DITHER>
struct PdmRagged {
    static __device__ __forceinline__ void run(uint32_t nt, uint32_t &a, uint32_t sp, const uint32_t *d,
                                               uint32_t &wlo, uint32_t &whi)
    {
        if (nt & BIT) {                                   // wave-uniform
            PdmRange<START, BIT, DITHER>::run(a, sp, d, wlo, whi);
            PdmRagged<BIT / 2, START + BIT, DITHER>::run(nt, a, sp, d, wlo, whi);
        } else {
            PdmRagged<BIT / 2, START, DITHER>::run(nt, a, sp, d, wlo, whi);
        }
    }
};
template <int START, bool DITHER>
struct PdmRagged<0, START, DITHER> {
    static __device__ __forceinline__ void run(uint32_t, uint32_t &, uint32_t, const uint32_t *, uint32_t &,
                                               uint32_t &) {}
};

template <bool DITHER>
__global__ __launch_bounds__(1024)
void pdm_bank_kernel(const uint32_t *__restrict__ setpoint,
                     uint32_t *__restrict__ accu,
                     const uint32_t *__restrict__ dither,
                     unsigned long long *__restrict__ bits64,
                     uint32_t words64_per_tick,   // n_pad / 64
                     uint32_t nticks,
                     uint32_t n)                  // real channels; the rest is padding
{
    // [buffer][tick][wave], padded.  Two buffers: tile k is written to S[k & 1], then ONE barrier,
    // then read; the buffer tile k+1 writes was last read in tile k-1, before every thread reached
    // tile k's barrier -- so no second barrier per tile is needed.
    __shared__ unsigned long long S[2][64][17];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t row = tid >> 4, col = tid & 15; // flush: 16 lanes x 8 B = one 128-B row
    // persistent workgroups over blocks of 1024 channels; the next block's setpoint/accu are
    // requested before the ticks of the current one (few-tick launches of big banks are a
    // read-modify-write stream: 64 Mi channels x 1 tick 204 -> 154 us = 5.3 TB/s)
    const uint32_t nblocks = words64_per_tick >> 4;
    uint32_t sp_next = 0, a_next = 0;
    if (blockIdx.x < nblocks) {
        sp_next = setpoint[blockIdx.x * 1024u + tid];
        a_next = accu[blockIdx.x * 1024u + tid];
    }
    uint32_t buf = 0;
    for (uint32_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const uint32_t ch = blk * 1024u + tid;
        const uint32_t sp = sp_next;
        uint32_t a = a_next;
        const uint32_t nb = min(blk + gridDim.x, nblocks - 1) * 1024u + tid;   // last trip re-reads its own
        sp_next = setpoint[nb];
        a_next = accu[nb];
        // padding channels (setpoint 0) would still pulse under dither: the storing thread masks
        // the lanes of "its" wave (col) that lie beyond the last real channel
        const uint32_t base = blk * 1024u + col * 64u;
        const unsigned long long vmask = base + 64u <= n ? ~0ull : (base < n ? (1ull << (n - base)) - 1ull : 0ull);

        for (uint32_t t0 = 0; t0 < nticks; t0 += 64) {
            const uint32_t nt = min(64u, nticks - t0);
            uint32_t wlo = 0, whi = 0;
            if (nt == 64) PdmTicks<0, DITHER>::run(a, sp, dither + t0, wlo, whi);
            else          PdmRagged<32, 0, DITHER>::run(nt, a, sp, dither + t0, wlo, whi);   // ragged tail
            S[buf][lane][wave] = ((unsigned long long)whi << 32) | wlo;
            __syncthreads();
            if (row < nt)
                bits64[(size_t)(t0 + row) * words64_per_tick + blk * 16u + col] = S[buf][row][col] & vmask;
            buf ^= 1;
        }
        accu[ch] = a;
    }
}

// ---------------------------------------------------------------------------
// Channel-stream output: streams[tick/32][channel], bit j of a word = the pulse of that
// channel at tick 32k+j (what a per-channel decimator or DAC model consumes).  No transpose
// is needed: a lane shifts its own channel's carry into its own word, w = w + w + carry, one
// v_addc_co_u32 -- 2 vector ops per channel-tick instead of 3.  Four channels per lane so
// that every carry mask is consumed >= 3 instructions after the v_add_co that wrote it
// (gfx950 VALU-SGPR hazard), and so that a lane stores 16 contiguous bytes per 32 ticks.
// ---------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void stream_step4(u32x4 &a, const u32x4 &x, u32x4 &w)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long m0, m1, m2, m3;
    asm("v_add_co_u32_e64 %0, %8, %0, %12\n\t"
        "v_add_co_u32_e64 %1, %9, %1, %13\n\t"
        "v_add_co_u32_e64 %2, %10, %2, %14\n\t"
        "v_add_co_u32_e64 %3, %11, %3, %15\n\t"
        "v_addc_co_u32_e64 %4, vcc, %4, %4, %8\n\t"
        "v_addc_co_u32_e64 %5, vcc, %5, %5, %9\n\t"
        "v_addc_co_u32_e64 %6, vcc, %6, %6, %10\n\t"
        "v_addc_co_u32_e64 %7, vcc, %7, %7, %11"
        : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(w.x), "+v"(w.y), "+v"(w.z), "+v"(w.w),
          "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(x.x), "v"(x.y), "v"(x.z), "v"(x.w)
        : "vcc");
#else
    (void)a; (void)x; (void)w;
#endif
}

template <bool DITHER>
__global__ __launch_bounds__(256)
void pdm_stream_kernel(const uint32_t *__restrict__ setpoint, uint32_t *__restrict__ accu,
                       const uint32_t *__restrict__ dither, uint32_t *__restrict__ streams,
                       uint32_t ngroups /* n_pad / 4 */, uint32_t nwords /* nticks / 32 */)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= ngroups) return;
    const u32x4 sp = reinterpret_cast<const u32x4 *>(setpoint)[g];
    u32x4 a = reinterpret_cast<const u32x4 *>(accu)[g];
    for (uint32_t k = 0; k < nwords; k++) {
        u32x4 w = 0;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            const u32x4 x = DITHER ? sp + dither[k * 32 + j] : sp;
            stream_step4(a, x, w);
        }
        // the oldest tick sits in bit 31: reverse so that bit j is tick 32k+j
        u32x4 o;
        o.x = __brev(w.x); o.y = __brev(w.y); o.z = __brev(w.z); o.w = __brev(w.w);
        reinterpret_cast<u32x4 *>(streams)[(size_t)k * ngroups + g] = o;
    }
    reinterpret_cast<u32x4 *>(accu)[g] = a;
}

}  // namespace

namespace smx {

int launch_pdm_streams(const uint32_t *d_setpoint, uint32_t *d_accu, const uint32_t *d_dither,
                       uint32_t *d_streams, uint32_t n_pad, uint32_t nticks, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || (nticks & 31)) {
        set_error("launch_pdm_streams: n_pad=%u nticks=%u (multiple of 32)", n_pad, nticks);
        return SMX_E_ARG;
    }
    if (nticks == 0) return SMX_OK;
    const uint32_t ngroups = n_pad / 4;
    const dim3 grid((ngroups + 255) / 256), block(256);
    if (d_dither)
        hipLaunchKernelGGL(pdm_stream_kernel<true>, grid, block, 0, stream, d_setpoint, d_accu, d_dither,
                           d_streams, ngroups, nticks / 32);
    else
        hipLaunchKernelGGL(pdm_stream_kernel<false>, grid, block, 0, stream, d_setpoint, d_accu, d_dither,
                           d_streams, ngroups, nticks / 32);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}


int launch_pdm_bank(const uint32_t *d_setpoint, uint32_t *d_accu,
                    const uint32_t *d_dither, uint32_t *d_bits, uint32_t n_pad,
                    uint32_t n, uint32_t nticks, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || n > n_pad) {
        set_error("launch_pdm_bank: n_pad=%u n=%u", n_pad, n);
        return SMX_E_ARG;
    }
    if (nticks == 0) return SMX_OK;
    // persistent workgroups (2 x 1024 threads are resident per CU; 1024 measured best: 512 loses 6 % on
    // 1 Mi channels x 4096 ticks, 256 loses 25 %)
    static const char *env = getenv("SMX_PDM_GRID");                    // tuning override
    uint32_t gx = env ? (uint32_t)atoi(env) : 1024u;
    if (gx > n_pad / 1024) gx = n_pad / 1024;
    if (gx < 1) gx = 1;
    const dim3 grid(gx), block(1024);
    auto *b64 = reinterpret_cast<unsigned long long *>(d_bits);
    if (d_dither)
        hipLaunchKernelGGL(pdm_bank_kernel<true>, grid, block, 0, stream, d_setpoint, d_accu,
                           d_dither, b64, n_pad / 64, nticks, n);
    else
        hipLaunchKernelGGL(pdm_bank_kernel<false>, grid, block, 0, stream, d_setpoint, d_accu,
                           d_dither, b64, n_pad / 64, nticks, n);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
