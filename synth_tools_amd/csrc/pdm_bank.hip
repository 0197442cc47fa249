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
    __shared__ unsigned long long S[64][17];      // [tick][wave], padded
    __shared__ unsigned long long VM[16];         // per wave: which lanes are real channels
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t ch = blockIdx.x * 1024u + tid;
    const uint32_t sp = setpoint[ch];
    uint32_t a = accu[ch];
    const uint32_t row = tid >> 4, col = tid & 15; // flush: 16 lanes x 8 B = one 128-B row
    // padding channels (setpoint 0) would still pulse under dither: mask them out
    const unsigned long long vm = __ballot(ch < n);
    if (lane == 0) VM[wave] = vm;

    for (uint32_t t0 = 0; t0 < nticks; t0 += 64) {
        const uint32_t nt = min(64u, nticks - t0);
        uint32_t wlo = 0, whi = 0;
        if (nt == 64) {
            PdmTicks<0, DITHER>::run(a, sp, dither + t0, wlo, whi);
        } else {
            // ragged tail: plain HIP (the compiler schedules its own hazards)
            for (uint32_t t = 0; t < nt; t++) {
                const uint32_t x = DITHER ? sp + dither[t0 + t] : sp;
                const uint32_t a1 = a + x;
                const unsigned long long m = __ballot(a1 < a);
                a = a1;
                if (lane == t) { wlo = (uint32_t)m; whi = (uint32_t)(m >> 32); }
            }
        }
        S[lane][wave] = ((unsigned long long)whi << 32) | wlo;
        __syncthreads();
        if (row < nt)
            bits64[(size_t)(t0 + row) * words64_per_tick + blockIdx.x * 16u + col] = S[row][col] & VM[col];
        __syncthreads();
    }
    accu[ch] = a;
}

}  // namespace

namespace smx {

int launch_pdm_bank(const uint32_t *d_setpoint, uint32_t *d_accu,
                    const uint32_t *d_dither, uint32_t *d_bits, uint32_t n_pad,
                    uint32_t n, uint32_t nticks, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || n > n_pad) {
        set_error("launch_pdm_bank: n_pad=%u n=%u", n_pad, n);
        return SMX_E_ARG;
    }
    if (nticks == 0) return SMX_OK;
    const dim3 grid(n_pad / 1024), block(1024);
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
