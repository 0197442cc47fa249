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
// of ticks.  Round 3: the accumulator is kept LAZILY, like the saw bank's phase: HBM holds accu0[] as of
// some tick, the host counts the ticks run since (T) and the device keeps the sum D of the dither words of those
// ticks; the accumulator a launch starts from is accu0 + T*setpoint + D (mod 2^32: the modulator is linear between
// pulses), and NOTHING is written back -- a launch reads 8 B per channel and writes its pulse bits
// (smx_pdm_read / _load / _set_setpoint materialise: accu0 += T*setpoint + D, T = 0, D = 0).  D lives in two
// device words used alternately (a launch with dither reads one and writes the other: late workgroups of the
// same launch must still see the old value).  The carry of
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
                     const uint32_t *__restrict__ accu,     // accu0[]: as of `elapsed` ticks ago
                     const uint32_t *__restrict__ dither,
                     unsigned long long *__restrict__ bits64,
                     uint32_t words64_per_tick,   // n_pad / 64
                     uint32_t nticks,
                     uint32_t n,                  // real channels; the rest is padding
                     uint32_t elapsed,            // ticks run since accu0 was valid
                     const uint32_t *__restrict__ dsum_in,   // sum of the dither words of those ticks
                     uint32_t *__restrict__ dsum_out)        // DITHER: receives dsum_in + this launch's dither words
{
    // [buffer][tick][wave], padded.  Two buffers: tile k is written to S[k & 1], then ONE barrier,
    // then read; the buffer tile k+1 writes was last read in tile k-1, before every thread reached
    // tile k's barrier -- so no second barrier per tile is needed.
    __shared__ unsigned long long S[2][64][17];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t row = tid >> 4, col = tid & 15; // flush: 16 lanes x 8 B = one 128-B row
    // persistent workgroups over blocks of 1024 channels; the next block's setpoint/accu are
    // requested before the ticks of the current one (few-tick launches of big banks are a memory stream: 64 Mi
    // channels x 1 tick 204 -> 154 us with this prefetch, 121 us since the accumulator is lazy and nothing is
    // written back; launches of <= 8 ticks on >= 2^20 channels now take pdm_fewticks_kernel)
    const uint32_t nblocks = words64_per_tick >> 4;
    const uint32_t dsum0 = *dsum_in;
    if (DITHER && blockIdx.x == 0 && wave == 0) {
        // this launch's dither words, summed by one wave, for the launches that follow (the other word of the pair)
        uint32_t d = 0;
        for (uint32_t t = lane; t < nticks; t += 64) d += dither[t];
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        if (lane == 0) *dsum_out = dsum0 + d;
    }
    uint32_t sp_next = 0, a_next = 0;
    if (blockIdx.x < nblocks) {
        sp_next = setpoint[blockIdx.x * 1024u + tid];
        a_next = accu[blockIdx.x * 1024u + tid];
    }
    uint32_t buf = 0;
    for (uint32_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const uint32_t sp = sp_next;
        uint32_t a = a_next + elapsed * sp + dsum0;          // the accumulator as of now (lazy: see the header)
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
    }
}

// One or two ticks per launch (the tick ABI) on a big bank: with the lazy accumulator the launch is a pure READ stream
// of 8 B per channel, and the 1024-thread tile machinery above (64-tick tiles, LDS transpose, a barrier per 1024
// channels) is in its way.  Here: one lane per channel, four 64-channel words per wave and trip (eight 4-byte loads in
// flight per lane), the wave's carry-outs of a tick are the compare mask of the add (__ballot: the rrx register, 64
// wide), and lane 0 stores it as one 64-bit word of the tick-major matrix.
template <bool DITHER, bool ONE>
__global__ __launch_bounds__(256)
void pdm_fewticks_kernel(const uint32_t *__restrict__ setpoint, const uint32_t *__restrict__ accu,
                         const uint32_t *__restrict__ dither, unsigned long long *__restrict__ bits64,
                         uint32_t words64_per_tick, uint32_t nticks /* <= 8 */, uint32_t n, uint32_t elapsed,
                         const uint32_t *__restrict__ dsum_in, uint32_t *__restrict__ dsum_out)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t dsum0 = *dsum_in;
    uint32_t d[8];
#pragma unroll
    for (int t = 0; t < 8; t++) d[t] = (DITHER && (uint32_t)t < nticks) ? dither[t] : 0u;     // wave-uniform
    if (DITHER && blockIdx.x == 0 && tid == 0) {
        uint32_t sum = dsum0;
#pragma unroll
        for (int t = 0; t < 8; t++) sum += d[t];             // words beyond nticks are 0
        *dsum_out = sum;
    }
    const uint32_t gw = blockIdx.x * 4u + (tid >> 6), nwaves = gridDim.x * 4u;
    for (uint32_t w0 = gw * 4u; w0 < words64_per_tick; w0 += nwaves * 4u) {
        uint32_t sp[4], a[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {                       // words64_per_tick is a multiple of 16: w0 + k is in range
            const uint32_t ch = (w0 + k) * 64u + lane;
            sp[k] = __builtin_nontemporal_load(setpoint + ch);
            a[k] = __builtin_nontemporal_load(accu + ch);
        }
        if constexpr (ONE) {
            // one tick: lane 0 stores each word as soon as its two loads are there (measured against the form below
            // on one box: 86.3 vs 89-91.6 us for 64 Mi channels -- the stream should not wait for all eight loads)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t base = (w0 + k) * 64u;
                const unsigned long long vm = base + 64u <= n ? ~0ull : (base < n ? (1ull << (n - base)) - 1ull : 0ull);
                const uint32_t x = sp[k] + d[0];
                const uint32_t acc1 = a[k] + elapsed * sp[k] + dsum0 + x;
                const unsigned long long m1 = __ballot(acc1 < x);          // carry out of accu += x
                if (lane == 0) bits64[w0 + k] = m1 & vm;
            }
            continue;
        }
        // The four words of a tick leave the wave as ONE store instruction: lanes 0..3 take the four carry masks (the
        // rrx registers of words w0..w0+3) and write 32 contiguous bytes -- one memory transaction per tick and trip
        // instead of four 8-byte ones from lane 0 (round 3: the store transactions, not the arithmetic, were what a
        // second tick cost: 64 Mi channels x 2 ticks 112.8 us against 85.0 for one tick).
        // padding channels (setpoint 0) would still pulse under dither: masked beyond the last real channel
        const uint32_t mybase = (w0 + (lane & 3u)) * 64u;
        const unsigned long long vmask = mybase + 64u <= n ? ~0ull : (mybase < n ? (1ull << (n - mybase)) - 1ull : 0ull);
        uint32_t acc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) acc[k] = a[k] + elapsed * sp[k] + dsum0;
#pragma unroll
        for (int t = 0; t < 8; t++) {                                      // (static indices: d[] stays in registers)
            if ((uint32_t)t < nticks) {                                    // wave-uniform
                unsigned long long m[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t x = sp[k] + d[t];
                    acc[k] += x;
                    m[k] = __ballot(acc[k] < x);                           // carry out of accu += x
                }
                unsigned long long v = m[0];
                v = lane == 1 ? m[1] : v;
                v = lane == 2 ? m[2] : v;
                v = lane == 3 ? m[3] : v;
                if (lane < 4) bits64[(size_t)t * words64_per_tick + w0 + lane] = v & vmask;
            }
        }
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
void pdm_stream_kernel(const uint32_t *__restrict__ setpoint, const uint32_t *__restrict__ accu,
                       const uint32_t *__restrict__ dither, uint32_t *__restrict__ streams,
                       uint32_t ngroups /* n_pad / 4 */, uint32_t nwords /* nticks / 32 */,
                       uint32_t elapsed, const uint32_t *__restrict__ dsum_in, uint32_t *__restrict__ dsum_out)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    const uint32_t dsum0 = *dsum_in;
    if (DITHER && blockIdx.x == 0 && threadIdx.x < 64) {
        uint32_t d = 0;
        for (uint32_t t = threadIdx.x; t < nwords * 32u; t += 64) d += dither[t];
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        if (threadIdx.x == 0) *dsum_out = dsum0 + d;
    }
    if (g >= ngroups) return;
    const u32x4 sp = reinterpret_cast<const u32x4 *>(setpoint)[g];
    u32x4 a = reinterpret_cast<const u32x4 *>(accu)[g] + elapsed * sp + dsum0;
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
}

// accu0 += elapsed * setpoint + D: the stored accumulators are current again (the host then resets elapsed and D).
__global__ __launch_bounds__(256)
void pdm_materialize_kernel(const uint32_t *__restrict__ setpoint, uint32_t *__restrict__ accu, uint32_t n_pad,
                            uint32_t elapsed, const uint32_t *__restrict__ dsum_in)
{
    const uint32_t dsum0 = *dsum_in;
    for (uint32_t c = blockIdx.x * 256u + threadIdx.x; c < n_pad; c += gridDim.x * 256u)
        accu[c] += elapsed * setpoint[c] + dsum0;
}

}  // namespace

namespace smx {

int launch_pdm_materialize(const uint32_t *d_setpoint, uint32_t *d_accu, uint32_t n_pad, uint32_t elapsed,
                           const uint32_t *d_dsum_in, hipStream_t stream)
{
    uint32_t gx = n_pad / 256;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(pdm_materialize_kernel, dim3(gx), dim3(256), 0, stream, d_setpoint, d_accu, n_pad, elapsed, d_dsum_in);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_pdm_streams(const uint32_t *d_setpoint, const uint32_t *d_accu, const uint32_t *d_dither,
                       uint32_t *d_streams, uint32_t n_pad, uint32_t nticks, uint32_t elapsed,
                       const uint32_t *d_dsum_in, uint32_t *d_dsum_out, hipStream_t stream)
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
                           d_streams, ngroups, nticks / 32, elapsed, d_dsum_in, d_dsum_out);
    else
        hipLaunchKernelGGL(pdm_stream_kernel<false>, grid, block, 0, stream, d_setpoint, d_accu, d_dither,
                           d_streams, ngroups, nticks / 32, elapsed, d_dsum_in, d_dsum_out);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}


int launch_pdm_bank(const uint32_t *d_setpoint, const uint32_t *d_accu,
                    const uint32_t *d_dither, uint32_t *d_bits, uint32_t n_pad,
                    uint32_t n, uint32_t nticks, uint32_t elapsed, const uint32_t *d_dsum_in, uint32_t *d_dsum_out,
                    hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || n > n_pad) {
        set_error("launch_pdm_bank: n_pad=%u n=%u", n_pad, n);
        return SMX_E_ARG;
    }
    if (nticks == 0) return SMX_OK;
    static const bool no_few = getenv("SMX_PDM_NO_FEWTICKS") != nullptr;          // A/B switch
    static const char *fmax = getenv("SMX_PDM_FEWTICKS_MAX");                     // tuning override (1..8)
    // (64 Mi channels, profiles/r03_pdm_few_sweep.txt: 1 tick 86 us against 120 for the tile kernel; with one 32-byte
    // store per tick and wave-trip 2 ticks 96 against 122, 3 ticks 104 against 122, 4 ticks 110 against 123, 6 ticks
    // 123-125 against 125, 8 ticks 137 against 129 -- with a 64-bit store per word from lane 0 it was 114 / 140 / 164 /
    // 212 / 265: the store transactions were what a further tick cost)
    const uint32_t few_max = fmax ? (uint32_t)atoi(fmax) : 4u;
    if (nticks <= few_max && nticks <= 8 && n_pad >= (1u << 20) && !no_few) {
        // the tick ABI on a big bank: a read stream (pdm_fewticks_kernel)
        auto *b64f = reinterpret_cast<unsigned long long *>(d_bits);
        const uint32_t w64 = n_pad / 64;
        uint32_t gx = w64 / 16;                              // one trip of 4 waves x 4 words per workgroup at least
        if (gx > 2048) gx = 2048;
#define SMX_FEW(D_, ONE_)                                                                                       \
    hipLaunchKernelGGL((pdm_fewticks_kernel<D_, ONE_>), dim3(gx), dim3(256), 0, stream, d_setpoint, d_accu,     \
                       d_dither, b64f, w64, nticks, n, elapsed, d_dsum_in, d_dsum_out)
        if (d_dither) { if (nticks == 1) SMX_FEW(true, true); else SMX_FEW(true, false); }
        else          { if (nticks == 1) SMX_FEW(false, true); else SMX_FEW(false, false); }
#undef SMX_FEW
        SMX_HIP(hipGetLastError());
        return SMX_OK;
    }
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
                           d_dither, b64, n_pad / 64, nticks, n, elapsed, d_dsum_in, d_dsum_out);
    else
        hipLaunchKernelGGL(pdm_bank_kernel<false>, grid, block, 0, stream, d_setpoint, d_accu,
                           d_dither, b64, n_pad / 64, nticks, n, elapsed, d_dsum_in, d_dsum_out);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
