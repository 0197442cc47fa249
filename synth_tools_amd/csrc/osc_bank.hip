// osc_bank.hip -- oscillator-side banks for gfx950 (MI355X):
//
// (1) pwmosc: the hard-synced saw/PWM phase accumulator of
//     stm32f103/mod_pdm.c:159-175, N voices:
//         duty  = phase >> 16
//         phase = (phase + speed + (phase >> 9)) & 0xFFFFFF
//     with OSC_HARD_SYNC (phase = 0) applied before a tick whenever the
//     analog oscillator's discharge pulse arrived since the previous tick
//     (mod_osc.c:60-62: the osc ISR is pre-empted by, hence lands between, PDM
//     ticks).  The feedback term phase >> 9 makes this a true recurrence in
//     time, so time stays sequential inside a lane.
//
// (2) osc events: the EXTI ISR of stm32f103/mod_osc.c:47-74 with the period
//     measurement of stm32f103/pmeas.h:64-100, N oscillators, E event slots.
//
// Mapping: one lane per oscillator, state in registers for the run.  pwmosc
// packs 4 adjacent oscillators per lane so a wave writes 256 contiguous duty
// bytes per tick; sync/valid masks use the pulse-matrix layout of the PDM
// bank (channel c -> bit c&31 of word c>>5 of a tick/event row).
#include "smx_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// One tick of 4 oscillators: duty byte = bits 16..23 of the 24-bit phase (3 byte-permutes pack
// four of them), then the recurrence.  m: my 4 hard-sync bits of this tick (bit k -> oscillator k).
__device__ __forceinline__ uint32_t pwmosc_tick4(u32x4 &ph, const u32x4 &sp, uint32_t m)
{
    if (m & 1) ph.x = 0;
    if (m & 2) ph.y = 0;
    if (m & 4) ph.z = 0;
    if (m & 8) ph.w = 0;
    const uint32_t lo = __builtin_amdgcn_perm(ph.y, ph.x, 0x0c0c0602u);   // [x.b2, y.b2, 0, 0]
    const uint32_t hi = __builtin_amdgcn_perm(ph.w, ph.z, 0x06020c0cu);   // [0, 0, z.b2, w.b2]
    ph = (ph + sp + (ph >> 9)) & 0xFFFFFFu;
    return lo | hi;
}

template <bool SYNC>
__global__ __launch_bounds__(256)
void pwmosc_kernel(uint32_t *__restrict__ phase, const uint32_t *__restrict__ speed,
                   const uint32_t *__restrict__ sync_bits, uint32_t words_per_row,
                   uint32_t *__restrict__ duty32, uint32_t ngroups, uint32_t nticks)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= ngroups) return;
    u32x4 ph = reinterpret_cast<u32x4 *>(phase)[g];
    const u32x4 sp = reinterpret_cast<const u32x4 *>(speed)[g];
    const uint32_t word = g >> 3, shift = (g & 7) * 4;      // my 4 sync bits inside a 32-channel word
    uint32_t t = 0;
    if (SYNC) {
        // the sync words of 8 ticks are requested together, ahead of the 8 dependent ticks that
        // use them (one load per tick inside the recurrence made the run 3x slower than no sync)
        const uint32_t *sb = sync_bits + word;
        for (; t + 8 <= nticks; t += 8) {
            uint32_t m[8];
#pragma unroll
            for (int k = 0; k < 8; k++) m[k] = sb[(size_t)(t + k) * words_per_row];
#pragma unroll
            for (int k = 0; k < 8; k++)
                duty32[(size_t)(t + k) * ngroups + g] = pwmosc_tick4(ph, sp, m[k] >> shift);
        }
    }
    for (; t < nticks; t++) {
        const uint32_t m = SYNC ? sync_bits[(size_t)t * words_per_row + word] >> shift : 0u;
        duty32[(size_t)t * ngroups + g] = pwmosc_tick4(ph, sp, m);
    }
    reinterpret_cast<u32x4 *>(phase)[g] = ph;
}

__global__ __launch_bounds__(256)
void osc_events_kernel(smx::PmeasArrays p, const uint32_t *__restrict__ cc,
                       const uint32_t *__restrict__ valid_bits, uint32_t words_per_row,
                       uint32_t n, uint32_t nevents, uint32_t log_max)
{
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n) return;
    uint32_t write = p.write[c], num = p.num[c], accu = p.accu[c], last_cc = p.last_cc[c];
    uint32_t sub = p.sub[c];
    uint32_t avg[2] = {p.avg0[c], p.avg1[c]}, npub[2] = {p.num0[c], p.num1[c]};
    const uint32_t max = 1u << log_max;
    // one event of the ISR for this oscillator
    auto event = [&](uint32_t now) __attribute__((always_inline)) {
        sub ^= 1;                                            // sub-osc divide by two, mod_osc.c:65
        const uint32_t meas = now - last_cc;                 // pmeas.h:67-68
        last_cc = now;
        const uint32_t accu1 = accu + meas;
        if (accu1 < max) {                                   // pmeas.h:77-80
            num++;
            accu = accu1;
        } else {                                             // pmeas.h:81-100
            const uint32_t w = write + 1;
            if (num > 0) {
                const uint32_t a = (accu << (32 - log_max)) / num;
                if (w & 1) { avg[1] = a; npub[1] = num; } else { avg[0] = a; npub[0] = num; }
                write = w;
            }
            num = 1;
            accu = meas;
        }
    };
    // the timestamps (and valid words) of 8 events are requested together, ahead of the state
    // machine that consumes them one by one (one load per event inside the loop: 107 us for
    // 1 Mi oscillators x 64 events)
    const uint32_t vword = c >> 5, vbit = c & 31;
    uint32_t e = 0;
    for (; e + 8 <= nevents; e += 8) {
        uint32_t now[8], vw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            now[k] = cc[(size_t)(e + k) * n + c];
            vw[k] = valid_bits ? valid_bits[(size_t)(e + k) * words_per_row + vword] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int k = 0; k < 8; k++)
            if ((vw[k] >> vbit) & 1) event(now[k]);
    }
    for (; e < nevents; e++) {
        if (valid_bits && !((valid_bits[(size_t)e * words_per_row + vword] >> vbit) & 1)) continue;
        event(cc[(size_t)e * n + c]);
    }
    p.write[c] = write; p.num[c] = num; p.accu[c] = accu; p.last_cc[c] = last_cc; p.sub[c] = sub;
    p.avg0[c] = avg[0]; p.avg1[c] = avg[1]; p.num0[c] = npub[0]; p.num1[c] = npub[1];
}

// (3) clock bank: the integer-divider square wave / MIDI clock of linux/clock.c:106-120,
//     N dividers.  One lane per clock.  The polarity of a wave's 64 clocks is ONE 64-bit scalar
//     mask: a frame is  roll = (phase >= hperiod)  [v_cmp -> SGPR pair],  polarity ^= roll,
//     midi tick = roll & polarity  [two scalar ops],  phase = phase + 1 - (roll ? hperiod : 0)
//     [v_cndmask + v_sub], and lane t keeps frame t's two masks (4 x v_writelane with a constant
//     lane select, as in the PDM bank); tiles of 64 frames are transposed through LDS into
//     frame-major bit matrices like the PDM bank's pulses.  7 vector instructions per 64 clock-frames.
template <int T>
__device__ __forceinline__ void keep_frame(uint32_t &wlo, uint32_t &whi, unsigned long long m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // the masks are scalar-ALU results; the s_nop covers a mask that the compiler took straight
    // from a vector compare (2 wait states between a VALU SGPR write and a VALU read on gfx950)
    asm("s_nop 1\n\t"
        "v_writelane_b32 %0, %2, %4\n\t"
        "v_writelane_b32 %1, %3, %4"
        : "+v"(wlo), "+v"(whi)
        : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)), "n"(T));
#else
    (void)wlo; (void)whi; (void)m;
#endif
}

template <int T>
struct ClockFrames {
    static __device__ __forceinline__ void run(uint32_t &ph, uint32_t hpm1, unsigned long long &pm,
                                               uint32_t &plo, uint32_t &phi, uint32_t &tlo, uint32_t &thi)
    {
        const bool roll = ph > hpm1;                          // ph >= hperiod, clock.c:108 (unsigned compare)
        const unsigned long long r = __ballot(roll);
        ph -= roll ? hpm1 : 0xFFFFFFFFu;                      // (ph - hperiod) + 1, or ph + 1
        pm ^= r;
        keep_frame<T>(plo, phi, pm);
        keep_frame<T>(tlo, thi, r & pm);                      // positive edge: MIDI clock 0xF8
        ClockFrames<T + 1>::run(ph, hpm1, pm, plo, phi, tlo, thi);
    }
};
template <>
struct ClockFrames<64> {
    static __device__ __forceinline__ void run(uint32_t &, uint32_t, unsigned long long &, uint32_t &, uint32_t &,
                                               uint32_t &, uint32_t &) {}
};

__global__ __launch_bounds__(256)
void clock_kernel(const uint32_t *__restrict__ hperiod, uint32_t *__restrict__ phase,
                  uint32_t *__restrict__ pol, unsigned long long *__restrict__ pol_bits,
                  unsigned long long *__restrict__ tick_bits, uint32_t words64_per_row,
                  uint32_t n, uint32_t nframes)
{
    __shared__ unsigned long long S[2][64][5];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t c = blockIdx.x * 256u + tid;
    const bool live = c < n;
    // a padding lane never rolls: hperiod 0 would make hpm1 wrap to "always" -- give it the longest period
    const uint32_t hp = live ? hperiod[c] : 0xFFFFFFFFu;
    uint32_t ph = live ? phase[c] : 0u;
    const uint32_t po0 = live ? pol[c] : 0u;
    // the mask form needs polarity in {0, 1} (clock.c's own values) and hperiod > 0 in every lane of
    // the wave; anything else (hperiod 0 rolls every frame, other polarity words) takes the generic
    // frames below, where polarity is only "zero / non-zero" in the bit matrices as well
    const bool fast = __all(hp != 0 && po0 <= 1u);
    unsigned long long pm = __ballot(po0 != 0);
    uint32_t po = po0;
    const uint32_t hpm1 = hp - 1;
    for (uint32_t t0 = 0; t0 < nframes; t0 += 64) {
        const uint32_t nt = min(64u, nframes - t0);
        uint32_t plo = 0, phi = 0, tlo = 0, thi = 0;
        if (nt == 64 && fast) {
            ClockFrames<0>::run(ph, hpm1, pm, plo, phi, tlo, thi);
        } else {
            if (fast) po = (uint32_t)((pm >> lane) & 1);      // after full tiles of the mask form
            for (uint32_t t = 0; t < nt; t++) {               // ragged tail / generic values
                const bool roll = live && ph >= hp;
                if (roll) { ph -= hp; po ^= 1u; }
                const unsigned long long mp = __ballot(po != 0);
                const unsigned long long mt = __ballot(roll && po == 1u);
                if (lane == t) { plo = (uint32_t)mp; phi = (uint32_t)(mp >> 32); tlo = (uint32_t)mt; thi = (uint32_t)(mt >> 32); }
                ph += 1;
            }
            pm = __ballot(po != 0);
        }
        S[0][lane][wave] = ((unsigned long long)phi << 32) | plo;
        S[1][lane][wave] = ((unsigned long long)thi << 32) | tlo;
        __syncthreads();
        // 64 rows x 4 words: one thread per (row, word)
        const uint32_t row = tid >> 2, col = tid & 3;
        if (row < nt) {
            const size_t o = (size_t)(t0 + row) * words64_per_row + blockIdx.x * 4u + col;
            pol_bits[o] = S[0][row][col];
            tick_bits[o] = S[1][row][col];
        }
        __syncthreads();
    }
    if (fast) po = (uint32_t)((pm >> lane) & 1);
    if (live) { phase[c] = ph; pol[c] = po; }
}

}  // namespace

namespace smx {

int launch_clock(const uint32_t *d_hperiod, uint32_t *d_phase, uint32_t *d_pol, uint32_t *d_pol_bits,
                 uint32_t *d_tick_bits, uint32_t n_pad, uint32_t n, uint32_t nframes, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || n > n_pad) { set_error("launch_clock: n_pad=%u n=%u", n_pad, n); return SMX_E_ARG; }
    if (nframes == 0) return SMX_OK;
    hipLaunchKernelGGL(clock_kernel, dim3(n_pad / 256), dim3(256), 0, stream, d_hperiod, d_phase, d_pol,
                       reinterpret_cast<unsigned long long *>(d_pol_bits),
                       reinterpret_cast<unsigned long long *>(d_tick_bits), n_pad / 64, n, nframes);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_pwmosc(uint32_t *d_phase, const uint32_t *d_speed, const uint32_t *d_sync_bits,
                  uint8_t *d_duty, uint32_t n_pad, uint32_t nticks, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023)) { set_error("launch_pwmosc: n_pad=%u", n_pad); return SMX_E_ARG; }
    if (nticks == 0) return SMX_OK;
    const uint32_t ngroups = n_pad / 4;
    const dim3 grid((ngroups + 255) / 256), block(256);
    auto *o = reinterpret_cast<uint32_t *>(d_duty);
    if (d_sync_bits)
        hipLaunchKernelGGL(pwmosc_kernel<true>, grid, block, 0, stream, d_phase, d_speed, d_sync_bits,
                           n_pad / 32, o, ngroups, nticks);
    else
        hipLaunchKernelGGL(pwmosc_kernel<false>, grid, block, 0, stream, d_phase, d_speed, d_sync_bits,
                           n_pad / 32, o, ngroups, nticks);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_osc_events(const PmeasArrays &p, const uint32_t *d_cc, const uint32_t *d_valid_bits,
                      uint32_t n, uint32_t nevents, uint32_t log_max, hipStream_t stream)
{
    if (n == 0 || log_max == 0 || log_max > 31) { set_error("launch_osc_events: n=%u log_max=%u", n, log_max); return SMX_E_ARG; }
    if (nevents == 0) return SMX_OK;
    hipLaunchKernelGGL(osc_events_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, p, d_cc,
                       d_valid_bits, (n + 31) / 32, n, nevents, log_max);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
