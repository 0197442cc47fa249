// pwm_bank.hip -- N-channel noise-shaped PWM bank with control-rate glide, for
// gfx950 (MI355X).
//
// Replaces the TIM ISR of stm32f103/mod_pdm_pwm.c:123-143 (3 channels there),
// the multi-bit noise shapers of stm32f103/pdm.h:10-77 and the control-rate
// line-segment update of stm32f103/mod_controlrate.c:28-40.  Per tick:
//     if (div_count == 0) { line[0] = line[1]; trigger the control update }
//     per channel: line[0].position += line[0].velocity            (glide)
//                  duty = pdmK_update(&pdm, position, out_shift, dither)
//     div_count = (div_count + 1) % CONTROL_DIV
//     (triggered) per channel: pos1 += vel1 << DIV_LOG;
//                              vel1 = (int32)(setpoint - pos1) >> DIV_LOG
// with pdmK (K = 1..4 integrators, pdm.h):
//     q = sK >> sh;  a = (q << sh) + dither;  s1 += in - a;  s2 += s1 - a; ...
// The control update is a lower-priority software interrupt in the firmware
// (mod_synth.c:78-80): it runs after the tick that triggered it.
//
// Mapping: 4 adjacent channels per lane (struct-of-arrays, 16-byte loads), all
// state in registers for the whole run; each tick a lane emits one 32-bit word
// holding its 4 duty bytes, so a wave writes 256 contiguous bytes of the
// tick-major duty matrix duty[tick][channel] (uint8).
#include "smx_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// v_perm_b32: pick 4 of the 8 bytes of {hi, lo}; selector byte k names the source
// of result byte k (0-3 = lo bytes, 4-7 = hi bytes, 0x0c = constant 0).
__device__ __forceinline__ uint32_t byte_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    (void)hi; (void)lo; (void)sel;
    return 0;
#endif
}

// SH24: out_shift == 24 (the firmware's value): the duty is the top byte of the
// last integrator, gathered from 4 channels with 3 byte-permutes.
template <int ORDER, bool DITHER, bool SH24>
__global__ __launch_bounds__(256)
void pwm_bank_kernel(smx::PwmArrays p, const uint32_t *__restrict__ dither,
                     uint32_t *__restrict__ duty32,   // [nticks][n_pad/4]
                     uint32_t ngroups,                // n_pad / 4
                     uint32_t nticks, uint32_t div_count, uint32_t div_log, uint32_t sh)
{
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= ngroups) return;
    const u32x4 sp = reinterpret_cast<const u32x4 *>(p.setpoint)[g];
    u32x4 pos0 = reinterpret_cast<const u32x4 *>(p.pos0)[g];
    u32x4 vel0 = reinterpret_cast<const u32x4 *>(p.vel0)[g];
    u32x4 pos1 = reinterpret_cast<const u32x4 *>(p.pos1)[g];
    u32x4 vel1 = reinterpret_cast<const u32x4 *>(p.vel1)[g];
    u32x4 s[ORDER];
#pragma unroll
    for (int k = 0; k < ORDER; k++) s[k] = reinterpret_cast<const u32x4 *>(p.s[k])[g];
    const uint32_t div_mask = (1u << div_log) - 1;
    if (SH24) sh = 24;
    const uint32_t hi_mask = ~0u << sh;               // (q << sh) == s & hi_mask

    // one ISR tick without the control-rate bookkeeping
    auto tick = [&](uint32_t t) {
        const uint32_t nd = DITHER ? 0u - dither[t] : 0u;
        pos0 += vel0;                                         // pdm_update_glide, :95-98
        const u32x4 last = s[ORDER - 1];                      // pdm.h: output = quantised last state
        // a = (q << sh) + dither;  every integrator adds (previous - a): keep -a
        const u32x4 na = nd - (last & hi_mask);
        s[0] += pos0 + na;
#pragma unroll
        for (int k = 1; k < ORDER; k++) s[k] += s[k - 1] + na;
        uint32_t word;
        if (SH24) {
            const uint32_t lo = byte_perm(last.y, last.x, 0x0c0c0703u);   // [x.b3, y.b3, 0, 0]
            const uint32_t hi = byte_perm(last.w, last.z, 0x07030c0cu);   // [0, 0, z.b3, w.b3]
            word = lo | hi;
        } else {
            const u32x4 q = last >> sh;
            word = (q.x & 0xFF) | ((q.y & 0xFF) << 8) | ((q.z & 0xFF) << 16) | (q.w << 24);
        }
        duty32[(size_t)t * ngroups + g] = word;
    };

    // Time is cut at the control-rate boundaries (every 1 << div_log ticks) so that the
    // long runs in between are branch-free and unrollable.
    uint32_t t = 0;
    while (t < nticks) {
        uint32_t seg = min(nticks - t, (div_mask + 1) - div_count);
        if (div_count == 0) {
            pos0 = pos1; vel0 = vel1;                         // PDM_COPY_LINE, mod_pdm_pwm.c:118-119
            tick(t);
            // control_trigger() -> lower-priority SWI runs after this tick:
            pos1 += vel1 << div_log;                          // pdm_update_line, mod_controlrate.c:28-40
            const u32x4 span = sp - pos1;
            vel1.x = (uint32_t)((int32_t)span.x >> div_log);
            vel1.y = (uint32_t)((int32_t)span.y >> div_log);
            vel1.z = (uint32_t)((int32_t)span.z >> div_log);
            vel1.w = (uint32_t)((int32_t)span.w >> div_log);
            t++; seg--;
            div_count = 1 & div_mask;
        }
        const uint32_t end = t + seg;
#pragma unroll 4
        for (; t < end; t++) tick(t);
        div_count = (div_count + seg) & div_mask;
    }
    reinterpret_cast<u32x4 *>(p.pos0)[g] = pos0;
    reinterpret_cast<u32x4 *>(p.vel0)[g] = vel0;
    reinterpret_cast<u32x4 *>(p.pos1)[g] = pos1;
    reinterpret_cast<u32x4 *>(p.vel1)[g] = vel1;
#pragma unroll
    for (int k = 0; k < ORDER; k++) reinterpret_cast<u32x4 *>(p.s[k])[g] = s[k];
}

template <int ORDER>
int launch_order(const smx::PwmArrays &p, const uint32_t *d_dither, uint8_t *d_duty, uint32_t n_pad,
                 uint32_t nticks, uint32_t div_count, uint32_t div_log, uint32_t sh, hipStream_t stream)
{
    const uint32_t ngroups = n_pad / 4;
    const dim3 grid((ngroups + 255) / 256), block(256);
    auto *o = reinterpret_cast<uint32_t *>(d_duty);
#define SMX_PWM_LAUNCH(D, S)                                                                  \
    hipLaunchKernelGGL((pwm_bank_kernel<ORDER, D, S>), grid, block, 0, stream, p, d_dither, o, \
                       ngroups, nticks, div_count, div_log, sh)
    if (d_dither) { if (sh == 24) SMX_PWM_LAUNCH(true, true); else SMX_PWM_LAUNCH(true, false); }
    else          { if (sh == 24) SMX_PWM_LAUNCH(false, true); else SMX_PWM_LAUNCH(false, false); }
#undef SMX_PWM_LAUNCH
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace

namespace smx {

int launch_pwm_bank(const PwmArrays &p, int order, const uint32_t *d_dither, uint8_t *d_duty,
                    uint32_t n_pad, uint32_t nticks, uint32_t div_count, uint32_t div_log,
                    uint32_t out_shift, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || div_log == 0 || div_log > 31 || out_shift > 31 ||
        div_count >= (1u << div_log)) {
        set_error("launch_pwm_bank: n_pad=%u div_log=%u out_shift=%u div_count=%u", n_pad, div_log,
                  out_shift, div_count);
        return SMX_E_ARG;
    }
    if (nticks == 0) return SMX_OK;
    if (order == 1) d_dither = nullptr;            // pdm1_update takes no dither (pdm.h:13)
    switch (order) {
    case 1: return launch_order<1>(p, d_dither, d_duty, n_pad, nticks, div_count, div_log, out_shift, stream);
    case 2: return launch_order<2>(p, d_dither, d_duty, n_pad, nticks, div_count, div_log, out_shift, stream);
    case 3: return launch_order<3>(p, d_dither, d_duty, n_pad, nticks, div_count, div_log, out_shift, stream);
    case 4: return launch_order<4>(p, d_dither, d_duty, n_pad, nticks, div_count, div_log, out_shift, stream);
    }
    set_error("launch_pwm_bank: order %d (1..4)", order);
    return SMX_E_ARG;
}

}  // namespace smx
