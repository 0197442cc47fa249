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

template <int ORDER, bool DITHER>
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

    for (uint32_t t = 0; t < nticks; t++) {
        const uint32_t d = DITHER ? dither[t] : 0u;
        const bool trigger = (div_count == 0);
        if (trigger) { pos0 = pos1; vel0 = vel1; }           // PDM_COPY_LINE, mod_pdm_pwm.c:118-119
        pos0 += vel0;                                         // pdm_update_glide, :95-98
        const u32x4 q = s[ORDER - 1] >> sh;                   // pdm.h: output = quantised last state
        const u32x4 a = (q << sh) + d;
        s[0] += pos0 - a;
#pragma unroll
        for (int k = 1; k < ORDER; k++) s[k] += s[k - 1] - a;
        duty32[(size_t)t * ngroups + g] =
            (q.x & 0xFF) | ((q.y & 0xFF) << 8) | ((q.z & 0xFF) << 16) | (q.w << 24);
        div_count = (div_count + 1) & div_mask;
        if (trigger) {                                        // pdm_update_line, mod_controlrate.c:28-40
            pos1 += vel1 << div_log;
            const u32x4 span = sp - pos1;
            vel1.x = (uint32_t)((int32_t)span.x >> div_log);
            vel1.y = (uint32_t)((int32_t)span.y >> div_log);
            vel1.z = (uint32_t)((int32_t)span.z >> div_log);
            vel1.w = (uint32_t)((int32_t)span.w >> div_log);
        }
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
    if (d_dither)
        hipLaunchKernelGGL((pwm_bank_kernel<ORDER, true>), grid, block, 0, stream, p, d_dither, o,
                           ngroups, nticks, div_count, div_log, sh);
    else
        hipLaunchKernelGGL((pwm_bank_kernel<ORDER, false>), grid, block, 0, stream, p, d_dither, o,
                           ngroups, nticks, div_count, div_log, sh);
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
