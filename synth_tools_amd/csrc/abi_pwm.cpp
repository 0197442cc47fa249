// abi_pwm.cpp -- part of the C-ABI of libsynth_mi355x.so (include/synth_mi355x.h): noise-shaped PWM bank
// Host side of the drop-in boundary.  No CPU compute fallback exists: every compute entry
// point needs a HIP device and fails with SMX_E_NOGPU otherwise.
#include "abi_internal.h"
// ---------------------------------------------------------------------------
// noise-shaped PWM bank: mod_pdm_pwm.c + pdm.h + mod_controlrate.c
// ---------------------------------------------------------------------------
struct smx_pwm {
    uint32_t n = 0, n_pad = 0;
    int order = 2, device = 0;
    uint32_t div_log = 12, out_shift = 24, div_count = 0;
    // struct controlrate (mod_controlrate.c:21-26): counts control ticks; a beat every 1024 of them
    uint32_t isr_count = 0, beat_pulse = 0, beat_handled = 0;
    smx::PwmArrays d{};
    uint32_t *d_dither = nullptr; uint32_t dither_cap = 0;
    uint8_t *d_duty = nullptr; size_t duty_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
};

static int pwm_slots(smx::PwmArrays &d, int order, void **slots[9])
{
    slots[0] = (void **)&d.setpoint; slots[1] = (void **)&d.pos0; slots[2] = (void **)&d.vel0;
    slots[3] = (void **)&d.pos1;     slots[4] = (void **)&d.vel1;
    for (int k = 0; k < order; k++) slots[5 + k] = (void **)&d.s[k];
    return 5 + order;
}

extern "C" smx_pwm *smx_pwm_create(uint32_t n_channels, int order, int device)
{
    if (n_channels == 0 || n_channels > 0xFFFFF000u || order < 1 || order > 4) {
        set_error("smx_pwm_create: n_channels=%u order=%d", n_channels, order);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_pwm_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_pwm_create: device %d of %d", device, ndev); return nullptr; }
    smx_pwm *p = new smx_pwm();
    p->n = n_channels;
    p->n_pad = smx::round_up(n_channels, 1024);
    p->order = order;
    p->device = device;
    const size_t bytes = (size_t)p->n_pad * 4;
    void **slots[9];
    const int ns = pwm_slots(p->d, order, slots);
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&p->ev_t0) == hipSuccess && hipEventCreate(&p->ev_t1) == hipSuccess;
    for (int i = 0; ok && i < ns; i++)
        ok = hipMalloc(slots[i], bytes) == hipSuccess &&
             hipMemsetAsync(*slots[i], 0, bytes, p->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(p->stream) == hipSuccess;
    if (!ok) {
        set_error("smx_pwm_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_pwm_destroy(p);
        return nullptr;
    }
    return p;
}

extern "C" void smx_pwm_destroy(smx_pwm *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    void **slots[9];
    const int ns = pwm_slots(p->d, p->order, slots);
    for (int i = 0; i < ns; i++)
        if (*slots[i]) (void)hipFree(*slots[i]);
    if (p->d_dither) (void)hipFree(p->d_dither);
    if (p->d_duty) (void)hipFree(p->d_duty);
    if (p->ev_t0) (void)hipEventDestroy(p->ev_t0);
    if (p->ev_t1) (void)hipEventDestroy(p->ev_t1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

extern "C" int smx_pwm_config(smx_pwm *p, uint32_t control_div_log, uint32_t out_shift)
{
    if (!p || control_div_log == 0 || control_div_log > 31 || out_shift > 31) {
        set_error("smx_pwm_config: div_log=%u out_shift=%u", control_div_log, out_shift);
        return SMX_E_ARG;
    }
    p->div_log = control_div_log;
    p->out_shift = out_shift;
    p->div_count &= (1u << control_div_log) - 1;
    return SMX_OK;
}

static int pwm_copy(smx_pwm *p, const struct smx_pwm_arrays *a, bool to_device)
{
    if (!p || !a) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipStreamSynchronize(p->stream));
    void **slots[9];
    const int ns = pwm_slots(p->d, p->order, slots);
    void *host[9] = {a->setpoint, a->pos0, a->vel0, a->pos1, a->vel1, a->s[0], a->s[1], a->s[2], a->s[3]};
    for (int i = 0; i < ns; i++) {
        if (!host[i]) continue;
        if (to_device) SMX_HIP(hipMemcpy(*slots[i], host[i], (size_t)p->n * 4, hipMemcpyHostToDevice));
        else           SMX_HIP(hipMemcpy(host[i], *slots[i], (size_t)p->n * 4, hipMemcpyDeviceToHost));
    }
    return SMX_OK;
}

extern "C" int smx_pwm_load(smx_pwm *p, const struct smx_pwm_arrays *a) { return pwm_copy(p, a, true); }
extern "C" int smx_pwm_read(smx_pwm *p, const struct smx_pwm_arrays *a) { return pwm_copy(p, a, false); }

extern "C" int smx_pwm_set_div_count(smx_pwm *p, uint32_t c)
{
    if (!p || c >= (1u << p->div_log)) return SMX_E_ARG;
    p->div_count = c;
    return SMX_OK;
}
extern "C" uint32_t smx_pwm_div_count(const smx_pwm *p) { return p ? p->div_count : 0; }

// struct controlrate (mod_controlrate.c:21-26)
extern "C" int smx_pwm_controlrate(const smx_pwm *p, uint32_t *isr_count, uint32_t *beat_pulse, uint32_t *beat_handled)
{
    if (!p) return SMX_E_ARG;
    if (isr_count) *isr_count = p->isr_count;
    if (beat_pulse) *beat_pulse = p->beat_pulse;
    if (beat_handled) *beat_handled = p->beat_handled;
    return SMX_OK;
}
// controlrate_poll / controlrate_beat_poll (mod_controlrate.c:64-72): the main loop handles one pending
// beat per call.  Returns 1 if a beat was handled, 0 if none was pending.
extern "C" int smx_pwm_controlrate_poll(smx_pwm *p)
{
    if (!p) return SMX_E_ARG;
    if (p->beat_handled != p->beat_pulse) { p->beat_handled++; return 1; }
    return 0;
}

// pdm_init, mod_pdm_pwm.c:147-160
extern "C" int smx_pwm_init(smx_pwm *p)
{
    if (!p) return SMX_E_ARG;
    std::vector<uint32_t> sp(p->n, pdm_safe_setpoint(0x40000000u)), z(p->n, 0u);
    sp[0] = 2000000000u;
    struct smx_pwm_arrays a = {sp.data(), z.data(), z.data(), z.data(), z.data(),
                               {z.data(), z.data(), z.data(), z.data()}};
    p->div_count = 0;
    p->isr_count = p->beat_pulse = p->beat_handled = 0;
    return smx_pwm_load(p, &a);
}

extern "C" int smx_pwm_set_setpoint(smx_pwm *p, uint32_t chan, uint32_t val)
{
    if (!p) return SMX_E_ARG;
    if (chan >= p->n) { set_error("smx_pwm_set_setpoint: chan %u >= %u", chan, p->n); return SMX_E_RANGE; }
    SMX_HIP(hipSetDevice(p->device));
    const uint32_t v = pdm_safe_setpoint(val);
    SMX_HIP(hipMemcpyAsync(p->d.setpoint + chan, &v, 4, hipMemcpyHostToDevice, p->stream));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

static int pwm_ensure(smx_pwm *p, uint32_t n_ticks)
{
    const size_t need = (size_t)n_ticks * p->n_pad;
    if (need > p->duty_cap) {
        SMX_HIP(hipStreamSynchronize(p->stream));
        if (p->d_duty) SMX_HIP(hipFree(p->d_duty));
        p->d_duty = nullptr; p->duty_cap = 0;
        SMX_HIP(hipMalloc((void **)&p->d_duty, need));
        p->duty_cap = need;
    }
    if (n_ticks > p->dither_cap) {
        SMX_HIP(hipStreamSynchronize(p->stream));
        if (p->d_dither) SMX_HIP(hipFree(p->d_dither));
        p->d_dither = nullptr; p->dither_cap = 0;
        SMX_HIP(hipMalloc((void **)&p->d_dither, (size_t)n_ticks * 4));
        p->dither_cap = n_ticks;
    }
    return SMX_OK;
}

extern "C" void *smx_pwm_dither_dev(smx_pwm *p, uint32_t n_ticks)
{
    if (!p || hipSetDevice(p->device) != hipSuccess || pwm_ensure(p, n_ticks) != SMX_OK) return nullptr;
    return p->d_dither;
}

extern "C" int smx_pwm_tick_n_async(smx_pwm *p, uint32_t n_ticks, int with_dither)
{
    if (!p) return SMX_E_ARG;
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pwm_ensure(p, n_ticks);
    if (rv) return rv;
    rv = smx::launch_pwm_bank(p->d, p->order, with_dither ? p->d_dither : nullptr, p->d_duty, p->n_pad,
                              n_ticks, p->div_count, p->div_log, p->out_shift, p->stream);
    if (rv) return rv;
    {   // control_update's beat divider (mod_controlrate.c:52-55): every control tick of this run --
        // a sample tick that starts with control_div_count == 0 (mod_pdm_pwm.c:129-137) -- does
        // `if (isr_count % 1024 == 0) beat_pulse++; isr_count++`
        const uint32_t div = 1u << p->div_log;
        const uint32_t first = (div - p->div_count) & (div - 1);            // offset of the run's first control tick
        const uint32_t k = first < n_ticks ? 1u + (n_ticks - 1u - first) / div : 0u;
        // multiples of 1024 in [isr_count, isr_count + k), isr_count being a 32-bit wrapping counter
        const uint64_t a = p->isr_count, b = a + k;
        p->beat_pulse += (uint32_t)((b + SMX_CONTROLRATE_BEAT_DIV - 1) / SMX_CONTROLRATE_BEAT_DIV -
                                    (a + SMX_CONTROLRATE_BEAT_DIV - 1) / SMX_CONTROLRATE_BEAT_DIV);
        p->isr_count = (uint32_t)b;
    }
    p->div_count = (uint32_t)(((uint64_t)p->div_count + n_ticks) & ((1u << p->div_log) - 1));
    return SMX_OK;
}

extern "C" int smx_pwm_sync(smx_pwm *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

extern "C" int smx_pwm_tick_n(smx_pwm *p, uint32_t n_ticks, const uint32_t *dither, uint8_t *duty)
{
    if (!p) return SMX_E_ARG;
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pwm_ensure(p, n_ticks);
    if (rv) return rv;
    if (dither)
        SMX_HIP(hipMemcpyAsync(p->d_dither, dither, (size_t)n_ticks * 4, hipMemcpyHostToDevice, p->stream));
    rv = smx_pwm_tick_n_async(p, n_ticks, dither != nullptr);
    if (rv) return rv;
    if (duty)
        SMX_HIP(hipMemcpy2DAsync(duty, p->n, p->d_duty, p->n_pad, p->n, n_ticks, hipMemcpyDeviceToHost,
                                 p->stream));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

extern "C" int smx_pwm_timer_start(smx_pwm *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipEventRecord(p->ev_t0, p->stream));
    return SMX_OK;
}

extern "C" int smx_pwm_timer_stop(smx_pwm *p, float *ms)
{
    if (!p || !ms) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipEventRecord(p->ev_t1, p->stream));
    SMX_HIP(hipEventSynchronize(p->ev_t1));
    SMX_HIP(hipEventElapsedTime(ms, p->ev_t0, p->ev_t1));
    return SMX_OK;
}

