// abi_pdm.cpp -- part of the C-ABI of libsynth_mi355x.so (include/synth_mi355x.h): carry-out PDM bank
// Host side of the drop-in boundary.  No CPU compute fallback exists: every compute entry
// point needs a HIP device and fails with SMX_E_NOGPU otherwise.
#include "abi_internal.h"
// ---------------------------------------------------------------------------
// carry-out PDM bank: stm32f103/mod_pdm.c
// ---------------------------------------------------------------------------
struct smx_pdm {
    uint32_t n = 0, n_pad = 0;
    int device = 0;
    uint32_t *d_setpoint = nullptr, *d_accu = nullptr;
    // Lazily materialised accumulators (pdm_bank.hip): accu[c] = d_accu[c] + elapsed * setpoint[c] + D, D = the sum of
    // the dither words of the `elapsed` ticks, kept on the device in d_dsum[dsum_cur] (two words used alternately).
    // Ticks only read; smx_pdm_read / _load / _set_setpoint materialise.
    uint32_t elapsed = 0;
    bool lazy = false;                           // ticks have run since the accumulators were last materialised
    uint32_t *d_dsum = nullptr;
    int dsum_cur = 0;
    uint32_t *d_dither = nullptr; uint32_t dither_cap = 0;
    uint32_t *d_bits = nullptr; size_t bits_cap = 0;     // bytes
    hipStream_t stream = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
};

extern "C" uint32_t pdm_safe_setpoint(uint32_t setpoint) { return setpoint; }  // mod_pdm.c:101-107

extern "C" smx_pdm *smx_pdm_create(uint32_t n_channels, int device)
{
    if (n_channels == 0 || n_channels > 0xFFFFF000u) { set_error("smx_pdm_create: n_channels=%u (1..2^32-4096)", n_channels); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_pdm_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_pdm_create: device %d of %d", device, ndev); return nullptr; }
    smx_pdm *p = new smx_pdm();
    p->n = n_channels;
    p->n_pad = smx::round_up(n_channels, 1024);
    p->device = device;
    const size_t bytes = (size_t)p->n_pad * 4;
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipMalloc((void **)&p->d_setpoint, bytes) == hipSuccess &&
              hipMalloc((void **)&p->d_accu, bytes) == hipSuccess &&
              hipMalloc((void **)&p->d_dsum, 8) == hipSuccess &&
              hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&p->ev_t0) == hipSuccess && hipEventCreate(&p->ev_t1) == hipSuccess &&
              hipMemsetAsync(p->d_setpoint, 0, bytes, p->stream) == hipSuccess &&
              hipMemsetAsync(p->d_accu, 0, bytes, p->stream) == hipSuccess &&
              hipMemsetAsync(p->d_dsum, 0, 8, p->stream) == hipSuccess &&
              hipStreamSynchronize(p->stream) == hipSuccess;
    if (!ok) {
        set_error("smx_pdm_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_pdm_destroy(p);
        return nullptr;
    }
    return p;
}

extern "C" void smx_pdm_destroy(smx_pdm *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->d_setpoint) (void)hipFree(p->d_setpoint);
    if (p->d_accu) (void)hipFree(p->d_accu);
    if (p->d_dsum) (void)hipFree(p->d_dsum);
    if (p->d_dither) (void)hipFree(p->d_dither);
    if (p->d_bits) (void)hipFree(p->d_bits);
    if (p->ev_t0) (void)hipEventDestroy(p->ev_t0);
    if (p->ev_t1) (void)hipEventDestroy(p->ev_t1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

// accu0 += elapsed * setpoint + D for every channel, elapsed = 0, D = 0: the stored accumulators are current again.
static int pdm_materialize(smx_pdm *p)
{
    if (!p->lazy) return SMX_OK;
    int rv = smx::launch_pdm_materialize(p->d_setpoint, p->d_accu, p->n_pad, p->elapsed, p->d_dsum + p->dsum_cur, p->stream);
    if (rv) return rv;
    SMX_HIP(hipMemsetAsync(p->d_dsum, 0, 8, p->stream));
    p->elapsed = 0;
    p->dsum_cur = 0;
    p->lazy = false;
    return SMX_OK;
}

extern "C" int smx_pdm_load(smx_pdm *p, const uint32_t *setpoint, const uint32_t *accu)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    {
        int rv = pdm_materialize(p);             // an array that is not replaced keeps its meaning
        if (rv) return rv;
    }
    SMX_HIP(hipStreamSynchronize(p->stream));
    if (setpoint) SMX_HIP(hipMemcpy(p->d_setpoint, setpoint, (size_t)p->n * 4, hipMemcpyHostToDevice));
    if (accu) SMX_HIP(hipMemcpy(p->d_accu, accu, (size_t)p->n * 4, hipMemcpyHostToDevice));
    return SMX_OK;
}

extern "C" int smx_pdm_read(smx_pdm *p, uint32_t *setpoint, uint32_t *accu)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    if (accu) {
        int rv = pdm_materialize(p);
        if (rv) return rv;
    }
    SMX_HIP(hipStreamSynchronize(p->stream));
    if (setpoint) SMX_HIP(hipMemcpy(setpoint, p->d_setpoint, (size_t)p->n * 4, hipMemcpyDeviceToHost));
    if (accu) SMX_HIP(hipMemcpy(accu, p->d_accu, (size_t)p->n * 4, hipMemcpyDeviceToHost));
    return SMX_OK;
}

// pdm_init, mod_pdm.c:320-326
extern "C" int smx_pdm_init(smx_pdm *p)
{
    if (!p) return SMX_E_ARG;
    std::vector<uint32_t> sp(p->n, pdm_safe_setpoint(0x40000000u)), ac(p->n, 0u);
    sp[0] = 2000000000u;
    return smx_pdm_load(p, sp.data(), ac.data());
}

// SETPOINT, mod_synth.c:104-111
extern "C" int smx_pdm_set_setpoint(smx_pdm *p, uint32_t chan, uint32_t val)
{
    if (!p) return SMX_E_ARG;
    if (chan >= p->n) { set_error("smx_pdm_set_setpoint: chan %u >= %u", chan, p->n); return SMX_E_RANGE; }
    SMX_HIP(hipSetDevice(p->device));
    {
        int rv = pdm_materialize(p);             // the ticks run so far used the old setpoint
        if (rv) return rv;
    }
    const uint32_t v = pdm_safe_setpoint(val);
    SMX_HIP(hipMemcpyAsync(p->d_setpoint + chan, &v, 4, hipMemcpyHostToDevice, p->stream));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

static int pdm_ensure(smx_pdm *p, uint32_t n_ticks)
{
    const size_t need = (size_t)n_ticks * (p->n_pad / 8);
    if (need > p->bits_cap) {
        SMX_HIP(hipStreamSynchronize(p->stream));
        if (p->d_bits) SMX_HIP(hipFree(p->d_bits));
        p->d_bits = nullptr; p->bits_cap = 0;
        SMX_HIP(hipMalloc((void **)&p->d_bits, need));
        p->bits_cap = need;
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

extern "C" void *smx_pdm_dither_dev(smx_pdm *p, uint32_t n_ticks)
{
    if (!p || hipSetDevice(p->device) != hipSuccess || pdm_ensure(p, n_ticks) != SMX_OK) return nullptr;
    return p->d_dither;
}

extern "C" void *smx_pdm_bits_dev(smx_pdm *p) { return p ? p->d_bits : nullptr; }

extern "C" int smx_pdm_tick_n_async(smx_pdm *p, uint32_t n_ticks, int with_dither)
{
    if (!p) return SMX_E_ARG;
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pdm_ensure(p, n_ticks);
    if (rv) return rv;
    rv = smx::launch_pdm_bank(p->d_setpoint, p->d_accu, with_dither ? p->d_dither : nullptr,
                              p->d_bits, p->n_pad, p->n, n_ticks, p->elapsed, p->d_dsum + p->dsum_cur,
                              p->d_dsum + (p->dsum_cur ^ 1), p->stream);
    if (rv) return rv;
    p->elapsed += n_ticks;                       // mod 2^32, like the accumulators
    p->lazy = true;
    if (with_dither) p->dsum_cur ^= 1;
    return SMX_OK;
}

extern "C" int smx_pdm_tick_n_streams_async(smx_pdm *p, uint32_t n_ticks, int with_dither)
{
    if (!p) return SMX_E_ARG;
    if (n_ticks & 31) { set_error("smx_pdm_tick_n_streams: n_ticks=%u is not a multiple of 32", n_ticks); return SMX_E_ARG; }
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pdm_ensure(p, n_ticks);                 // same byte count as the tick-major matrix
    if (rv) return rv;
    rv = smx::launch_pdm_streams(p->d_setpoint, p->d_accu, with_dither ? p->d_dither : nullptr, p->d_bits,
                                 p->n_pad, n_ticks, p->elapsed, p->d_dsum + p->dsum_cur, p->d_dsum + (p->dsum_cur ^ 1),
                                 p->stream);
    if (rv) return rv;
    p->elapsed += n_ticks;
    p->lazy = true;
    if (with_dither) p->dsum_cur ^= 1;
    return SMX_OK;
}

extern "C" int smx_pdm_tick_n_streams(smx_pdm *p, uint32_t n_ticks, const uint32_t *dither, uint32_t *streams)
{
    if (!p) return SMX_E_ARG;
    if (n_ticks & 31) { set_error("smx_pdm_tick_n_streams: n_ticks=%u is not a multiple of 32", n_ticks); return SMX_E_ARG; }
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pdm_ensure(p, n_ticks);
    if (rv) return rv;
    if (dither)
        SMX_HIP(hipMemcpyAsync(p->d_dither, dither, (size_t)n_ticks * 4, hipMemcpyHostToDevice, p->stream));
    rv = smx_pdm_tick_n_streams_async(p, n_ticks, dither != nullptr);
    if (rv) return rv;
    if (streams)
        SMX_HIP(hipMemcpy2DAsync(streams, (size_t)p->n * 4, p->d_bits, (size_t)p->n_pad * 4, (size_t)p->n * 4,
                                 n_ticks / 32, hipMemcpyDeviceToHost, p->stream));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

extern "C" int smx_pdm_sync(smx_pdm *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

extern "C" int smx_pdm_tick_n(smx_pdm *p, uint32_t n_ticks, const uint32_t *dither, uint32_t *bits)
{
    if (!p) return SMX_E_ARG;
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pdm_ensure(p, n_ticks);
    if (rv) return rv;
    if (dither)
        SMX_HIP(hipMemcpyAsync(p->d_dither, dither, (size_t)n_ticks * 4, hipMemcpyHostToDevice, p->stream));
    rv = smx_pdm_tick_n_async(p, n_ticks, dither != nullptr);
    if (rv) return rv;
    if (bits) {
        const size_t words = (p->n + 31) / 32;
        SMX_HIP(hipMemcpy2DAsync(bits, words * 4, p->d_bits, (size_t)p->n_pad / 8, words * 4, n_ticks,
                                 hipMemcpyDeviceToHost, p->stream));
    }
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

extern "C" int smx_pdm_timer_start(smx_pdm *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipEventRecord(p->ev_t0, p->stream));
    return SMX_OK;
}

extern "C" int smx_pdm_timer_stop(smx_pdm *p, float *ms)
{
    if (!p || !ms) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipEventRecord(p->ev_t1, p->stream));
    SMX_HIP(hipEventSynchronize(p->ev_t1));
    SMX_HIP(hipEventElapsedTime(ms, p->ev_t0, p->ev_t1));
    return SMX_OK;
}

// mod_pdm.c:271-286: the reference's rrx register holds channel c at bit
// 32-nb+c; shifted right by (32-nb-4) that is bit 4+c.
extern "C" uint32_t smx_pdm_bsrr_word(uint32_t pulse_bits, uint32_t nb)
{
    if (nb == 0 || nb > 12) return 0;
    const uint32_t mask = ((1u << nb) - 1) << 4;
    const uint32_t set = (pulse_bits << 4) & mask;
    const uint32_t clr = (~set) & mask;
    return set | (clr << 16);
}


// ---------------------------------------------------------------------------
// mod_pdm.c as ONE module: its timer ISR (stm32f103/mod_pdm.c:177-194) does, per tick,
//     pdm_update();                      the carry-out channels' pulses          (:259-286)
//     val = pwm_update();                the fixed-rate PWM channel's duty       (:166-175; hard sync: mod_osc.c:60-62)
//     if (control_div_count == 0) control_trigger();
//     control_div_count = (control_div_count + 1) % CONTROL_DIV;                 (CONTROL_DIV 256, :164)
// Here: a PDM bank and an oscillator bank ticking in lockstep (their kernels run side by side on the two banks'
// streams: the channels and the oscillators share no state) and the divider on the host.  control_trigger() pends
// the software interrupt whose handler is control_update (mod_controlrate.c:52-55): its beat divider
// (`if (isr_count % 1024 == 0) beat_pulse++; isr_count++`) is kept per module like smx_pwm_controlrate's.
// ---------------------------------------------------------------------------
struct smx_modpdm {
    smx_pdm *pdm = nullptr;
    smx_osc *osc = nullptr;
    uint32_t control_div_count = 0;                 // mod_pdm.c:165
    uint32_t isr_count = 0, beat_pulse = 0, beat_handled = 0;   // struct controlrate, mod_controlrate.c:21-26
};
static constexpr uint32_t MODPDM_CONTROL_DIV = 256;                 // mod_pdm.c:164

extern "C" smx_modpdm *smx_modpdm_create(uint32_t n_channels, uint32_t n_osc, int device)
{
    if (n_osc == 0) { set_error("smx_modpdm_create: n_osc=0 (the module has a PWM channel)"); return nullptr; }
    smx_modpdm *m = new smx_modpdm();
    m->pdm = smx_pdm_create(n_channels, device);
    m->osc = m->pdm ? smx_osc_create(n_osc, device) : nullptr;      // pwm_phase 0, pwm_speed 256 * 13 (mod_pdm.c:160-161)
    if (!m->pdm || !m->osc || smx_pdm_init(m->pdm) != SMX_OK) {     // pdm_init: setpoints (mod_pdm.c:320-326)
        smx_modpdm_destroy(m);
        return nullptr;
    }
    return m;
}

extern "C" void smx_modpdm_destroy(smx_modpdm *m)
{
    if (!m) return;
    smx_pdm_destroy(m->pdm);
    smx_osc_destroy(m->osc);
    delete m;
}

extern "C" smx_pdm *smx_modpdm_pdm(smx_modpdm *m) { return m ? m->pdm : nullptr; }
extern "C" smx_osc *smx_modpdm_osc(smx_modpdm *m) { return m ? m->osc : nullptr; }
extern "C" uint32_t smx_modpdm_control_div_count(const smx_modpdm *m) { return m ? m->control_div_count : 0; }

extern "C" int smx_modpdm_controlrate(const smx_modpdm *m, uint32_t *isr_count, uint32_t *beat_pulse, uint32_t *beat_handled)
{
    if (!m) return SMX_E_ARG;
    if (isr_count) *isr_count = m->isr_count;
    if (beat_pulse) *beat_pulse = m->beat_pulse;
    if (beat_handled) *beat_handled = m->beat_handled;
    return SMX_OK;
}

// n_ticks of the ISR.  bits: host uint32[n_ticks][ceil(n_channels/32)] or NULL; duty: host uint8[n_ticks][n_osc] or
// NULL; dither / sync_bits as smx_pdm_tick_n / smx_osc_tick_n (NULL: none).  *control_triggers receives the number
// of control_trigger() calls of this run (ticks that began with control_div_count == 0).
extern "C" int smx_modpdm_tick_n(smx_modpdm *m, uint32_t n_ticks, const uint32_t *dither, const uint32_t *sync_bits,
                                 uint32_t *bits, uint8_t *duty, uint32_t *control_triggers)
{
    if (!m) return SMX_E_ARG;
    if (control_triggers) *control_triggers = 0;
    if (n_ticks == 0) return SMX_OK;
    smx_pdm *p = m->pdm;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pdm_ensure(p, n_ticks);
    if (rv) return rv;
    // the channels' kernel is queued on the PDM bank's stream ...
    if (dither)
        SMX_HIP(hipMemcpyAsync(p->d_dither, dither, (size_t)n_ticks * 4, hipMemcpyHostToDevice, p->stream));
    rv = smx_pdm_tick_n_async(p, n_ticks, dither != nullptr);
    if (rv) return rv;
    if (bits) {
        const size_t words = (p->n + 31) / 32;
        SMX_HIP(hipMemcpy2DAsync(bits, words * 4, p->d_bits, (size_t)p->n_pad / 8, words * 4, n_ticks,
                                 hipMemcpyDeviceToHost, p->stream));
    }
    // ... and runs beside the oscillators' (their own stream; this call waits for that one)
    rv = smx_osc_tick_n(m->osc, n_ticks, sync_bits, duty);
    if (rv) return rv;
    SMX_HIP(hipStreamSynchronize(p->stream));
    // the divider: ticks t of this run with (control_div_count + t) % CONTROL_DIV == 0
    const uint32_t first = (MODPDM_CONTROL_DIV - m->control_div_count) % MODPDM_CONTROL_DIV;
    const uint32_t k = first < n_ticks ? 1u + (n_ticks - 1u - first) / MODPDM_CONTROL_DIV : 0u;
    const uint64_t a = m->isr_count, b = a + k;                     // control_update's beat divider over k calls
    m->beat_pulse += (uint32_t)((b + SMX_CONTROLRATE_BEAT_DIV - 1) / SMX_CONTROLRATE_BEAT_DIV -
                                (a + SMX_CONTROLRATE_BEAT_DIV - 1) / SMX_CONTROLRATE_BEAT_DIV);
    m->isr_count = (uint32_t)b;
    m->control_div_count = (uint32_t)(((uint64_t)m->control_div_count + n_ticks) % MODPDM_CONTROL_DIV);
    if (control_triggers) *control_triggers = k;
    return SMX_OK;
}
