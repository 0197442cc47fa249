// abi_osc.cpp -- part of the C-ABI of libsynth_mi355x.so (include/synth_mi355x.h): oscillator bank and clock bank
// Host side of the drop-in boundary.  No CPU compute fallback exists: every compute entry
// point needs a HIP device and fails with SMX_E_NOGPU otherwise.
#include "abi_internal.h"
// ---------------------------------------------------------------------------
// oscillator bank: mod_pdm.c pwm_update + hard sync, mod_osc.c ISR, pmeas.h
// ---------------------------------------------------------------------------
static void pmeas_slots(smx::PmeasArrays &d, void **slots[9])
{
    slots[0] = (void **)&d.write; slots[1] = (void **)&d.avg0; slots[2] = (void **)&d.avg1;
    slots[3] = (void **)&d.num0;  slots[4] = (void **)&d.num1; slots[5] = (void **)&d.num;
    slots[6] = (void **)&d.accu;  slots[7] = (void **)&d.last_cc; slots[8] = (void **)&d.sub;
}

extern "C" smx_osc *smx_osc_create(uint32_t n, int device)
{
    if (n == 0 || n > 0xFFFFF000u) { set_error("smx_osc_create: n=%u (1..2^32-4096)", n); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_osc_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_osc_create: device %d of %d", device, ndev); return nullptr; }
    smx_osc *o = new smx_osc();
    o->n = n;
    o->n_pad = smx::round_up(n, 1024);
    o->device = device;
    const size_t bytes = (size_t)o->n_pad * 4;
    void **slots[9];
    pmeas_slots(o->pm, slots);
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&o->d_phase, bytes) == hipSuccess &&
              hipMalloc((void **)&o->d_speed, bytes) == hipSuccess &&
              hipMemsetAsync(o->d_phase, 0, bytes, o->stream) == hipSuccess;
    for (int i = 0; ok && i < 9; i++)
        ok = hipMalloc(slots[i], bytes) == hipSuccess && hipMemsetAsync(*slots[i], 0, bytes, o->stream) == hipSuccess;
    if (ok) {
        std::vector<uint32_t> sp(o->n_pad, 256u * 13u);          // pwm_speed, mod_pdm.c:161
        ok = hipMemcpyAsync(o->d_speed, sp.data(), bytes, hipMemcpyHostToDevice, o->stream) == hipSuccess &&
             hipStreamSynchronize(o->stream) == hipSuccess;
    }
    if (!ok) {
        set_error("smx_osc_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_osc_destroy(o);
        return nullptr;
    }
    return o;
}

extern "C" void smx_osc_destroy(smx_osc *o)
{
    if (!o) return;
    (void)hipSetDevice(o->device);
    if (o->stream) (void)hipStreamSynchronize(o->stream);
    void **slots[9];
    pmeas_slots(o->pm, slots);
    for (int i = 0; i < 9; i++)
        if (*slots[i]) (void)hipFree(*slots[i]);
    if (o->d_phase) (void)hipFree(o->d_phase);
    if (o->d_speed) (void)hipFree(o->d_speed);
    if (o->d_tmp) (void)hipFree(o->d_tmp);
    if (o->d_tmp2) (void)hipFree(o->d_tmp2);
    if (o->d_duty) (void)hipFree(o->d_duty);
    if (o->stream) (void)hipStreamDestroy(o->stream);
    delete o;
}

extern "C" int smx_osc_set_log_max(smx_osc *o, uint32_t log_max)
{
    if (!o || log_max == 0 || log_max > 31) { set_error("smx_osc_set_log_max: %u (1..31)", log_max); return SMX_E_ARG; }
    o->log_max = log_max;
    return SMX_OK;
}

extern "C" int smx_osc_load_pwm(smx_osc *o, const uint32_t *phase, const uint32_t *speed)
{
    if (!o) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(o->device));
    SMX_HIP(hipStreamSynchronize(o->stream));
    if (phase) SMX_HIP(hipMemcpy(o->d_phase, phase, (size_t)o->n * 4, hipMemcpyHostToDevice));
    if (speed) SMX_HIP(hipMemcpy(o->d_speed, speed, (size_t)o->n * 4, hipMemcpyHostToDevice));
    return SMX_OK;
}

extern "C" int smx_osc_read_pwm(smx_osc *o, uint32_t *phase, uint32_t *speed)
{
    if (!o) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(o->device));
    SMX_HIP(hipStreamSynchronize(o->stream));
    if (phase) SMX_HIP(hipMemcpy(phase, o->d_phase, (size_t)o->n * 4, hipMemcpyDeviceToHost));
    if (speed) SMX_HIP(hipMemcpy(speed, o->d_speed, (size_t)o->n * 4, hipMemcpyDeviceToHost));
    return SMX_OK;
}

extern "C" int smx_osc_tick_n(smx_osc *o, uint32_t n_ticks, const uint32_t *sync_bits, uint8_t *duty)
{
    if (!o) return SMX_E_ARG;
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(o->device));
    int rv = dev_reserve((void **)&o->d_duty, &o->duty_cap, (size_t)n_ticks * o->n_pad, o->stream);
    if (rv) return rv;
    const uint32_t *d_sync = nullptr;
    if (sync_bits) {
        const size_t row = (size_t)o->n_pad / 8, words = (o->n + 31) / 32;
        rv = dev_reserve(&o->d_tmp, &o->tmp_cap, (size_t)n_ticks * row, o->stream);
        if (rv) return rv;
        SMX_HIP(hipMemsetAsync(o->d_tmp, 0, (size_t)n_ticks * row, o->stream));
        SMX_HIP(hipMemcpy2DAsync(o->d_tmp, row, sync_bits, words * 4, words * 4, n_ticks,
                                 hipMemcpyHostToDevice, o->stream));
        d_sync = (const uint32_t *)o->d_tmp;
    }
    rv = smx::launch_pwmosc(o->d_phase, o->d_speed, d_sync, o->d_duty, o->n_pad, n_ticks, o->stream);
    if (rv) return rv;
    if (duty)
        SMX_HIP(hipMemcpy2DAsync(duty, o->n, o->d_duty, o->n_pad, o->n, n_ticks, hipMemcpyDeviceToHost,
                                 o->stream));
    SMX_HIP(hipStreamSynchronize(o->stream));
    return SMX_OK;
}

extern "C" int smx_osc_events(smx_osc *o, uint32_t n_events, const uint32_t *cc, const uint32_t *valid_bits)
{
    if (!o || (n_events && !cc)) return SMX_E_ARG;
    if (n_events == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(o->device));
    int rv = dev_reserve(&o->d_tmp2, &o->tmp2_cap, (size_t)n_events * o->n * 4, o->stream);
    if (rv) return rv;
    SMX_HIP(hipMemcpyAsync(o->d_tmp2, cc, (size_t)n_events * o->n * 4, hipMemcpyHostToDevice, o->stream));
    const uint32_t *d_valid = nullptr;
    if (valid_bits) {
        const size_t bytes = (size_t)n_events * ((o->n + 31) / 32) * 4;
        rv = dev_reserve(&o->d_tmp, &o->tmp_cap, bytes, o->stream);
        if (rv) return rv;
        SMX_HIP(hipMemcpyAsync(o->d_tmp, valid_bits, bytes, hipMemcpyHostToDevice, o->stream));
        d_valid = (const uint32_t *)o->d_tmp;
    }
    rv = smx::launch_osc_events(o->pm, (const uint32_t *)o->d_tmp2, d_valid, o->n, n_events, o->log_max, o->stream);
    if (rv) return rv;
    SMX_HIP(hipStreamSynchronize(o->stream));
    return SMX_OK;
}

static int pmeas_copy(smx_osc *o, const struct smx_pmeas_arrays *a, bool to_device)
{
    if (!o || !a) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(o->device));
    SMX_HIP(hipStreamSynchronize(o->stream));
    void **slots[9];
    pmeas_slots(o->pm, slots);
    void *host[9] = {a->write, a->avg0, a->avg1, a->num0, a->num1, a->num, a->accu, a->last_cc, a->sub};
    for (int i = 0; i < 9; i++) {
        if (!host[i]) continue;
        if (to_device) SMX_HIP(hipMemcpy(*slots[i], host[i], (size_t)o->n * 4, hipMemcpyHostToDevice));
        else           SMX_HIP(hipMemcpy(host[i], *slots[i], (size_t)o->n * 4, hipMemcpyDeviceToHost));
    }
    return SMX_OK;
}

extern "C" int smx_osc_load_pmeas(smx_osc *o, const struct smx_pmeas_arrays *a) { return pmeas_copy(o, a, true); }
extern "C" int smx_osc_read_pmeas(smx_osc *o, const struct smx_pmeas_arrays *a) { return pmeas_copy(o, a, false); }

// ---------------------------------------------------------------------------
// clock bank: linux/clock.c:58-62, 106-120
// ---------------------------------------------------------------------------
struct smx_clock {
    uint32_t n = 0, n_pad = 0;
    int device = 0;
    uint32_t *d_hperiod = nullptr, *d_phase = nullptr, *d_pol = nullptr;
    void *d_pbits = nullptr; size_t pbits_cap = 0;
    void *d_tbits = nullptr; size_t tbits_cap = 0;
    hipStream_t stream = nullptr;
};

extern "C" uint32_t smx_bpm_to_hperiod(uint32_t sr, uint32_t bpm) { return bpm ? (sr * 5) / (bpm * 4) : 0; }

extern "C" smx_clock *smx_clock_create(uint32_t n, int device)
{
    if (n == 0 || n > 0xFFFFF000u) { set_error("smx_clock_create: n=%u", n); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_clock_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_clock_create: device %d of %d", device, ndev); return nullptr; }
    smx_clock *c = new smx_clock();
    c->n = n;
    c->n_pad = smx::round_up(n, 1024);
    c->device = device;
    const size_t bytes = (size_t)c->n_pad * 4;
    std::vector<uint32_t> ones(c->n_pad, 1u);                      // clock_pol = 1, clock.c:62
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&c->d_hperiod, bytes) == hipSuccess &&
              hipMalloc((void **)&c->d_phase, bytes) == hipSuccess &&
              hipMalloc((void **)&c->d_pol, bytes) == hipSuccess &&
              // cleared and filled on the bank's own stream, then waited for: nothing of this object is ever
              // ordered by the null stream (which a non-blocking stream does not wait for)
              hipMemsetAsync(c->d_hperiod, 0, bytes, c->stream) == hipSuccess &&
              hipMemsetAsync(c->d_phase, 0, bytes, c->stream) == hipSuccess &&
              hipMemcpyAsync(c->d_pol, ones.data(), bytes, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
              hipStreamSynchronize(c->stream) == hipSuccess;
    if (!ok) {
        set_error("smx_clock_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_clock_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void smx_clock_destroy(smx_clock *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_hperiod) (void)hipFree(c->d_hperiod);
    if (c->d_phase) (void)hipFree(c->d_phase);
    if (c->d_pol) (void)hipFree(c->d_pol);
    if (c->d_pbits) (void)hipFree(c->d_pbits);
    if (c->d_tbits) (void)hipFree(c->d_tbits);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int smx_clock_load(smx_clock *c, const uint32_t *hperiod, const int32_t *phase, const uint32_t *pol)
{
    if (!c) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(c->device));
    SMX_HIP(hipStreamSynchronize(c->stream));
    if (hperiod) SMX_HIP(hipMemcpy(c->d_hperiod, hperiod, (size_t)c->n * 4, hipMemcpyHostToDevice));
    if (phase) SMX_HIP(hipMemcpy(c->d_phase, phase, (size_t)c->n * 4, hipMemcpyHostToDevice));
    if (pol) SMX_HIP(hipMemcpy(c->d_pol, pol, (size_t)c->n * 4, hipMemcpyHostToDevice));
    return SMX_OK;
}

extern "C" int smx_clock_read(smx_clock *c, uint32_t *hperiod, int32_t *phase, uint32_t *pol)
{
    if (!c) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(c->device));
    SMX_HIP(hipStreamSynchronize(c->stream));
    if (hperiod) SMX_HIP(hipMemcpy(hperiod, c->d_hperiod, (size_t)c->n * 4, hipMemcpyDeviceToHost));
    if (phase) SMX_HIP(hipMemcpy(phase, c->d_phase, (size_t)c->n * 4, hipMemcpyDeviceToHost));
    if (pol) SMX_HIP(hipMemcpy(pol, c->d_pol, (size_t)c->n * 4, hipMemcpyDeviceToHost));
    return SMX_OK;
}

extern "C" int smx_clock_run(smx_clock *c, uint32_t n_frames, uint32_t *pol_bits, uint32_t *tick_bits)
{
    if (!c) return SMX_E_ARG;
    if (n_frames == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(c->device));
    const size_t row = (size_t)c->n_pad / 8, words = (c->n + 31) / 32;
    int rv;
    if ((rv = dev_reserve(&c->d_pbits, &c->pbits_cap, (size_t)n_frames * row, c->stream))) return rv;
    if ((rv = dev_reserve(&c->d_tbits, &c->tbits_cap, (size_t)n_frames * row, c->stream))) return rv;
    rv = smx::launch_clock(c->d_hperiod, c->d_phase, c->d_pol, (uint32_t *)c->d_pbits, (uint32_t *)c->d_tbits,
                           c->n_pad, c->n, n_frames, c->stream);
    if (rv) return rv;
    if (pol_bits)
        SMX_HIP(hipMemcpy2DAsync(pol_bits, words * 4, c->d_pbits, row, words * 4, n_frames, hipMemcpyDeviceToHost, c->stream));
    if (tick_bits)
        SMX_HIP(hipMemcpy2DAsync(tick_bits, words * 4, c->d_tbits, row, words * 4, n_frames, hipMemcpyDeviceToHost, c->stream));
    SMX_HIP(hipStreamSynchronize(c->stream));
    return SMX_OK;
}
