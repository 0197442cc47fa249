// abi_core.cpp -- part of the C-ABI of libsynth_mi355x.so (include/synth_mi355x.h): errors, note tables, the Linux drop-in names
// Host side of the drop-in boundary.  No CPU compute fallback exists: every compute entry
// point needs a HIP device and fails with SMX_E_NOGPU otherwise.
#include "abi_internal.h"
#include <cstdarg>
#include <mutex>

namespace smx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace smx

// ---------------------------------------------------------------------------
// note tables: linux/synth.c:69-125.  The reference folds the top octave at
// compile time in double; the same double products evaluated at load time
// give the same IEEE results.
// ---------------------------------------------------------------------------
static uint32_t g_note_tab[12];
extern "C" const uint8_t midi_tab[128] = {
#define SMX_NOTE(o, n) (uint8_t)((((o) & 15) << 4) | ((n) & 15))
#define SMX_OCT(o)                                                                     \
    SMX_NOTE(o, 0), SMX_NOTE(o, 1), SMX_NOTE(o, 2), SMX_NOTE(o, 3), SMX_NOTE(o, 4),    \
    SMX_NOTE(o, 5), SMX_NOTE(o, 6), SMX_NOTE(o, 7), SMX_NOTE(o, 8), SMX_NOTE(o, 9),    \
    SMX_NOTE(o, 10), SMX_NOTE(o, 11)
    SMX_NOTE(10, 4), SMX_NOTE(10, 5), SMX_NOTE(10, 6), SMX_NOTE(10, 7),
    SMX_NOTE(10, 8), SMX_NOTE(10, 9), SMX_NOTE(10, 10), SMX_NOTE(10, 11),
    SMX_OCT(9), SMX_OCT(8), SMX_OCT(7), SMX_OCT(6), SMX_OCT(5),
    SMX_OCT(4), SMX_OCT(3), SMX_OCT(2), SMX_OCT(1), SMX_OCT(0),
#undef SMX_OCT
#undef SMX_NOTE
};
static std::once_flag g_tab_once;
static void init_note_tab()
{
    const double semitone_down = 0.9438743126816935;          // 2^(-1/12)
    double x = (12543.853951415975 / 48000.0) * 4294967296.0;  // MIDI 127 @48 kHz, 32-bit phasor
    for (int i = 11; i >= 0; i--) {
        g_note_tab[i] = (uint32_t)x;
        x = semitone_down * x;
    }
}

extern "C" phasor_t note_to_inc(int note)
{
    std::call_once(g_tab_once, init_note_tab);
    const int on = midi_tab[note & 127];
    return g_note_tab[on & 15] >> (on >> 4);
}

// ---------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------
extern "C" const char *smx_last_error(void) { return smx::g_err; }
extern "C" int smx_version(void) { return SMX_VERSION; }
extern "C" int smx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
// every stream of the device idle (what a benchmark brackets its timed region with)
extern "C" int smx_device_synchronize(int device)
{
    SMX_HIP(hipSetDevice(device));
    SMX_HIP(hipDeviceSynchronize());
    return SMX_OK;
}

// ---------------------------------------------------------------------------
// Linux drop-in: linux/synth.c:42-45, 145-165, 196-206
// ---------------------------------------------------------------------------
static smx_bank *g_dropin = nullptr;      // 64-voice scratch bank, JACK RT thread only
static std::mutex g_dropin_mu;

extern "C" int voice_alloc(struct synth *x)
{
    for (unsigned v = 0; v < 64; v++)
        if (x->voice[v].note_inc == 0) return (int)v;
    return 0;
}

extern "C" void synth_note_on(struct synth *x, int note)
{
    const int v = voice_alloc(x);
    x->note2voice[note % 128] = v;
    x->voice[v].note_inc = note_to_inc(note % 128);
}

extern "C" void synth_note_off(struct synth *x, int note)
{
    const int v = x->note2voice[note % 128];
    x->note2voice[note % 128] = 0;
    x->voice[v].note_inc = 0;
}

extern "C" void synth_init(struct synth *x) { memset(x, 0, sizeof(*x)); }

extern "C" void synth_midi_event(struct synth *x, const uint8_t *msg, size_t size)
{
    if (size != 3) return;
    if (msg[0] == 0x90) {                     // note on, channel 0 (linux/synth.c:246-256)
        if (msg[2] == 0) synth_note_off(x, msg[1]);
        else synth_note_on(x, msg[1]);
    } else if (msg[0] == 0x80) {              // note off, channel 0 (:257-261)
        synth_note_off(x, msg[1]);
    }                                         // CC 23..31 on 0xB0: accepted, no action (:240-245)
}

// Run the caller's 64 voices through the scratch bank: upload, n frames, download.
static void dropin_run(struct synth *x, float *vec, int n, bool square)
{
    std::lock_guard<std::mutex> lock(g_dropin_mu);
    if (!g_dropin) {
        g_dropin = smx_bank_create(64, 0);
        if (!g_dropin) SMX_ASSERT_OK(SMX_E_NOGPU, "synth_run: smx_bank_create");
    }
    uint32_t inc[64], state[64];
    for (int v = 0; v < 64; v++) { inc[v] = x->voice[v].note_inc; state[v] = x->voice[v].note_state; }
    if (square) {
        SMX_ASSERT_OK(smx_bank_load(g_dropin, inc, state), "sum_tick_square: load");
        SMX_ASSERT_OK(smx_bank_run_square(g_dropin, vec, n), "sum_tick_square: run");
    } else {
        SMX_ASSERT_OK(smx::bank_dropin_run(g_dropin, inc, state, vec, n), "synth_run: run");
    }
    // the advanced phases in closed form (an off voice does not advance): no read-back needed
    for (int v = 0; v < 64; v++) x->voice[v].note_state = state[v] + (uint32_t)n * inc[v];
}

extern "C" void synth_run(struct synth *x, float *vec, int n)
{
    if (n <= 0) return;
    dropin_run(x, vec, n, false);
}

// linux/synth.c:169-181 and :182-195 are global symbols in the reference too: one sample.
extern "C" float sum_tick_saw(struct synth *x)
{
    float v = 0.0f;
    dropin_run(x, &v, 1, false);
    return v;
}

extern "C" float sum_tick_square(struct synth *x)
{
    float v = 0.0f;
    dropin_run(x, &v, 1, true);
    return v;
}
