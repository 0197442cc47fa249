/* synth_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See synth_oracle.h for scope, provenance and the parity-pinning status of
 * every function.  Citations are relative to /root/reference.
 * Build: gcc -O2 -fwrapv -ffp-contract=off -fPIC -shared (oracle/Makefile). */
#include "synth_oracle.h"
#include <string.h>

/* ======================================================================== */
/* linux/synth.c                                                            */
/* ======================================================================== */

/* linux/synth.c:69-98.  The reference builds the top octave (MIDI notes
 * 116..127) at compile time in double: entry 11 is note 127's frequency
 * scaled to a 32-bit phasor at 48 kHz, every lower semitone is the next one
 * times 2^(-1/12), nested left to right (N10 = SEMI*N11, N9 = SEMI*N10 ...),
 * and each is truncated to uint32 by the static initialiser. */
void orc_note_tab(uint32_t out[12]) {
    const double semi = 0.9438743126816935;
    double x = (12543.853951415975 / 48000.0) * 4294967296.0;
    for (int i = 11; i >= 0; i--) {
        out[i] = (uint32_t)x;
        x = semi * x;
    }
}

/* linux/synth.c:100-115.  Packed (octave<<4 | semitone): notes 0..7 are the
 * tail of octave 10 (semitones 4..11), then octaves 9 down to 0. */
uint8_t orc_midi_tab(int note) {
    note &= 127;
    int octave, n;
    if (note < 8) { octave = 10; n = note + 4; }
    else          { octave = 9 - (note - 8) / 12; n = (note - 8) % 12; }
    return (uint8_t)(((octave & 15) << 4) | (n & 15));
}

/* linux/synth.c:118-125 (without the LOG at :123). */
uint32_t orc_note_to_inc(int note) {
    uint32_t tab[12];
    orc_note_tab(tab);
    int on = orc_midi_tab(note & 127);
    return tab[on & 15] >> (on >> 4);
}

/* linux/synth.c:145-154: first voice with inc==0, else steal voice 0. */
int orc_voice_alloc(const uint32_t *inc, uint32_t n) {
    for (uint32_t v = 0; v < n; v++)
        if (inc[v] == 0) return (int)v;
    return 0;
}

/* linux/synth.c:156-160.  Phase (state) is NOT reset. */
void orc_note_on(int *note2voice, uint32_t *inc, uint32_t n, int note) {
    int v = orc_voice_alloc(inc, n);
    note2voice[note % 128] = v;
    inc[v] = orc_note_to_inc(note % 128);
}

/* linux/synth.c:161-165.  A note never played maps to voice 0 and silences it. */
void orc_note_off(int *note2voice, uint32_t *inc, uint32_t n, int note) {
    (void)n;
    int v = note2voice[note % 128];
    note2voice[note % 128] = 0;
    inc[v] = 0;
}

/* linux/synth.c:169-179: integer part of sum_tick_saw.  State is read
 * BEFORE the increment; voices with inc==0 neither contribute nor advance. */
int32_t orc_sum_tick_saw(const uint32_t *inc, uint32_t *state, uint32_t n) {
    int32_t sum = 0;
    for (uint32_t v = 0; v < n; v++) {
        if (inc[v]) {
            int32_t p = (int32_t)state[v];
            sum += (p >> 4);
            state[v] += inc[v];
        }
    }
    return sum;
}
/* The same tick without the branch (the term is masked; adding an increment of 0 is no advance): what orc_synth_run
 * uses on the tests' big banks, where `if (inc)` on a half-active bank is a coin toss per voice (a 2^25-voice test:
 * 33 s with the branch, 7 s without).  tests/test_oracle_golden.py holds it against orc_sum_tick_saw; the timed CPU
 * baselines of bench.py (banks of up to 2^20 voices) keep the reference's loop as it is. */
static int32_t sum_tick_saw_branch_free(const uint32_t *inc, uint32_t *state, uint32_t n) {
    int32_t sum = 0;
    for (uint32_t v = 0; v < n; v++) {
        const uint32_t i = inc[v];
        const int32_t on = -(int32_t)(i != 0);          /* all ones when the voice is on */
        const int32_t p = (int32_t)state[v];
        sum += (p >> 4) & on;
        state[v] += i;
    }
    return sum;
}

/* linux/synth.c:180: (1.0/2^32) * (float)sum, double product, float return. */
float orc_bus_to_float(int32_t sum) {
    return (float)((1.0 / 4294967296.0) * (double)((float)sum));
}

/* linux/synth.c:182-195 (unused by synth_run; OR of the sign bits). */
float orc_sum_tick_square(const uint32_t *inc, uint32_t *state, uint32_t n) {
    uint32_t accu = 0;
    for (uint32_t v = 0; v < n; v++) {
        if (inc[v]) {
            accu |= state[v] & 0x80000000u;
            state[v] += inc[v];
        }
    }
    return (float)((1.0 / 4294967296.0) * (double)((float)accu));
}

/* linux/synth.c:196-202: sample-outer, voice-inner. */
void orc_synth_run(const uint32_t *inc, uint32_t *state, uint32_t n,
                   float *vec, int32_t *bus, int nframes) {
    for (int i = 0; i < nframes; i++) {
        int32_t s = n > (1u << 20) ? sum_tick_saw_branch_free(inc, state, n) : orc_sum_tick_saw(inc, state, n);
        if (bus) bus[i] = s;
        if (vec) vec[i] = orc_bus_to_float(s);
    }
}

/* linux/synth.c:236-258: 3-byte events on channel 0 only. */
void orc_midi_event(int *note2voice, uint32_t *inc, uint32_t n,
                    const uint8_t *msg, size_t size) {
    if (size != 3) return;
    if (msg[0] == 0xB0 && msg[1] >= 23 && msg[1] <= 31) {
        /* CC: the reference does nothing here (:240-245) */
    } else if (msg[0] == 0x90) {
        if (msg[2] == 0) orc_note_off(note2voice, inc, n, msg[1]);
        else             orc_note_on (note2voice, inc, n, msg[1]);
    } else if (msg[0] == 0x80) {
        orc_note_off(note2voice, inc, n, msg[1]);
    }
}

/* ======================================================================== */
/* stm32f103/mod_pdm.c                                                      */
/* ======================================================================== */

/* mod_pdm.c:230-244: sp = setpoint + dither (wraps); "adds" accu += sp and
 * the carry flag is the output pulse. */
static inline uint32_t pdm_channel_step(uint32_t setpoint, uint32_t *accu,
                                        uint32_t dither) {
    uint32_t sp = setpoint + dither;
    uint32_t a0 = *accu;
    uint32_t a1 = a0 + sp;
    *accu = a1;
    return a1 < a0;          /* unsigned carry out of the 32-bit add */
}

void orc_pdm_tick(const uint32_t *setpoint, uint32_t *accu, uint32_t n,
                  uint32_t dither, uint32_t *bits) {
    uint32_t words = (n + 31) >> 5;
    memset(bits, 0, words * sizeof(uint32_t));
    for (uint32_t c = 0; c < n; c++)
        bits[c >> 5] |= pdm_channel_step(setpoint[c], &accu[c], dither) << (c & 31);
}

void orc_pdm_run(const uint32_t *setpoint, uint32_t *accu, uint32_t n,
                 const uint32_t *dither, uint32_t nticks, uint32_t *bits) {
    uint32_t words = (n + 31) >> 5;
    for (uint32_t t = 0; t < nticks; t++)
        orc_pdm_tick(setpoint, accu, n, dither ? dither[t] : 0,
                     bits + (size_t)t * words);
}

/* mod_pdm.c:259-286: "rrx" shifts each carry into the MSB of a shift register
 * in channel order, then the register is aligned so that channel c lands on
 * pin 4+c, and a set/clear BSRR word is formed. */
uint32_t orc_pdm_bsrr(const uint32_t *setpoint, uint32_t *accu, uint32_t nb,
                      uint32_t dither) {
    uint32_t shiftreg = 0;
    for (uint32_t c = 0; c < nb; c++) {
        uint32_t carry = pdm_channel_step(setpoint[c], &accu[c], dither);
        shiftreg = (carry << 31) | (shiftreg >> 1);
    }
    uint32_t set  = shiftreg >> (32 - nb - 4);
    uint32_t mask = ((1u << nb) - 1) << 4;
    uint32_t clr  = (~set) & mask;
    return set | (clr << 16);
}

/* mod_pdm.c:167-175: 24-bit phase with a curvature (phase>>9) feedback. */
uint32_t orc_pwm_update(uint32_t *phase, uint32_t speed) {
    uint32_t ph = *phase;
    uint32_t duty = ph >> 16;
    *phase = (ph + speed + (ph >> 9)) & 0xFFFFFFu;
    return duty;
}

/* ======================================================================== */
/* stm32f103/pdm.h                                                          */
/* ======================================================================== */

/* pdm.h:13-24.  Output is the quantised LAST state (one sample delay). */
uint32_t orc_pdm1_update(uint32_t *s, uint32_t input, uint32_t sh) {
    uint32_t q = s[0] >> sh;
    s[0] += input - (q << sh);
    return q;
}
/* pdm.h:32-40 */
uint32_t orc_pdm2_update(uint32_t *s, uint32_t input, uint32_t sh, uint32_t dither) {
    uint32_t q = s[1] >> sh;
    uint32_t a = (q << sh) + dither;
    s[0] += input - a;
    s[1] += s[0] - a;
    return q;
}
/* pdm.h:48-57 */
uint32_t orc_pdm3_update(uint32_t *s, uint32_t input, uint32_t sh, uint32_t dither) {
    uint32_t q = s[2] >> sh;
    uint32_t a = (q << sh) + dither;
    s[0] += input - a;
    s[1] += s[0] - a;
    s[2] += s[1] - a;
    return q;
}
/* pdm.h:67-77 */
uint32_t orc_pdm4_update(uint32_t *s, uint32_t input, uint32_t sh, uint32_t dither) {
    uint32_t q = s[3] >> sh;
    uint32_t a = (q << sh) + dither;
    s[0] += input - a;
    s[1] += s[0] - a;
    s[2] += s[1] - a;
    s[3] += s[2] - a;
    return q;
}

/* ======================================================================== */
/* stm32f103/mod_pdm_pwm.c + mod_controlrate.c                              */
/* ======================================================================== */

/* mod_controlrate.c:28-40, for every channel.  Runs as a software interrupt
 * of lower priority than the PDM ISR (mod_synth.c:78-80), i.e. after the
 * tick that triggered it has finished. */
static void pwm_bank_control_update(struct orc_pwm_bank *b) {
    for (uint32_t c = 0; c < b->n; c++) {
        b->pos1[c] += (uint32_t)b->vel1[c] << b->div_log;
        int32_t span = (int32_t)(b->setpoint[c] - b->pos1[c]);
        b->vel1[c] = span >> b->div_log;
    }
}

/* mod_pdm_pwm.c:123-143 */
void orc_pwm_bank_run(struct orc_pwm_bank *b, const uint32_t *dither,
                      uint32_t nticks, uint8_t *duty) {
    uint32_t div = 1u << b->div_log;
    for (uint32_t t = 0; t < nticks; t++) {
        uint32_t d = dither ? dither[t] : 0;
        int trigger = 0;
        if (b->div_count == 0) {
            for (uint32_t c = 0; c < b->n; c++) {    /* PDM_COPY_LINE :118-119 */
                b->pos0[c] = b->pos1[c];
                b->vel0[c] = b->vel1[c];
            }
            trigger = 1;                              /* control_trigger :136 */
        }
        for (uint32_t c = 0; c < b->n; c++) {         /* PDM_UPDATE_CHANNEL :108-116 */
            b->pos0[c] += (uint32_t)b->vel0[c];       /* glide :95-98 */
            uint32_t s[4] = {0, 0, 0, 0}, q;
            for (uint32_t k = 0; k < b->order; k++) s[k] = b->s[k][c];
            switch (b->order) {                       /* PDM_UPDATE = pdm<ORDER>_update :87 */
            case 1:  q = orc_pdm1_update(s, b->pos0[c], b->out_shift); break;   /* no dither input */
            case 2:  q = orc_pdm2_update(s, b->pos0[c], b->out_shift, d); break;
            case 3:  q = orc_pdm3_update(s, b->pos0[c], b->out_shift, d); break;
            default: q = orc_pdm4_update(s, b->pos0[c], b->out_shift, d); break;
            }
            for (uint32_t k = 0; k < b->order; k++) b->s[k][c] = s[k];
            if (duty) duty[(size_t)t * b->n + c] = (uint8_t)q;
        }
        b->div_count = (b->div_count + 1) % div;
        if (trigger) pwm_bank_control_update(b);
    }
}

/* ======================================================================== */
/* stm32f103/pmeas.h + mod_osc.c                                            */
/* ======================================================================== */

/* pmeas.h:64-100 */
void orc_pmeas_update(struct orc_pmeas *p, uint32_t cc) {
    uint32_t meas = cc - p->last_cc;
    p->last_cc = cc;
    uint32_t accu = p->accu;
    uint32_t accu1 = accu + meas;
    uint32_t max = 1u << p->log_max;
    if (accu1 < max) {
        p->num++;
        p->accu = accu1;
    } else {
        uint32_t write = p->write + 1;
        if (p->num > 0) {
            p->avg[write & 1] = (accu << (32 - p->log_max)) / p->num;
            p->num_pub[write & 1] = p->num;
            p->write = write;
        }
        p->num = 1;
        p->accu = meas;
    }
}

/* mod_osc.c:47-74: sub-osc divide-by-two, then the period measurement.
 * (The hard-sync reset of the PWM phase, :60-62, is applied by the caller to
 * the phase it owns: see orc_pwm_update.) */
void orc_osc_event(struct orc_pmeas *p, uint32_t cc) {
    p->sub ^= 1;
    orc_pmeas_update(p, cc);
}

void orc_pwmosc_run(uint32_t *phase, const uint32_t *speed, uint32_t n,
                    const uint32_t *sync_bits, uint32_t nticks, uint8_t *duty) {
    uint32_t words = (n + 31) >> 5;
    for (uint32_t t = 0; t < nticks; t++)
        for (uint32_t c = 0; c < n; c++) {
            if (sync_bits && ((sync_bits[(size_t)t * words + (c >> 5)] >> (c & 31)) & 1))
                phase[c] = 0;                                  /* OSC_HARD_SYNC */
            uint32_t d = orc_pwm_update(&phase[c], speed[c]);
            if (duty) duty[(size_t)t * n + c] = (uint8_t)d;
        }
}

void orc_osc_bank_events(struct orc_pmeas *p, uint32_t n, const uint32_t *cc,
                         const uint32_t *valid_bits, uint32_t nevents) {
    uint32_t words = (n + 31) >> 5;
    for (uint32_t e = 0; e < nevents; e++)
        for (uint32_t c = 0; c < n; c++)
            if (!valid_bits || ((valid_bits[(size_t)e * words + (c >> 5)] >> (c & 31)) & 1))
                orc_osc_event(&p[c], cc[(size_t)e * n + c]);
}

/* ======================================================================== */
/* linux/clock.c                                                            */
/* ======================================================================== */
uint32_t orc_bpm_to_hperiod(uint32_t sr, uint32_t bpm) { return (sr * 5) / (bpm * 4); }   /* clock.c:58 */

void orc_clock_run(const uint32_t *hperiod, int32_t *phase, uint32_t *pol, uint32_t n,
                   uint32_t nframes, uint32_t *pol_bits, uint32_t *tick_bits) {
    uint32_t words = (n + 31) >> 5;
    memset(pol_bits, 0, (size_t)nframes * words * 4);
    memset(tick_bits, 0, (size_t)nframes * words * 4);
    for (uint32_t t = 0; t < nframes; t++)
        for (uint32_t c = 0; c < n; c++) {
            /* clock.c:108: `int clock_phase >= jack_nframes_t clock_hperiod` compares unsigned */
            if ((uint32_t)phase[c] >= hperiod[c]) {
                phase[c] -= (int32_t)hperiod[c];
                pol[c] ^= 1;
                if (pol[c] == 1) tick_bits[(size_t)t * words + (c >> 5)] |= 1u << (c & 31);
            }
            if (pol[c]) pol_bits[(size_t)t * words + (c >> 5)] |= 1u << (c & 31);
            phase[c] += 1;
        }
}

/* ======================================================================== */
/* generic/cproc.h                                                          */
/* ======================================================================== */
void orc_acc_update(uint32_t *out, uint32_t in) { *out += in; }
void orc_edge_update(uint32_t *out, uint32_t *last, uint32_t in) {
    *out = (in != *last);
    *last = in;
}

void orc_cproc_run(const struct orc_cproc_node *nodes, uint32_t n_nodes, uint32_t n_inst,
                   uint32_t n_inputs, uint32_t *state, const uint32_t *input,
                   const uint32_t *g, uint32_t nticks, uint32_t out_node, uint32_t *out) {
    for (uint32_t t = 0; t < nticks; t++) {
        uint32_t gt = g ? g[t] : 0xFFFFFFFFu;
        for (uint32_t i = 0; i < n_inst; i++) {
            for (uint32_t k = 0; k < n_nodes; k++) {            /* allocation order */
                if (!(gt & nodes[k].cond)) continue;             /* PROC_COND */
                uint32_t src = nodes[k].in;
                uint32_t in = (src & ORC_CPROC_INPUT)
                    ? input[((size_t)t * n_inputs + (src & 0x7FFFFFFFu)) * n_inst + i]
                    : state[((size_t)src * 2 + 0) * n_inst + i];
                uint32_t *o = &state[((size_t)k * 2 + 0) * n_inst + i];
                uint32_t *l = &state[((size_t)k * 2 + 1) * n_inst + i];
                if (nodes[k].proc == ORC_PROC_ACC) orc_acc_update(o, in);
                else if (nodes[k].proc == ORC_PROC_EDGE) orc_edge_update(o, l, in);
                else if (nodes[k].proc == ORC_PROC_GPIN) *o = in;   /* hw_cproc_stm32f103.h:12-14, the pin = an input word */
            }
            if (out) out[(size_t)t * n_inst + i] = state[((size_t)out_node * 2) * n_inst + i];
        }
    }
}

/* ======================================================================== */
/* poly voice -- BUILD-DEFINED EXTENSION; no reference.  */
/* ======================================================================== */
void orc_poly_run(struct orc_poly_bank *b, int32_t *bus_lr, int nframes) {
    memset(bus_lr, 0, sizeof(int32_t) * 2 * (size_t)nframes);
    for (uint32_t v = 0; v < b->n; v++) {
        uint32_t inc = b->inc[v];
        if (!inc) continue;                       /* 0 == off, as linux/synth.c:32 */
        uint32_t phase = b->phase[v], level = b->level[v], stage = b->stage[v];
        float y = b->y[v], a = b->a[v];
        uint32_t ar = b->ar[v], dr = b->dr[v], sl = b->sl[v], rr = b->rr[v];
        int32_t pl = (int32_t)(b->pan[v] & 0xFFFF), pr = (int32_t)(b->pan[v] >> 16);
        /* gate is a control-rate input, picked up at the start of the block */
        if (b->gate[v]) { if (stage == ORC_ENV_IDLE || stage == ORC_ENV_R) stage = ORC_ENV_A; }
        else            { if (stage != ORC_ENV_IDLE) stage = ORC_ENV_R; }
        for (int i = 0; i < nframes; i++) {
            float x = (float)(int32_t)phase * 0x1p-31f;
            phase += inc;
            float t = x - y;
            y = y + a * t;                        /* two roundings, never fused */
            switch (stage) {
            case ORC_ENV_A: {
                uint32_t nl = level + ar;
                if (nl < level) { level = 0xFFFFFFFFu; stage = ORC_ENV_D; }
                else level = nl;
                break; }
            case ORC_ENV_D:
                if (level <= sl || level - sl <= dr) { level = sl; stage = ORC_ENV_S; }
                else level -= dr;
                break;
            case ORC_ENV_S: level = sl; break;
            case ORC_ENV_R:
                if (level <= rr) { level = 0; stage = ORC_ENV_IDLE; }
                else level -= rr;
                break;
            default: level = 0; break;
            }
            float g = (float)(level >> 8) * 0x1p-24f;
            float o = y * g;
            int32_t q = (int32_t)(o * 524288.0f);
            bus_lr[2 * i]     += q * pl;
            bus_lr[2 * i + 1] += q * pr;
        }
        b->phase[v] = phase; b->level[v] = level; b->stage[v] = stage; b->y[v] = y;
    }
}
