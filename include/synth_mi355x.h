/* synth_mi355x.h -- C-ABI of libsynth_mi355x.so, the MI355X (gfx950) drop-in
 * for the per-sample voice loops of zwizwa/synth_tools.
 *
 * Plain C: pointers and sizes only, no HIP/torch/C++ types.  Host glue (the
 * JACK client, the Erlang port program, firmware-shaped control code) stays C
 * and calls HIP through this boundary.  Every entry point cites the reference
 * interface (file:line under /root/reference) that it replaces or widens.
 *
 * Conventions
 *  - "bank" objects are opaque handles; voice/channel state is resident in
 *    HBM, struct-of-arrays, one lane per voice.
 *  - The four Linux names (synth_init/note_on/note_off/run) are exported
 *    unchanged, with the reference's prototypes and its caller-owned
 *    `struct synth` (linux/synth.c:31-45).  They are void, like the
 *    reference; a HIP failure aborts the process with a message, which is the
 *    reference's ASSERT convention (linux/erl_tools_system.h:15,24-27).
 *  - All other calls return 0 on success and a negative code on error, the
 *    firmware handlers' convention (mod_synth.c:91,106-107,113);
 *    smx_last_error() gives the text.  There is NO CPU fallback: with no
 *    usable GPU every compute call fails with SMX_E_NOGPU.
 *  - Integer results are bit-exact with the reference CPU loops.
 *  - Threading: like the reference (everything runs on the JACK RT thread,
 *    linux/synth.c:277-282), a handle is used by one thread at a time; distinct
 *    handles are independent.  The reference-named drop-in calls serialise on
 *    an internal lock.
 */
#ifndef SYNTH_MI355X_H
#define SYNTH_MI355X_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SMX_OK          0
#define SMX_E_ARG     (-1)   /* bad argument (mod_synth.c:91)            */
#define SMX_E_RANGE   (-2)   /* index out of range (mod_synth.c:107)     */
#define SMX_E_STATE   (-3)   /* wrong state / too many args (:113)       */
#define SMX_E_NOGPU   (-10)  /* no HIP device / HIP runtime error        */
#define SMX_E_COMM    (-11)  /* RCCL error                               */

const char *smx_last_error(void);
int  smx_device_count(void);             /* 0 when no GPU is visible      */
#define SMX_VERSION 1                    /* of this header; smx_version() = the library's */
int  smx_version(void);
int  smx_device_synchronize(int device); /* all streams of the device idle */

/* ======================================================================== */
/* 1. Linux drop-in: linux/synth.c:31-45, 145-208                           */
/* ======================================================================== */
typedef uint32_t phasor_t;               /* linux/synth.c:29 */
struct voice {                           /* linux/synth.c:31-34 */
    phasor_t note_inc;                   /* 0 == off */
    phasor_t note_state;
};
struct synth {                           /* linux/synth.c:35-38 */
    int note2voice[128];
    struct voice voice[64];
};
/* Same names, prototypes and observable behaviour as linux/synth.c:42-45.
 * The caller owns *x (host memory); note_on/note_off only touch *x (as the
 * reference does); synth_run mirrors voice[] to the GPU, runs the bank kernel
 * for n frames, and writes vec[0..n) and the advanced note_state back. */
void synth_note_on (struct synth *x, int note);      /* linux/synth.c:156-160 */
void synth_note_off(struct synth *x, int note);      /* linux/synth.c:161-165 */
void synth_init    (struct synth *x);                /* linux/synth.c:204-206 */
void synth_run     (struct synth *x, float *vec, int n); /* linux/synth.c:196-202 */
/* The per-sample tick functions, linux/synth.c:169-181 and :182-195 (global symbols in
 * the reference; synth_run calls the first n times). */
float sum_tick_saw(struct synth *x);
float sum_tick_square(struct synth *x);
/* linux/synth.c:118-125 and :145-154, also global symbols in the reference. */
phasor_t note_to_inc(int note);
int      voice_alloc(struct synth *x);
extern const uint8_t midi_tab[128];                  /* linux/synth.c:106-115 */
/* MIDI dispatch of process_midi, linux/synth.c:236-258, for one event. */
void synth_midi_event(struct synth *x, const uint8_t *msg, size_t size);

/* ======================================================================== */
/* 2. Saw voice bank: linux/synth.c widened from 64 to N voices             */
/* ======================================================================== */
typedef struct smx_bank smx_bank;

/* n_voices >= 1.  device = HIP ordinal.  State zeroed (linux/synth.c:205). */
smx_bank *smx_bank_create(uint32_t n_voices, int device);
void      smx_bank_destroy(smx_bank *b);
uint32_t  smx_bank_voices(const smx_bank *b);

/* Bulk load / read back of inc[] and state[] (host arrays of n_voices).
 * Either pointer may be NULL to skip that array. */
int smx_bank_load(smx_bank *b, const uint32_t *inc, const uint32_t *state);
int smx_bank_read(smx_bank *b, uint32_t *inc, uint32_t *state);
/* smx_bank_load(inc, state) + smx_bank_run(vec, bus, n) with a single host
 * synchronisation (what the drop-in synth_run does for blocks longer than 1024 frames; shorter ones are one
 * launch with the voices as kernel arguments).  Both arrays are required; the caller's arrays may change as soon as
 * the call returns.  Not for a sharded bank (SMX_E_STATE: its loads are collective, smx_bank_load). */
int smx_bank_load_run(smx_bank *b, const uint32_t *inc, const uint32_t *state, float *vec,
                      int32_t *bus, int n);

/* note_on/off over N voices with the reference's allocator semantics:
 * first free voice, steal voice 0 when full, phase not reset, note_off of a
 * never-played note silences voice 0 (linux/synth.c:145-165). */
int smx_bank_note_on (smx_bank *b, int note);
int smx_bank_note_off(smx_bank *b, int note);
/* MIDI dispatch of process_midi (linux/synth.c:236-258) for one event; the updates
 * are queued on the bank's stream ahead of the next block, without a host sync. */
int smx_bank_midi_event(smx_bank *b, const uint8_t *msg, size_t size);
/* The event loop of process_midi (linux/synth.c:246-258) for all events of a block at once:
 * msgs3 holds n_events consecutive 3-byte messages (process_midi ignores every other size, so
 * the caller leaves those out).  Same result as n_events calls of smx_bank_midi_event, applied
 * by one copy and one kernel on the bank's stream. */
int smx_bank_midi_events(smx_bank *b, const uint8_t *msgs3, size_t n_events);

/* synth_run over the bank (linux/synth.c:196-202).  vec: host float[n] or
 * NULL; bus: host int32[n] or NULL (the integer sum before the 2^-32 scale,
 * linux/synth.c:170-179).  Synchronous. */
int smx_bank_run(smx_bank *b, float *vec, int32_t *bus, int n);

/* Block mode of smx_bank_run.  SMX_BLOCK_SYNC (default) is the reference's behaviour:
 * the call returns this block's samples.  SMX_BLOCK_PIPELINED launches block k and
 * returns block k-1 (silence on the first call), so a real-time thread never waits for
 * a kernel it has just launched: one block of added latency. */
#define SMX_BLOCK_SYNC      0
#define SMX_BLOCK_PIPELINED 1
int smx_bank_set_block_mode(smx_bank *b, int mode);

/* How long blocks of a big launch (more than 32 frames, >= 2^16 voices, >= 2^30 voice-samples per launch; see the end
 * of this comment for 17..32 frames)
 * find the 32-bit wraps of the phases -- the only non-linear part of sum_tick_saw
 * (linux/synth.c:172-179); every form gives the same bits.  STEPPING adds inc frame by frame and
 * counts the carry-outs.  EVENTS locates each wrap directly (first at floor(~phase/inc), then every
 * floor((2^32-1)/inc) or one more frames): much less work for banks of mostly low voices, more for
 * banks of high ones (up to 3x the stepping form on a bank of only very high notes).  AUTO (default) keeps
 * a statistic of the increments on the device and picks per launch: events only while the largest increment
 * is below MIDI note ~110 and the mean is at most 2 wraps per voice and 64 frames.  The statistic is exact as
 * of the last long block and is kept conservative by every note event in between (a new increment or a
 * running sum above the bound selects stepping at once), so the event form never runs on a bank outside
 * the rule: on every bank the rule admits it was measured at most as slow as stepping -- a
 * real-time caller (the JACK process callback, linux/synth.c:277-282) gets at most the stepping form's time.
 * smx_bank_load(inc) applies the rule to the loaded increments (exact sum and maximum), so the first long block after
 * a load already runs the right form.  STEPPING / EVENTS pin a form.
 * Blocks of 17..32 frames of banks of 2^25 voices and more take the same two forms as one 32-frame chunk under AUTO
 * and EVENTS (a piano-range bank: 61 instead of 48 % of the HBM peak on 64 Mi voices); with STEPPING pinned they run
 * the direct form, like every shorter block. */
#define SMX_FORM_AUTO     0
#define SMX_FORM_STEPPING 1
#define SMX_FORM_EVENTS   2
int smx_bank_set_block_form(smx_bank *b, int form);
/* The form the next long block will run (SMX_FORM_STEPPING / SMX_FORM_EVENTS); synchronises. */
int smx_bank_next_block_form(smx_bank *b);

/* Asynchronous form: enqueue one block of n frames on the bank's stream and
 * leave the int32 bus in device memory (smx_bank_bus_dev).  No host sync.
 * (Some block shapes leave the last fold of their partial sums to the next block's launch; smx_bank_bus_dev,
 * _fetch, _sync and the all-reduce calls enqueue it when it is still owed, so the bus they hand out is complete.
 * Take the device pointer AFTER the run_async call it belongs to, through smx_bank_bus_dev.) */
int   smx_bank_run_async(smx_bank *b, int n);
void *smx_bank_bus_dev(smx_bank *b);     /* device int32[n] of the last block */
int   smx_bank_sync(smx_bank *b);
/* sum_tick_square (linux/synth.c:182-195) for n frames; vec host float[n].  With a communicator: the OR over all
 * ranks' voices (collective). */
int smx_bank_run_square(smx_bank *b, float *vec, int n);

/* Timing on the bank's own stream (HIP events; ms). */
int smx_bank_timer_start(smx_bank *b);
int smx_bank_timer_stop(smx_bank *b, float *ms);
/* Kernel-only time of the last smx_bank_run_async calls since timer_start is
 * the same interval when nothing else is enqueued on the stream. */

/* ---- multi-GPU: per-GPU mix, then one int32 sum over xGMI (RCCL) -------- */
/* One process per GPU; rank r's bank holds its contiguous shard of the voices (the voice loop
 * of linux/synth.c:172-179 is the only data-parallel axis and the mix its only exchange).
 * SPMD contract: once a bank has a communicator, every rank makes the SAME sequence of
 * smx_bank_run / _run_async / _allreduce_async / _fetch / _sync / _set_block_mode /
 * _set_comm_group calls with the same frame counts (like MPI collectives): the calls decide
 * when the queued sums are issued, and all ranks must issue the same collectives. */
#define SMX_UNIQUE_ID_BYTES 128
int smx_comm_unique_id(uint8_t id[SMX_UNIQUE_ID_BYTES]);   /* rank 0; hand the bytes to the other ranks */
int smx_bank_comm_init(smx_bank *b, int rank, int nranks,
                       const uint8_t id[SMX_UNIQUE_ID_BYTES]);
/* Note routing over the shards ("note-on routing is host-side").  smx_bank_shard declares this bank the
 * shard [first_voice, first_voice + n) of a global bank of total_voices voices (equal shards of a multiple of 64
 * voices: first_voice = rank x n, total_voices = ranks x n).  From then on smx_bank_note_on / _note_off /
 * _midi_event(s) run the reference's allocator (linux/synth.c:145-165: first free voice, steal voice 0 when full,
 * stray note-off silences voice 0) over the WHOLE global bank on every rank -- every rank is given the same events
 * in the same order, so every rank takes the same decisions -- and each rank applies to its device arrays only what
 * falls into its own range.  smx_bank_load(inc) on a sharded bank is collective: the ranks exchange which of their
 * voices are free.  Call it after smx_bank_comm_init, on every rank. */
int smx_bank_shard(smx_bank *b, uint32_t first_voice, uint32_t total_voices);
/* Ranks the communicator really spans (ncclCommCount); 0 without a communicator. */
int smx_bank_comm_ranks(const smx_bank *b);
/* All-reduce (sum, int32) of the last block's bus across ranks, in place in device memory, on
 * a second stream ordered after the block's kernel.  Requests are queued and issued as ONE
 * all-reduce per group of consecutive blocks (the bus ring is contiguous; default 8 blocks:
 * fewer, larger collectives -- xGMI is latency-bound at these sizes), or at once when a result
 * is needed (smx_bank_fetch / smx_bank_sync / smx_bank_run).  With a communicator
 * smx_bank_run returns the sum over all ranks in both block modes; in SMX_BLOCK_PIPELINED the
 * reduce and the copy to the host run on the second stream behind the kernel, so the caller
 * waits for neither; in SMX_BLOCK_SYNC the block's sum is issued on the bank's own stream right behind
 * its kernel (the caller waits for it anyway: no hop to the second stream and back). */
int smx_bank_allreduce_async(smx_bank *b, int n);
/* Blocks per collective, 1..16; default 8 (1: every block's sum is issued at once).  A group hides the
 * collective's latency L behind `blocks` kernels: it pays when blocks x (kernel time) >= L. */
int smx_bank_set_comm_group(smx_bank *b, int blocks);
/* Counters: collectives issued so far and the block sums they carried. */
int smx_bank_comm_stats(const smx_bank *b, unsigned long long *collectives, unsigned long long *block_sums);
/* Measurement aid, no reference counterpart: the cost of ONE bus sum of n_words int32 (1..2^20) on this bank's
 * communicator with no kernel in between, averaged over `reps` all-reduces.  *us_sync: host time per all-reduce when
 * each is waited for (what a synchronous smx_bank_run pays on top of its kernel); *us_queued: per all-reduce when
 * `reps` are queued back to back and waited for once (what a group of kernels has to hide in throughput mode).
 * Collective: every rank calls it with the same arguments.  Issues and waits for everything queued before. */
int smx_bank_comm_probe(smx_bank *b, uint32_t n_words, uint32_t reps, float *us_sync, float *us_queued);
/* The (reduced) bus of the last block on the host, converted as linux/synth.c:180.  Blocks of up to 4096 frames
 * come back without a copy engine: the stream's last (one-workgroup) kernel writes the sums to coherent pinned host
 * memory and then a sequence number, and this call polls that number -- bounded: if the GPU has not answered within
 * SMX_PUBLISH_TIMEOUT_MS (environment, default 10000) the call returns SMX_E_NOGPU instead of spinning forever.
 * SMX_NO_PUBLISH=1 selects the older hipMemcpyAsync + hipStreamSynchronize path (same bits).  The drop-in
 * synth_run(struct synth *) uses the same hand-over from a single launch (its 64 voices are kernel arguments). */
int smx_bank_fetch(smx_bank *b, float *vec, int32_t *bus, int n);

/* ======================================================================== */
/* 3. Carry-out PDM bank: stm32f103/mod_pdm.c:198-286                       */
/* ======================================================================== */
typedef struct smx_pdm smx_pdm;
smx_pdm *smx_pdm_create(uint32_t n_channels, int device);
void     smx_pdm_destroy(smx_pdm *p);
/* pdm_init (mod_pdm.c:296-326): every setpoint 0x40000000, channel 0
 * 2000000000, accumulators 0. */
int smx_pdm_init(smx_pdm *p);
/* mod_synth.c:104-111 (SETPOINT): -2 if chan out of range. */
int smx_pdm_set_setpoint(smx_pdm *p, uint32_t chan, uint32_t val);
uint32_t pdm_safe_setpoint(uint32_t setpoint);       /* mod_pdm.c:101-107 */
/* (The accumulators are kept lazily on the device -- accu0 + ticks * setpoint + the sum of the ticks' dither words,
 * mod 2^32: the modulator is linear between pulses -- so a tick launch only reads; smx_pdm_load, smx_pdm_read(accu)
 * and smx_pdm_set_setpoint first bring the stored accumulators up to date, one pass over the bank.) */
int smx_pdm_load(smx_pdm *p, const uint32_t *setpoint, const uint32_t *accu);
int smx_pdm_read(smx_pdm *p, uint32_t *setpoint, uint32_t *accu);
/* n_ticks of the PDM ISR body (mod_pdm.c:259-264).  dither: host
 * uint32[n_ticks] or NULL (= 0); the value is shared by all channels within
 * a tick, as in the reference, and is added unmasked (the reference masks
 * its generator with 0x0FFFFFFF, mod_pdm.c:261; the generator itself is
 * uc_tools' and is not part of this library).  bits: host
 * uint32[n_ticks * ceil(n/32)] or NULL: tick-major pulse words, channel c in
 * bit (c&31) of word (c>>5).  Synchronous. */
int smx_pdm_tick_n(smx_pdm *p, uint32_t n_ticks, const uint32_t *dither,
                   uint32_t *bits);
/* The same ticks with channel-stream output (what a per-channel decimator or DAC model
 * reads; the intent of linux/test_pdm.c:1-10): streams[k * n + c], bit j = the pulse of
 * channel c at tick 32k+j.  n_ticks must be a multiple of 32.  streams: host
 * uint32[(n_ticks/32) * n] or NULL.  This layout needs no transpose on the GPU (2 instead
 * of 3 vector ops per channel-tick). */
int smx_pdm_tick_n_streams(smx_pdm *p, uint32_t n_ticks, const uint32_t *dither, uint32_t *streams);
int smx_pdm_tick_n_streams_async(smx_pdm *p, uint32_t n_ticks, int with_dither);
/* Asynchronous, output stays in HBM (smx_pdm_bits_dev). */
int   smx_pdm_tick_n_async(smx_pdm *p, uint32_t n_ticks, int with_dither);
void *smx_pdm_bits_dev(smx_pdm *p);
void *smx_pdm_dither_dev(smx_pdm *p, uint32_t n_ticks);  /* device uint32[n_ticks] */
int   smx_pdm_sync(smx_pdm *p);
int   smx_pdm_timer_start(smx_pdm *p);
int   smx_pdm_timer_stop(smx_pdm *p, float *ms);
/* The reference's GPIO BSRR word (mod_pdm.c:271-286) from one tick's pulse
 * word, for nb <= 12 channels on pins 4.. */
uint32_t smx_pdm_bsrr_word(uint32_t pulse_bits, uint32_t nb);

/* ======================================================================== */
/* 4. Poly voice bank (BASELINE config 4): saw -> 1-pole LPF -> ADSR -> stereo */
/*    BUILD-DEFINED EXTENSION: the reference has no filter/envelope (FIXME at  */
/*    linux/synth.c:150-152); defined in DESIGN.md §3.4 / oracle orc_poly_run. */
/* ======================================================================== */
typedef struct smx_poly smx_poly;
/* Host-side view of the per-voice arrays (each n_voices long). */
struct smx_poly_arrays {
    uint32_t *inc, *phase;           /* phasor as linux/synth.c:31-34; inc 0 == off   */
    float    *y, *a;                 /* filter state and coefficient (0..1)            */
    uint32_t *level, *stage;         /* envelope level, stage 0 idle 1 A 2 D 3 S 4 R   */
    uint32_t *gate;                  /* 0/1, sampled at the start of every block       */
    uint32_t *ar, *dr, *sl, *rr;     /* per-sample rates and sustain level (u32)       */
    uint32_t *pan;                   /* pan_l | pan_r << 16, each 0..256               */
};
smx_poly *smx_poly_create(uint32_t n_voices, int device);
void      smx_poly_destroy(smx_poly *p);
/* NULL members are skipped. */
int smx_poly_load(smx_poly *p, const struct smx_poly_arrays *a);
int smx_poly_read(smx_poly *p, const struct smx_poly_arrays *a);
/* n frames (any n; run in launches of <= 64).  vec_lr: host float[2n] interleaved L,R
 * (bus * 2^-32 as linux/synth.c:180) or NULL; bus_lr: host int32[2n] or NULL. */
int smx_poly_run(smx_poly *p, float *vec_lr, int32_t *bus_lr, int n);
int smx_poly_run_async(smx_poly *p, int n);      /* n <= 64, bus stays on the device */
int smx_poly_sync(smx_poly *p);
int smx_poly_timer_start(smx_poly *p);
int smx_poly_timer_stop(smx_poly *p, float *ms);

/* ======================================================================== */
/* 5. Noise-shaped PWM bank: stm32f103/mod_pdm_pwm.c:76-160, pdm.h:10-77,    */
/*    mod_controlrate.c:28-57                                                */
/* ======================================================================== */
typedef struct smx_pwm smx_pwm;
/* order = number of integrators of the noise shaper, 1..4 (pdm1..pdm4_update,
 * pdm.h:13,32,48,67); the firmware uses 2 (PDM_ORDER, mod_pdm_pwm.c:85).
 * Defaults: CONTROL_DIV_LOG 12 (mod_pdm_pwm.c:76), out_shift 32-PDM_DIV_LOG = 24
 * (mod_pdm_pwm.c:115, mod_synth.c:29). */
smx_pwm *smx_pwm_create(uint32_t n_channels, int order, int device);
void     smx_pwm_destroy(smx_pwm *p);
int      smx_pwm_config(smx_pwm *p, uint32_t control_div_log, uint32_t out_shift);
/* pdm_init, mod_pdm_pwm.c:147-160: setpoints 0x40000000, channel 0 2000000000,
 * lines and integrators 0, control_div_count 0. */
int smx_pwm_init(smx_pwm *p);
/* SETPOINT, mod_synth.c:104-111: -2 if chan is out of range. */
int smx_pwm_set_setpoint(smx_pwm *p, uint32_t chan, uint32_t val);
/* struct channel (mod_pdm_pwm.c:89-93) as host arrays of n_channels; velocities
 * are int32 stored in uint32.  NULL members are skipped. */
struct smx_pwm_arrays {
    uint32_t *setpoint;
    uint32_t *pos0, *vel0;           /* line[0] */
    uint32_t *pos1, *vel1;           /* line[1] */
    uint32_t *s[4];                  /* pdm integrators s1..s4 (first `order` used) */
};
int smx_pwm_load(smx_pwm *p, const struct smx_pwm_arrays *a);
int smx_pwm_read(smx_pwm *p, const struct smx_pwm_arrays *a);
int smx_pwm_set_div_count(smx_pwm *p, uint32_t control_div_count);
uint32_t smx_pwm_div_count(const smx_pwm *p);
/* The control-rate ISR's beat divider, `struct controlrate` (mod_controlrate.c:19-26, 52-55): every control
 * tick (a sample tick that starts with control_div_count == 0) runs `if (isr_count % 1024 == 0) beat_pulse++;
 * isr_count++`; controlrate_poll (:64-72) handles one pending beat per call of the main loop. */
#define SMX_CONTROLRATE_BEAT_DIV 1024
int smx_pwm_controlrate(const smx_pwm *p, uint32_t *isr_count, uint32_t *beat_pulse, uint32_t *beat_handled);
int smx_pwm_controlrate_poll(smx_pwm *p);          /* 1: a beat was handled, 0: none pending */
/* n_ticks of the ISR (mod_pdm_pwm.c:123-143).  dither: host uint32[n_ticks] or
 * NULL (= 0), shared by all channels of a tick (the firmware masks its
 * generator with 0x3FF, mod_pdm_pwm.c:127).  duty: host uint8[n_ticks *
 * n_channels] tick-major (low 8 bits of the shaper output) or NULL. */
int smx_pwm_tick_n(smx_pwm *p, uint32_t n_ticks, const uint32_t *dither, uint8_t *duty);
int   smx_pwm_tick_n_async(smx_pwm *p, uint32_t n_ticks, int with_dither);
void *smx_pwm_dither_dev(smx_pwm *p, uint32_t n_ticks);
int   smx_pwm_sync(smx_pwm *p);
int   smx_pwm_timer_start(smx_pwm *p);
int   smx_pwm_timer_stop(smx_pwm *p, float *ms);

/* ======================================================================== */
/* 6. Oscillator bank: stm32f103/mod_pdm.c:159-175 (pwm_update + hard sync), */
/*    mod_osc.c:47-103 (osc ISR, sub-osc), pmeas.h:10-108 (period measurement)*/
/* ======================================================================== */
typedef struct smx_osc smx_osc;
/* osc_init (mod_osc.c:82-103): log_max 26, everything else 0; pwm_phase 0,
 * pwm_speed 256*13 (mod_pdm.c:160-161). */
smx_osc *smx_osc_create(uint32_t n_oscillators, int device);
void     smx_osc_destroy(smx_osc *o);
/* MEASURE with one argument sets log_max (mod_synth.c:112-117). 1..31. */
int smx_osc_set_log_max(smx_osc *o, uint32_t log_max);
int smx_osc_load_pwm(smx_osc *o, const uint32_t *phase, const uint32_t *speed);
int smx_osc_read_pwm(smx_osc *o, uint32_t *phase, uint32_t *speed);
/* n_ticks of pwm_update (mod_pdm.c:167-175).  sync_bits: host
 * uint32[n_ticks * ceil(n/32)] or NULL; a set bit (oscillator c -> bit c&31 of
 * word c>>5 of row t) applies OSC_HARD_SYNC before tick t.  duty: host
 * uint8[n_ticks * n] tick-major or NULL. */
int smx_osc_tick_n(smx_osc *o, uint32_t n_ticks, const uint32_t *sync_bits, uint8_t *duty);
/* n_events slots of the osc ISR (mod_osc.c:47-74): oscillator c takes slot e with
 * cycle-counter timestamp cc[e*n + c] iff its valid bit is set (same layout;
 * NULL = every oscillator takes every slot): sub-osc toggle + pmeas_update. */
int smx_osc_events(smx_osc *o, uint32_t n_events, const uint32_t *cc, const uint32_t *valid_bits);
/* struct pmeas_state as host arrays of n (NULL members skipped). */
struct smx_pmeas_arrays {
    uint32_t *write;                 /* low bit points at the current measurement */
    uint32_t *avg0, *avg1;           /* meas[0..1].avg, (32-log_max) fractional bits */
    uint32_t *num0, *num1;           /* meas[0..1].num */
    uint32_t *num, *accu, *last_cc;  /* ISR-side state */
    uint32_t *sub;                   /* sub-oscillator output bit (GPIOB pin 10) */
};
int smx_osc_load_pmeas(smx_osc *o, const struct smx_pmeas_arrays *a);
int smx_osc_read_pmeas(smx_osc *o, const struct smx_pmeas_arrays *a);

/* ---- clock bank: linux/clock.c:58-62, 106-120 ----------------------------- */
/* N integer-divider square clocks.  Initial state as the reference: phase 0,
 * polarity 1 (clock.c:61-62). */
typedef struct smx_clock smx_clock;
uint32_t smx_bpm_to_hperiod(uint32_t sample_rate, uint32_t bpm);    /* BPM_TO_HPERIOD, clock.c:58 */
smx_clock *smx_clock_create(uint32_t n_clocks, int device);
void smx_clock_destroy(smx_clock *c);
int smx_clock_load(smx_clock *c, const uint32_t *hperiod, const int32_t *phase, const uint32_t *pol);
int smx_clock_read(smx_clock *c, uint32_t *hperiod, int32_t *phase, uint32_t *pol);
/* n_frames of the generator loop (clock.c:106-120).  pol_bits: the square wave
 * (audio_out_buf[t] = clock_pol), tick_bits: the frames at which the reference sends
 * MIDI clock 0xF8 (polarity turned 1).  Both host uint32[n_frames * ceil(n/32)],
 * frame-major, clock c in bit c&31 of word c>>5; either may be NULL. */
int smx_clock_run(smx_clock *c, uint32_t n_frames, uint32_t *pol_bits, uint32_t *tick_bits);

/* ---- 6b. mod_pdm.c as ONE module: its timer ISR HW_TIM_ISR(TIM_PDM) (stm32f103/mod_pdm.c:177-194) runs, per tick,
 * pdm_update() (the carry-out channels), val = pwm_update() (the fixed-rate PWM channel, mod_pdm.c:166-175, hard-synced
 * by the oscillator ISR, mod_osc.c:60-62) and, every CONTROL_DIV = 256 ticks (mod_pdm.c:164-165, 184-192),
 * control_trigger().  smx_modpdm is a PDM bank of n_channels and an oscillator bank of n_osc ticking in lockstep
 * (one kernel each, side by side) plus that divider; the banks are created in the firmware's initial state (pdm_init:
 * mod_pdm.c:320-326; pwm_phase 0, pwm_speed 256*13) and are reached through the borrowed handles for
 * smx_pdm_load / _read / _set_setpoint and smx_osc_load_pwm / _read_pwm.  control_trigger() pends control_update
 * (mod_controlrate.c:52-55), whose beat divider is reported like smx_pwm_controlrate's. */
typedef struct smx_modpdm smx_modpdm;
smx_modpdm *smx_modpdm_create(uint32_t n_channels, uint32_t n_osc, int device);
void        smx_modpdm_destroy(smx_modpdm *m);
smx_pdm    *smx_modpdm_pdm(smx_modpdm *m);          /* borrowed handles */
smx_osc    *smx_modpdm_osc(smx_modpdm *m);
/* n_ticks of the ISR.  dither: host uint32[n_ticks] or NULL (as smx_pdm_tick_n); sync_bits: OSC_HARD_SYNC bit matrix
 * or NULL (as smx_osc_tick_n); bits: host uint32[n_ticks][ceil(n_channels/32)] or NULL; duty: host
 * uint8[n_ticks][n_osc] or NULL; *control_triggers: control_trigger() calls of this run. */
int smx_modpdm_tick_n(smx_modpdm *m, uint32_t n_ticks, const uint32_t *dither, const uint32_t *sync_bits,
                      uint32_t *bits, uint8_t *duty, uint32_t *control_triggers);
uint32_t smx_modpdm_control_div_count(const smx_modpdm *m);
int smx_modpdm_controlrate(const smx_modpdm *m, uint32_t *isr_count, uint32_t *beat_pulse, uint32_t *beat_handled);

/* ======================================================================== */
/* 7. Firmware control surface, hosted: stm32f103/mod_synth.c:50-137 and the   */
/*    packet entry stm32f103/synth.c:27-42                                     */
/* ======================================================================== */
/* The firmware's `synth_init(struct cbuf*)` / `synth_handle_tag_u32` clash by
 * name with the Linux `synth_init(struct synth*)`; they live in different
 * programs in the reference.  Here the firmware side is namespaced smx_fw_*.
 * The wire layout of TAG_U32 and the parameter-table setter are uc_tools'
 * (tag_u32.h, parameter.h: not in the reference tree); the layout used here is the
 * one the reference's own example shows (mod_synth.c:98:
 * <<16#FFF50002:32, 100:32, 1:32>> = tag:16, nb_from:8, nb_args:8, from[], args[]
 * big-endian words, then payload bytes). */
#define SMX_TAG_U32 0xFFF5u              /* erl/jack_client.erl:30 */
typedef struct smx_fw smx_fw;
/* firmware synth_init (mod_synth.c:63-86): pdm_init + pdm_start (noise-shaped
 * PWM bank, the module mod_synth.c:38 compiles), osc_init, controlrate_init.
 * The reference has 3 channels and 1 oscillator. */
smx_fw *smx_fw_create(uint32_t n_channels, uint32_t n_oscillators, int device);
void    smx_fw_destroy(smx_fw *f);
smx_pwm *smx_fw_pwm(smx_fw *f);          /* borrowed handles */
smx_osc *smx_fw_osc(smx_fw *f);
/* synth_handle_tag_u32 (mod_synth.c:89-137), same return codes:
 * 100 MODE on/off; 101 SETPOINT chan val (-2 bad chan); 102 MEASURE [log_max]
 * with an optional continuation payload (-3 if more than 2 args); anything else
 * = parameter table {id, value} (id 0 = osc_setpoint, mod_synth.c:50-56). */
int smx_fw_handle_tag_u32(smx_fw *f, const uint32_t *args, uint32_t nb_args,
                          const uint8_t *bytes, uint32_t nb_bytes);
/* handle_tag (stm32f103/synth.c:27-42) for one {packet,4} payload (without the
 * 4-byte length): TAG_U32 is dispatched, other tags are logged and ignored. */
int smx_fw_handle_packet(smx_fw *f, const uint8_t *buf, uint32_t len);
int smx_fw_running(const smx_fw *f);     /* MODE state (pdm_start/pdm_stop) */
uint32_t smx_fw_parameter(const smx_fw *f, uint32_t id);
/* The PDM timer ISR n times (mod_pdm_pwm.c:123-143); returns the number of ticks
 * executed: n_ticks when running, 0 when stopped; negative on error. */
int smx_fw_tick_n(smx_fw *f, uint32_t n_ticks, const uint32_t *dither, uint8_t *duty);
/* osc_poll (mod_osc.c:77-80, pmeas.h:30-61) for one oscillator: returns 1 and the
 * newly published average when read != write, and hands back the oldest queued
 * MEASURE continuation (cont_len 0 if none); 0 when nothing is new. */
int smx_fw_poll(smx_fw *f, uint32_t osc, uint32_t *avg, uint32_t *num,
                uint8_t *cont, uint32_t cont_cap, uint32_t *cont_len);

/* ======================================================================== */
/* 8. cproc dataflow bank: generic/cproc.h:72-155, mod_bpmodular.c:36-45,72-78 */
/* ======================================================================== */
/* N instances of one static chain of processors (PROC_COND bindings in allocation
 * order).  All atoms are uint32 words (cproc.h:128); state starts at zero
 * (cproc.h:65-66,73). */
#define SMX_PROC_ACC   1u                /* out += in              (cproc.h:134-144) */
#define SMX_PROC_EDGE  2u                /* out = in != last; last = in (cproc.h:146-155) */
#define SMX_PROC_GPIN  3u                /* out = input word (hw_cproc_stm32f103.h:8-14, the GPIO pin read replaced by an external input word) */
#define SMX_PROC_GPOUT 4u                /* patcher only: sink (hw_cproc_stm32f103.h:16-22); its input is what smx_patch_tick returns */
#define SMX_CPROC_INPUT(k) (0x80000000u | (uint32_t)(k))   /* external input word k */
#define SMX_CPROC_MAX_NODES 32
struct smx_cproc_node {
    uint32_t proc;                       /* SMX_PROC_*                               */
    uint32_t in;                         /* SMX_CPROC_INPUT(k) or an earlier node     */
    uint32_t cond;                       /* subgraph mask: runs when (g & cond) != 0 */
};
typedef struct smx_cproc smx_cproc;
smx_cproc *smx_cproc_create(uint32_t n_instances, const struct smx_cproc_node *nodes,
                            uint32_t n_nodes, uint32_t n_inputs, int device);
void smx_cproc_destroy(smx_cproc *c);
/* n_ticks of cproc_update(input, g) (linux/test_cproc.c:13-17) for every instance.
 * input: host uint32[n_ticks][n_inputs][n_instances]; g: host uint32[n_ticks] or
 * NULL (synchronous graph: every node runs); out: host uint32[n_ticks][n_instances]
 * = `out` of node out_node after each tick (cproc_output), or NULL. */
int smx_cproc_tick_n(smx_cproc *c, uint32_t n_ticks, const uint32_t *input, const uint32_t *g,
                     uint32_t out_node, uint32_t *out);
/* state[node][2][n_instances] = {out, last} */
int smx_cproc_read_state(smx_cproc *c, uint32_t *state);
int smx_cproc_load_state(smx_cproc *c, const uint32_t *state);

/* Dynamic patcher: instances allocated one by one and connected by node index, run in allocation order
 * (stm32f103/mod_bpmodular.c:36-45 struct proc/inst, :72-78 tick, :84-113 apply, :218-228 reset/tick,
 * :153-190 state get/set).  n_instances copies of the network, one per lane.  Classes: SMX_PROC_ACC,
 * SMX_PROC_EDGE (cproc.h:134-155), SMX_PROC_GPIN (no input; config = the input word it reads each tick, in
 * place of the GPIO pin) and SMX_PROC_GPOUT (one input, no state; a sink).  (The reference's own class
 * table, mod_bpmodular_procs.c, is generated and not in its tree; these four are all the DEF_PROCs it holds.)
 * The reference answers "bad_ref" / "bad_node" / "alloc_fail" (its bump allocator holds 1024 words, :27;
 * an instance takes 1 + state fields + inputs of them, :88); here the codes below, and at most
 * SMX_CPROC_MAX_NODES instances. */
#define SMX_PATCH_BAD_REF    (-11)
#define SMX_PATCH_BAD_NODE   (-12)
#define SMX_PATCH_ALLOC_FAIL (-13)
typedef struct smx_patch smx_patch;
smx_patch *smx_patch_create(uint32_t n_instances, uint32_t n_inputs, int device);
void smx_patch_destroy(smx_patch *p);
/* class/<cls>/apply: -> node index (>= 0) or one of the codes above; in[]: existing node indices (n_in of
 * them: acc 1, edge 1, gpin 0, gpout 1); config: the input word of a gpin (< n_inputs), else ignored. */
int smx_patch_apply(smx_patch *p, uint32_t cls, const uint32_t *in, uint32_t n_in, uint32_t config);
uint32_t smx_patch_count(const smx_patch *p);
int smx_patch_reset(smx_patch *p);                      /* patch/reset */
/* patch/tick, n_ticks times.  input: host uint32[n_ticks][n_inputs][n_instances] (what the gpins read) or
 * NULL when the patch has no gpin; out: host uint32[n_ticks][n_instances] = what gpout node `gpout` wrote
 * at each tick, or NULL. */
int smx_patch_tick(smx_patch *p, uint32_t n_ticks, const uint32_t *input, uint32_t gpout, uint32_t *out);
/* inst/<node>/state/<field>/get|set for every copy of the network: vals host uint32[n_instances];
 * field 0 is `out` (acc: out; edge: out, last; gpin: out; gpout: none). */
int smx_patch_state_get(smx_patch *p, uint32_t node, uint32_t field, uint32_t *vals);
int smx_patch_state_set(smx_patch *p, uint32_t node, uint32_t field, const uint32_t *vals);

#ifdef __cplusplus
}
#endif
#endif
