// smx_common.h -- shared host-side plumbing for libsynth_mi355x.so
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "../../include/synth_mi355x.h"

namespace smx {

void set_error(const char *fmt, ...);

// Non-void ABI calls: record the message and return SMX_E_NOGPU.
#define SMX_HIP(expr)                                                          \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            smx::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,       \
                           hipGetErrorString(e_));                             \
            return SMX_E_NOGPU;                                                \
        }                                                                      \
    } while (0)

// void reference-named calls: the reference's ASSERT convention
// (linux/erl_tools_system.h:15,24-27): log and exit(1).
#define SMX_ASSERT_OK(rv, what)                                                \
    do {                                                                       \
        if ((rv) != SMX_OK) {                                                  \
            fprintf(stderr, "libsynth_mi355x: %s failed (%d): %s\n", what,     \
                    (int)(rv), smx_last_error());                              \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static inline uint32_t round_up(uint32_t x, uint32_t m) { return (x + m - 1) / m * m; }

// ---- kernel launchers (defined in the .hip files) --------------------------
// Saw bank (saw_bank.hip).  n_pad is a multiple of 1024.  HBM holds inc[] and state0[];
// tbase = frames elapsed since state0 was valid (phase = state0 + tbase*inc): a launch only
// reads.  d_bus[0..nframes) must be zero on entry; the launch zeroes d_bus_next[0..nframes)
// for its successor.  d_scratch: saw_scratch_bytes(max frames) bytes of ZEROED device memory
// (partial-sum slots of the carry formulation; each launch leaves them zero again), or NULL
// to force the direct formulation.
size_t saw_scratch_bytes(uint32_t max_frames);
// The slot variants of the direct formulation (>= 2^20 voices, 5+ frames, below the carry crossover) end in a
// small fold of their slots into the bus.  With a SawPending that fold is DEFERRED: the next slot launch does it
// in its first workgroups (the scratch holds two slot regions, used alternately), anything else that needs the
// bus -- a fetch, an all-reduce, a launch of another form -- calls launch_saw_flush first.  The stream then
// carries one kernel per block instead of two.  Until the fold has run, `bus` holds no valid sums and
// `bus_next` is not yet zeroed.
// Where a block's LAST kernel may hand its bus to the host itself (round 3): device pointers of coherent pinned host
// memory (see launch_saw_publish); hflag == nullptr: nobody asked.  Only kernels that finish a block of <= 64 frames in
// ONE workgroup can do it (the fold of the direct form's slots, the finalize of the carry formulations).
struct SawPublish {
    int32_t *hbus = nullptr;
    uint32_t *hflag = nullptr;
    uint32_t seq = 0;
};
struct SawPending {
    void *partial = nullptr;            // slots of the block whose fold is owed (nullptr: nothing owed)
    int32_t *bus = nullptr, *bus_next = nullptr;
    uint32_t nframes = 0;
    uint32_t region = 0;                // slot region the NEXT slot launch fills
    size_t region_stride = 0;           // bytes between the two regions: saw_scratch_region_bytes(cap the scratch has)
};
size_t saw_scratch_region_bytes(uint32_t max_frames);
// pub: the fold, if one is owed and covers <= 64 frames, also publishes the bus (returns true through *published).
int launch_saw_flush(SawPending *pend, hipStream_t stream, const SawPublish *pub = nullptr, bool *published = nullptr);
// d_hbus / d_hflag: DEVICE pointers of coherent pinned host memory.  The kernel writes n bus words there, then seq to
// *d_hflag (system-scope release): the host polls the flag instead of waiting for a copy and a stream.
int launch_saw_publish(const int32_t *d_bus, int32_t *d_hbus, uint32_t *d_hflag, uint32_t n, uint32_t seq, hipStream_t stream);
// struct synth's 64 voices (linux/synth.c:31-40) for one block of n <= 1024 frames in ONE launch: the voices are kernel
// arguments, the bus goes straight to pinned host memory (published as above).  Nothing in HBM is read or written.
int launch_saw_dropin(const uint32_t inc[64], const uint32_t state[64], int32_t *d_hbus, uint32_t *d_hflag, uint32_t n,
                      uint32_t seq, hipStream_t stream);
// long_block_form: SMX_FORM_AUTO / SMX_FORM_STEPPING / SMX_FORM_EVENTS (include/synth_mi355x.h)
// host_flag: two pinned (device-visible) words that receive, after every long block, the form the device
// would pick next (0 stepping, 1 events) and host_tag, the caller's own number of this block; or NULL.
int launch_saw_bank(const uint32_t *d_inc, const uint32_t *d_state0, int32_t *d_bus,
                    int32_t *d_bus_next, uint32_t n_pad, uint32_t nframes, uint32_t tbase,
                    void *d_scratch, int long_block_form, uint32_t *host_flag, uint32_t host_tag,
                    hipStream_t stream, SawPending *pend = nullptr,       // pend == nullptr: every launch folds its own slots
                    const SawPublish *pub = nullptr, bool *published = nullptr);   // a finalize that ends the block may publish it
// leading bytes of the scratch area that hold the formulation flag (zero: stepping form) and the bank's sum of
// increments; after clearing them (new increments) launch_saw_sum_inc recomputes the sum
size_t saw_scratch_header_bytes();
int launch_saw_sum_inc(const uint32_t *d_inc, uint32_t n_pad, void *d_scratch, hipStream_t stream);
int launch_square_bank(const uint32_t *d_inc, const uint32_t *d_state0, uint32_t *d_or_bus,
                       uint32_t n_pad, uint32_t nframes, uint32_t tbase, hipStream_t stream);
// one voice's increment changes at elapsed time tbase; state0 += tbase*inc for all voices
// (d_scratch: the bank's scratch area or NULL -- its header's form pick is kept conservative under note events)
int launch_saw_rebase(uint32_t *d_inc, uint32_t *d_state0, uint32_t voice, uint32_t new_inc,
                      uint32_t tbase, void *d_scratch, uint32_t n_pad, hipStream_t stream);
// npairs (voice, final increment) pairs with distinct voices, in device memory
int launch_saw_rebase_batch(uint32_t *d_inc, uint32_t *d_state0, const uint32_t *d_pairs, uint32_t npairs,
                            uint32_t tbase, void *d_scratch, uint32_t n_pad, hipStream_t stream);
int launch_saw_materialize(const uint32_t *d_inc, uint32_t *d_state0, uint32_t n_pad, uint32_t tbase,
                           hipStream_t stream);
// Carry-out PDM bank (pdm_bank.hip).  n_pad multiple of 1024; d_bits rows are n_pad/8 bytes.
// The accumulators are lazy: d_accu holds accu0[] as of `elapsed` ticks ago, *d_dsum_in the sum of the dither words of
// those ticks; a launch reads both and writes nothing back.  With dither it leaves *d_dsum_in + (its own dither
// words' sum) in *d_dsum_out (the other word of a pair: late workgroups still read the old one).
int launch_pdm_bank(const uint32_t *d_setpoint, const uint32_t *d_accu,
                    const uint32_t *d_dither /*nullable*/, uint32_t *d_bits,
                    uint32_t n_pad, uint32_t n, uint32_t nticks, uint32_t elapsed,
                    const uint32_t *d_dsum_in, uint32_t *d_dsum_out, hipStream_t stream);

// Same ticks, channel-stream output: d_streams[nticks/32][n_pad] (bit j = tick 32k+j).
int launch_pdm_streams(const uint32_t *d_setpoint, const uint32_t *d_accu, const uint32_t *d_dither,
                       uint32_t *d_streams, uint32_t n_pad, uint32_t nticks, uint32_t elapsed,
                       const uint32_t *d_dsum_in, uint32_t *d_dsum_out, hipStream_t stream);
// accu0 += elapsed * setpoint + *d_dsum_in for every channel (the caller then resets elapsed and the dither sums)
int launch_pdm_materialize(const uint32_t *d_setpoint, uint32_t *d_accu, uint32_t n_pad, uint32_t elapsed,
                           const uint32_t *d_dsum_in, hipStream_t stream);

// Poly voice bank (poly_bank.hip): device SoA arrays, n_pad entries each.
struct PolyArrays {
    uint32_t *inc, *phase;
    float *y, *a;
    uint32_t *level, *stage, *gate, *ar, *dr, *sl, *rr, *pan;
};
// The fold a launch left to its successor (poly_bank.hip): the copies it filled and the bus they belong to.
struct PolyPending {
    int32_t *slots = nullptr;           // nullptr: nothing owed
    int32_t *bus = nullptr;
    uint32_t region = 0;                // which of the two regions of copies the NEXT launch fills
};
// d_slots: poly_scratch_bytes() of zeroed device memory = two regions of copies (kept zero between uses);
// d_bus_lr: poly_bus_bytes(); rows [0, 2*nframes) receive the block's stereo sums.
// pend == nullptr: the launch folds its own copies (a second kernel) and d_bus_lr is complete when it has run.
// pend != nullptr: the fold is deferred -- the next launch_poly_bank with the same pend does it (pass the OTHER bus
// buffer there), or launch_poly_flush.
size_t poly_scratch_bytes();
size_t poly_bus_bytes();
int launch_poly_flush(PolyPending *pend, hipStream_t stream);
int launch_poly_bank(const PolyArrays &p, int32_t *d_bus_lr, int32_t *d_slots, uint32_t n_pad,
                     uint32_t nframes, hipStream_t stream, PolyPending *pend = nullptr);

// Noise-shaped PWM bank (pwm_bank.hip): device SoA arrays, n_pad entries each.
struct PwmArrays {
    uint32_t *setpoint, *pos0, *vel0, *pos1, *vel1;
    uint32_t *s[4];                     // integrators s1..sORDER
};
int launch_pwm_bank(const PwmArrays &p, int order, const uint32_t *d_dither, uint8_t *d_duty,
                    uint32_t n_pad, uint32_t nticks, uint32_t div_count, uint32_t div_log,
                    uint32_t out_shift, hipStream_t stream);

// Oscillator banks (osc_bank.hip).
struct PmeasArrays {                    // struct pmeas_state (pmeas.h:16-28) + sub-osc bit, SoA
    uint32_t *write, *avg0, *avg1, *num0, *num1, *num, *accu, *last_cc, *sub;
};
int launch_pwmosc(uint32_t *d_phase, const uint32_t *d_speed, const uint32_t *d_sync_bits,
                  uint8_t *d_duty, uint32_t n_pad, uint32_t nticks, hipStream_t stream);
int launch_osc_events(const PmeasArrays &p, const uint32_t *d_cc, const uint32_t *d_valid_bits,
                      uint32_t n, uint32_t nevents, uint32_t log_max, hipStream_t stream);

int launch_clock(const uint32_t *d_hperiod, uint32_t *d_phase, uint32_t *d_pol, uint32_t *d_pol_bits,
                 uint32_t *d_tick_bits, uint32_t n_pad, uint32_t n, uint32_t nframes, hipStream_t stream);

// cproc dataflow bank (cproc_bank.hip)
#define SMX_CPROC_MAX_NODES 32
struct CprocNode { uint32_t proc, in, cond; };
struct CprocProgram {
    uint32_t n_nodes, n_inputs;
    CprocNode nodes[SMX_CPROC_MAX_NODES];
};
int launch_cproc(const CprocProgram &prog, uint32_t *d_state, const uint32_t *d_input,
                 const uint32_t *d_g, uint32_t *d_out, uint32_t n_pad, uint32_t nticks,
                 uint32_t out_node, hipStream_t stream);

}  // namespace smx
