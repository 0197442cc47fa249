// saw_bank.hip -- N-voice phase-accumulator saw bank for gfx950 (MI355X).
//
// Replaces the sample-outer / voice-inner loops of linux/synth.c:169-202
// (sum_tick_saw, synth_run).  Per voice and sample, exactly as the reference:
//     p = (int)state;  sum += p >> 4;  state += inc;        (inc == 0: skip)
// with `sum` a wrapping 32-bit integer (the reference's `int sum` overflows
// for >= 16 loud voices and wraps in practice), which makes the mix
// associative: any reduction order gives the same bits.
//
// Mapping: one lane per voice (VW voices per lane per trip, struct-of-arrays
// inc[]/state[] so a wave's load is one contiguous 256 B / 1 KiB segment).
// A lane keeps its voices' phase in registers for the whole block of frames
// and TC per-frame partial sums in VGPRs; HBM is touched once per voice per
// block (8 B read, nothing written).  The block's partial sums go through an LDS
// [TC][64+1] matrix (conflict-free ds_add, padded rows), are folded by a
// 4-lane shuffle and leave the CU as one integer atomic per frame.
//
// The phasor is linear, state(t) = state0 + t*inc (mod 2^32), and the kernels use that twice:
//  * Time is split into chunks of 64 frames on blockIdx.y (state at chunk start in closed
//    form), so chunks are independent and small banks still fill the chip.
//  * The advanced phase is never written back.  HBM keeps state0[] and the bank keeps one
//    counter T of frames elapsed since state0 was valid; a launch reads inc[] and state0[]
//    (8 B per voice) and starts every voice at state0 + T*inc.  A block therefore moves 8
//    instead of 12 bytes per voice and is a pure read stream.  The phase is materialised
//    (state0 += T*inc, T = 0) only when the host reads or reloads the bank, and a note
//    event rebases one voice: state0 += T*(inc_old - inc_new) (saw_rebase_kernel), which
//    keeps state0 + T*inc continuous -- the reference's "note_on does not reset the phase".
//
// Long blocks of big banks (> 32 frames, >= 2^30 voice-samples; from 17 frames on >= 2^25 voices) take a second formulation
// (saw_bank_carry_kernel) that needs 1.5 instead of 3 vector ops per voice-sample:
// with u = state ^ 0x80000000 (offset binary) the arithmetic shift becomes a logical
// one, (int)state >> 4 == (u >> 4) - 2^27, and because every term is a floor,
//     sum_v (u_v >> 4) == (sum_v u_v - sum_v (u_v & 15)) >> 4        exactly.
// sum_v u_v(t) == U0 + t*I - 2^32 * W(t) with U0 = sum u_v(t0), I = sum inc_v and
// W(t) = number of 32-bit wraps of all phases before frame t; the low nibbles
// (u_v(t) & 15) == ((u_v(t0) & 15) + t*(inc_v & 15)) & 15 depend on 8 bits per voice,
// so their sum comes from a 256-bin histogram.  Per voice-sample only the phase add
// and the count of its carry-out remain: one 64-bit add of {wraps so far, phase} += inc
// per voice (the carry lands in the high word) and two v_add3 per four voices that add
// the cumulative counts into the frame's counter (carry_step4_wide; carry_step4 is the
// older loop over carry masks).  Everything is reduced in integers, so the result is the same
// bits as the reference loop; a small second kernel combines the per-workgroup
// partial sums into the int32 bus.
//
// W(t) is the ONLY per-sample non-linearity, and it can also be had without stepping: the wraps
// of a voice lie at frame floor(~u/inc) and then every floor((2^32-1)/inc) or one more frames.
// The EVENTS form of the carry kernel (one chunk of 32, 64 or 128 frames: blocks of 17..32, 33..64, 65..128 frames)
// and saw_bank_event_long_kernel (256-frame chunks for launches of 129 frames and more, 1024-frame chunks from 1024)
// locate them and add them to a histogram -- work per
// WRAP instead of per sample, 1.1 wraps per voice per 64 frames on a piano-range bank.  Which
// form runs is decided per launch ON THE DEVICE from the bank's own increment statistics (both
// are queued, one returns at once); both are exact on any bank.
#include "smx_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Streaming accesses: a bank larger than the caches is touched exactly once per
// launch, so its lines are loaded non-temporally (measured +10 % HBM read rate).
template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
// acc += p0 + p1 + p2 + p3.  Left to itself the compiler writes two v_add3_u32; PLAIN spells it as four
// two-operand v_add_u32 (pair sums first, so the chain on the accumulator stays two adds long).  Which one is faster
// depends on how many waves a SIMD holds (same-box A/B of the whole kernel, tools/ab_add3.sh): with 8 waves per SIMD
// -- banks of 2^24 voices and more -- the launch is bound by issue THROUGHPUT and the plain adds win, because
// v_add_u32 issues in 2.9 cycles and v_add3_u32 in 4.7 (64 Mi voices: 16 frames 97.2 -> 94.5 us, 32 frames 156 ->
// 140, 64 frames 289 -> 241); with the 4 waves per SIMD that a 2^20..2^22-voice bank leaves, a wave's own issue rate
// bounds the launch and the shorter instruction stream wins (1 Mi voices x 64 frames: 9.5 us with v_add3, 10.3 with
// plain adds).  The kernels pass their NT flag (set from 2^24 voices up) as PLAIN.
template <bool PLAIN>
__device__ __forceinline__ void acc_add4(int32_t &acc, int32_t p0, int32_t p1, int32_t p2, int32_t p3)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (PLAIN) {
        int32_t t0, t1;
        asm("v_add_u32 %0, %1, %2" : "=v"(t0) : "v"(p0), "v"(p1));
        asm("v_add_u32 %0, %1, %2" : "=v"(t1) : "v"(p2), "v"(p3));
        asm("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(t0));
        asm("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(t1));
        return;
    }
#endif
    acc += p0 + p1;
    acc += p2 + p3;
}
// Partial sums of one 64-frame chunk.  Workgroups add into one of a few slots per chunk
// (few adders per address); saw_bank_finalize_kernel sums the slots and clears them.
#ifndef SAW_PFD
#define SAW_PFD 2
#endif
constexpr int SAW_SLOTS = 64;       // slots per chunk (unused ones stay zero)
struct SawPartial {
    unsigned long long L[64];     // sum over voices of (u_v(t) & 15)
    unsigned long long U0;        // sum u_v(t0)
    uint32_t W[64];               // carries out of the phase add at frame t (t -> t+1)
    uint32_t maxinc, pad_;        // largest increment seen (statistic for the next launch's formulation)
};
// The first bytes of the scratch area (uint32 words):
//   [0] form of the next long-block launch under AUTO (0: stepping, 1: wrap events), read by both forms at
//       their start -- the one that is not selected returns at once;
//   [1] which slot layout the launch filled (for the finalize kernel);
//   [2] number of long blocks finalized so far (copied to the host's mirror: tells a fresh pick from an old one);
//   [3] largest increment as of the last load (saw_sum_inc_kernel; only read by saw_stats_init_kernel);
//   [4..5] exact sum of all increments: computed when the increments are loaded (saw_sum_inc_kernel), kept
//          current by the note-event kernels; the finalize kernels take the bank's I from here (the main kernels
//          do not add the increments up again at every launch).
// The finalize kernel writes [0] from the launch's own largest increment and the sum of increments.
// Note events keep it CONSERVATIVE in between: a new increment above the bound, or a running sum above the
// bound, clears [0] at once (saw_stats_note), so that the event form never runs on a bank the rule would not
// admit -- that is what bounds AUTO's run time.  Both forms are exact for any bank; the flag
// only steers speed.
constexpr uint32_t SAW_EVENTS_MAX_INC = 13u << 25;          // 6.5 wraps per 64 frames (MIDI note 110 at 48 kHz)
constexpr size_t SAW_SCRATCH_HEADER = 64;

// Fold of one 64-frame row of slots (256 threads): sum the SAW_SLOTS copies, clear them for the next launch,
// write the bus and keep the "next bus buffer is zero" contract.
__device__ __forceinline__
void saw_direct_fold(SawPartial *__restrict__ partial, int32_t *__restrict__ bus, int32_t *__restrict__ bus_next,
                     uint32_t nframes, uint32_t row, uint32_t (*Ws)[64], smx::SawPublish pub = smx::SawPublish{})
{
    const uint32_t tid = threadIdx.x, t = tid & 63, part = tid >> 6;
    SawPartial *p = partial + (size_t)row * SAW_SLOTS;
    uint32_t wv[SAW_SLOTS / 4];
#pragma unroll
    for (int k = 0; k < SAW_SLOTS / 4; k++) wv[k] = p[part + 4 * k].W[t];
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < SAW_SLOTS / 4; k++) { w += wv[k]; p[part + 4 * k].W[t] = 0; }
    Ws[part][t] = w;
    __syncthreads();
    const uint32_t f = row * 64u + t;
    if (part == 0 && f < nframes) {
        const int32_t v = (int32_t)(Ws[0][t] + Ws[1][t] + Ws[2][t] + Ws[3][t]);
        bus[f] = v;
        bus_next[f] = 0;
        if (pub.hflag) __builtin_nontemporal_store(v, &pub.hbus[f]);
    }
    // a block of <= 64 frames ends here, in one wave of one workgroup: that wave hands the bus to the host itself
    // (sums first, system-wide fence, then the sequence word: see saw_publish_kernel)
    if (pub.hflag && part == 0) {
        __threadfence_system();
        if (t == 0) __hip_atomic_store(pub.hflag, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The fold a slot launch still owes for its PREDECESSOR (smx::SawPending): rows of `partial`, nullptr for none.
struct SawOwed {
    SawPartial *partial;
    int32_t *bus, *bus_next;
    uint32_t nframes;
};

// SLOT: instead of one atomic per frame per workgroup on the bus itself (same-address integer
// atomics serialise at ~20 ns: 2048 workgroups ending together cost 40 us), the workgroup adds
// its frame sums into one of SAW_SLOTS copies (SawPartial::W) and saw_direct_finalize_kernel
// folds them.  Used for >= 2^20-voice banks from 5 frames up (launch_vw).
template <int TC, int VW, bool NT, bool SLOT>
__global__ __launch_bounds__(256)
void saw_bank_kernel(const uint32_t *__restrict__ inc,
                     const uint32_t *__restrict__ st_in,   // state0[]
                     int32_t *__restrict__ bus,
                     int32_t *__restrict__ bus_next,   // zeroed here for the NEXT launch
                     uint32_t ngroups,      // n_pad / VW
                     uint32_t nframes,      // total frames of this block
                     uint32_t tbase,        // frames elapsed since state0 was valid
                     SawPartial *__restrict__ partial,   // SLOT only: zeroed slots, SAW_SLOTS per chunk
                     SawOwed owed)                       // SLOT only: the previous block's slots, folded here
{
    __shared__ int32_t M[TC][65];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    // Time is cut into chunks of TC frames on blockIdx.y (the phasor is linear: the phase at the chunk's
    // first frame is state0 + (tbase + f0) * inc).  Big banks use one chunk per 64 frames; small banks
    // take shorter chunks so that a block of 64 frames still gives the chip a few hundred workgroups.
    const uint32_t f0 = blockIdx.y * (uint32_t)TC;  // first frame of this chunk
    const uint32_t t0 = tbase + f0;                 // phase offset of this chunk

    for (uint32_t i = tid; i < TC * 65; i += 256) (&M[0][0])[i] = 0;
    // the bus is accumulated with atomics, so it must start at zero: each launch
    // clears the buffer its successor will use (saves a fill kernel per step)
    if (!SLOT && blockIdx.x == 0 && blockIdx.y == 0)
        for (uint32_t i = tid; i < nframes; i += 256) bus_next[i] = 0;

    int32_t acc[TC];
#pragma unroll
    for (int t = 0; t < TC; t++) acc[t] = 0;

    // VW == 4: ngroups is a multiple of 256 (n_pad of 1024), so the grid-stride loop runs over
    // whole workgroup rows with a wave-uniform trip count, and the next row's 32 bytes per lane
    // are requested before the arithmetic on the current row (software prefetch).
    // From 8 frames up (SLOT variants) the prefetch runs TWO rows ahead: one row of arithmetic
    // (160..640 cycles x 8 waves per SIMD) is shorter than the HBM latency, and with one row in
    // flight per wave the chip holds 16 MB in flight, ~5 TB/s at most.
    // PFD rows are kept in flight per wave (a small queue of register rows, rotated per trip).
    constexpr int PFD = SLOT ? SAW_PFD : 1;
    const uint32_t nrows = ngroups >> 8;
    u32x4 qa[PFD], qb[PFD];
    if constexpr (VW == 4) {
#pragma unroll
        for (int k = 0; k < PFD; k++) { qa[k] = 0; qb[k] = 0; }
        if (blockIdx.x < nrows) {
#pragma unroll
            for (int k = 0; k < PFD; k++) {
                const uint32_t r = min(blockIdx.x + k * gridDim.x, nrows - 1) * 256u + tid;   // short banks re-read a row
                qa[k] = stream_load<NT>(reinterpret_cast<const u32x4 *>(inc) + r);
                qb[k] = stream_load<NT>(reinterpret_cast<const u32x4 *>(st_in) + r);
            }
        }
    }
    if constexpr (SLOT) {
        // the previous slot launch left its fold to this one (other slot region, so no clash with this launch's
        // atomics): the first workgroups take one 64-frame row each while their own first rows are in flight
        const uint32_t wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (owed.partial && wg < (owed.nframes + 63u) / 64u) {
            __shared__ uint32_t Ws[4][64];
            saw_direct_fold(owed.partial, owed.bus, owed.bus_next, owed.nframes, wg, Ws);
        }
    }
    const uint32_t g_first = (VW == 4) ? blockIdx.x : blockIdx.x * 256u + tid;
    const uint32_t g_end = (VW == 4) ? nrows : ngroups;
    const uint32_t g_step = (VW == 4) ? gridDim.x : gridDim.x * 256u;
    for (uint32_t gi = g_first; gi < g_end; gi += g_step) {
        const uint32_t g = (VW == 4) ? gi * 256u + tid : gi;
        uint32_t vi[VW], vs[VW];
        if constexpr (VW == 4) {
            const u32x4 a = qa[0], b = qb[0];
#pragma unroll
            for (int k = 0; k + 1 < PFD; k++) { qa[k] = qa[k + 1]; qb[k] = qb[k + 1]; }
            const uint32_t rn = min(gi + PFD * gridDim.x, nrows - 1) * 256u + tid;           // last trips re-read a row
            qa[PFD - 1] = stream_load<NT>(reinterpret_cast<const u32x4 *>(inc) + rn);
            qb[PFD - 1] = stream_load<NT>(reinterpret_cast<const u32x4 *>(st_in) + rn);
            vi[0] = a.x; vi[1] = a.y; vi[2] = a.z; vi[3] = a.w;
            vs[0] = b.x; vs[1] = b.y; vs[2] = b.z; vs[3] = b.w;
        } else {
            vi[0] = stream_load<NT>(inc + g);
            vs[0] = stream_load<NT>(st_in + g);
        }
#pragma unroll
        for (int k = 0; k < VW; k++) {
            // an inactive voice (inc == 0) contributes nothing: park it at 0
            vs[k] = vi[k] ? vs[k] + t0 * vi[k] : 0u;
        }
#pragma unroll
        for (int t = 0; t < TC; t++) {
            if constexpr (VW == 4) {
                acc_add4<NT>(acc[t], (int32_t)vs[0] >> 4, (int32_t)vs[1] >> 4, (int32_t)vs[2] >> 4, (int32_t)vs[3] >> 4);
            } else {
                acc[t] += ((int32_t)vs[0] >> 4);
            }
#pragma unroll
            for (int k = 0; k < VW; k++) vs[k] += vi[k];
        }
    }

    __syncthreads();
#pragma unroll
    for (int t = 0; t < TC; t++) atomicAdd(&M[t][lane], acc[t]);
    __syncthreads();

    // 4 threads per frame, 16 columns each, folded by two shuffles
    const uint32_t t = tid >> 2, q = tid & 3;
    int32_t s = 0;
    if (t < TC) {
#pragma unroll
        for (int j = 0; j < 16; j++) s += M[t][q * 16 + j];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if constexpr (SLOT) {
        // slots are kept per 64 frames: chunk f0 lands in row f0 / 64 at offset f0 % 64 (TC divides 64)
        SawPartial *out = partial + (size_t)(f0 >> 6) * SAW_SLOTS + (blockIdx.x % SAW_SLOTS);
        if (q == 0 && t < TC && f0 + t < nframes) atomicAdd(&out->W[(f0 & 63u) + t], (uint32_t)s);
    } else {
        if (q == 0 && t < TC && f0 + t < nframes) atomicAdd(&bus[f0 + t], s);
    }
}

// Fold the slots of the direct formulation (one workgroup per 64-frame chunk), clear them for the
// next launch, write the bus and keep the "next bus buffer is zero" contract.
__global__ __launch_bounds__(256)
void saw_direct_finalize_kernel(SawPartial *__restrict__ partial, int32_t *__restrict__ bus,
                                int32_t *__restrict__ bus_next, uint32_t nframes, smx::SawPublish pub)
{
    __shared__ uint32_t Ws[4][64];
    saw_direct_fold(partial, bus, bus_next, nframes, blockIdx.x, Ws, pub);
}

// Few-frame blocks (the tick ABI: 1..4 frames) of banks with >= 2^20 voices: the same direct
// formulation in 1024-thread workgroups.  A few-frame launch is a pure read stream plus ONE
// bus atomic per frame per workgroup; big workgroups give the stream the same bytes in flight
// with a quarter of the workgroups (256 x 1024 threads: 6.9 TB/s), and a quarter of the
// serialised atomics.  Partial sums are folded with wave shuffles instead of the LDS matrix.
template <int TC, bool NT>
__global__ __launch_bounds__(1024)
void saw_tick_kernel(const uint32_t *__restrict__ inc, const uint32_t *__restrict__ st_in,
                     int32_t *__restrict__ bus, int32_t *__restrict__ bus_next,
                     uint32_t ngroups, uint32_t nframes, uint32_t tbase)
{
    __shared__ int32_t W[16][TC];
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x == 0 && tid < nframes) bus_next[tid] = 0;     // same contract as saw_bank_kernel
    int32_t acc[TC];
#pragma unroll
    for (int t = 0; t < TC; t++) acc[t] = 0;
    const u32x4 *inc4 = reinterpret_cast<const u32x4 *>(inc);
    const u32x4 *st4 = reinterpret_cast<const u32x4 *>(st_in);
    const uint32_t nrows = ngroups >> 10;                        // ngroups is a multiple of 1024 here
    u32x4 a_next = 0, b_next = 0;
    if (blockIdx.x < nrows) {
        a_next = stream_load<NT>(inc4 + blockIdx.x * 1024u + tid);
        b_next = stream_load<NT>(st4 + blockIdx.x * 1024u + tid);
    }
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const u32x4 a = a_next, b = b_next;
        const uint32_t rn = min(row + gridDim.x, nrows - 1) * 1024u + tid;   // software prefetch
        a_next = stream_load<NT>(inc4 + rn);
        b_next = stream_load<NT>(st4 + rn);
        u32x4 s = b + tbase * a;
        // an inactive voice (inc == 0) contributes nothing: park it at 0
        s.x = a.x ? s.x : 0u; s.y = a.y ? s.y : 0u; s.z = a.z ? s.z : 0u; s.w = a.w ? s.w : 0u;
#pragma unroll
        for (int t = 0; t < TC; t++) {
            acc_add4<false>(acc[t], (int32_t)s.x >> 4, (int32_t)s.y >> 4, (int32_t)s.z >> 4, (int32_t)s.w >> 4);
            s += a;
        }
    }
#pragma unroll
    for (int t = 0; t < TC; t++) {
        int32_t v = acc[t];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((tid & 63) == 0) W[tid >> 6][t] = v;
    }
    __syncthreads();
    if (tid < TC && tid < nframes) {
        int32_t v = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) v += W[w][tid];
        atomicAdd(&bus[tid], v);
    }
}

// ---------------------------------------------------------------------------
// carry-count formulation (see the header comment)
// ---------------------------------------------------------------------------
// Four voices advance one frame: 4 x v_add_co_u32 (carry-outs as SGPR masks).  The
// carries of voices 0/1 are counted per lane by v_addc_co_u32; those of voices 2/3
// are returned as a scalar population count.  All SGPR masks are consumed >= 3
// instructions after they are written (gfx950 needs 2 wait states between a VALU
// SGPR write and a VALU read; scalar reads are interlocked).
__device__ __forceinline__ uint32_t carry_step4(uint32_t &u0, uint32_t &u1, uint32_t &u2, uint32_t &u3,
                                                uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3,
                                                uint32_t &cnt)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long m0, m1, m2, m3;
    asm("v_add_co_u32_e64 %0, %5, %0, %9\n\t"
        "v_add_co_u32_e64 %1, %6, %1, %10\n\t"
        "v_add_co_u32_e64 %2, %7, %2, %11\n\t"
        "v_add_co_u32_e64 %3, %8, %3, %12\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %5\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %6"
        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(cnt), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(i0), "v"(i1), "v"(i2), "v"(i3)
        : "vcc");
    return (uint32_t)(__builtin_popcountll(m2) + __builtin_popcountll(m3));
#else
    (void)u0; (void)u1; (void)u2; (void)u3; (void)i0; (void)i1; (void)i2; (void)i3; (void)cnt;
    return 0;
#endif
}

// The same four voices as {wraps so far, phase} pairs: phase += inc as ONE 64-bit multiply-add whose carry lands in
// the high word (v_mad_u64_u32 issues at the rate of one v_add_co: 4.5 cycles, tools/ubench/valu_rates.hip), and the
// voices' CUMULATIVE wrap counts enter the frame's counter with two v_add3 -- 6 vector instructions per 4 voices and
// frame like carry_step4, but none of them waits for an SGPR mask and nothing is left to the scalar unit
// (tools/ubench/saw_v2_proto.hip: 21.1 vs 19.1-20.0 T voice-samples/s; in the kernel, stepping form pinned: 64 Mi
// voices x 64 frames 225.6 -> 212 us, 16 Mi x 64 72.1 -> 65.2, 8 Mi x 128 79.1 -> 71.5, 256 Ki x 4096 78.5 -> 72.0;
// two voices through pairs and two through masks + s_bcnt1 measured as well: better only for 64 Mi x 128, 385 vs 411).
// The counters then hold cumulative counts; the caller turns them into per-frame counts once per launch.
__device__ __forceinline__ void carry_step4_wide(unsigned long long &q0, unsigned long long &q1, unsigned long long &q2,
                                                 unsigned long long &q3, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3,
                                                 uint32_t &cnt)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %4, 1, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %5, 1, %1\n\t"
        "v_mad_u64_u32 %2, vcc, %6, 1, %2\n\t"
        "v_mad_u64_u32 %3, vcc, %7, 1, %3"
        : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
        : "v"(i0), "v"(i1), "v"(i2), "v"(i3)
        : "vcc");
    cnt += (uint32_t)(q0 >> 32) + (uint32_t)(q1 >> 32);
    cnt += (uint32_t)(q2 >> 32) + (uint32_t)(q3 >> 32);
#else
    (void)q0; (void)q1; (void)q2; (void)q3; (void)i0; (void)i1; (void)i2; (void)i3; (void)cnt;
#endif
}
// floor(a / d) for a quotient known to be below 2^11, with ONE reciprocal per divisor shared by all quotients of a
// voice: the estimate (float)a * rd, rd = rcp((float)d) * (1 - 2^-18), lies below the true ratio by less than
// 2^-17.5 of it (three roundings of 2^-24 .. 2^-23 each against a bias of 2^-18), i.e. by less than 0.006 for ratios
// below 2^11 -- so its floor is the quotient or one less, and one compare settles it.  The generic 32-bit division
// the compiler emits is ~25 instructions; this is 6.  (Only voices that wrap within the chunk are divided, and for
// those both quotients are below the chunk length or irrelevant: see the callers.)
__device__ __forceinline__ float rcp_biased(uint32_t d)
{
    return __builtin_amdgcn_rcpf((float)d) * 0.99999618530273437500f;      // 1 - 2^-18
}
__device__ __forceinline__ uint32_t div_small(uint32_t a, uint32_t d, float rd)
{
    uint32_t q = (uint32_t)((float)a * rd);
    const uint32_t r = a - q * d;                 // q <= the true quotient: no wrap
    return r >= d ? q + 1u : q;
}

// MULTI: more than one 64-frame chunk per launch (blockIdx.y).
// TC: frames computed per chunk: 64; 32 for blocks of 17..32 frames of >= 2^25-voice banks (the stepping chunk equals
//   the direct form there within 1-3 %: 64 Mi voices x 32 frames 139.6 against 138.5 us; what pays is the event form);
//   128 (EVENTS only) for blocks of 65..128 frames.
// EVENTS: the wraps are not found by stepping the phases but located directly:
//   first wrap of a voice at frame  n1 = floor(~u / inc)            (u: offset-binary phase at chunk start)
//   then gaps of                    Q + (r <= R),  Q = floor((2^32-1)/inc),  R = 2^32-1 - Q*inc
//   residual                        r <- r + (r <= R ? E : E - inc),  E = inc-1-R,  r0 = u + (n1+1)*inc
// ~8 vector instructions and one LDS add per WRAP (plus two 32-bit divisions per voice) instead of
// 1.5 per voice-SAMPLE: a piano-range bank wraps 1.1 times per voice in 64 frames -- 64 Mi voices
// x 64 frames 236 -> 18x us -- but the loop runs as long as the busiest voice of the wave, so banks
// with many high voices are slower this way (all voices at 12 wraps: 3x).  The finalize kernel
// keeps the statistic that picks the form (mode_flag; see SAW_SCRATCH_HEADER).
// NTH: threads per workgroup.  The event form keeps a 2 KB list per wave next to the counting matrix: with 256 threads
// that is 25.9 KB per workgroup, 6 workgroups = 6 waves per SIMD on a CU's 160 KB; 512 threads share one matrix and
// one histogram (34 KB per workgroup, 4 of them: 8 waves per SIMD, which the kernel's 61 vector registers allow).
template <bool NT, bool MULTI, int TC, bool EVENTS, bool WIDE = false, int NTH = 256>
__global__ __launch_bounds__(NTH)
void saw_bank_carry_kernel(const uint32_t *__restrict__ inc, const uint32_t *__restrict__ st_in,
                           SawPartial *__restrict__ partial, uint32_t ngroups, uint32_t tbase,
                           const uint32_t *__restrict__ mode_flag, uint32_t *__restrict__ ran_long,
                           uint32_t nframes)            // EVENTS: the block's frames (wraps beyond them are not located)
{
    static_assert(TC == 64 || TC == 32 || (TC == 128 && EVENTS),
                  "frames per chunk (32: blocks of 17..32 frames, one chunk; 128: blocks of 65..128 frames, event form)");
    static_assert(TC == 64 || !MULTI, "a 32- or 128-frame chunk is the whole block");
    constexpr uint32_t LG = TC == 128 ? 7 : TC == 64 ? 6 : 5;      // log2(TC)
    constexpr int TCM = TC > 64 ? TC : 64;         // frames of the counting matrix (whole 64-frame slot rows)
    __shared__ uint32_t M[TCM][65];                // [frame][lane] carry counts; column 64: scalar counts
    __shared__ uint32_t H[256];                    // histogram of (phase & 15, inc & 15)
    __shared__ unsigned long long S[2];            // U0 of the chunk's first (and, TC == 128, second) 64-frame row
    __shared__ uint32_t MX;                        // largest increment
    __shared__ uint2 EL[EVENTS ? (NTH / 64) * 256 : 1];     // EVENTS: per wave, the (phase, inc) of the voices that wrap
    if (mode_flag && (*mode_flag != 0u) != EVENTS) return;     // the other form runs this launch
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    if (ran_long && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *ran_long = 0;   // 64-frame slot layout
    const uint32_t t0 = tbase + (MULTI ? blockIdx.y * 64u : 0u);   // phase offset of this chunk
    for (uint32_t i = tid; i < (uint32_t)TCM * 65u; i += NTH) (&M[0][0])[i] = 0;
    if (tid < 256) H[tid] = 0;
    if (tid < 2) S[tid] = 0;
    if (tid == 0) MX = 0;
    __syncthreads();

    uint32_t cnt[EVENTS ? 1 : TC];                 // per-lane carry counts (voices 0/1)
    uint32_t W[EVENTS ? 1 : 32];                   // wave-uniform counts (voices 2/3): frames t | t+32 << 16
    if constexpr (!EVENTS) {
#pragma unroll
        for (int t = 0; t < TC; t++) cnt[t] = 0;
#pragma unroll
        for (int t = 0; t < 32; t++) W[t] = 0;
    }
    unsigned long long sumU = 0, sumU2 = 0;        // sumU2 (TC == 128): the phases at the second row's first frame
    uint32_t mx = 0;

    // ngroups is a multiple of 256 (n_pad of 1024): whole workgroup rows, so the trip count is
    // wave-uniform and the scalar counters W stay in SGPRs
    const uint32_t nrows = ngroups / (uint32_t)NTH;     // the launcher picks NTH so that it divides ngroups
    // software prefetch: the next row's 32 bytes per lane are requested before the ~6000
    // cycles of arithmetic on the current row, so HBM latency never sits on the critical path
    const u32x4 *inc4 = reinterpret_cast<const u32x4 *>(inc);
    const u32x4 *st4 = reinterpret_cast<const u32x4 *>(st_in);
    u32x4 a_next = 0, b_next = 0;
    if (blockIdx.x < nrows) {
        a_next = stream_load<NT>(inc4 + blockIdx.x * (uint32_t)NTH + tid);
        b_next = stream_load<NT>(st4 + blockIdx.x * (uint32_t)NTH + tid);
    }
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const u32x4 a = a_next, b = b_next;
        // unconditional prefetch (the last trip re-reads its own row): no branch, no join
        const uint32_t rn = min(row + gridDim.x, nrows - 1) * (uint32_t)NTH + tid;
        a_next = stream_load<NT>(inc4 + rn);
        b_next = stream_load<NT>(st4 + rn);
        // an inactive voice (inc == 0) is parked at phase 0: it contributes (0 >> 4) = 0
        uint32_t u0 = (a.x ? b.x + t0 * a.x : 0u) ^ 0x80000000u;
        uint32_t u1 = (a.y ? b.y + t0 * a.y : 0u) ^ 0x80000000u;
        uint32_t u2 = (a.z ? b.z + t0 * a.z : 0u) ^ 0x80000000u;
        uint32_t u3 = (a.w ? b.w + t0 * a.w : 0u) ^ 0x80000000u;
        sumU += (unsigned long long)u0 + u1 + u2 + u3;
        if constexpr (TC == 128)      // offset-binary phases 64 frames on (mod 2^32 each): the second slot row's U0
            sumU2 += (unsigned long long)(u0 + (a.x << 6)) + (u1 + (a.y << 6)) + (u2 + (a.z << 6)) + (u3 + (a.w << 6));
        atomicAdd(&H[((u0 & 15) << 4) | (a.x & 15)], 1u);
        atomicAdd(&H[((u1 & 15) << 4) | (a.y & 15)], 1u);
        atomicAdd(&H[((u2 & 15) << 4) | (a.z & 15)], 1u);
        atomicAdd(&H[((u3 & 15) << 4) | (a.w & 15)], 1u);
        mx = max(max(mx, a.x), max(max(a.y, a.z), a.w));
        if constexpr (EVENTS) {
            // (1) Which voices wrap in this chunk at all, and how often?  K = (u + 64*inc) >> 32.  In a
            //     piano-range bank more than half do not wrap, and they need neither the divisions nor the
            //     loop: the wave compacts the (u, inc) pairs of its wrapping voices into its own LDS list, the
            //     busy ones (K >= 3) first, then the others ...
            const uint32_t vi[4] = {a.x, a.y, a.z, a.w}, vu[4] = {u0, u1, u2, u3};
            uint2 *list = &EL[(tid >> 6) * 256];
            bool heavy[4], light[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t lo = vu[k] + (vi[k] << LG);
                const uint32_t K = (vi[k] >> (32 - LG)) + (lo < vu[k] ? 1u : 0u);
                heavy[k] = K >= 3u;
                light[k] = K - 1u < 2u;
            }
            uint32_t nw = 0;                                  // wave-uniform
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool w = pass == 0 ? heavy[k] : light[k];
                    const unsigned long long m = __ballot(w);
                    const uint32_t pos = nw + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (w) list[pos] = make_uint2(vu[k], vi[k]);
                    nw += (uint32_t)__builtin_popcountll(m);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // (2) ... and the lanes take entries lane, lane + 64, ... slot by slot: the same number of wrapping
            //     voices per lane (+-1) whatever their place in the bank, and because the busy voices sit together
            //     in the first slot(s), a slot's loop runs as long as ITS busiest voice, not the wave's (a
            //     piano-range row: ~5.5 rounds over the first slot and 2 over the second, instead of 6 over both)
            for (uint32_t k = 0; 64u * k < nw; k++) {            // wave-uniform trip count
                const uint32_t e = lane + 64u * k;
                const uint2 en = list[e < nw ? e : 0u];
                const uint32_t d = en.y;                          // > 0: an off voice never wraps
                const float rd = rcp_biased(d);
                // the voice wraps within the chunk, so ~u < TC d: the first quotient is below TC (64 or 32)
                const uint32_t n1 = div_small(~en.x, d, rd);
                // Q < TC exactly when d >= 2^(32-LG); a smaller increment wraps at most once per chunk and any
                // gap beyond the chunk is as good as any other (2^30 keeps et + gap from wrapping)
                const uint32_t eq = (d >> (32 - LG)) ? div_small(0xFFFFFFFFu, d, rd) : (1u << 30);
                const uint32_t erm = 0xFFFFFFFFu - eq * d;        // (meaningless, and unused, in the second case)
                const uint32_t ee = d - 1u - erm;
                uint32_t er = en.x + (n1 + 1u) * d;               // mod 2^32: the phase right after the first wrap
                uint32_t et = e < nw ? n1 : 0xFFFFFFFFu;
                // every gap is at least one frame, so TC rounds always suffice: the bound makes the
                // loop finite whatever the data
                // the chunk's frames that the block needs: a 65-frame block in a 128-frame chunk, a 40-frame block in a
                // 64-frame one, the last chunk of a MULTI launch
                const uint32_t tlim = min((uint32_t)TC, nframes - (MULTI ? blockIdx.y * 64u : 0u));
                for (int round = 0; round < TC && __any(et < tlim); round++) {
                    if (et < tlim) {
                        atomicAdd(&M[et][lane], 1u);             // own column: no lane ever shares an address
                        const bool c = er <= erm;
                        et += eq + (c ? 1u : 0u);
                        er += c ? ee : ee - d;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // list is free for the next row
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (WIDE) {
            unsigned long long q0 = u0, q1 = u1, q2 = u2, q3 = u3;       // high words: wraps so far in this chunk
#pragma unroll
            for (int t = 0; t < TC; t++) carry_step4_wide(q0, q1, q2, q3, a.x, a.y, a.z, a.w, cnt[t]);
        } else {
#pragma unroll
            for (int t = 0; t < TC; t++) {
                const uint32_t c = carry_step4(u0, u1, u2, u3, a.x, a.y, a.z, a.w, cnt[t]);
                W[t & 31] += (t < 32) ? c : (c << 16);
            }
        }
    }

    if constexpr (!EVENTS) {
        if constexpr (WIDE) {
            // the 64-bit forms counted cumulative wraps (frames 0..t): back to wraps AT frame t
#pragma unroll
            for (int t = TC - 1; t > 0; t--) cnt[t] -= cnt[t - 1];
        }
        // per-lane counts -> M[t][lane]; per-wave scalar counts -> M[t][64]
#pragma unroll
        for (int t = 0; t < TC; t++) atomicAdd(&M[t][lane], cnt[t]);
        if (!WIDE && lane == 0) {
#pragma unroll
            for (int t = 0; t < 32; t++) {
                atomicAdd(&M[t][64], W[t] & 0xFFFFu);
                atomicAdd(&M[t + 32][64], W[t] >> 16);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if (lane == 0) atomicMax(&MX, mx);
    for (int o = 32; o > 0; o >>= 1) sumU += __shfl_xor(sumU, o);
    if (lane == 0) atomicAdd(&S[0], sumU);
    if constexpr (TC == 128) {
        for (int o = 32; o > 0; o >>= 1) sumU2 += __shfl_xor(sumU2, o);
        if (lane == 0) atomicAdd(&S[1], sumU2);
    }
    __syncthreads();

    // a 128-frame chunk fills two 64-frame slot rows, as two chunks of a MULTI launch would: the second row's frames
    // count from its own first frame (the finalize kernel's W(t) and t * I are relative to the row's U0)
#pragma unroll
    for (int half = 0; half < TCM / 64; half++) {
        SawPartial *out = partial + (size_t)(blockIdx.y + half) * SAW_SLOTS + (blockIdx.x % SAW_SLOTS);
        if (tid < 256) {   // carries per frame: 4 lanes x 16 columns (+ column 64)
            const uint32_t t = tid >> 2, q = tid & 3;
            uint32_t s = (q == 0) ? M[64 * half + t][64] : 0u;
#pragma unroll
            for (int j = 0; j < 16; j++) s += M[64 * half + t][q * 16 + j];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (q == 0) atomicAdd(&out->W[t], s);
        }
        if (tid < 256) {   // low-nibble sums per frame from the histogram: 4 lanes x 64 bins per frame.  A workgroup
            // sees at most 400 rows x 1024 voices < 2^19 voices (launch_saw_bank; 2^20 with 1024 threads: 64 bins x
            // count x 15 stays below 2^32), a bin count fits 24 bits: 32-bit v_mad_u32_u24 sums, widened at the end.
            // (the second row's histogram is the first one's: 64 * (inc & 15) = 0 mod 16)
            const uint32_t t = tid >> 2, q = tid & 3;
            uint32_t l32 = 0;
#pragma unroll 8
            for (uint32_t k = 0; k < 64; k++) {
                const uint32_t bin = q * 64 + k;
                l32 = __umul24(H[bin], ((bin >> 4) + t * (bin & 15)) & 15) + l32;
            }
            unsigned long long l = l32;
            l += __shfl_xor(l, 1);
            l += __shfl_xor(l, 2);
            if (q == 0) atomicAdd(&out->L[t], l);
        }
        if (tid == 0) { atomicAdd(&out->U0, S[half]); atomicMax(&out->maxinc, MX); }
    }
}

// The event form for launches of 129 frames and more: 256-frame chunks, so that the divisions are
// paid once per 256 frames (wraps are located only up to the frames the block needs: tlim).  The wraps go to ONE histogram per workgroup (a per-lane matrix of 256
// frames would not fit), the nibble histogram travels to the slot as it is, and the finalize
// kernel does the rest.  Slot layout of a 256-frame chunk:
// (TL = 256; launches of 1024 frames and more use 1024-frame chunks, TL = 1024.)
template <uint32_t TL>
struct SawPartialL {
    unsigned long long U0;        // sum u_v(t0)
    uint32_t maxinc, pad_;
    uint32_t H[256];              // voices per (phase & 15, inc & 15) class
    uint32_t W[TL];               // wraps at frame t (t -> t+1)
};
// Shortest launch that takes the long event form.  Round 3: every launch of more than 64 frames -- the kernel locates
// wraps only up to the frames the block needs (tlim), so 65..255 frames are ONE pass over the bank instead of two to
// four 64-frame chunks (64 Mi voices, piano range: 65 frames 243 -> 178 us, 128 frames 244 -> 210, 192 frames 360 ->
// 242, 255 frames 465 -> 273; 16 Mi voices x 128 frames 80.9 -> 68.7: profiles/r03_long_min.txt).  Rounds 1-2: 256.
constexpr uint32_t SAW_LONG_DEFAULT = 65;

template <bool NT, uint32_t TL>
__global__ __launch_bounds__(256)
void saw_bank_event_long_kernel(const uint32_t *__restrict__ inc, const uint32_t *__restrict__ st_in,
                                SawPartialL<TL> *__restrict__ partial, uint32_t ngroups, uint32_t tbase,
                                const uint32_t *__restrict__ mode_flag, uint32_t *__restrict__ ran_long,
                                uint32_t nframes)
{
    static_assert(TL == 256 || TL == 1024, "chunk lengths of the long event form");
    constexpr uint32_t LG = TL == 256 ? 8 : 10;
    __shared__ uint32_t hist[TL];
    __shared__ uint32_t H[256];
    __shared__ unsigned long long S[2];
    __shared__ uint32_t MX;
    __shared__ uint2 EL[4 * 256];
    __shared__ uint32_t CNT[4 * 16];
    if (mode_flag && *mode_flag == 0u) return;                 // the stepping form runs this launch
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *ran_long = TL == 256 ? 1u : 2u;   // slot layout of this launch
    const uint32_t t0 = tbase + blockIdx.y * TL;
    // frames of this chunk that the block needs (the last chunk of a launch, or a launch shorter than the chunk:
    // 128..255 frames run as one 256-frame chunk -- the bank is read once instead of two to four times): wraps
    // beyond them are not located
    const uint32_t tlim = min(TL, nframes - blockIdx.y * TL);
    for (uint32_t i = tid; i < TL; i += 256) hist[i] = 0;
    H[tid] = 0;
    if (tid < 2) S[tid] = 0;
    if (tid == 0) MX = 0;
    __syncthreads();

    unsigned long long sumU = 0;
    uint32_t mx = 0;
    const uint32_t nrows = ngroups >> 8;
    const u32x4 *inc4 = reinterpret_cast<const u32x4 *>(inc);
    const u32x4 *st4 = reinterpret_cast<const u32x4 *>(st_in);
    u32x4 a_next = 0, b_next = 0;
    if (blockIdx.x < nrows) {
        a_next = stream_load<NT>(inc4 + blockIdx.x * 256u + tid);
        b_next = stream_load<NT>(st4 + blockIdx.x * 256u + tid);
    }
    uint2 *list = &EL[(tid >> 6) * 256];
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const u32x4 a = a_next, b = b_next;
        const uint32_t rn = min(row + gridDim.x, nrows - 1) * 256u + tid;
        a_next = stream_load<NT>(inc4 + rn);
        b_next = stream_load<NT>(st4 + rn);
        const uint32_t vi[4] = {a.x, a.y, a.z, a.w};
        uint32_t vu[4];
        vu[0] = (a.x ? b.x + t0 * a.x : 0u) ^ 0x80000000u;
        vu[1] = (a.y ? b.y + t0 * a.y : 0u) ^ 0x80000000u;
        vu[2] = (a.z ? b.z + t0 * a.z : 0u) ^ 0x80000000u;
        vu[3] = (a.w ? b.w + t0 * a.w : 0u) ^ 0x80000000u;
        sumU += (unsigned long long)vu[0] + vu[1] + vu[2] + vu[3];
        mx = max(max(mx, a.x), max(max(a.y, a.z), a.w));
        // (1) The wave sorts its wrapping voices by the binary order of their wrap count K (a counting
        //     sort over 9 classes in LDS): dealt out lane by lane afterwards, slot k of every lane then
        //     holds voices of about the same K, and the slots are worked off one after the other -- a
        //     slot's loop runs as long as ITS busiest voice, not as long as the wave's.
        uint32_t *cnt = &CNT[(tid >> 6) * 16];
        if (lane < 16) cnt[lane] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t cls[4], pos[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            atomicAdd(&H[((vu[k] & 15) << 4) | (vi[k] & 15)], 1u);
            // wraps within the chunk: K = (u + TL*inc) >> 32
            const uint32_t lo = vu[k] + (vi[k] << LG);
            const uint32_t K = (vi[k] >> (32 - LG)) + (lo < vu[k] ? 1u : 0u);
            cls[k] = K ? 31u - (uint32_t)__builtin_clz(K) : 0xFFFFFFFFu;       // 0..LG, or none
            pos[k] = 0;
            if (K) pos[k] = atomicAdd(&cnt[cls[k]], 1u);                       // place within the class
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // exclusive prefix over the classes (lanes 0..15), total = number of wrapping voices
        uint32_t c = lane < 16 ? cnt[lane] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += up;
        }
        const uint32_t nw = __shfl(incl, 15);                 // wave-uniform
        __builtin_amdgcn_wave_barrier();
        if (lane < 16) cnt[lane] = incl - c;                   // class bases
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (cls[k] != 0xFFFFFFFFu) list[cnt[cls[k]] + pos[k]] = make_uint2(vu[k], vi[k]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (2) slot by slot: entry lane + 64*k, its two divisions, its wraps
        for (uint32_t k = 0; 64u * k < nw; k++) {              // wave-uniform trip count
            const uint32_t e = lane + 64u * k;
            const uint2 en = list[e < nw ? e : 0u];
            const uint32_t d = en.y;
            const float rd = rcp_biased(d);
            const uint32_t n1 = div_small(~en.x, d, rd);      // < TL: the voice wraps within the chunk
            // Q < TL exactly when d >= 2^(32-LG); below that the next wrap lies beyond the chunk anyway
            const uint32_t eq = (d >> (32 - LG)) ? div_small(0xFFFFFFFFu, d, rd) : (1u << 30);
            const uint32_t erm = 0xFFFFFFFFu - eq * d;
            const uint32_t ee = d - 1u - erm;
            uint32_t er = en.x + (n1 + 1u) * d;               // the phase right after the first wrap
            uint32_t et = e < nw ? n1 : 0xFFFFFFFFu;
            // every gap is at least one frame: TL rounds always suffice
            for (uint32_t round = 0; round < TL && __any(et < tlim); round++) {
                if (et < tlim) {
                    atomicAdd(&hist[et], 1u);
                    const bool cc = er <= erm;
                    et += eq + (cc ? 1u : 0u);
                    er += cc ? ee : ee - d;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // list and counters are free for the next row
        __builtin_amdgcn_wave_barrier();
    }
    for (int o = 32; o > 0; o >>= 1) {
        sumU += __shfl_xor(sumU, o);
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    }
    if (lane == 0) { atomicAdd(&S[0], sumU); atomicMax(&MX, mx); }
    __syncthreads();
    SawPartialL<TL> *out = partial + (size_t)blockIdx.y * SAW_SLOTS + (blockIdx.x % SAW_SLOTS);
    for (uint32_t i = tid; i < TL; i += 256)
        if (hist[i]) atomicAdd(&out->W[i], hist[i]);
    if (H[tid]) atomicAdd(&out->H[tid], H[tid]);
    if (tid == 0) { atomicAdd(&out->U0, S[0]); atomicMax(&out->maxinc, MX); }
}

// End of a long block: the exact statistics of this launch (header words above) and the host's mirror
// {pick, number of long blocks finalized}.
// host_tag: the HOST's number of this long block (it counts the long blocks it launches): the host ignores picks
// whose tag is not younger than its last note event, so a finalization that was in flight when the increments
// changed can never pin a form.
__device__ __forceinline__ void saw_stats_publish(uint32_t *__restrict__ hdr, uint32_t *__restrict__ host_flag,
                                                  uint32_t pick, uint32_t host_tag)
{
    hdr[0] = pick;
    hdr[2] = hdr[2] + 1u;                                         // long blocks finalized since the header was cleared
    if (host_flag) { host_flag[0] = pick; host_flag[1] = host_tag; }   // pinned host copy: lets the host skip the form that would return at once
}
// I = the sum of all increments of the bank (header words [4..5], exact: see SAW_SCRATCH_HEADER)
__device__ __forceinline__ unsigned long long saw_stats_sum_inc(const uint32_t *__restrict__ hdr)
{
    return *reinterpret_cast<const unsigned long long *>(hdr + 4);
}
// A note event changes one increment: keep the pick conservative until the next long block recomputes it.
__device__ __forceinline__ void saw_stats_note(uint32_t *__restrict__ hdr, uint32_t nvoices, uint32_t old_inc, uint32_t new_inc)
{
    if (new_inc >= SAW_EVENTS_MAX_INC) hdr[0] = 0u;
    const unsigned long long d = (unsigned long long)new_inc - (unsigned long long)old_inc;       // mod 2^64
    const unsigned long long s = atomicAdd(reinterpret_cast<unsigned long long *>(hdr + 4), d) + d;
    if (s > ((unsigned long long)nvoices << 27)) hdr[0] = 0u;
}

// Finalize of a TL-frame chunk of the long event form (called from saw_bank_finalize_kernel):
// thread tid owns the frames FPT*tid .. FPT*tid + FPT-1 of the chunk.
template <uint32_t TL>
__device__ __forceinline__ void saw_finalize_long(SawPartialL<TL> *__restrict__ partial, int32_t *__restrict__ bus,
                                                  int32_t *__restrict__ bus_next, uint32_t nframes,
                                                  uint32_t nvoices, uint32_t *__restrict__ mode_flag,
                                                  uint32_t *__restrict__ host_flag, uint32_t host_tag)
{
    constexpr uint32_t FPT = TL / 256;
    __shared__ uint32_t Hs[256], Wsum[4];
    __shared__ unsigned long long US[1];
    __shared__ uint32_t MXs;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SawPartialL<TL> *p = partial + (size_t)blockIdx.x * SAW_SLOTS;
    uint32_t w[FPT], h = 0;
#pragma unroll
    for (uint32_t j = 0; j < FPT; j++) w[j] = 0;
    for (int k0 = 0; k0 < SAW_SLOTS; k0 += 8) {
        uint32_t wv[8][FPT], hv[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (uint32_t j = 0; j < FPT; j++) wv[k][j] = p[k0 + k].W[FPT * tid + j];
            hv[k] = p[k0 + k].H[tid];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
#pragma unroll
            for (uint32_t j = 0; j < FPT; j++) { w[j] += wv[k][j]; p[k0 + k].W[FPT * tid + j] = 0; }
            h += hv[k];
            p[k0 + k].H[tid] = 0;
        }
    }
    Hs[tid] = h;
    if (wave == 0) {                                          // lane s folds slot s's scalars
        unsigned long long u0 = p[lane].U0;
        uint32_t m = p[lane].maxinc;
        p[lane].U0 = 0; p[lane].maxinc = 0;
        for (int o = 32; o > 0; o >>= 1) {
            u0 += __shfl_xor(u0, o);
            m = max(m, (uint32_t)__shfl_xor((int)m, o));
        }
        if (lane == 0) { US[0] = u0; MXs = m; }
    }
    // exclusive prefix of the wraps over the chunk: the thread's own frames, wave scan, waves' totals
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < FPT; j++) mine += w[j];
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += up;
    }
    if (lane == 63) Wsum[wave] = incl;
    __syncthreads();
    uint32_t wraps = incl - mine;
    for (uint32_t k = 0; k < wave; k++) wraps += Wsum[k];
    const unsigned long long U0 = US[0], I = saw_stats_sum_inc(mode_flag);
#pragma unroll
    for (uint32_t j = 0; j < FPT; j++) {
        const uint32_t t = FPT * tid + j;
        // low-nibble sum at frame t: the classes (lo, li) contribute ((lo + t*li) & 15) each
        unsigned long long L = 0;
#pragma unroll 8
        for (uint32_t bin = 0; bin < 256; bin++)
            L += (unsigned long long)Hs[bin] * (((bin >> 4) + t * (bin & 15)) & 15);
        const unsigned long long x = U0 + (unsigned long long)t * I - ((unsigned long long)(wraps & 15u) << 32) - L;
        const uint32_t r = (uint32_t)(x >> 4) - (nvoices << 27);
        const uint32_t f = blockIdx.x * TL + t;
        if (f < nframes) {
            bus[f] = (int32_t)r;
            bus_next[f] = 0;
        }
        wraps += w[j];
    }
    if (blockIdx.x == 0 && tid == 0 && mode_flag) {
        const uint32_t f = (MXs < SAW_EVENTS_MAX_INC && I <= ((unsigned long long)nvoices << 27)) ? 1u : 0u;
        saw_stats_publish(mode_flag, host_flag, f, host_tag);
    }
}

// One workgroup per 64-frame chunk: add the chunk's slots (and clear them for the next
// launch) and emit
//   bus[t0+t] = ((U0 + t*I - 2^32*W(t) - L(t)) >> 4) - nvoices * 2^27        (mod 2^32)
__global__ __launch_bounds__(256)
void saw_bank_finalize_kernel(SawPartial *__restrict__ partial,
                              int32_t *__restrict__ bus, int32_t *__restrict__ bus_next,
                              uint32_t nframes, uint32_t nvoices, uint32_t *__restrict__ mode_flag,
                              const uint32_t *__restrict__ ran_long, uint32_t *__restrict__ host_flag,
                              uint32_t host_tag, smx::SawPublish pub)
{
    // which slot layout did this launch fill?  (written by the main kernel that ran, stable here)
    if (ran_long && *ran_long != 0u) {
        if (*ran_long == 1u) {
            if (blockIdx.x * 256u < nframes)
                saw_finalize_long<256>(reinterpret_cast<SawPartialL<256> *>(partial), bus, bus_next, nframes, nvoices, mode_flag, host_flag, host_tag);
        } else {
            if (blockIdx.x * 1024u < nframes)
                saw_finalize_long<1024>(reinterpret_cast<SawPartialL<1024> *>(partial), bus, bus_next, nframes, nvoices, mode_flag, host_flag, host_tag);
        }
        return;
    }
    if (blockIdx.x * 64u >= nframes) return;
    __shared__ unsigned long long Ls[4][64], Us[4][1];
    __shared__ uint32_t Ws[4][64], Mx[4];
    const uint32_t tid = threadIdx.x, t = tid & 63, part = tid >> 6;
    SawPartial *p = partial + (size_t)blockIdx.x * SAW_SLOTS;
    // (requested together with the slots: this kernel is a chain of memory round trips, ~2 us each, on the tail of
    // every long block -- the layout word above is not even read when the launch cannot have filled a long layout)
    const unsigned long long I = saw_stats_sum_inc(mode_flag);
    // all loads first (16 independent ones per thread in flight), then the clearing stores
    unsigned long long lv[SAW_SLOTS / 4], uv[SAW_SLOTS / 4];
    uint32_t wv[SAW_SLOTS / 4];
#pragma unroll
    for (int k = 0; k < SAW_SLOTS / 4; k++) {
        const SawPartial *q = p + part + 4 * k;
        lv[k] = q->L[t];
        wv[k] = q->W[t];
        uv[k] = (t == 0) ? q->U0 : 0ull;
    }
    uint32_t mxv[SAW_SLOTS / 4];
#pragma unroll
    for (int k = 0; k < SAW_SLOTS / 4; k++) mxv[k] = (t == 0) ? p[part + 4 * k].maxinc : 0u;
    unsigned long long l = 0, u0 = 0;
    uint32_t w = 0, mx = 0;
#pragma unroll
    for (int k = 0; k < SAW_SLOTS / 4; k++) {
        SawPartial *q = p + part + 4 * k;
        l += lv[k]; w += wv[k]; u0 += uv[k]; mx = max(mx, mxv[k]);
        q->L[t] = 0;
        q->W[t] = 0;
        if (t == 0) { q->U0 = 0; q->maxinc = 0; }
    }
    Ls[part][t] = l;
    Ws[part][t] = w;
    if (t == 0) { Us[part][0] = u0; Mx[part] = mx; }
    __syncthreads();
    if (part == 0) {
        const unsigned long long L = Ls[0][t] + Ls[1][t] + Ls[2][t] + Ls[3][t];
        const unsigned long long U0 = Us[0][0] + Us[1][0] + Us[2][0] + Us[3][0];
        // W(t): carries of the frames before t (only mod 16 matters) -- exclusive prefix sum over
        // the 64 lanes of this wave (part == 0 is exactly wave 0; lane == t)
        const uint32_t mine = Ws[0][t] + Ws[1][t] + Ws[2][t] + Ws[3][t];
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (t >= (uint32_t)o) incl += up;
        }
        const uint32_t wraps = incl - mine;
        const unsigned long long x = U0 + (unsigned long long)t * I - ((unsigned long long)(wraps & 15u) << 32) - L;
        const uint32_t r = (uint32_t)(x >> 4) - (nvoices << 27);
        const uint32_t f = blockIdx.x * 64u + t;
        if (f < nframes) {
            bus[f] = (int32_t)r;
            bus_next[f] = 0;                        // same contract as saw_bank_kernel
            if (pub.hflag) __builtin_nontemporal_store((int32_t)r, &pub.hbus[f]);
        }
        // a block of <= 64 frames ends in this wave: it hands the bus to the host itself (saw_publish_kernel's
        // protocol: sums, system-wide fence, sequence word)
        if (pub.hflag) {
            __threadfence_system();
            if (t == 0) __hip_atomic_store(pub.hflag, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        // Form of the next long-block launch: wrap events pay while no voice wraps more than ~6
        // times per 64 frames (inc < 6.5 * 2^26: up to MIDI note 110 at 48 kHz) and the bank's mean
        // is at most 2 wraps per voice (sum of inc <= voices * 2^27).  Measured on 64 Mi voices x
        // 64 frames (stepping: 236 us): piano-range bank 18x us, all voices at 2 wraps 123 us, at 6
        // wraps 269 us, 1 % of the voices at 12 wraps 243 us, all at 12 wraps 750 us.
        if (blockIdx.x == 0 && t == 0 && mode_flag) {
            const uint32_t m = max(max(Mx[0], Mx[1]), max(Mx[2], Mx[3]));
            const uint32_t fl = (m < SAW_EVENTS_MAX_INC && I <= ((unsigned long long)nvoices << 27)) ? 1u : 0u;
            saw_stats_publish(mode_flag, host_flag, fl, host_tag);
        }
    }
}

// sum_tick_square (linux/synth.c:182-195): OR of the active voices' sign bits.
// Unused by the reference's synth_run; kept as a bank variant.  bus word t
// receives 0x80000000 if any active voice has its sign bit set at frame t.
__global__ __launch_bounds__(256)
void square_bank_kernel(const uint32_t *__restrict__ inc,
                        const uint32_t *__restrict__ st_in,
                        uint32_t *__restrict__ or_bus,
                        uint32_t n_pad, uint32_t nframes, uint32_t tbase)
{
    const uint32_t f0 = blockIdx.y * 64u;
    const uint32_t t0 = tbase + f0;
    const uint32_t nf = min(64u, nframes - f0);
    unsigned long long any_lo = 0;   // bit t: some active voice negative at t0+t
    for (uint32_t v = blockIdx.x * 256u + threadIdx.x; v < n_pad; v += gridDim.x * 256u) {
        const uint32_t i = inc[v];
        uint32_t s = st_in[v] + t0 * i;
        if (i) {
            for (uint32_t t = 0; t < nf; t++) {
                any_lo |= (unsigned long long)(s >> 31) << t;
                s += i;
            }
        }
    }
    // wave OR, then one atomic per set frame
    for (int o = 32; o > 0; o >>= 1) any_lo |= __shfl_xor(any_lo, o);
    if ((threadIdx.x & 63) == 0) {
        for (uint32_t t = 0; t < nf; t++)
            if ((any_lo >> t) & 1) atomicOr(&or_bus[f0 + t], 0x80000000u);
    }
}

// Lazy state maintenance (see the header comment).
// A note event on one voice at elapsed time T: keep state0 + T*inc continuous.
__global__ void saw_rebase_kernel(uint32_t *__restrict__ inc, uint32_t *__restrict__ state0,
                                  uint32_t voice, uint32_t new_inc, uint32_t tbase,
                                  uint32_t *__restrict__ hdr, uint32_t nvoices)     // hdr: scratch header or NULL
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint32_t old = inc[voice];
        state0[voice] += tbase * (old - new_inc);
        inc[voice] = new_inc;
        if (hdr) saw_stats_note(hdr, nvoices, old, new_inc);
    }
}
// A block's worth of note events at once: pairs[2k] = voice, pairs[2k+1] = its increment after the
// last event that touched it (the host applies the events to its allocator in order and keeps one
// pair per voice: rebasing is linear in inc and all events of a batch share T, so only the final
// increment matters).  Voices are distinct, one lane per pair.
__global__ __launch_bounds__(256)
void saw_rebase_batch_kernel(uint32_t *__restrict__ inc, uint32_t *__restrict__ state0,
                             const uint32_t *__restrict__ pairs, uint32_t npairs, uint32_t tbase,
                             uint32_t *__restrict__ hdr, uint32_t nvoices)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= npairs) return;
    const uint32_t voice = pairs[2 * k], new_inc = pairs[2 * k + 1];
    const uint32_t old = inc[voice];
    state0[voice] += tbase * (old - new_inc);
    inc[voice] = new_inc;
    if (hdr) saw_stats_note(hdr, nvoices, old, new_inc);
}
// I = sum of all increments, into the scratch header (after the increments were loaded; the header was cleared),
// and their maximum into word [3]: saw_stats_init_kernel, queued behind this one, turns both into the form pick of
// the bank's first long block (round 3: the pick used to start at "stepping" and only a long block's finalize set it).
__global__ __launch_bounds__(256)
void saw_sum_inc_kernel(const uint32_t *__restrict__ inc, uint32_t n_pad, uint32_t *__restrict__ hdr)
{
    unsigned long long s = 0;
    uint32_t mx = 0;
    const u32x4 *inc4 = reinterpret_cast<const u32x4 *>(inc);          // n_pad is a multiple of 1024
    for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < n_pad / 4; g += gridDim.x * 256u) {
        const u32x4 a = inc4[g];
        s += (unsigned long long)a.x + a.y + a.z + a.w;
        mx = max(max(mx, a.x), max(max(a.y, a.z), a.w));
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    }
    if ((threadIdx.x & 63) == 0) {
        if (s) atomicAdd(reinterpret_cast<unsigned long long *>(hdr + 4), s);
        if (mx) atomicMax(hdr + 3, mx);
    }
}
// The rule of saw_bank_finalize_kernel on the loaded increments (exact: sum and maximum of the whole bank).
__global__ void saw_stats_init_kernel(uint32_t *__restrict__ hdr, uint32_t nvoices)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const unsigned long long I = saw_stats_sum_inc(hdr);
        hdr[0] = (hdr[3] < SAW_EVENTS_MAX_INC && I <= ((unsigned long long)nvoices << 27)) ? 1u : 0u;
    }
}
// Materialise every phase: state0 += T*inc (the host then resets T to 0).
__global__ __launch_bounds__(256)
void saw_materialize_kernel(const uint32_t *__restrict__ inc, uint32_t *__restrict__ state0,
                            uint32_t n_pad, uint32_t tbase)
{
    for (uint32_t v = blockIdx.x * 256u + threadIdx.x; v < n_pad; v += gridDim.x * 256u)
        state0[v] += tbase * inc[v];
}

// Publish a finished bus to the host WITHOUT a copy engine and without hipStreamSynchronize (round 3): one small
// workgroup at the end of the block's work on the stream writes the n sums to coherent pinned host memory, makes them
// visible system-wide and then writes the block's sequence number; the host polls that word (abi_saw.cpp).  Measured
// (tools/ubench/sync_latency.hip): launch + 256-byte hipMemcpyAsync + hipStreamSynchronize 15.2 us, launch + this
// kernel + poll 11.1 us, a producing kernel that publishes by itself 8.1 us.
__global__ __launch_bounds__(256)
void saw_publish_kernel(const int32_t *__restrict__ bus, int32_t *__restrict__ hbus, uint32_t *__restrict__ hflag,
                        uint32_t n, uint32_t seq)
{
    for (uint32_t i = threadIdx.x; i < n; i += 256u) __builtin_nontemporal_store(bus[i], &hbus[i]);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(hflag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The reference's own operating point -- struct synth's 64 voices, one process() block (linux/synth.c:169-202,
// 261-276) -- as ONE launch: the 64 {inc, state} pairs travel as kernel arguments (512 bytes: no upload), they are
// wave-uniform and live in scalar registers, one lane per FRAME sums (int)(state + t*inc) >> 4 over the voices
// that are on (the phasor is linear, so frame t needs no frame t-1), and the kernel publishes the bus to pinned
// host memory itself (see saw_publish_kernel).  Blocks of up to 1024 frames; one workgroup.
struct DropinVoices { uint32_t inc[64], state[64]; };
__global__ __launch_bounds__(1024)
void saw_dropin_kernel(DropinVoices a, int32_t *__restrict__ hbus, uint32_t *__restrict__ hflag, uint32_t n, uint32_t seq)
{
    const uint32_t t = threadIdx.x;
    if (t < n) {
        uint32_t sum = 0;
#pragma unroll
        for (int v = 0; v < 64; v++) {
            const uint32_t inc = a.inc[v];                                   // scalar: the same voice for every lane
            const int32_t ph = (int32_t)(a.state[v] + t * inc);
            sum += inc ? (uint32_t)(ph >> 4) : 0u;                           // inc == 0: the voice is off
        }
        __builtin_nontemporal_store((int32_t)sum, &hbus[t]);
    }
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(hflag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Grid: persistent workgroups, grid-stride over voices.  Two effects set the size:
//  * every workgroup ends with one integer atomic per frame on the same bus words, and those
//    serialise (~20 ns each): a 1 Mi-voice block of 64 frames takes 12.6 us with 256
//    workgroups and 23 us with 1024.  So a workgroup should own at least 4 rows of 256 lanes;
//  * the pure read stream of a few-frame block runs fastest with 3-4 workgroups per CU (64 Mi
//    voices, 1 frame: 768-1024 workgroups 6.8 TB/s, 512: 6.3, 2048: 6.0); from 8 frames up the
//    arithmetic wants every SIMD full (2048).
static uint32_t grid_size(uint32_t tc, uint32_t rows, uint32_t gy)
{
    static const char *env = getenv("SMX_SAW_GRID");      // tuning override
    uint32_t cap = env ? (uint32_t)atoi(env) : (tc <= 4 ? 1024u : 2048u);
    cap = (cap + gy - 1) / gy;
    uint32_t gx = rows / 4;
    if (gx < 128) gx = 128;
    if (gx > cap) gx = cap;
    if (gx > rows) gx = rows;
    return gx < 1 ? 1 : gx;
}

// Run the fold a slot launch left behind (smx::SawPending) as a kernel of its own.
int flush_pending(smx::SawPending *pend, hipStream_t stream, const smx::SawPublish *pub = nullptr, bool *published = nullptr)
{
    if (!pend || !pend->partial) return SMX_OK;
    smx::SawPublish p{};
    if (pub && pub->hflag && pend->nframes <= 64) {          // one workgroup ends the block: it may publish it
        p = *pub;
        if (published) *published = true;
    }
    hipLaunchKernelGGL(saw_direct_finalize_kernel, dim3((pend->nframes + 63) / 64), dim3(256), 0, stream,
                       static_cast<SawPartial *>(pend->partial), pend->bus, pend->bus_next, pend->nframes, p);
    pend->partial = nullptr;
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

template <int TC, int VW, bool NT>
int launch_tc(const uint32_t *inc, const uint32_t *si, int32_t *bus, int32_t *bus_next,
              uint32_t n_pad, uint32_t nframes, uint32_t tbase, SawPartial *partial, hipStream_t stream,
              smx::SawPending *pend)
{
    const uint32_t ngroups = n_pad / VW;
    const uint32_t gy = (nframes + TC - 1) / TC;                 // chunks of TC frames
    const uint32_t gy64 = (nframes + 63) / 64;                   // slot rows (one per 64 frames)
    const uint32_t rows = (ngroups + 255) / 256;
    if constexpr (VW == 4 && TC >= 8) {
        static const bool no_slots = getenv("SMX_SAW_NO_SLOTS") != nullptr;     // A/B switch
        if (partial && !no_slots) {
            // slots: the bus atomics no longer bound the workgroup count -- one row per workgroup
            // while the chip has room (8 workgroups per CU), grid-stride above that
            static const char *env = getenv("SMX_SAW_SLOT_GRID");               // tuning override
            // (5..8 frames are still a read stream first: 1024 workgroups, as in tick mode --
            // 64 Mi voices x 5 frames 79 us = 6.8 TB/s vs 90 us with 2048)
            uint32_t gx = ((env ? (uint32_t)atoi(env) : (TC <= 8 ? 1024u : 2048u)) + gy - 1) / gy;
            if (gx > rows) gx = rows;
            SawOwed owed{nullptr, nullptr, nullptr, 0};
            if (pend && pend->region_stride) {
                // deferred fold: this launch folds its predecessor's slots and fills the other region
                if (pend->partial && (pend->nframes + 63) / 64 > gx * gy) {
                    const int rv = flush_pending(pend, stream);
                    if (rv) return rv;
                }
                if (pend->partial)
                    owed = SawOwed{static_cast<SawPartial *>(pend->partial), pend->bus, pend->bus_next, pend->nframes};
                partial = reinterpret_cast<SawPartial *>(reinterpret_cast<char *>(partial) + pend->region * pend->region_stride);
            }
            hipLaunchKernelGGL((saw_bank_kernel<TC, VW, NT, true>), dim3(gx, gy), dim3(256), 0, stream,
                               inc, si, bus, bus_next, ngroups, nframes, tbase, partial, owed);
            if (pend && pend->region_stride) {
                pend->partial = partial;
                pend->bus = bus;
                pend->bus_next = bus_next;
                pend->nframes = nframes;
                pend->region ^= 1u;
            } else {
                hipLaunchKernelGGL(saw_direct_finalize_kernel, dim3(gy64), dim3(256), 0, stream, partial, bus,
                                   bus_next, nframes, smx::SawPublish{});
            }
            SMX_HIP(hipGetLastError());
            return SMX_OK;
        }
    }
    {
        const int rv = flush_pending(pend, stream);     // this launch adds to the bus the owed fold still has to zero
        if (rv) return rv;
    }
    const uint32_t gx = grid_size(TC, rows, gy);
    hipLaunchKernelGGL((saw_bank_kernel<TC, VW, NT, false>), dim3(gx, gy), dim3(256), 0, stream,
                       inc, si, bus, bus_next, ngroups, nframes, tbase, (SawPartial *)nullptr,
                       SawOwed{nullptr, nullptr, nullptr, 0});
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

// tc_cap: longest chunk (frames per workgroup pass); blocks longer than that run as several chunks on
// blockIdx.y.  64 for big banks; small banks take shorter chunks (launch_saw_bank).
template <int VW, bool NT>
int launch_vw(const uint32_t *inc, const uint32_t *si, int32_t *bus, int32_t *bus_next,
              uint32_t n_pad, uint32_t nframes, uint32_t tbase, SawPartial *partial, uint32_t tc_cap, hipStream_t stream,
              smx::SawPending *pend)
{
    const uint32_t nf = nframes < tc_cap ? nframes : tc_cap;
    if (nf > 32) return launch_tc<64, VW, NT>(inc, si, bus, bus_next, n_pad, nframes, tbase, partial, stream, pend);
    if (nf > 16) return launch_tc<32, VW, NT>(inc, si, bus, bus_next, n_pad, nframes, tbase, partial, stream, pend);
    if (nf > 8)  return launch_tc<16, VW, NT>(inc, si, bus, bus_next, n_pad, nframes, tbase, partial, stream, pend);
    if (nf > 4)  return launch_tc<8, VW, NT>(inc, si, bus, bus_next, n_pad, nframes, tbase, partial, stream, pend);
    if (nf > 2)  return launch_tc<4, VW, NT>(inc, si, bus, bus_next, n_pad, nframes, tbase, partial, stream, pend);
    if (nf > 1)  return launch_tc<2, VW, NT>(inc, si, bus, bus_next, n_pad, nframes, tbase, partial, stream, pend);
    return launch_tc<1, VW, NT>(inc, si, bus, bus_next, n_pad, nframes, tbase, partial, stream, pend);
}

}  // namespace

namespace smx {

size_t saw_scratch_header_bytes() { return SAW_SCRATCH_HEADER; }

size_t saw_scratch_region_bytes(uint32_t max_frames)
{
    return (size_t)SAW_SLOTS * ((max_frames + 63) / 64 + 1) * sizeof(SawPartial);
}

// header + two slot regions (the second one only for the deferred fold of the direct form's slot launches;
// the carry formulations use the area from the first region on, with nothing owed)
size_t saw_scratch_bytes(uint32_t max_frames) { return SAW_SCRATCH_HEADER + 2 * saw_scratch_region_bytes(max_frames); }

int launch_saw_flush(SawPending *pend, hipStream_t stream, const SawPublish *pub, bool *published)
{
    return flush_pending(pend, stream, pub, published);
}

int launch_saw_bank(const uint32_t *d_inc, const uint32_t *d_state_in, int32_t *d_bus,
                    int32_t *d_bus_next, uint32_t n_pad, uint32_t nframes, uint32_t tbase,
                    void *d_scratch, int long_block_form, uint32_t *host_flag, uint32_t host_tag,
                    hipStream_t stream, SawPending *pend, const SawPublish *pub, bool *published)
{
    if (n_pad == 0 || (n_pad & 1023) || nframes == 0) {
        set_error("launch_saw_bank: n_pad=%u nframes=%u", n_pad, nframes);
        return SMX_E_ARG;
    }
    static const bool no_carry = getenv("SMX_SAW_NO_CARRY") != nullptr;      // A/B switch
    // measured crossover on MI355X: the carry formulation's fixed cost (histogram fold, slot
    // atomics, further kernels) pays off from about 2^30 voice-samples per launch (16 Mi voices x
    // 64 frames: direct 78 us, stepping 77 us, wrap events 69 us; 8 Mi voices: 48 / 52 / 49 us)
    static const char *cm = getenv("SMX_SAW_CARRY_MIN_LOG2");           // tuning override
    static const unsigned carry_min_log2 = cm ? (unsigned)atoi(cm) : 30u;
    // (round 2, with the cheaper event form: 8 Mi voices x 64 frames 42.8 -> 40.6 us as well, x 128 frames 78 -> 70;
    // 4 Mi voices stay with the direct form: 24.6 vs 31.2 us)
    const bool big = (unsigned long long)n_pad * nframes >= (1ull << carry_min_log2) ||
                     (!cm && n_pad >= (1u << 23) && nframes >= 64);
    // (banks from 2^16 voices: 256 Ki voices x 4096 frames 70 -> 44 us, x 16384 frames 260 -> 120 us)
    // Blocks of 17..32 frames (round 3): under AUTO / EVENTS they take the same path as ONE 32-frame chunk -- the
    // event form's work goes with the number of wraps, and a piano-range bank wraps 0.56 times per voice in 32
    // frames (64 Mi voices x 32 frames: direct form 136 us = 49 % of HBM; see DESIGN 3.2b for the event form's
    // figure).  A caller that pins the stepping form keeps the direct form there (equal within 3 %, and its
    // fold is deferred).  SMX_SAW_NO_SHORT_EVENTS=1: as before (direct form up to 32 frames).
    static const bool no_short = getenv("SMX_SAW_NO_SHORT_EVENTS") != nullptr;          // A/B switch
    const bool short_chunk = nframes <= 32;
    const bool carry_frames = nframes > 32 || (nframes > 16 && !no_short && long_block_form != SMX_FORM_STEPPING);
    // (measured, piano-range banks, 17 / 24 / 32 frames alike -- the event form's cost is per voice, not per frame:
    // 2^26 voices 138 -> 109 us, 2^25 77 -> 65, 2^24 44.4 -> 42.0, 2^23 26.3 -> 30.5: from 2^25 voices.  A bank above
    // the rule's bound steps its 32 frames in 139.6 us against 138.5 for the direct form: profiles/r03_short_events.txt)
    const bool big_short = cm ? big : n_pad >= (1u << 25);
    if (carry_frames && n_pad >= (1u << 16) && (short_chunk ? big_short : big) && d_scratch && !no_carry) {
        // carry-count formulation: 1.5 vector ops per voice-sample
        const uint32_t ngroups = n_pad / 4;
        const uint32_t gy = (nframes + 63) / 64;
        static const char *cg = getenv("SMX_SAW_CARRY_GRID");           // tuning override
        // 4096 workgroups in total for multi-chunk launches; a single chunk runs 1-2 % faster with
        // 2048 (all resident at once) as long as that keeps the trip count within the counters' range
        uint32_t total = cg ? (uint32_t)atoi(cg) : 4096u;
        if (!cg && gy == 1 && (ngroups + 2048u * 256u - 1) / (2048u * 256u) <= 400) total = 2048u;
        // banks of up to 2^24 voices: half as many again (8 Mi voices x 64 frames 43.6 -> 40.9 us, 16 Mi 69.5 -> 66.6;
        // from 32 Mi voices up 1024 .. 2560 workgroups are within 1 %: profiles/r03_carry_grid.txt)
        const bool mid_bank = !cg && gy == 1 && n_pad <= (1u << 24);
        if (mid_bank) total = 1024u;
        uint32_t gx = (total + gy - 1) / gy;
        if (gx > (ngroups + 255) / 256) gx = (ngroups + 255) / 256;
        // the packed 16-bit scalar counters take 128 carries per trip: stay below 400 trips
        // (banks that would need more workgroups than the scratch holds use the direct form)
        const uint32_t trips = (ngroups + gx * 256u - 1) / (gx * 256u);
        if (trips <= 400 && (size_t)SAW_SLOTS * gy * sizeof(SawPartial) <= saw_scratch_region_bytes(nframes)) {
            {
                const int rv = flush_pending(pend, stream);       // the slots must be all zero, the bus cleared
                if (rv) return rv;
            }
            auto *flag = static_cast<uint32_t *>(d_scratch);
            auto *part = reinterpret_cast<SawPartial *>(static_cast<char *>(d_scratch) + SAW_SCRATCH_HEADER);   // all zero between launches
            const bool no_events = long_block_form == SMX_FORM_STEPPING;
            const bool force_events = long_block_form == SMX_FORM_EVENTS;
            // the event form's rows take unequal time: more, shorter workgroups balance better
            // (64 Mi voices: 64 frames 198 us with 2048, 189 with 4096; 1024 frames 2.68 / 2.56 ms with 4096 / 8192)
            // (banks of up to 2^24 voices, one chunk: 2048 -- 8 Mi voices x 64 frames 33.4 -> 30.5 us, 16 Mi 48.3 -> 46.5)
            uint32_t gx_ev = ((cg ? (uint32_t)atoi(cg) : (mid_bank ? 2048u : gy == 1 ? 4096u : 8192u)) + gy - 1) / gy;
            if (gx_ev > (ngroups + 255) / 256) gx_ev = (ngroups + 255) / 256;
            uint32_t *ran_long = flag + 1;                    // which slot layout the launch filled
            static const bool wide = getenv("SMX_SAW_NO_WIDE") == nullptr;          // A/B switch: carry masks instead of 64-bit pairs
            // the event form in 512-thread workgroups (8 instead of 6 waves per SIMD); A/B: SMX_SAW_EVENTS_256=1
            static const bool ev256_env = getenv("SMX_SAW_EVENTS_256") != nullptr;
            // (1024-thread workgroups measured too: equal -- piano-range bank 124.8 vs 125.6 us on one box)
            const bool ev512 = !ev256_env && (ngroups % 512u) == 0;
#define SMX_CARRY_LAUNCH_W(NT_, MULTI_, TC_, EV_, W_, FLAG_)                                                  \
    hipLaunchKernelGGL((saw_bank_carry_kernel<NT_, MULTI_, TC_, EV_, W_>), dim3((EV_) ? gx_ev : gx, gy),    \
                       dim3(256), 0, stream, d_inc, d_state_in, part, ngroups, tbase, FLAG_, ran_long, nframes)
#define SMX_CARRY_LAUNCH(NT_, MULTI_, TC_, EV_, FLAG_)                                                        \
    do {                                                                                                      \
        if ((EV_) && ev512)                                                                              \
            hipLaunchKernelGGL((saw_bank_carry_kernel<NT_, MULTI_, TC_, true, false, 512>),                   \
                               dim3((gx_ev + 1) / 2, gy), dim3(512), 0, stream, d_inc, d_state_in, part,      \
                               ngroups, tbase, FLAG_, ran_long, nframes);                                     \
        else if ((EV_) || !wide) SMX_CARRY_LAUNCH_W(NT_, MULTI_, TC_, EV_, false, FLAG_);                     \
        else                SMX_CARRY_LAUNCH_W(NT_, MULTI_, TC_, false, true, FLAG_);                         \
    } while (0)
            const bool nt = n_pad >= (1u << 24);
            static const bool no_long = getenv("SMX_SAW_NO_LONG_EVENTS") != nullptr;          // A/B switch
            static const char *lm = getenv("SMX_SAW_LONG_MIN");                               // tuning override (frames)
            static const uint32_t saw_long = lm ? (uint32_t)atoi(lm) : SAW_LONG_DEFAULT;
            if (nframes >= saw_long && !no_long) {
                // long launches: the stepping form in 64-frame chunks and the event form in 256-frame
                // chunks (divisions paid once per 256 frames) are queued, the flag picks one; the
                // finalize kernel reads from `ran_long` which slot layout was filled
                const uint32_t tl = nframes >= 1024 ? 1024u : 256u;          // chunk length of the event form
                const uint32_t gyl = (nframes + tl - 1) / tl;
                const uint32_t *f = (no_events || force_events) ? nullptr : flag;
                if (!force_events) {
                    if (nt) SMX_CARRY_LAUNCH(true, true, 64, false, f); else SMX_CARRY_LAUNCH(false, true, 64, false, f);
                }
                // 65..128 frames (round 3): the 64-frame event kernel with a 128-frame counting matrix in 1024-thread
                // workgroups (two slot rows, filled as two chunks would: the stepping form's layout) -- its per-voice
                // cost is the 64-frame kernel's, which the 256-frame kernel with its counting sort does not reach
                // (SMX_SAW_NO_EVENTS_128=1: the 256-frame chunk)
                static const bool no_ev128 = getenv("SMX_SAW_NO_EVENTS_128") != nullptr;          // A/B switch
                if (!no_events && nframes <= 128 && !no_ev128 && (ngroups % 1024u) == 0) {
                    uint32_t gx128 = (cg ? (uint32_t)atoi(cg) : (mid_bank ? 2048u : 4096u)) / 4u;     // 1024-thread workgroups
                    if (gx128 > ngroups / 1024u) gx128 = ngroups / 1024u;
                    if (gx128 < 1) gx128 = 1;
                    if (nt)
                        hipLaunchKernelGGL((saw_bank_carry_kernel<true, false, 128, true, false, 1024>), dim3(gx128), dim3(1024),
                                           0, stream, d_inc, d_state_in, part, ngroups, tbase, f, ran_long, nframes);
                    else
                        hipLaunchKernelGGL((saw_bank_carry_kernel<false, false, 128, true, false, 1024>), dim3(gx128), dim3(1024),
                                           0, stream, d_inc, d_state_in, part, ngroups, tbase, f, ran_long, nframes);
                } else if (!no_events) {
                    uint32_t gxl = ((cg ? (uint32_t)atoi(cg) : 8192u) + gyl - 1) / gyl;
                    if (gxl > (ngroups + 255) / 256) gxl = (ngroups + 255) / 256;
#define SMX_LONG_LAUNCH(NT_, TL_)                                                                            \
    hipLaunchKernelGGL((saw_bank_event_long_kernel<NT_, TL_>), dim3(gxl, gyl), dim3(256), 0, stream, d_inc, \
                       d_state_in, reinterpret_cast<SawPartialL<TL_> *>(part), ngroups, tbase, f, ran_long, nframes)
                    if (tl == 1024) { if (nt) SMX_LONG_LAUNCH(true, 1024); else SMX_LONG_LAUNCH(false, 1024); }
                    else            { if (nt) SMX_LONG_LAUNCH(true, 256);  else SMX_LONG_LAUNCH(false, 256); }
#undef SMX_LONG_LAUNCH
                }
            } else if (short_chunk) {
                // one 32-frame chunk (17..32 frames): as below
                const uint32_t *f = force_events ? nullptr : flag;
                if (!force_events) { if (nt) SMX_CARRY_LAUNCH(true, false, 32, false, f); else SMX_CARRY_LAUNCH(false, false, 32, false, f); }
                if (nt) SMX_CARRY_LAUNCH(true, false, 32, true, f); else SMX_CARRY_LAUNCH(false, false, 32, true, f);
            } else {
                // 64-frame chunks: both forms are queued, the device-side flag picks one (the other
                // returns at once); the finalize kernel refreshes the flag from this launch's statistics
                const uint32_t *f = (no_events || force_events) ? nullptr : flag;
                if (!force_events) {
                    if (gy > 1) { if (nt) SMX_CARRY_LAUNCH(true, true, 64, false, f);  else SMX_CARRY_LAUNCH(false, true, 64, false, f); }
                    else        { if (nt) SMX_CARRY_LAUNCH(true, false, 64, false, f); else SMX_CARRY_LAUNCH(false, false, 64, false, f); }
                }
                if (!no_events) {
                    if (gy > 1) { if (nt) SMX_CARRY_LAUNCH(true, true, 64, true, f);  else SMX_CARRY_LAUNCH(false, true, 64, true, f); }
                    else        { if (nt) SMX_CARRY_LAUNCH(true, false, 64, true, f); else SMX_CARRY_LAUNCH(false, false, 64, true, f); }
                }
            }
#undef SMX_CARRY_LAUNCH
#undef SMX_CARRY_LAUNCH_W
            // a single 64-frame chunk ends in ONE workgroup of the finalize kernel: it may hand the bus to the host
            SawPublish fin_pub{};
            const bool long_layout_possible = nframes >= saw_long && !no_long;
            if (pub && pub->hflag && gy == 1 && !long_layout_possible) {
                fin_pub = *pub;
                if (published) *published = true;
            }
            hipLaunchKernelGGL(saw_bank_finalize_kernel, dim3(gy), dim3(256), 0, stream, part, d_bus,
                               d_bus_next, nframes, n_pad, flag, long_layout_possible ? ran_long : nullptr, host_flag,
                               host_tag, fin_pub);
            SMX_HIP(hipGetLastError());
            return SMX_OK;
        }
    }
    static const char *tm = getenv("SMX_SAW_TICK_MAX_LOG2");           // tuning override
    static const unsigned tick_max_log2 = tm ? (unsigned)atoi(tm) : 33u;     // no upper bound (round 2: see below)
    if (nframes <= 4 && n_pad >= (1u << 20) && (unsigned long long)n_pad < (1ull << tick_max_log2) && (n_pad & 4095) == 0) {
        // tick ABI on a bank of 2^20 voices or more: 1024-thread streaming workgroups, >= 4 rows each, 256
        // of them = one per CU (measured: 16 Mi voices 19.6 us vs 27.4 us with 256-thread workgroups; 64 Mi
        // voices 74.7 us = 7.18 TB/s vs 78.3 us for the 1024 x 256-thread grid of saw_bank_kernel, 256 Mi
        // voices 302 vs 317 us, 32 and 128 Mi voices equal; a grid that is not a multiple of the 256 CUs,
        // e.g. 320, loses 35 %)
        const uint32_t ngroups = n_pad / 4, nrows = ngroups >> 10;
        static const char *tg = getenv("SMX_SAW_TICK_GRID");           // tuning override
        uint32_t gx = nrows / 4;
        const uint32_t cap = tg ? (uint32_t)atoi(tg) : 256u;
        if (gx > cap) gx = cap;
        if (gx < 1) gx = 1;
        const bool nt = n_pad >= (1u << 24);
        {
            const int rv = flush_pending(pend, stream);
            if (rv) return rv;
        }
#define SMX_TICK_LAUNCH(TC_, NT_)                                                              \
    hipLaunchKernelGGL((saw_tick_kernel<TC_, NT_>), dim3(gx), dim3(1024), 0, stream, d_inc,    \
                       d_state_in, d_bus, d_bus_next, ngroups, nframes, tbase)
        if (nframes == 1)      { if (nt) SMX_TICK_LAUNCH(1, true); else SMX_TICK_LAUNCH(1, false); }
        else if (nframes == 2) { if (nt) SMX_TICK_LAUNCH(2, true); else SMX_TICK_LAUNCH(2, false); }
        else                   { if (nt) SMX_TICK_LAUNCH(4, true); else SMX_TICK_LAUNCH(4, false); }
#undef SMX_TICK_LAUNCH
        SMX_HIP(hipGetLastError());
        return SMX_OK;
    }
    // 4 voices per lane once there are enough voices to fill the chip that way;
    // non-temporal streaming once the bank (12 B/voice) cannot live in the 256 MiB
    // Infinity Cache between launches anyway
    // slots need one SawPartial row per 64-frame chunk in the scratch
    SawPartial *part = nullptr;
    if (d_scratch && (size_t)SAW_SLOTS * ((nframes + 63) / 64) * sizeof(SawPartial) <= saw_scratch_region_bytes(nframes))
        part = reinterpret_cast<SawPartial *>(static_cast<char *>(d_scratch) + SAW_SCRATCH_HEADER);
    static const char *tcc = getenv("SMX_SAW_TC_CAP");                  // tuning overrides
    static const char *svw = getenv("SMX_SAW_SMALL_VW");
    static const char *stc = getenv("SMX_SAW_SMALL_TC");
    // 2 Mi .. 8 Mi voices: 16-frame chunks on blockIdx.y.  A 64-frame chunk keeps 64 per-frame sums per lane (100
    // vector registers: 5 waves per SIMD) and gives a 2 Mi-voice bank only 2048 workgroups; four 16-frame chunks are
    // 8192 workgroups of a 61-register kernel, and a bank of this size is re-read from the Infinity Cache, not from HBM
    // (tools/explore_tc_cap.py, 64-frame blocks: 2 Mi voices 15.1 -> 12.0 us, 4 Mi 23.7 -> 20.9 us; 1 Mi voices equal,
    // 8.0 us either way; banks from 16 Mi voices up would read HBM once per chunk and keep 64)
    const uint32_t tc_cap = tcc ? (uint32_t)atoi(tcc) : (n_pad >= (1u << 21) && n_pad < (1u << 24)) ? 16u : 64u;
    if (n_pad >= (1u << 24))
        return launch_vw<4, true>(d_inc, d_state_in, d_bus, d_bus_next, n_pad, nframes, tbase, part, tc_cap, stream, pend);
    if (n_pad >= (1u << 20))
        return launch_vw<4, false>(d_inc, d_state_in, d_bus, d_bus_next, n_pad, nframes, tbase, part, tc_cap, stream, pend);
    // Small banks (< 2^20 voices): 4 voices per lane as well (16-byte loads) and chunks of 16 frames on
    // blockIdx.y, so that a 64-frame block of 65 536 voices is 64 x 4 workgroups with 64 bus atomics per
    // frame instead of 128 x 1 with 128: the launch is bound by the same-address atomics at its end
    // (measured, 64 frames: 2^14 voices 4.0 -> 3.2 us, 2^16 5.7 -> 3.4 us, 2^18 9.2 -> 5.3 us).
    const uint32_t small_tc = stc ? (uint32_t)atoi(stc) : 16u;
    const bool small_vw4 = svw ? atoi(svw) == 4 : n_pad >= (1u << 13);
    if (small_vw4)
        return launch_vw<4, false>(d_inc, d_state_in, d_bus, d_bus_next, n_pad, nframes, tbase, nullptr, small_tc, stream, pend);
    return launch_vw<1, false>(d_inc, d_state_in, d_bus, d_bus_next, n_pad, nframes, tbase, nullptr, small_tc, stream, pend);
}

int launch_square_bank(const uint32_t *d_inc, const uint32_t *d_state_in, uint32_t *d_or_bus,
                       uint32_t n_pad, uint32_t nframes, uint32_t tbase, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || nframes == 0) return SMX_E_ARG;
    uint32_t gx = n_pad / 256;
    const uint32_t gy = (nframes + 63) / 64;
    const uint32_t cap = (2048 + gy - 1) / gy;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(square_bank_kernel, dim3(gx, gy), dim3(256), 0, stream,
                       d_inc, d_state_in, d_or_bus, n_pad, nframes, tbase);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_saw_rebase(uint32_t *d_inc, uint32_t *d_state0, uint32_t voice, uint32_t new_inc,
                      uint32_t tbase, void *d_scratch, uint32_t n_pad, hipStream_t stream)
{
    hipLaunchKernelGGL(saw_rebase_kernel, dim3(1), dim3(64), 0, stream, d_inc, d_state0, voice, new_inc, tbase,
                       static_cast<uint32_t *>(d_scratch), n_pad);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_saw_rebase_batch(uint32_t *d_inc, uint32_t *d_state0, const uint32_t *d_pairs, uint32_t npairs,
                            uint32_t tbase, void *d_scratch, uint32_t n_pad, hipStream_t stream)
{
    if (npairs == 0) return SMX_OK;
    hipLaunchKernelGGL(saw_rebase_batch_kernel, dim3((npairs + 255) / 256), dim3(256), 0, stream, d_inc,
                       d_state0, d_pairs, npairs, tbase, static_cast<uint32_t *>(d_scratch), n_pad);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_saw_sum_inc(const uint32_t *d_inc, uint32_t n_pad, void *d_scratch, hipStream_t stream)
{
    if (!d_scratch) return SMX_OK;
    uint32_t gx = n_pad / 1024;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(saw_sum_inc_kernel, dim3(gx), dim3(256), 0, stream, d_inc, n_pad, static_cast<uint32_t *>(d_scratch));
    hipLaunchKernelGGL(saw_stats_init_kernel, dim3(1), dim3(64), 0, stream, static_cast<uint32_t *>(d_scratch), n_pad);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_saw_publish(const int32_t *d_bus, int32_t *d_hbus, uint32_t *d_hflag, uint32_t n, uint32_t seq, hipStream_t stream)
{
    hipLaunchKernelGGL(saw_publish_kernel, dim3(1), dim3(256), 0, stream, d_bus, d_hbus, d_hflag, n, seq);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_saw_dropin(const uint32_t inc[64], const uint32_t state[64], int32_t *d_hbus, uint32_t *d_hflag, uint32_t n,
                      uint32_t seq, hipStream_t stream)
{
    if (n == 0 || n > 1024) { set_error("launch_saw_dropin: n=%u (1..1024)", n); return SMX_E_ARG; }
    DropinVoices a;
    memcpy(a.inc, inc, sizeof(a.inc));
    memcpy(a.state, state, sizeof(a.state));
    hipLaunchKernelGGL(saw_dropin_kernel, dim3(1), dim3((n + 63u) & ~63u), 0, stream, a, d_hbus, d_hflag, n, seq);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_saw_materialize(const uint32_t *d_inc, uint32_t *d_state0, uint32_t n_pad, uint32_t tbase,
                           hipStream_t stream)
{
    uint32_t gx = n_pad / 256;
    if (gx > 2048) gx = 2048;
    hipLaunchKernelGGL(saw_materialize_kernel, dim3(gx), dim3(256), 0, stream, d_inc, d_state0, n_pad, tbase);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
