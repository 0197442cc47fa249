// poly_bank.hip -- poly voice bank: saw phasor -> 1-pole low-pass -> ADSR gain ->
// fixed-point stereo mix, for gfx950 (MI355X).  BASELINE config 4.
//
// BUILD-DEFINED EXTENSION.  The reference has no filter and no envelope (only a
// FIXME at linux/synth.c:150-152); SURVEY.md §8 a-9 asks the build to specify
// them.  The definition lives in oracle/synth_oracle.c (orc_poly_run) and is
// restated here operation for operation:
//   x = (float)(int32)phase * 2^-31;  phase += inc            (inc == 0: voice off)
//   t = x - y;  y = y + a*t                                   (two roundings, never fused)
//   ADSR on a u32 level with per-sample u32 rates (integer state machine)
//   o = y * ((float)(level >> 8) * 2^-24);  q = (int32)(o * 2^19)   (cvt toward zero)
//   bus_l += q * pan_l;  bus_r += q * pan_r                   (wrapping int32, pan 0..256)
// The integer mix keeps the result order-independent, like the saw bank's.
//
// Mapping: one lane per voice, all voice state in registers for the block
// (44 B read, 16 B written per voice per launch), per-frame stereo partial sums
// accumulated with conflict-free LDS atomics into M[2*frames][64+1], folded by
// 4 lanes per row.  Integer atomics on ONE address serialise at ~20 ns each, and
// 1024 workgroups ending together on the same 128 bus words cost 20 us of the
// first version's 33 us launch (config 4): a workgroup therefore adds its rows into one of
// POLY_SLOTS copies of the bus (32 atomics per address at most).
//
// Round 3: the fold of those copies is DEFERRED (like the saw bank's slot fold, saw_bank.hip): a launch does not
// queue a second kernel behind itself any more -- the NEXT launch folds its predecessor's copies into the
// predecessor's bus while its own voices' loads are in flight (every workgroup takes a few of the 128 rows x 8
// groups of 4 copies: one 4-word read, one atomic on the bus word, four zeroing stores), the scratch holds two
// regions of copies used alternately, and whoever needs a bus before the next launch (smx_poly_run's copy, a
// sync) runs poly_finalize_kernel on the spot (abi_poly.cpp).  A stream of un-fetched blocks is one kernel per
// block instead of two.  The 64-frame block (the JACK operating point) has its own instantiation with the frame
// loop unrolled (LDS offsets become immediates, no loop counter, no pointer arithmetic), and the voice loads are
// issued before the LDS matrix is cleared.
//
// Per voice-sample (all bit-exact restatements of the definition above):
//   t  = fma((float)(int)phase, 2^-31, -y)   == x - y: the product is exact, one rounding
//   ADSR: arrived = (level ^ flip) < thr; level = arrived ? reach : level + delta  (4 operations in
//   every stage; flip/thr/delta/reach change on arrival, behind a wave-uniform branch)
//   q  = (int)(y * ((float)(level >> 8) * 2^-5))  == (int)((y*g) * 2^19): power-of-two scalings
//        commute with the rounding (results below 2^-107 truncate to 0 either way)
//   q*pan with v_mul_i32_i24 (|q| <= 2^19+, pan < 2^16: the low 32 bits are the int32 product)
#include "smx_common.h"

namespace {

enum { ENV_IDLE = 0, ENV_A = 1, ENV_D = 2, ENV_S = 3, ENV_R = 4 };
enum { POLY_SLOTS = 32, POLY_ROWS = 128, POLY_PARTS = 8 /* groups of 4 copies */ };

// What the previous launch left to this one: its copies (to be folded into its bus and cleared).
struct PolyOwed {
    int32_t *slots;      // nullptr: nothing owed
    int32_t *bus;        // zeroed by the launch that filled `slots`
};

template <int NT, bool F64>
__global__ __launch_bounds__(NT)
void poly_bank_kernel(smx::PolyArrays p, int32_t *__restrict__ slots /* [POLY_SLOTS][128], zero */,
                      int32_t *__restrict__ own_bus /* cleared here when the fold of THIS launch is deferred; or nullptr */,
                      PolyOwed owed, uint32_t n_pad, uint32_t nframes_rt /* <= 64 */)
{
    __shared__ int32_t M[POLY_ROWS][65];
    const uint32_t nframes = F64 ? 64u : nframes_rt;
    const uint32_t tid = threadIdx.x, lane = tid & 63;

    // (0) the predecessor's fold, spread over all workgroups: unit u = (row u & 127, copies 4*(u >> 7) .. +3)
    if (owed.slots) {
        for (uint32_t u = blockIdx.x + gridDim.x * tid; u < POLY_ROWS * POLY_PARTS; u += gridDim.x * NT) {
            const uint32_t row = u & (POLY_ROWS - 1), part = u >> 7;
            int32_t *s = owed.slots + (size_t)part * 4 * POLY_ROWS + row;
            const int32_t a0 = s[0], a1 = s[POLY_ROWS], a2 = s[2 * POLY_ROWS], a3 = s[3 * POLY_ROWS];
            s[0] = 0; s[POLY_ROWS] = 0; s[2 * POLY_ROWS] = 0; s[3 * POLY_ROWS] = 0;
            const int32_t sum = a0 + a1 + a2 + a3;
            if (sum) atomicAdd(&owed.bus[row], sum);
        }
    }
    if (own_bus && blockIdx.x == 0 && tid < POLY_ROWS) own_bus[tid] = 0;     // the next launch adds this block's sums here

    // (1) this workgroup's first row of voices: all twelve loads in flight while the LDS matrix is cleared
    // (gridDim.x * NT <= n_pad, so the first row always exists)
    struct Voice {
        uint32_t inc, phase, level, stage, ar, dr, sl, rr, pan, gate;
        float y, a;
    };
    auto load = [&](uint32_t v) {
        Voice w;
        w.inc = p.inc[v]; w.phase = p.phase[v]; w.level = p.level[v]; w.stage = p.stage[v];
        w.y = p.y[v]; w.a = p.a[v]; w.ar = p.ar[v]; w.dr = p.dr[v]; w.sl = p.sl[v]; w.rr = p.rr[v];
        w.pan = p.pan[v]; w.gate = p.gate[v];
        return w;
    };
    uint32_t v = blockIdx.x * (uint32_t)NT + tid;
    Voice w = load(v);
    for (uint32_t i = tid; i < 2 * nframes * 65; i += NT) (&M[0][0])[i] = 0;
    __syncthreads();

    for (;;) {
      if (w.inc) {                                           // 0 == off: frozen, silent
        const uint32_t inc = w.inc;
        uint32_t phase = w.phase, level = w.level, stage = w.stage;
        float y = w.y;
        const float a = w.a;
        const uint32_t ar = w.ar, dr = w.dr, sl = w.sl, rr = w.rr;
        const uint32_t pan = w.pan;
        int32_t pl = (int32_t)(pan & 0xFFFF), pr = (int32_t)(pan >> 16);
        asm volatile("" : "+v"(pl), "+v"(pr));               // registers of their own: plain v_mul_i32_i24, no SDWA selects
        // gate: control-rate input, sampled at the start of the block
        if (w.gate) { if (stage == ENV_IDLE || stage == ENV_R) stage = ENV_A; }
        else        { if (stage != ENV_IDLE) stage = ENV_R; }

        // ADSR as "arrived = (level ^ flip) < thr; level = arrived ? reach_val : level + delta; on
        // arrival enter the next stage" -- the same four vector operations in every stage:
        //   A (ar > 0): arrives when level + ar wraps  <=>  ~level < ar      -> level MAX, next D
        //   D: arrives when level <= sl + dr           <=>  level < sl+dr+1  -> level sl,  next S
        //   R: arrives when level <= rr                <=>  level < rr+1     -> level 0,   next idle
        //   hold (S, idle, A with ar 0, D/R with rate 0 above their target): thr 0, never arrives
        //   a D/R stage whose bound sl+dr+1 does not fit 32 bits, or that has rate 0 and sits at or
        //   below its target, arrives on the next frame whatever the level: thr MAX, and flip chosen
        //   so that the (unchanging) level compares below it.
        // Round 3: a stage that only goes DOWN or holds (D, S, R, idle, A with ar 0) is also
        //   level = max(level -sat down, floor)     (D: down dr, floor sl; R: down rr, floor 0; holds: 0, 0)
        // two operations, no comparison, and its arrival needs no detection at all: the level simply stays at its
        // floor, which is what the next stage (S / idle) holds anyway; the stage word is set right after the block
        // (D that sits at sl is S, R that sits at 0 is idle -- a D/R that has not arrived is strictly above its
        // target).  Only a rising attack needs the general form, so 8-frame chunks run it while some lane of the wave
        // is attacking and the short form otherwise (an attack can only begin at a block's start: the gate is
        // sampled there).
        uint32_t flip, thr, delta, reach_val, next, down, floor_;
        auto enter = [&](uint32_t st) {
            // (masks instead of select chains: the compiler turns a chain of `st == K ? x : ...` over several
            // variables into a table in scratch memory indexed by st -- 168 bytes of private segment and a launch
            // three times as long)
            const uint32_t isA = 0u - (uint32_t)(st == ENV_A), isD = 0u - (uint32_t)(st == ENV_D);
            const uint32_t isR = 0u - (uint32_t)(st == ENV_R), isS = 0u - (uint32_t)(st == ENV_S);
            const uint32_t isI = 0u - (uint32_t)(st == ENV_IDLE);
            stage = st;
            level = (level & ~(isS | isI)) | (sl & isS);                 // S: level = sl; idle: level = 0
            const uint32_t rate = (ar & isA) | (dr & isD) | (rr & isR);
            const uint32_t target = sl & isD;
            down = (dr & isD) | (rr & isR);
            floor_ = target;
            const bool is_down = (isD | isR) != 0u;
            const uint64_t lim = (uint64_t)target + rate + 1;
            const bool always = is_down && (rate ? lim > 0xFFFFFFFFull : level <= target);
            const bool hold = !always && rate == 0;
            flip = isA;
            thr = isA ? ar : (uint32_t)lim;
            delta = isA ? ar : 0u - rate;
            if (always) { thr = 0xFFFFFFFFu; flip = level == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u; }
            if (hold) { thr = 0; flip = 0; delta = 0; }
            reach_val = isA | target;                                   // A: MAX; D: sl; R: 0
            next = (ENV_D & isA) | (ENV_S & isD);                        // A -> D, D -> S, R -> idle (0)
        };
        enter(stage);

        int32_t *m = &M[0][lane];
        auto voice_out = [&](uint32_t i) {
            const float g = __fmul_rn((float)(level >> 8), 0x1p-5f);
            const int32_t q = (int32_t)__fmul_rn(y, g);
            atomicAdd(m + 130 * i, __mul24(q, pl));
            atomicAdd(m + 130 * i + 65, __mul24(q, pr));
        };
        auto filter = [&]() {
            const float t = __fmaf_rn((float)(int32_t)phase, 0x1p-31f, -y);
            phase += inc;
            y = __fadd_rn(y, __fmul_rn(a, t));
        };
        auto frame = [&](uint32_t i) {                                  // any stage
            filter();
            const bool arrived = (level ^ flip) < thr;
            const uint32_t nl = level + delta;
            level = arrived ? reach_val : nl;
            if (__builtin_expect(__ballot(arrived) != 0ull, 0)) {       // rare: a lane of the wave changes its stage
                if (arrived) enter(next);
            }
            voice_out(i);
        };
        auto frame_down = [&](uint32_t i) {                             // no lane of the wave is attacking
            filter();
            level = max(__builtin_elementwise_sub_sat(level, down), floor_);
            voice_out(i);
        };
        bool rising = __any(stage == ENV_A && ar != 0u);
        bool fast_tail;                                                 // did the block's last frames run the short form?
        if (F64) {
            for (uint32_t i0 = 0; i0 < 64; i0 += 8) {
                fast_tail = !rising;
                if (rising) {
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) frame(i0 + j);
                    rising = __any(stage == ENV_A && ar != 0u);
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < 8; j++) frame_down(i0 + j);
                }
            }
        } else {
            fast_tail = !rising;
            for (uint32_t i = 0; i < nframes; i++) {
                if (rising) frame(i); else frame_down(i);
            }
        }
        // Arrivals the short form did not announce: a D that sits at sl is S, an R that sits at 0 is idle.  Only after
        // frames of the short form (a lane then spent at least one frame in its stage): an attack that arrives in the
        // block's very last frame of the general form has just ENTERED D and must stay D whatever its level.
        if (fast_tail) {
            if (stage == ENV_D && level == sl) stage = ENV_S;
            if (stage == ENV_R && level == 0u) stage = ENV_IDLE;
        }
        p.phase[v] = phase; p.level[v] = level; p.stage[v] = stage; p.y[v] = y;
      }
      v += gridDim.x * NT;
      if (v >= n_pad) break;
      w = load(v);
    }

    __syncthreads();
    // rows (frame, channel): passes of NT/4 rows, 4 lanes x 16 columns per row
    int32_t *slot = slots + (blockIdx.x % POLY_SLOTS) * POLY_ROWS;
    for (uint32_t r0 = 0; r0 < 2 * nframes; r0 += NT / 4) {
        const uint32_t row = r0 + (tid >> 2), q4 = tid & 3;
        int32_t s = 0;
        if (row < POLY_ROWS) {
#pragma unroll
            for (int j = 0; j < 16; j++) s += M[row][q4 * 16 + j];
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q4 == 0 && row < 2 * nframes && s) atomicAdd(&slot[row], s);
    }
}

// Fold on the spot (somebody needs the bus now): bus = sum of the copies, copies cleared.
__global__ __launch_bounds__(128)
void poly_finalize_kernel(int32_t *__restrict__ slots, int32_t *__restrict__ bus_lr)
{
    const uint32_t row = threadIdx.x;
    int32_t v[POLY_SLOTS];
#pragma unroll
    for (int k = 0; k < POLY_SLOTS; k++) v[k] = slots[k * POLY_ROWS + row];
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < POLY_SLOTS; k++) { s += v[k]; slots[k * POLY_ROWS + row] = 0; }
    bus_lr[row] = s;
}

template <int NT>
void poly_launch(bool f64, uint32_t gx, hipStream_t stream, const smx::PolyArrays &p, int32_t *slots, int32_t *own_bus,
                 PolyOwed owed, uint32_t n_pad, uint32_t nframes)
{
    if (f64) hipLaunchKernelGGL((poly_bank_kernel<NT, true>), dim3(gx), dim3(NT), 0, stream, p, slots, own_bus, owed, n_pad, nframes);
    else     hipLaunchKernelGGL((poly_bank_kernel<NT, false>), dim3(gx), dim3(NT), 0, stream, p, slots, own_bus, owed, n_pad, nframes);
}

}  // namespace

namespace smx {

size_t poly_region_bytes() { return (size_t)POLY_SLOTS * POLY_ROWS * sizeof(int32_t); }
size_t poly_scratch_bytes() { return 2 * poly_region_bytes(); }
size_t poly_bus_bytes() { return (size_t)POLY_ROWS * sizeof(int32_t); }

int launch_poly_flush(PolyPending *pend, hipStream_t stream)
{
    if (!pend || !pend->slots) return SMX_OK;
    hipLaunchKernelGGL(poly_finalize_kernel, dim3(1), dim3(POLY_ROWS), 0, stream, pend->slots, pend->bus);
    pend->slots = nullptr;
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

int launch_poly_bank(const PolyArrays &p, int32_t *d_bus_lr, int32_t *d_slots, uint32_t n_pad,
                     uint32_t nframes, hipStream_t stream, PolyPending *pend)
{
    if (n_pad == 0 || (n_pad & 1023) || nframes == 0 || nframes > 64) {
        set_error("launch_poly_bank: n_pad=%u nframes=%u", n_pad, nframes);
        return SMX_E_ARG;
    }
    // ~18 vector ops per voice-sample on a 3-deep fp dependency chain: one row of NT voices per workgroup while
    // the chip has room (LDS: 4 workgroups of 33 KB per CU), grid-stride above that.
    static const uint32_t env_nt = [] { const char *e = getenv("SMX_POLY_NT"); return e ? (uint32_t)atoi(e) : 0u; }();
    static const uint32_t env_gx = [] { const char *e = getenv("SMX_POLY_GRID"); return e ? (uint32_t)atoi(e) : 0u; }();
    // 512 threads: 8 waves share one LDS matrix and one fold (measured against 256 / 1024, tools/explore_poly.py:
    // 256 Ki voices x 64 frames 11.8 / 12.2 / 11.9 us, 4 Mi voices 93 / 107 / 93 us)
    const uint32_t nt = (env_nt == 512 || env_nt == 1024 || env_nt == 256) ? env_nt : 512u;
    const uint32_t rows = n_pad / nt;
    uint32_t gx = rows;
    const uint32_t cap = nt == 1024 ? 512u : 1024u;       // resident workgroups: 4 per CU (LDS), 8 waves per SIMD
    if (gx > cap) gx = cap;
    if (env_gx) gx = env_gx < rows ? env_gx : rows;
    // the region this launch fills; with `pend` the fold is left to the next launch (or to launch_poly_flush)
    PolyOwed owed{nullptr, nullptr};
    int32_t *slots = d_slots, *own_bus = nullptr;
    if (pend) {
        if (pend->slots) { owed.slots = pend->slots; owed.bus = pend->bus; }
        slots = d_slots + (size_t)pend->region * POLY_SLOTS * POLY_ROWS;
        own_bus = d_bus_lr;
    }
    const bool f64 = nframes == 64;
    if (nt == 256)      poly_launch<256>(f64, gx, stream, p, slots, own_bus, owed, n_pad, nframes);
    else if (nt == 512) poly_launch<512>(f64, gx, stream, p, slots, own_bus, owed, n_pad, nframes);
    else                poly_launch<1024>(f64, gx, stream, p, slots, own_bus, owed, n_pad, nframes);
    if (pend) {
        pend->slots = slots;
        pend->bus = d_bus_lr;
        pend->region ^= 1u;
    } else {
        hipLaunchKernelGGL(poly_finalize_kernel, dim3(1), dim3(POLY_ROWS), 0, stream, slots, d_bus_lr);
    }
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
