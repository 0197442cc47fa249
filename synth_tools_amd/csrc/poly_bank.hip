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
// former 33 us launch (config 4): a workgroup therefore adds its rows into one of
// POLY_SLOTS copies of the bus (32 atomics per address), and poly_finalize_kernel
// folds the copies, writes the bus and re-zeroes the slots for the next launch.
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
enum { POLY_SLOTS = 32 };

__global__ __launch_bounds__(256)
void poly_bank_kernel(smx::PolyArrays p, int32_t *__restrict__ slots /* [POLY_SLOTS][128], zero */,
                      uint32_t n_pad, uint32_t nframes /* <= 64 */)
{
    __shared__ int32_t M[128][65];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i < 2 * nframes * 65; i += 256) (&M[0][0])[i] = 0;
    __syncthreads();

    for (uint32_t v = blockIdx.x * 256u + tid; v < n_pad; v += gridDim.x * 256u) {
        const uint32_t inc = p.inc[v];
        if (!inc) continue;                                  // 0 == off: frozen, silent
        uint32_t phase = p.phase[v], level = p.level[v], stage = p.stage[v];
        float y = p.y[v];
        const float a = p.a[v];
        const uint32_t ar = p.ar[v], dr = p.dr[v], sl = p.sl[v], rr = p.rr[v];
        const uint32_t pan = p.pan[v];
        const int32_t pl = (int32_t)(pan & 0xFFFF), pr = (int32_t)(pan >> 16);
        // gate: control-rate input, sampled at the start of the block
        if (p.gate[v]) { if (stage == ENV_IDLE || stage == ENV_R) stage = ENV_A; }
        else           { if (stage != ENV_IDLE) stage = ENV_R; }

        // ADSR as "arrived = (level ^ flip) < thr; level = arrived ? reach_val : level + delta; on
        // arrival enter the next stage" -- the same four vector operations in every stage:
        //   A (ar > 0): arrives when level + ar wraps  <=>  ~level < ar      -> level MAX, next D
        //   D: arrives when level <= sl + dr           <=>  level < sl+dr+1  -> level sl,  next S
        //   R: arrives when level <= rr                <=>  level < rr+1     -> level 0,   next idle
        //   hold (S, idle, A with ar 0, D/R with rate 0 above their target): thr 0, never arrives
        //   a D/R stage whose bound sl+dr+1 does not fit 32 bits, or that has rate 0 and sits at or
        //   below its target, arrives on the next frame whatever the level: thr MAX, and flip chosen
        //   so that the (unchanging) level compares below it.
        uint32_t flip, thr, delta, reach_val, next;
        auto enter = [&](uint32_t st) {
            stage = st;
            if (st == ENV_S) level = sl;
            if (st == ENV_IDLE) level = 0;
            const uint32_t rate = st == ENV_A ? ar : st == ENV_D ? dr : st == ENV_R ? rr : 0u;
            const uint32_t target = st == ENV_D ? sl : 0u;
            const bool down = st == ENV_D || st == ENV_R;
            const uint64_t lim = (uint64_t)target + rate + 1;
            const bool always = down && (rate ? lim > 0xFFFFFFFFull : level <= target);
            const bool hold = !always && rate == 0;
            flip = (st == ENV_A) ? 0xFFFFFFFFu : 0u;
            thr = st == ENV_A ? ar : (uint32_t)lim;
            delta = st == ENV_A ? ar : 0u - rate;
            if (always) { thr = 0xFFFFFFFFu; flip = level == 0xFFFFFFFFu ? 0xFFFFFFFFu : 0u; }
            if (hold) { thr = 0; flip = 0; delta = 0; }
            reach_val = st == ENV_A ? 0xFFFFFFFFu : target;
            next = st == ENV_A ? (uint32_t)ENV_D : st == ENV_D ? (uint32_t)ENV_S : (uint32_t)ENV_IDLE;
        };
        enter(stage);

        int32_t *m = &M[0][lane];
        for (uint32_t i = 0; i < nframes; i++) {
            const float t = __fmaf_rn((float)(int32_t)phase, 0x1p-31f, -y);
            phase += inc;
            y = __fadd_rn(y, __fmul_rn(a, t));
            const bool arrived = (level ^ flip) < thr;
            const uint32_t nl = level + delta;
            level = arrived ? reach_val : nl;
            if (__any(arrived)) {
                if (arrived) enter(next);
            }
            const float g = __fmul_rn((float)(level >> 8), 0x1p-5f);
            const int32_t q = (int32_t)__fmul_rn(y, g);
            atomicAdd(m, __mul24(q, pl));
            atomicAdd(m + 65, __mul24(q, pr));
            m += 130;
        }
        p.phase[v] = phase; p.level[v] = level; p.stage[v] = stage; p.y[v] = y;
    }

    __syncthreads();
    // rows (frame, channel): up to 2 passes of 64 rows, 4 lanes x 16 columns per row
    int32_t *slot = slots + (blockIdx.x % POLY_SLOTS) * 128;
    for (uint32_t pass = 0; pass * 64 < 2 * nframes; pass++) {
        const uint32_t row = pass * 64 + (tid >> 2), q4 = tid & 3;
        int32_t s = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) s += M[row][q4 * 16 + j];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q4 == 0 && row < 2 * nframes) atomicAdd(&slot[row], s);
    }
}

__global__ __launch_bounds__(128)
void poly_finalize_kernel(int32_t *__restrict__ slots, int32_t *__restrict__ bus_lr, uint32_t nrows)
{
    const uint32_t row = threadIdx.x;
    int32_t v[POLY_SLOTS];
#pragma unroll
    for (int k = 0; k < POLY_SLOTS; k++) v[k] = slots[k * 128 + row];
    int32_t s = 0;
#pragma unroll
    for (int k = 0; k < POLY_SLOTS; k++) { s += v[k]; slots[k * 128 + row] = 0; }
    if (row < nrows) bus_lr[row] = s;
}

}  // namespace

namespace smx {

size_t poly_scratch_bytes() { return (size_t)POLY_SLOTS * 128 * sizeof(int32_t); }

int launch_poly_bank(const PolyArrays &p, int32_t *d_bus_lr, int32_t *d_slots, uint32_t n_pad,
                     uint32_t nframes, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || nframes == 0 || nframes > 64) {
        set_error("launch_poly_bank: n_pad=%u nframes=%u", n_pad, nframes);
        return SMX_E_ARG;
    }
    // ~20 dependent vector ops per voice-sample: one row of 256 voices per workgroup while the
    // chip has room (4 workgroups of 33 KB LDS per CU), grid-stride above that.
    const uint32_t rows = n_pad / 256;
    uint32_t gx = rows;
    if (gx > 1024) gx = 1024;
    static const uint32_t env_gx = [] { const char *e = getenv("SMX_POLY_GRID"); return e ? (uint32_t)atoi(e) : 0u; }();
    if (env_gx) gx = env_gx < rows ? env_gx : rows;
    hipLaunchKernelGGL(poly_bank_kernel, dim3(gx), dim3(256), 0, stream, p, d_slots, n_pad, nframes);
    hipLaunchKernelGGL(poly_finalize_kernel, dim3(1), dim3(128), 0, stream, d_slots, d_bus_lr, 2 * nframes);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
