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
// 4 lanes per row and sent to the bus with one integer atomic per row.
#include "smx_common.h"

namespace {

enum { ENV_IDLE = 0, ENV_A = 1, ENV_D = 2, ENV_S = 3, ENV_R = 4 };

__global__ __launch_bounds__(256)
void poly_bank_kernel(smx::PolyArrays p, int32_t *__restrict__ bus_lr, uint32_t n_pad,
                      uint32_t nframes /* <= 64 */)
{
    __shared__ int32_t M[128][65];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i < 128 * 65; i += 256) (&M[0][0])[i] = 0;
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

        // ADSR as "move the level towards a target at a rate; on arrival enter the next stage".
        // The per-frame work is the same few operations in every stage; the stage-specific
        // parameters live in registers and are rewritten only when some lane of the wave
        // arrives (rare: a stage lasts 10^2..10^5 frames), behind a wave-uniform branch.
        //   A: up,   rate ar, arrival = 32-bit wrap  -> level MAX,  next D
        //   D: down, rate dr, target sl              -> level sl,   next S
        //   R: down, rate rr, target 0               -> level 0,    next idle
        //   S, idle: hold (up at rate 0 never arrives); level is sl / 0 on entry
        bool up;
        uint32_t rate, target, reach_val, next;
        auto enter = [&](uint32_t st) {
            stage = st;
            up = (st == ENV_A) || (st == ENV_S) || (st == ENV_IDLE);
            rate = st == ENV_A ? ar : st == ENV_D ? dr : st == ENV_R ? rr : 0u;
            target = st == ENV_D ? sl : 0u;
            reach_val = st == ENV_A ? 0xFFFFFFFFu : st == ENV_D ? sl : 0u;
            next = st == ENV_A ? (uint32_t)ENV_D : st == ENV_D ? (uint32_t)ENV_S : (uint32_t)ENV_IDLE;
        };
        enter(stage);
        if (stage == ENV_S) level = sl;                      // "S: level = sl" / "idle: level = 0"
        if (stage == ENV_IDLE) level = 0;

        for (uint32_t i = 0; i < nframes; i++) {
            const float x = __fmul_rn((float)(int32_t)phase, 0x1p-31f);
            phase += inc;
            const float t = __fsub_rn(x, y);
            y = __fadd_rn(y, __fmul_rn(a, t));
            const uint32_t nl_u = level + rate, nl_d = level - rate;
            const bool arrived = up ? (nl_u < level) : (level <= target || level - target <= rate);
            level = arrived ? reach_val : (up ? nl_u : nl_d);
            if (__any(arrived)) {
                if (arrived) enter(next);
            }
            const float g = __fmul_rn((float)(level >> 8), 0x1p-24f);
            const float o = __fmul_rn(y, g);
            const int32_t q = (int32_t)__fmul_rn(o, 524288.0f);
            atomicAdd(&M[2 * i][lane], q * pl);
            atomicAdd(&M[2 * i + 1][lane], q * pr);
        }
        p.phase[v] = phase; p.level[v] = level; p.stage[v] = stage; p.y[v] = y;
    }

    __syncthreads();
    // 128 rows (frame, channel): 2 passes of 64 rows, 4 lanes x 16 columns per row
    for (uint32_t pass = 0; pass < 2; pass++) {
        const uint32_t row = pass * 64 + (tid >> 2), q4 = tid & 3;
        int32_t s = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) s += M[row][q4 * 16 + j];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (q4 == 0 && row < 2 * nframes) atomicAdd(&bus_lr[row], s);
    }
}

}  // namespace

namespace smx {

int launch_poly_bank(const PolyArrays &p, int32_t *d_bus_lr, uint32_t n_pad, uint32_t nframes,
                     hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || nframes == 0 || nframes > 64) {
        set_error("launch_poly_bank: n_pad=%u nframes=%u", n_pad, nframes);
        return SMX_E_ARG;
    }
    // ~35 dependent vector ops per voice-sample: parallelism matters more than the serialised
    // bus atomics at the end (256 Ki voices x 64 frames: 33 us with 512-1024 workgroups, 46 us
    // with 256); only few-frame blocks prefer fewer, longer workgroups (1 frame: 9.7 vs 17 us).
    const uint32_t rows = n_pad / 256;
    uint32_t gx = nframes <= 4 ? rows / 4 : rows;
    if (gx < 128) gx = 128;
    if (gx > 1024) gx = 1024;                      // 4 workgroups per CU (33 KB LDS each)
    if (gx > rows) gx = rows;
    hipLaunchKernelGGL(poly_bank_kernel, dim3(gx), dim3(256), 0, stream, p, d_bus_lr, n_pad, nframes);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
