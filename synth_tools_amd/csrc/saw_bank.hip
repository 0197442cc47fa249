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
// block (8 B read, 4 B write).  The block's partial sums go through an LDS
// [TC][64+1] matrix (conflict-free ds_add, padded rows), are folded by a
// 4-lane shuffle and leave the CU as one integer atomic per frame.
//
// Time is split into chunks of 64 frames on blockIdx.y: the phasor is linear,
// state(t0) = state + t0*inc (mod 2^32), so chunks are independent and small
// banks still fill the chip.  State is ping-ponged (state_in -> state_out) so
// that chunks never read what another chunk has already advanced.
#include "smx_common.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Streaming accesses: a bank larger than the caches is touched exactly once per
// launch, so its lines are loaded/stored non-temporally (measured +7 % HBM rate).
template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void stream_store(T v, T *p)
{
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <int TC, int VW, bool NT>
__global__ __launch_bounds__(256)
void saw_bank_kernel(const uint32_t *__restrict__ inc,
                     const uint32_t *__restrict__ st_in,
                     uint32_t *__restrict__ st_out,
                     int32_t *__restrict__ bus,
                     int32_t *__restrict__ bus_next,   // zeroed here for the NEXT launch
                     uint32_t ngroups,      // n_pad / VW
                     uint32_t nframes)      // total frames of this block
{
    __shared__ int32_t M[TC][65];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t t0 = blockIdx.y * 64u;   // >0 only when TC == 64

    for (uint32_t i = tid; i < TC * 65; i += 256) (&M[0][0])[i] = 0;
    // the bus is accumulated with atomics, so it must start at zero: each launch
    // clears the buffer its successor will use (saves a fill kernel per step)
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (uint32_t i = tid; i < nframes; i += 256) bus_next[i] = 0;

    int32_t acc[TC];
#pragma unroll
    for (int t = 0; t < TC; t++) acc[t] = 0;

    for (uint32_t g = blockIdx.x * 256u + tid; g < ngroups; g += gridDim.x * 256u) {
        uint32_t vi[VW], vs[VW];
        if constexpr (VW == 4) {
            const u32x4 a = stream_load<NT>(reinterpret_cast<const u32x4 *>(inc) + g);
            const u32x4 b = stream_load<NT>(reinterpret_cast<const u32x4 *>(st_in) + g);
            vi[0] = a.x; vi[1] = a.y; vi[2] = a.z; vi[3] = a.w;
            vs[0] = b.x; vs[1] = b.y; vs[2] = b.z; vs[3] = b.w;
        } else {
            vi[0] = stream_load<NT>(inc + g);
            vs[0] = stream_load<NT>(st_in + g);
        }
        if (blockIdx.y == 0) {
            // final state in closed form; inc == 0 leaves the phase untouched
            if constexpr (VW == 4) {
                u32x4 o;
                o.x = vs[0] + nframes * vi[0]; o.y = vs[1] + nframes * vi[1];
                o.z = vs[2] + nframes * vi[2]; o.w = vs[3] + nframes * vi[3];
                stream_store<NT>(o, reinterpret_cast<u32x4 *>(st_out) + g);
            } else {
                stream_store<NT>(vs[0] + nframes * vi[0], st_out + g);
            }
        }
#pragma unroll
        for (int k = 0; k < VW; k++) {
            // an inactive voice (inc == 0) contributes nothing: park it at 0
            vs[k] = vi[k] ? vs[k] + t0 * vi[k] : 0u;
        }
#pragma unroll
        for (int t = 0; t < TC; t++) {
            if constexpr (VW == 4) {
                acc[t] += ((int32_t)vs[0] >> 4) + ((int32_t)vs[1] >> 4);
                acc[t] += ((int32_t)vs[2] >> 4) + ((int32_t)vs[3] >> 4);
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
    if (q == 0 && t < TC && t0 + t < nframes) atomicAdd(&bus[t0 + t], s);
}

// sum_tick_square (linux/synth.c:182-195): OR of the active voices' sign bits.
// Unused by the reference's synth_run; kept as a bank variant.  bus word t
// receives 0x80000000 if any active voice has its sign bit set at frame t.
__global__ __launch_bounds__(256)
void square_bank_kernel(const uint32_t *__restrict__ inc,
                        const uint32_t *__restrict__ st_in,
                        uint32_t *__restrict__ st_out,
                        uint32_t *__restrict__ or_bus,
                        uint32_t n_pad, uint32_t nframes)
{
    const uint32_t t0 = blockIdx.y * 64u;
    const uint32_t nf = min(64u, nframes - t0);
    unsigned long long any_lo = 0;   // bit t: some active voice negative at t0+t
    for (uint32_t v = blockIdx.x * 256u + threadIdx.x; v < n_pad; v += gridDim.x * 256u) {
        const uint32_t i = inc[v];
        const uint32_t s0 = st_in[v];
        if (blockIdx.y == 0) st_out[v] = s0 + nframes * i;
        uint32_t s = s0 + t0 * i;
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
            if ((any_lo >> t) & 1) atomicOr(&or_bus[t0 + t], 0x80000000u);
    }
}

// Grid: persistent workgroups, grid-stride over voices.  In the HBM-bound regime
// (few frames per launch) 2 workgroups per CU stream fastest (measured: 768 x 256
// threads = 6.2 TB/s vs 5.6 TB/s at 2048); the VALU-bound regime wants every SIMD full.
static uint32_t grid_cap(uint32_t tc, uint32_t gy)
{
    static const char *env = getenv("SMX_SAW_GRID");      // tuning override
    uint32_t cap = env ? (uint32_t)atoi(env) : (tc <= 16 ? 768u : 2048u);
    cap = (cap + gy - 1) / gy;
    return cap < 1 ? 1 : cap;
}

template <int TC, int VW, bool NT>
int launch_tc(const uint32_t *inc, const uint32_t *si, uint32_t *so, int32_t *bus, int32_t *bus_next,
              uint32_t n_pad, uint32_t nframes, hipStream_t stream)
{
    const uint32_t ngroups = n_pad / VW;
    uint32_t gx = (ngroups + 255) / 256;
    const uint32_t gy = (nframes + 63) / 64;
    const uint32_t cap = grid_cap(TC, gy);
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL((saw_bank_kernel<TC, VW, NT>), dim3(gx, gy), dim3(256), 0, stream,
                       inc, si, so, bus, bus_next, ngroups, nframes);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

template <int VW, bool NT>
int launch_vw(const uint32_t *inc, const uint32_t *si, uint32_t *so, int32_t *bus, int32_t *bus_next,
              uint32_t n_pad, uint32_t nframes, hipStream_t stream)
{
    if (nframes > 32) return launch_tc<64, VW, NT>(inc, si, so, bus, bus_next, n_pad, nframes, stream);
    if (nframes > 16) return launch_tc<32, VW, NT>(inc, si, so, bus, bus_next, n_pad, nframes, stream);
    if (nframes > 8)  return launch_tc<16, VW, NT>(inc, si, so, bus, bus_next, n_pad, nframes, stream);
    if (nframes > 4)  return launch_tc<8, VW, NT>(inc, si, so, bus, bus_next, n_pad, nframes, stream);
    if (nframes > 2)  return launch_tc<4, VW, NT>(inc, si, so, bus, bus_next, n_pad, nframes, stream);
    if (nframes > 1)  return launch_tc<2, VW, NT>(inc, si, so, bus, bus_next, n_pad, nframes, stream);
    return launch_tc<1, VW, NT>(inc, si, so, bus, bus_next, n_pad, nframes, stream);
}

}  // namespace

namespace smx {

int launch_saw_bank(const uint32_t *d_inc, const uint32_t *d_state_in,
                    uint32_t *d_state_out, int32_t *d_bus, int32_t *d_bus_next,
                    uint32_t n_pad, uint32_t nframes, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || nframes == 0) {
        set_error("launch_saw_bank: n_pad=%u nframes=%u", n_pad, nframes);
        return SMX_E_ARG;
    }
    // 4 voices per lane once there are enough voices to fill the chip that way;
    // non-temporal streaming once the bank (12 B/voice) cannot live in the 256 MiB
    // Infinity Cache between launches anyway
    if (n_pad >= (1u << 24))
        return launch_vw<4, true>(d_inc, d_state_in, d_state_out, d_bus, d_bus_next, n_pad, nframes, stream);
    if (n_pad >= (1u << 20))
        return launch_vw<4, false>(d_inc, d_state_in, d_state_out, d_bus, d_bus_next, n_pad, nframes, stream);
    return launch_vw<1, false>(d_inc, d_state_in, d_state_out, d_bus, d_bus_next, n_pad, nframes, stream);
}

int launch_square_bank(const uint32_t *d_inc, const uint32_t *d_state_in,
                       uint32_t *d_state_out, uint32_t *d_or_bus, uint32_t n_pad,
                       uint32_t nframes, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 1023) || nframes == 0) return SMX_E_ARG;
    uint32_t gx = n_pad / 256;
    const uint32_t gy = (nframes + 63) / 64;
    const uint32_t cap = (2048 + gy - 1) / gy;
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(square_bank_kernel, dim3(gx, gy), dim3(256), 0, stream,
                       d_inc, d_state_in, d_state_out, d_or_bus, n_pad, nframes);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
