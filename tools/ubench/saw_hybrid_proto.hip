// Prototype: a BOUNDED hybrid of the stepping and the located-wrap forms of the 64-frame saw block.
// Per wave row (256 voices, 4 per lane) the voices with inc >= 2^LG ("high": they may wrap more than 2^(LG-26) times in
// 64 frames) are compacted into the wave's LDS list and STEPPED in ceil(high / 64) slots instead of 4; the others get
// their one or two wraps LOCATED (floor(~u / inc) by one biased reciprocal, as the event form does).  A row with more
// than 192 high voices steps all four slots as the stepping form does, so no bank costs more than stepping plus the
// classification.  Measures only the wrap counting (the part the forms differ in): W[t] = wraps at frame t over the
// whole bank, the hybrid's checked against the stepping form's and both against a CPU loop on a small bank.
//   hipcc --offload-arch=gfx950 -O3 -o saw_hybrid_proto tools/ubench/saw_hybrid_proto.hip && ./saw_hybrid_proto
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rcp_biased(uint32_t d) { return __builtin_amdgcn_rcpf((float)d) * 0.99999618530273437500f; }
__device__ __forceinline__ uint32_t div_small(uint32_t a, uint32_t d, float rd)
{
    uint32_t q = (uint32_t)((float)a * rd);
    const uint32_t r = a - q * d;
    return r >= d ? q + 1u : q;
}
// {wraps so far, phase} += inc as one 64-bit multiply-add (the product's stepping form)
__device__ __forceinline__ void step1(unsigned long long &q, uint32_t inc)
{
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(q) : "v"(inc) : "vcc");
#endif
}

template <int K>
__device__ __forceinline__ void step_slots(const uint32_t (&u)[4], const uint32_t (&inc)[4], uint32_t (&cnt)[64])
{
    unsigned long long q[4];
#pragma unroll
    for (int k = 0; k < K; k++) q[k] = u[k];
#pragma unroll
    for (int t = 0; t < 64; t++) {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < K; k++) { step1(q[k], inc[k]); s += (uint32_t)(q[k] >> 32); }
        cnt[t] += s;                                   // cumulative counts; differenced at the end
    }
}

// MODE 0: stepping (4 slots).  MODE 1: hybrid, threshold 2^LG.
template <int MODE, int LG>
__global__ __launch_bounds__(256)
void wraps_kernel(const u32x4 *__restrict__ inc4, const u32x4 *__restrict__ st4, uint32_t nrows, uint32_t *__restrict__ Wout,
                  uint32_t *__restrict__ stats)
{
    __shared__ uint32_t M[64][65];
    __shared__ uint2 EL[4 * 256];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    for (uint32_t i = tid; i < 64 * 65; i += 256) (&M[0][0])[i] = 0;
    __syncthreads();
    uint32_t cnt[64];
#pragma unroll
    for (int t = 0; t < 64; t++) cnt[t] = 0;
    uint32_t slots_stepped = 0;
    // software prefetch, as in the product's kernels: the next row's 32 bytes per lane are requested before the
    // arithmetic on the current row
    u32x4 a_next = 0, b_next = 0;
    if (blockIdx.x < nrows) {
        a_next = __builtin_nontemporal_load(inc4 + (size_t)blockIdx.x * 256u + tid);
        b_next = __builtin_nontemporal_load(st4 + (size_t)blockIdx.x * 256u + tid);
    }
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const u32x4 a = a_next, b = b_next;
        const uint32_t rn = (row + gridDim.x < nrows ? row + gridDim.x : nrows - 1);
        a_next = __builtin_nontemporal_load(inc4 + (size_t)rn * 256u + tid);
        b_next = __builtin_nontemporal_load(st4 + (size_t)rn * 256u + tid);
        const uint32_t vi[4] = {a.x, a.y, a.z, a.w}, vu[4] = {b.x, b.y, b.z, b.w};
        if (MODE == 0) {
            step_slots<4>(vu, vi, cnt);
            slots_stepped += 4;
            continue;
        }
        bool high[4];
        uint32_t nh = 0;
        unsigned long long bm[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            high[k] = (vi[k] >> LG) != 0u;
            bm[k] = __ballot(high[k]);
            nh += (uint32_t)__builtin_popcountll(bm[k]);
        }
        if (nh > 192u) {                               // four slots either way: step everything, as the stepping form does
            step_slots<4>(vu, vi, cnt);
            slots_stepped += 4;
            continue;
        }
        // the high voices of the wave, compacted, then dealt out lane by lane
        uint2 *list = &EL[(tid >> 6) * 256];
        uint32_t base = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm[k] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm[k], 0u));
            if (high[k]) list[pos] = make_uint2(vu[k], vi[k]);
            base += (uint32_t)__builtin_popcountll(bm[k]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t hu[4] = {0, 0, 0, 0}, hi_[4] = {0, 0, 0, 0};
        const uint32_t kslots = (nh + 63u) >> 6;       // wave-uniform, 0..3
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t e = lane + 64u * k;
            if ((uint32_t)k < kslots && e < nh) { const uint2 en = list[e]; hu[k] = en.x; hi_[k] = en.y; }
        }
        if (kslots == 1) step_slots<1>(hu, hi_, cnt);
        else if (kslots == 2) step_slots<2>(hu, hi_, cnt);
        else if (kslots == 3) step_slots<3>(hu, hi_, cnt);
        slots_stepped += kslots;
        // the low voices: at most 2^(LG-26) wraps in 64 frames (LG <= 27: one or two), located
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t d = vi[k], u = vu[k];
            const uint32_t lo = u + (d << 6);
            const uint32_t wraps = (d >> 26) + (lo < u ? 1u : 0u);
            if (!high[k] && wraps) {
                const float rd = rcp_biased(d);
                const uint32_t n1 = div_small(~u, d, rd);
                atomicAdd(&M[n1][lane], 1u);
                if (LG > 26 && (d >> 26)) {
                    const uint32_t eq = div_small(0xFFFFFFFFu, d, rd);
                    const uint32_t erm = 0xFFFFFFFFu - eq * d;
                    const uint32_t er = u + (n1 + 1u) * d;
                    const uint32_t n2 = n1 + eq + (er <= erm ? 1u : 0u);
                    if (n2 < 64u) atomicAdd(&M[n2][lane], 1u);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int t = 63; t > 0; t--) cnt[t] -= cnt[t - 1];
#pragma unroll
    for (int t = 0; t < 64; t++) atomicAdd(&M[t][lane], cnt[t]);
    __syncthreads();
    const uint32_t t = tid >> 2, q = tid & 3;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) s += M[t][q * 16 + j];
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (q == 0 && s) atomicAdd(&Wout[t], s);
    if (lane == 0 && stats) atomicAdd(stats, slots_stepped);
}

static uint32_t note_inc(int note) { return (uint32_t)(440.0 * std::pow(2.0, (note - 69) / 12.0) / 48000.0 * 4294967296.0); }
static uint64_t sm64(uint64_t &x) { uint64_t z = (x += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <int MODE, int LG>
static float run(const uint32_t *d_inc, const uint32_t *d_st, uint32_t n, uint32_t *d_w, uint32_t *d_stats, std::vector<uint32_t> &w, double *slots)
{
    const uint32_t nrows = n / 1024;
    const uint32_t grid = nrows < 2048 ? nrows : 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int warm = 0; warm < 3; warm++)
        hipLaunchKernelGGL((wraps_kernel<MODE, LG>), dim3(grid), dim3(256), 0, 0, (const u32x4 *)d_inc, (const u32x4 *)d_st, nrows, d_w, d_stats);
    (void)hipMemset(d_w, 0, 256); (void)hipMemset(d_stats, 0, 4);
    (void)hipDeviceSynchronize();
    const int reps = 10;
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL((wraps_kernel<MODE, LG>), dim3(grid), dim3(256), 0, 0, (const u32x4 *)d_inc, (const u32x4 *)d_st, nrows, d_w, d_stats);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    w.resize(64);
    (void)hipMemcpy(w.data(), d_w, 256, hipMemcpyDeviceToHost);
    for (auto &x : w) x /= reps;                        // every launch adds the same counts
    uint32_t st = 0;
    (void)hipMemcpy(&st, d_stats, 4, hipMemcpyDeviceToHost);
    *slots = (double)st / reps / (n / 256.0);           // slots stepped per wave row
    return ms / reps * 1e3f;
}

int main()
{
    struct Bank { const char *name; int lo, hi; double high_share; };
    const Bank banks[] = {{"piano range (notes 21..108)", 21, 108, -1}, {"low notes only (21..76)", 21, 76, -1},
                          {"high notes only (100..127)", 100, 127, -1}, {"3/4 high rows (100..127) + 1/4 piano", 21, 108, 0.75}};
    for (uint32_t n : {1u << 16, 1u << 26}) {
        for (const Bank &bk : banks) {
            std::vector<uint32_t> inc(n), st(n);
            uint64_t seed = 0x5EED0000 + n;
            for (uint32_t v = 0; v < n; v++) {
                const uint64_t r = sm64(seed);
                int note = bk.lo + (int)(r % (uint64_t)(bk.hi - bk.lo + 1));
                if (bk.high_share > 0 && ((v >> 8) & 3) != 3) note = 100 + (int)(r % 28);
                inc[v] = note_inc(note);
                st[v] = (uint32_t)(r >> 32);
            }
            uint32_t *d_inc, *d_st, *d_w, *d_stats;
            (void)hipMalloc(&d_inc, (size_t)n * 4); (void)hipMalloc(&d_st, (size_t)n * 4); (void)hipMalloc(&d_w, 256); (void)hipMalloc(&d_stats, 4);
            (void)hipMemcpy(d_inc, inc.data(), (size_t)n * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(d_st, st.data(), (size_t)n * 4, hipMemcpyHostToDevice);
            std::vector<uint32_t> w0, w26, w27;
            double s0, s26, s27;
            const float t0 = run<0, 26>(d_inc, d_st, n, d_w, d_stats, w0, &s0);
            const float t26 = run<1, 26>(d_inc, d_st, n, d_w, d_stats, w26, &s26);
            const float t27 = run<1, 27>(d_inc, d_st, n, d_w, d_stats, w27, &s27);
            bool ok = w0 == w26 && w0 == w27;
            if (n == (1u << 16)) {                       // the stepping kernel itself against a CPU loop
                std::vector<uint32_t> ref(64, 0);
                for (uint32_t v = 0; v < n; v++) {
                    uint32_t u = st[v];
                    for (int t = 0; t < 64; t++) { const uint32_t nx = u + inc[v]; ref[t] += nx < u; u = nx; }
                }
                ok = ok && ref == w0;
            }
            printf("%9u voices, %-40s stepping %7.1f us (%.2f slots/row) | hybrid 2^26 %7.1f us (%.2f) | hybrid 2^27 %7.1f us (%.2f) | wraps per frame %s\n",
                   n, bk.name, t0, s0, t26, s26, t27, s27, ok ? "equal" : "DIFFER");
            (void)hipFree(d_inc); (void)hipFree(d_st); (void)hipFree(d_w); (void)hipFree(d_stats);
        }
    }
    return 0;
}
