// Scratch prototype: wrap-EVENT formulation of the saw bank for long launches.
// Instead of stepping every phase T times and counting the carry-outs, each voice's wrap frames
// are located directly and added to a per-workgroup histogram hist[t] (W(t) = prefix sum):
//   first wrap at frame   n1 = floor(~u0 / inc)
//   then gaps of          Q + (r <= R),   Q = floor((2^32-1)/inc), R = 2^32-1 - Q*inc
//   residual              r <- r + (r <= R ? E : E - inc),  E = inc - 1 - R,  r0 = u0 + (n1+1)*inc (mod 2^32)
// The prototype checks hist[] against brute-force stepping and times both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int T>
__global__ __launch_bounds__(256) void ev_kernel(const u32x4 *__restrict__ inc4, const u32x4 *__restrict__ st4,
                                                 uint32_t *__restrict__ ghist, uint32_t nrows, uint32_t t0)
{
    __shared__ uint32_t hist[T];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < T; i += 256) hist[i] = 0;
    __syncthreads();
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const u32x4 a = __builtin_nontemporal_load(&inc4[row * 256u + tid]);
        const u32x4 b = __builtin_nontemporal_load(&st4[row * 256u + tid]);
        uint32_t inc[4] = {a.x, a.y, a.z, a.w}, st[4] = {b.x, b.y, b.z, b.w};
        uint32_t t[4], r[4], Q[4], Rm[4], E[4];
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = inc[k];
            const uint32_t u0 = (i ? st[k] + t0 * i : 0u) ^ 0x80000000u;
            const uint32_t d = i ? i : 1u;
            const uint32_t n1 = ~u0 / d;
            Q[k] = 0xFFFFFFFFu / d;
            Rm[k] = 0xFFFFFFFFu - Q[k] * d;
            E[k] = d - 1u - Rm[k];
            Q[k] = Q[k] < (1u << 30) ? Q[k] : (1u << 30);   // no wrap of t + gap (inc == 1)
            r[k] = u0 + (n1 + 1u) * d;                 // mod 2^32: the phase right after the first wrap
            t[k] = i ? n1 : 0xFFFFFFFFu;               // off: no events
            any |= t[k] < (uint32_t)T;
        }
        while (__any(any)) {
            any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (t[k] < (uint32_t)T) {
                    atomicAdd(&hist[t[k]], 1u);
                    const bool c = r[k] <= Rm[k];
                    t[k] += Q[k] + (c ? 1u : 0u);
                    r[k] += c ? E[k] : E[k] - inc[k];
                    any |= t[k] < (uint32_t)T;
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < T; i += 256)
        if (hist[i]) atomicAdd(&ghist[(blockIdx.x & 63) * T + i], hist[i]);
}

// brute force: step every phase T times, count carry-outs per frame (LDS histogram, then global)
template <int T>
__global__ __launch_bounds__(256) void step_kernel(const u32x4 *__restrict__ inc4, const u32x4 *__restrict__ st4,
                                                   uint32_t *__restrict__ ghist, uint32_t nrows, uint32_t t0)
{
    __shared__ uint32_t hist[T];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < T; i += 256) hist[i] = 0;
    __syncthreads();
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const u32x4 a = inc4[row * 256u + tid], b = st4[row * 256u + tid];
        uint32_t inc[4] = {a.x, a.y, a.z, a.w}, st[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t u = (inc[k] ? st[k] + t0 * inc[k] : 0u) ^ 0x80000000u;
            for (uint32_t t = 0; t < (uint32_t)T; t++) {
                const uint32_t n = u + inc[k];
                if (n < u) atomicAdd(&hist[t], 1u);
                u = n;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < T; i += 256)
        if (hist[i]) atomicAdd(&ghist[(blockIdx.x & 63) * T + i], hist[i]);
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

static uint64_t sm64(uint64_t &x) { uint64_t z = (x += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

template <int T>
void run(uint32_t n, const char *label, std::vector<uint32_t> &hinc, std::vector<uint32_t> &hst, bool verify)
{
    uint32_t *inc, *st, *h1, *h2;
    (void)hipMalloc(&inc, (size_t)n * 4); (void)hipMalloc(&st, (size_t)n * 4);
    (void)hipMalloc(&h1, 64 * T * 4); (void)hipMalloc(&h2, 64 * T * 4);
    (void)hipMemcpy(inc, hinc.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(st, hst.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    const uint32_t nrows = n / 1024;
    const uint32_t t0 = 12345;
    if (verify) {
        (void)hipMemset(h1, 0, 64 * T * 4); (void)hipMemset(h2, 0, 64 * T * 4);
        hipLaunchKernelGGL((ev_kernel<T>), dim3(2048), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)st, h1, nrows, t0);
        hipLaunchKernelGGL((step_kernel<T>), dim3(2048), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)st, h2, nrows, t0);
        std::vector<uint32_t> a(64 * T), b(64 * T);
        (void)hipMemcpy(a.data(), h1, 64 * T * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(b.data(), h2, 64 * T * 4, hipMemcpyDeviceToHost);
        uint64_t bad = 0, total = 0;
        for (int t = 0; t < T; t++) { uint64_t x = 0, y = 0; for (int s = 0; s < 64; s++) { x += a[s * T + t]; y += b[s * T + t]; } bad += x != y; total += y; }
        printf("%s T=%d: verify %s (%llu wraps, %.2f per voice)\n", label, T, bad ? "MISMATCH" : "ok", (unsigned long long)total, (double)total / n);
    }
    for (int gx : {4096}) {
        float e = timeit([&] { hipLaunchKernelGGL((ev_kernel<T>), dim3(gx), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)st, h1, nrows, t0); }, 5);
        printf("%s T=%d grid %d: events %.1f us = %.1f Tsamples/s\n", label, T, gx, e * 1e3, (double)n * T / e / 1e9);
    }
    (void)hipFree(inc); (void)hipFree(st); (void)hipFree(h1); (void)hipFree(h2);
}

int main() {
    const uint32_t n = 1u << 26;
    std::vector<uint32_t> hinc(n), hst(n);
    uint64_t s = 42;
    // piano-range bank: notes 21..108, inc = 2^32 * f / 48000
    uint32_t tab[128];
    for (int m = 0; m < 128; m++) tab[m] = (uint32_t)(4294967296.0 * (440.0 * pow(2.0, (m - 69) / 12.0)) / 48000.0);
    for (uint32_t v = 0; v < n; v++) { uint64_t r = sm64(s); hinc[v] = tab[21 + (uint32_t)(r >> 40) % 88]; hst[v] = (uint32_t)r; }
    for (uint32_t v = 5; v < n; v += 100003) hinc[v] = 0;           // a few voices off
    hinc[7] = 0xFFFFFFFFu; hinc[9] = 1; hinc[11] = 0x80000000u;   // extremes
    run<256>(1u << 22, "4Mi piano", hinc, hst, true);
    run<1024>(1u << 22, "4Mi piano", hinc, hst, true);
    run<1024>(n, "64Mi piano", hinc, hst, false);
    run<256>(n, "64Mi piano", hinc, hst, false);
    run<128>(n, "64Mi piano", hinc, hst, false);
    run<64>(n, "64Mi piano", hinc, hst, true);
    run<32>(n, "64Mi piano", hinc, hst, false);
    // crossover: piano bank with a fraction of the voices replaced by a high increment (K wraps per 64 frames)
    for (double k : {2.0, 6.0, 12.0, 17.0, 32.0, 64.0}) {
        for (double frac : {0.01, 1.0}) {
            std::vector<uint32_t> x(hinc.begin(), hinc.begin() + n);
            const uint32_t hi = (uint32_t)(k / 64.0 * 4294967295.0);
            uint64_t s2 = 7;
            for (uint32_t v = 0; v < n; v++) if ((sm64(s2) >> 11) * (1.0 / 9007199254740992.0) < frac) x[v] = hi;
            char lab[64]; snprintf(lab, sizeof lab, "K=%g frac=%g", k, frac);
            run<64>(n, lab, x, hst, false);
        }
    }
    // sorted by increment (what grouping by octave would give)
    std::vector<uint32_t> sinc(hinc.begin(), hinc.begin() + n);
    std::sort(sinc.begin(), sinc.end());
    run<1024>(n, "64Mi sorted", sinc, hst, false);
    return 0;
}
