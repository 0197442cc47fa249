// Scratch prototype: carry-count formulation of the 64-frame saw block.
// Measures the inner loop only; epilogue just keeps the counters alive.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 4 voices advance one frame; NV of them have their carries counted per lane by v_addc,
// the rest leave as SGPR masks for s_bcnt1.
template <int NV>
__device__ __forceinline__ uint32_t step4(uint32_t &u0, uint32_t &u1, uint32_t &u2, uint32_t &u3,
                                      uint32_t i0, uint32_t i1, uint32_t i2, uint32_t i3, uint32_t &cnt)
{
    uint32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long m0, m1, m2, m3;
    if (NV == 2) {
    asm("v_add_co_u32_e64 %0, %5, %0, %9\n\t"
        "v_add_co_u32_e64 %1, %6, %1, %10\n\t"
        "v_add_co_u32_e64 %2, %7, %2, %11\n\t"
        "v_add_co_u32_e64 %3, %8, %3, %12\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %5\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %6"
        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(cnt), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(i0), "v"(i1), "v"(i2), "v"(i3) : "vcc");
    c = (uint32_t)(__builtin_popcountll(m2) + __builtin_popcountll(m3));
    } else if (NV == 4) {
    asm("v_add_co_u32_e64 %0, %5, %0, %9\n\t"
        "v_add_co_u32_e64 %1, %6, %1, %10\n\t"
        "v_add_co_u32_e64 %2, %7, %2, %11\n\t"
        "v_add_co_u32_e64 %3, %8, %3, %12\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %5\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %6\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %7\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %8"
        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(cnt), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(i0), "v"(i1), "v"(i2), "v"(i3) : "vcc");
    } else if (NV == 1) {
    asm("v_add_co_u32_e64 %0, %5, %0, %9\n\t"
        "v_add_co_u32_e64 %1, %6, %1, %10\n\t"
        "v_add_co_u32_e64 %2, %7, %2, %11\n\t"
        "v_add_co_u32_e64 %3, %8, %3, %12\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %5"
        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(cnt), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(i0), "v"(i1), "v"(i2), "v"(i3) : "vcc");
    c = (uint32_t)(__builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3));
    } else {
    asm("v_add_co_u32_e64 %0, %4, %0, %8\n\t"
        "v_add_co_u32_e64 %1, %5, %1, %9\n\t"
        "v_add_co_u32_e64 %2, %6, %2, %10\n\t"
        "v_add_co_u32_e64 %3, %7, %3, %11"
        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "=&s"(m0), "=&s"(m1), "=&s"(m2), "=&s"(m3)
        : "v"(i0), "v"(i1), "v"(i2), "v"(i3) : "vcc");
    c = (uint32_t)(__builtin_popcountll(m0) + __builtin_popcountll(m1) + __builtin_popcountll(m2) + __builtin_popcountll(m3));
    }
#endif
    return c;
}

template <int NV>
__global__ __launch_bounds__(256)
void v2_k(const u32x4 *__restrict__ inc, const u32x4 *__restrict__ si, u32x4 *__restrict__ so,
          uint32_t *__restrict__ dump, uint32_t ngroups)
{
    uint32_t cnt[64];
    uint32_t W[32];                       // two 16-bit counters per word: frame t and t+32
#pragma unroll
    for (int t = 0; t < 64; t++) cnt[t] = 0;
#pragma unroll
    for (int t = 0; t < 32; t++) W[t] = 0;
    for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < ngroups; g += gridDim.x * 256u) {
        u32x4 a = __builtin_nontemporal_load(&inc[g]);
        u32x4 b = __builtin_nontemporal_load(&si[g]);
        u32x4 o; o.x = b.x + 64u * a.x; o.y = b.y + 64u * a.y; o.z = b.z + 64u * a.z; o.w = b.w + 64u * a.w;
        __builtin_nontemporal_store(o, &so[g]);
        uint32_t u0 = b.x ^ 0x80000000u, u1 = b.y ^ 0x80000000u, u2 = b.z ^ 0x80000000u, u3 = b.w ^ 0x80000000u;
#pragma unroll
        for (int t = 0; t < 64; t++) {
            const uint32_t c = step4<NV>(u0, u1, u2, u3, a.x, a.y, a.z, a.w, cnt[t]);
            if (NV < 4) W[t & 31] += (t < 32) ? c : (c << 16);
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int t = 0; t < 64; t++) acc += cnt[t] * (t + 1);
#pragma unroll
    for (int t = 0; t < 32; t++) acc += W[t] * 3;
    dump[blockIdx.x * 256 + threadIdx.x] = acc;
}


// Mixed: voices 0/1 step as {wraps so far, phase} += inc in one 64-bit multiply-add (the carry lands in the high
// word), their cumulative counts enter the per-frame counter with one v_add3; voices 2/3 as before (v_add_co +
// s_bcnt1).  VARIANT 1: all four voices through the 64-bit form (two v_add3 per frame).
// NOTE: cnt[t] then holds CUMULATIVE carries (what the finalize kernel derives by a prefix sum anyway).
template <int VARIANT>
__global__ __launch_bounds__(256)
void v5_k(const u32x4 *__restrict__ inc, const u32x4 *__restrict__ si, u32x4 *__restrict__ so,
          uint32_t *__restrict__ dump, uint32_t ngroups)
{
    uint32_t cnt[64];
    uint32_t W[32];
#pragma unroll
    for (int t = 0; t < 64; t++) cnt[t] = 0;
#pragma unroll
    for (int t = 0; t < 32; t++) W[t] = 0;
    for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < ngroups; g += gridDim.x * 256u) {
        u32x4 a = __builtin_nontemporal_load(&inc[g]);
        u32x4 b = __builtin_nontemporal_load(&si[g]);
        u32x4 o; o.x = b.x + 64u * a.x; o.y = b.y + 64u * a.y; o.z = b.z + 64u * a.z; o.w = b.w + 64u * a.w;
        __builtin_nontemporal_store(o, &so[g]);
        unsigned long long q0 = b.x ^ 0x80000000u, q1 = b.y ^ 0x80000000u, q2 = b.z ^ 0x80000000u, q3 = b.w ^ 0x80000000u;
        uint32_t u2 = (uint32_t)q2, u3 = (uint32_t)q3;
#pragma unroll
        for (int t = 0; t < 64; t++) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (VARIANT == 0) {
                unsigned long long m2, m3;
                asm("v_mad_u64_u32 %0, vcc, %6, 1, %0\n\t"
                    "v_mad_u64_u32 %1, vcc, %7, 1, %1\n\t"
                    "v_add_co_u32_e64 %2, %4, %2, %8\n\t"
                    "v_add_co_u32_e64 %3, %5, %3, %9"
                    : "+v"(q0), "+v"(q1), "+v"(u2), "+v"(u3), "=&s"(m2), "=&s"(m3)
                    : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w) : "vcc");
                const uint32_t c = (uint32_t)(__builtin_popcountll(m2) + __builtin_popcountll(m3));
                W[t & 31] += (t < 32) ? c : (c << 16);
                cnt[t] += (uint32_t)(q0 >> 32) + (uint32_t)(q1 >> 32);
            } else {
                asm("v_mad_u64_u32 %0, vcc, %4, 1, %0\n\t"
                    "v_mad_u64_u32 %1, vcc, %5, 1, %1\n\t"
                    "v_mad_u64_u32 %2, vcc, %6, 1, %2\n\t"
                    "v_mad_u64_u32 %3, vcc, %7, 1, %3"
                    : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                    : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w) : "vcc");
                cnt[t] += (uint32_t)(q0 >> 32) + (uint32_t)(q1 >> 32);
                cnt[t] += (uint32_t)(q2 >> 32) + (uint32_t)(q3 >> 32);
            }
#endif
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int t = 0; t < 64; t++) acc += cnt[t] * (t + 1);
#pragma unroll
    for (int t = 0; t < 32; t++) acc += W[t] * 3;
    dump[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int NV>
__global__ __launch_bounds__(256)
void v3_k(const u32x4 *__restrict__ inc, const u32x4 *__restrict__ si, u32x4 *__restrict__ so,
          uint32_t *__restrict__ dump, uint32_t ngroups)
{
    __shared__ uint32_t M[64][65];
    for (uint32_t i = threadIdx.x; i < 64 * 65; i += 256) (&M[0][0])[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    uint32_t W[32];
#pragma unroll
    for (int t = 0; t < 32; t++) W[t] = 0;
    const uint32_t nrows = ngroups >> 8;
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const uint32_t g = row * 256u + threadIdx.x;
        u32x4 a = __builtin_nontemporal_load(&inc[g]);
        u32x4 b = __builtin_nontemporal_load(&si[g]);
        u32x4 o = b + 64u * a;
        __builtin_nontemporal_store(o, &so[g]);
        uint32_t u0 = b.x ^ 0x80000000u, u1 = b.y ^ 0x80000000u, u2 = b.z ^ 0x80000000u, u3 = b.w ^ 0x80000000u;
#pragma unroll
        for (int pass = 0; pass < 2; pass++) {
            uint32_t cnt[32];
#pragma unroll
            for (int t = 0; t < 32; t++) cnt[t] = 0;
#pragma unroll
            for (int t = 0; t < 32; t++) {
                const uint32_t c = step4<NV>(u0, u1, u2, u3, a.x, a.y, a.z, a.w, cnt[t]);
                if (NV < 4) W[t] += pass ? (c << 16) : c;
            }
#pragma unroll
            for (int t = 0; t < 32; t++) atomicAdd(&M[pass * 32 + t][lane], cnt[t]);
        }
    }
    __syncthreads();
    uint32_t acc = 0;
    for (int t = 0; t < 64; t++) acc += M[t][lane] * (t + 1);
#pragma unroll
    for (int t = 0; t < 32; t++) acc += W[t] * 3;
    dump[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <bool NT>
__global__ __launch_bounds__(256)
void classic_k(const u32x4 *__restrict__ inc, const u32x4 *__restrict__ si, u32x4 *__restrict__ so,
               uint32_t *__restrict__ dump, uint32_t ngroups)
{
    int32_t acc[64];
#pragma unroll
    for (int t = 0; t < 64; t++) acc[t] = 0;
    for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < ngroups; g += gridDim.x * 256u) {
        u32x4 a = __builtin_nontemporal_load(&inc[g]);
        u32x4 b = __builtin_nontemporal_load(&si[g]);
        u32x4 o; o.x = b.x + 64u * a.x; o.y = b.y + 64u * a.y; o.z = b.z + 64u * a.z; o.w = b.w + 64u * a.w;
        __builtin_nontemporal_store(o, &so[g]);
#pragma unroll
        for (int t = 0; t < 64; t++) {
            acc[t] += ((int32_t)b.x >> 4) + ((int32_t)b.y >> 4);
            acc[t] += ((int32_t)b.z >> 4) + ((int32_t)b.w >> 4);
            b += a;
        }
    }
    uint32_t x = 0;
#pragma unroll
    for (int t = 0; t < 64; t++) x += acc[t] * (t + 1);
    dump[blockIdx.x * 256 + threadIdx.x] = x;
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}

int main(int argc, char **argv) {
    const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : (1u << 26);
    const uint32_t ng = n / 4;
    uint32_t *inc, *s0, *s1, *dump;
    (void)hipMalloc(&inc, (size_t)n * 4); (void)hipMalloc(&s0, (size_t)n * 4); (void)hipMalloc(&s1, (size_t)n * 4);
    (void)hipMalloc(&dump, 8192 * 256 * 4);
    std::vector<uint32_t> h(n);
    for (uint32_t i = 0; i < n; i++) h[i] = (i * 2654435761u) >> 5 | 1;
    (void)hipMemcpy(inc, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    for (uint32_t i = 0; i < n; i++) h[i] = i * 40503u + 12345u;
    (void)hipMemcpy(s0, h.data(), (size_t)n * 4, hipMemcpyHostToDevice);
    for (int gx : {2048, 4096, 8192}) {
        float v2 = timeit([&] { hipLaunchKernelGGL(v2_k<2>, dim3(gx), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)s0, (u32x4 *)s1, dump, ng); }, 10);
        float w2 = timeit([&] { hipLaunchKernelGGL(v3_k<2>, dim3(gx), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)s0, (u32x4 *)s1, dump, ng); }, 10);
        float w1 = timeit([&] { hipLaunchKernelGGL(v3_k<1>, dim3(gx), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)s0, (u32x4 *)s1, dump, ng); }, 10);
        float w4 = timeit([&] { hipLaunchKernelGGL(v3_k<4>, dim3(gx), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)s0, (u32x4 *)s1, dump, ng); }, 10);
        float m0 = timeit([&] { hipLaunchKernelGGL(v5_k<0>, dim3(gx), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)s0, (u32x4 *)s1, dump, ng); }, 10);
        float m1 = timeit([&] { hipLaunchKernelGGL(v5_k<1>, dim3(gx), dim3(256), 0, 0, (const u32x4 *)inc, (const u32x4 *)s0, (u32x4 *)s1, dump, ng); }, 10);
        printf("grid %5d: mixed 64-bit (2 voices mad_u64 + add3, 2 add_co + s_bcnt) %.1f | all four 64-bit %.1f Gs/s\n", gx, n * 64.0 / m0 / 1e6, n * 64.0 / m1 / 1e6);
        printf("grid %5d: v2<2> %.1f | 2-pass NV2 %.1f NV1 %.1f NV4 %.1f Gs/s\n", gx, n * 64.0 / v2 / 1e6, n * 64.0 / w2 / 1e6, n * 64.0 / w1 / 1e6, n * 64.0 / w4 / 1e6);
    }
    return 0;
}
