// Scratch microbenchmark: what does the host wait for in a synchronous block (launch -> bus in host memory)?
//   A  kernel + hipStreamSynchronize                                  (no data)
//   B  kernel + hipMemcpyAsync D2H (256 B, pinned) + hipStreamSynchronize     (what smx_bank_fetch does)
//   C  kernel + publish kernel writing pinned host memory + flag, host polls the flag
//   D  ONE kernel that writes the pinned host memory + flag itself, host polls
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <algorithm>
#include <vector>

__global__ void work(int32_t *bus, int n) { if (threadIdx.x < (unsigned)n) atomicAdd(&bus[threadIdx.x], (int)threadIdx.x + 1); }
__global__ void publish(const int32_t *bus, volatile int32_t *hbus, volatile uint32_t *flag, int n, uint32_t seq)
{
    if (threadIdx.x < (unsigned)n) hbus[threadIdx.x] = bus[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *flag = seq;
}
__global__ void work_publish(int32_t *bus, volatile int32_t *hbus, volatile uint32_t *flag, int n, uint32_t seq)
{
    if (threadIdx.x < (unsigned)n) hbus[threadIdx.x] = (int)threadIdx.x + (int)seq;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *flag = seq;
}

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int32_t *bus; (void)hipMalloc(&bus, 4096); (void)hipMemset(bus, 0, 4096);
    int32_t *hb; (void)hipHostMalloc((void **)&hb, 4096, hipHostMallocDefault);
    int32_t *hc; uint32_t *hf;
    (void)hipHostMalloc((void **)&hc, 4096, hipHostMallocCoherent | hipHostMallocMapped);
    (void)hipHostMalloc((void **)&hf, 64, hipHostMallocCoherent | hipHostMallocMapped);
    int32_t *dc; uint32_t *df;
    (void)hipHostGetDevicePointer((void **)&dc, hc, 0); (void)hipHostGetDevicePointer((void **)&df, hf, 0);
    *hf = 0;
    (void)hipDeviceSynchronize();
    const int N = 2000, n = 64;
    auto stats = [&](const char *name, std::vector<double> &v) {
        std::sort(v.begin(), v.end());
        double m = 0; for (double x : v) m += x; m /= v.size();
        printf("%-70s mean %6.2f us  median %6.2f  p99 %6.2f  max %6.2f\n", name, m, v[v.size() / 2], v[v.size() * 99 / 100], v.back());
    };
    std::vector<double> v;
    for (int mode = 0; mode < 4; mode++) {
        v.clear();
        uint32_t seq = *hf;
        for (int i = 0; i < N + 100; i++) {
            const double t0 = now();
            if (mode == 0) { hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, bus, n); (void)hipStreamSynchronize(s); }
            if (mode == 1) { hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, bus, n); (void)hipMemcpyAsync(hb, bus, n * 4, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); }
            if (mode == 2) {
                seq++;
                hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, s, bus, n);
                hipLaunchKernelGGL(publish, dim3(1), dim3(64), 0, s, bus, dc, df, n, seq);
                while (*(volatile uint32_t *)hf != seq) { }
            }
            if (mode == 3) {
                seq++;
                hipLaunchKernelGGL(work_publish, dim3(1), dim3(64), 0, s, bus, dc, df, n, seq);
                while (*(volatile uint32_t *)hf != seq) { }
            }
            const double t1 = now();
            if (i >= 100) v.push_back(t1 - t0);
        }
        (void)hipStreamSynchronize(s);
        const char *names[] = {"A kernel + hipStreamSynchronize", "B kernel + hipMemcpyAsync D2H 256 B + hipStreamSynchronize (today)",
                               "C kernel + publish kernel to pinned host + flag, host polls", "D one kernel writing pinned host + flag, host polls"};
        stats(names[mode], v);
    }
    return 0;
}
