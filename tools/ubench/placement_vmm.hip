// Scratch microbenchmark: does a recycled hipMalloc region stream slower than a fresh one, and does memory mapped through
// the virtual-memory API (hipMemCreate handles of a chosen size) behave differently?  Kernel: the tick kernel's read
// stream over two arrays of 256 MiB (64 Mi voices).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void stream2(const u32x4 *__restrict__ a4, const u32x4 *__restrict__ b4, uint32_t nrows, uint32_t *sink)
{
    const uint32_t tid = threadIdx.x;
    u32x4 acc = 0, an = 0, bn = 0;
    if (blockIdx.x < nrows) { an = __builtin_nontemporal_load(a4 + blockIdx.x * 1024u + tid); bn = __builtin_nontemporal_load(b4 + blockIdx.x * 1024u + tid); }
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const u32x4 a = an, b = bn;
        const uint32_t rn = min(row + gridDim.x, nrows - 1) * 1024u + tid;
        an = __builtin_nontemporal_load(a4 + rn); bn = __builtin_nontemporal_load(b4 + rn);
        acc ^= a + b;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x5EED5EEDu) *sink = 1;
}

static const size_t BYTES = (size_t)256 << 20;
static uint32_t *sink;
static float timeit(void *a, void *b)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t nrows = (uint32_t)(BYTES / 16 / 1024);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(stream2, dim3(256), dim3(1024), 0, 0, (const u32x4 *)a, (const u32x4 *)b, nrows, sink);
    (void)hipDeviceSynchronize(); (void)hipEventRecord(e0);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(stream2, dim3(256), dim3(1024), 0, 0, (const u32x4 *)a, (const u32x4 *)b, nrows, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 200 * 1e3f;
}

struct Vmm { void *va = nullptr; std::vector<hipMemGenericAllocationHandle_t> h; size_t chunk = 0; };
static int vmm_alloc(Vmm &v, size_t chunk)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (chunk < gran) chunk = gran;
    v.chunk = chunk;
    CK(hipMemAddressReserve(&v.va, BYTES, chunk, nullptr, 0));
    for (size_t off = 0; off < BYTES; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char *)v.va + off, chunk, 0, h, 0));
        v.h.push_back(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(v.va, BYTES, &acc, 1));
    return 0;
}
static void vmm_free(Vmm &v)
{
    (void)hipMemUnmap(v.va, BYTES);
    for (auto h : v.h) (void)hipMemRelease(h);
    (void)hipMemAddressFree(v.va, BYTES);
    v = Vmm();
}

int main()
{
    CK(hipMalloc(&sink, 4));
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    (void)hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
    (void)hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("VMM granularity: minimum %zu, recommended %zu bytes\n", gmin, grec);
    void *a[8], *b[8];
    printf("hipMalloc, 8 pairs kept alive:");
    for (int k = 0; k < 8; k++) { CK(hipMalloc(&a[k], BYTES)); CK(hipMalloc(&b[k], BYTES)); printf(" %.2f", timeit(a[k], b[k])); }
    printf(" us\n");
    for (int k = 0; k < 8; k++) { (void)hipFree(a[k]); (void)hipFree(b[k]); }
    for (int r = 0; r < 3; r++) {
        void *x, *y; CK(hipMalloc(&x, BYTES)); CK(hipMalloc(&y, BYTES));
        printf("hipMalloc after freeing everything (recycled): %.2f us\n", timeit(x, y));
        (void)hipFree(x); (void)hipFree(y);
    }
    for (size_t chunk : {(size_t)2 << 20, (size_t)32 << 20, BYTES}) {
        Vmm x, y;
        if (vmm_alloc(x, chunk) || vmm_alloc(y, chunk)) return 1;
        printf("VMM, handles of %4zu MiB: %.2f us", x.chunk >> 20, timeit(x.va, y.va));
        printf("  again: %.2f us\n", timeit(x.va, y.va));
        vmm_free(x); vmm_free(y);
    }
    { void *x, *y; CK(hipMalloc(&x, BYTES)); CK(hipMalloc(&y, BYTES)); printf("hipMalloc once more: %.2f us\n", timeit(x, y)); (void)hipFree(x); (void)hipFree(y); }
    // both arrays in ONE allocation, the second one skewed against the first
    for (int rep = 0; rep < 2; rep++) {
        char *base; CK(hipMalloc((void **)&base, 2 * BYTES + ((size_t)64 << 20)));
        printf("one allocation, skew of the second array:");
        for (size_t skew : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)262144, (size_t)1 << 20, (size_t)3 << 20, (size_t)17 << 20})
            printf("  %zu: %.2f", skew, timeit(base, base + BYTES + skew));
        printf(" us\n");
        (void)hipFree(base);
    }
    return 0;
}
