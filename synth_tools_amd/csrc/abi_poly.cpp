// abi_poly.cpp -- part of the C-ABI of libsynth_mi355x.so (include/synth_mi355x.h): poly voice bank
// Host side of the drop-in boundary.  No CPU compute fallback exists: every compute entry
// point needs a HIP device and fails with SMX_E_NOGPU otherwise.
#include "abi_internal.h"
// ---------------------------------------------------------------------------
// poly voice bank (build-defined extension; BASELINE config 4)
// ---------------------------------------------------------------------------
struct smx_poly {
    uint32_t n = 0, n_pad = 0;
    int device = 0;
    smx::PolyArrays d{};               // device arrays
    int32_t *d_bus2 = nullptr;         // two buses of int32[2*64], used alternately (the fold of block k is done by launch k+1)
    int32_t *d_bus = nullptr;          // the bus of the last block
    uint32_t bus_k = 0;
    int32_t *d_slots = nullptr;        // bus copies the workgroups add into (poly_bank.hip), kept zero
    smx::PolyPending pend;             // the fold the last launch left to its successor
    int32_t *h_bus = nullptr;          // pinned
    hipStream_t stream = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
};

static void poly_slots(smx::PolyArrays &d, void **slots[12])
{
    slots[0] = (void **)&d.inc;   slots[1] = (void **)&d.phase; slots[2] = (void **)&d.y;
    slots[3] = (void **)&d.a;     slots[4] = (void **)&d.level; slots[5] = (void **)&d.stage;
    slots[6] = (void **)&d.gate;  slots[7] = (void **)&d.ar;    slots[8] = (void **)&d.dr;
    slots[9] = (void **)&d.sl;    slots[10] = (void **)&d.rr;   slots[11] = (void **)&d.pan;
}

extern "C" smx_poly *smx_poly_create(uint32_t n_voices, int device)
{
    if (n_voices == 0 || n_voices > 0xFFFFF000u) { set_error("smx_poly_create: n_voices=%u (1..2^32-4096)", n_voices); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_poly_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_poly_create: device %d of %d", device, ndev); return nullptr; }
    smx_poly *p = new smx_poly();
    p->n = n_voices;
    p->n_pad = smx::round_up(n_voices, 1024);
    p->device = device;
    const size_t bytes = (size_t)p->n_pad * 4;
    void **slots[12];
    poly_slots(p->d, slots);
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&p->ev_t0) == hipSuccess && hipEventCreate(&p->ev_t1) == hipSuccess &&
              hipMalloc((void **)&p->d_bus2, 2 * smx::poly_bus_bytes()) == hipSuccess &&
              hipMemsetAsync(p->d_bus2, 0, 2 * smx::poly_bus_bytes(), p->stream) == hipSuccess &&
              hipMalloc((void **)&p->d_slots, smx::poly_scratch_bytes()) == hipSuccess &&
              hipMemsetAsync(p->d_slots, 0, smx::poly_scratch_bytes(), p->stream) == hipSuccess &&
              hipHostMalloc((void **)&p->h_bus, 128 * 4, hipHostMallocDefault) == hipSuccess;
    for (int i = 0; ok && i < 12; i++)
        ok = hipMalloc(slots[i], bytes) == hipSuccess &&
             hipMemsetAsync(*slots[i], 0, bytes, p->stream) == hipSuccess;   // inc 0: all voices off
    ok = ok && hipStreamSynchronize(p->stream) == hipSuccess;
    if (!ok) {
        set_error("smx_poly_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_poly_destroy(p);
        return nullptr;
    }
    return p;
}

extern "C" void smx_poly_destroy(smx_poly *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    void **slots[12];
    poly_slots(p->d, slots);
    for (int i = 0; i < 12; i++)
        if (*slots[i]) (void)hipFree(*slots[i]);
    if (p->d_bus2) (void)hipFree(p->d_bus2);
    if (p->d_slots) (void)hipFree(p->d_slots);
    if (p->h_bus) (void)hipHostFree(p->h_bus);
    if (p->ev_t0) (void)hipEventDestroy(p->ev_t0);
    if (p->ev_t1) (void)hipEventDestroy(p->ev_t1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

static int poly_copy(smx_poly *p, const struct smx_poly_arrays *a, bool to_device)
{
    if (!p || !a) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipStreamSynchronize(p->stream));
    void **slots[12];
    poly_slots(p->d, slots);
    void *host[12] = {a->inc, a->phase, a->y, a->a, a->level, a->stage,
                      a->gate, a->ar, a->dr, a->sl, a->rr, a->pan};
    for (int i = 0; i < 12; i++) {
        if (!host[i]) continue;
        if (to_device) SMX_HIP(hipMemcpy(*slots[i], host[i], (size_t)p->n * 4, hipMemcpyHostToDevice));
        else           SMX_HIP(hipMemcpy(host[i], *slots[i], (size_t)p->n * 4, hipMemcpyDeviceToHost));
    }
    return SMX_OK;
}

extern "C" int smx_poly_load(smx_poly *p, const struct smx_poly_arrays *a) { return poly_copy(p, a, true); }
extern "C" int smx_poly_read(smx_poly *p, const struct smx_poly_arrays *a) { return poly_copy(p, a, false); }

extern "C" int smx_poly_run_async(smx_poly *p, int n)
{
    if (!p || n <= 0 || n > 64) { set_error("smx_poly_run_async: n=%d (1..64)", n); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(p->device));
    static const bool no_defer = getenv("SMX_POLY_NO_DEFER") != nullptr;      // A/B switch: every launch folds its own copies
    p->d_bus = p->d_bus2 + (size_t)(p->bus_k++ & 1u) * (smx::poly_bus_bytes() / 4);
    return smx::launch_poly_bank(p->d, p->d_bus, p->d_slots, p->n_pad, (uint32_t)n, p->stream, no_defer ? nullptr : &p->pend);
}

// Whoever is about to read the last block's bus first runs the fold its launch deferred (a no-op when nothing is owed).
static int poly_flush(smx_poly *p) { return smx::launch_poly_flush(&p->pend, p->stream); }

extern "C" int smx_poly_sync(smx_poly *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    int rv = poly_flush(p);                          // after a sync the last block's bus holds its sums
    if (rv) return rv;
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

extern "C" int smx_poly_run(smx_poly *p, float *vec_lr, int32_t *bus_lr, int n)
{
    if (!p || n <= 0) { set_error("smx_poly_run: bad args"); return SMX_E_ARG; }
    for (int done = 0; done < n;) {
        const int nf = n - done < 64 ? n - done : 64;
        int rv = smx_poly_run_async(p, nf);
        if (rv) return rv;
        rv = poly_flush(p);
        if (rv) return rv;
        SMX_HIP(hipMemcpyAsync(p->h_bus, p->d_bus, (size_t)nf * 8, hipMemcpyDeviceToHost, p->stream));
        SMX_HIP(hipStreamSynchronize(p->stream));
        if (bus_lr) memcpy(bus_lr + 2 * done, p->h_bus, (size_t)nf * 8);
        if (vec_lr) for (int i = 0; i < 2 * nf; i++) vec_lr[2 * done + i] = bus_to_float(p->h_bus[i]);
        done += nf;
    }
    return SMX_OK;
}

extern "C" int smx_poly_timer_start(smx_poly *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipEventRecord(p->ev_t0, p->stream));
    return SMX_OK;
}

extern "C" int smx_poly_timer_stop(smx_poly *p, float *ms)
{
    if (!p || !ms) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipEventRecord(p->ev_t1, p->stream));
    SMX_HIP(hipEventSynchronize(p->ev_t1));
    SMX_HIP(hipEventElapsedTime(ms, p->ev_t0, p->ev_t1));
    return SMX_OK;
}

