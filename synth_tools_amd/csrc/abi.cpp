// abi.cpp -- the C-ABI of libsynth_mi355x.so (include/synth_mi355x.h).
//
// Host side of the drop-in boundary: owns HBM-resident struct-of-arrays voice
// state, the HIP streams/events, the note allocator (linux/synth.c:145-165
// semantics widened to N voices) and the RCCL communicator for the
// multi-GPU bus sum.  No CPU compute fallback exists: every compute entry
// point needs a HIP device and fails with SMX_E_NOGPU otherwise.
#include "smx_common.h"
#include <rccl/rccl.h>
#include <cstdarg>
#include <vector>
#include <mutex>

namespace smx {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// First-free search over N voices in O(log64 N): the reference scans its 64
// voices linearly (linux/synth.c:147-149); a bank has up to 2^32.
class FreeMap {
public:
    void reset(uint32_t n, bool all_free)
    {
        n_ = n;
        levels_.clear();
        uint32_t bits = n;
        do {
            uint32_t words = (bits + 63) / 64;
            levels_.emplace_back(words, 0ull);
            bits = words;
        } while (bits > 1);
        if (all_free)
            for (uint32_t v = 0; v < n; v++) set_leaf_only(v);
        rebuild_summaries();
    }
    void load(const uint32_t *inc, uint32_t n)
    {
        reset(n, false);
        for (uint32_t v = 0; v < n; v++)
            if (inc[v] == 0) set_leaf_only(v);
        rebuild_summaries();
    }
    void set_free(uint32_t v, bool is_free)
    {
        uint32_t idx = v;
        for (size_t l = 0; l < levels_.size(); l++) {
            uint64_t &w = levels_[l][idx >> 6];
            const uint64_t bit = 1ull << (idx & 63);
            if (is_free) w |= bit; else w &= ~bit;
            const bool any = w != 0;
            idx >>= 6;
            if (l + 1 < levels_.size()) {
                const bool was = (levels_[l + 1][idx >> 6] >> (idx & 63)) & 1;
                if (was == any) break;
                is_free = any;
            }
        }
    }
    // index of the first free voice, or -1
    int64_t first_free() const
    {
        if (levels_.empty() || levels_.back()[0] == 0) return -1;
        uint32_t idx = 0;
        for (size_t l = levels_.size(); l-- > 0;) {
            const uint64_t w = levels_[l][idx];
            idx = idx * 64 + (uint32_t)__builtin_ctzll(w);
        }
        return idx < n_ ? (int64_t)idx : -1;
    }
private:
    void set_leaf_only(uint32_t v) { levels_[0][v >> 6] |= 1ull << (v & 63); }
    void rebuild_summaries()
    {
        for (size_t l = 1; l < levels_.size(); l++) {
            std::fill(levels_[l].begin(), levels_[l].end(), 0ull);
            for (size_t i = 0; i < levels_[l - 1].size(); i++)
                if (levels_[l - 1][i]) levels_[l][i >> 6] |= 1ull << (i & 63);
        }
    }
    uint32_t n_ = 0;
    std::vector<std::vector<uint64_t>> levels_;
};

}  // namespace smx

using smx::set_error;

// ---------------------------------------------------------------------------
// note tables: linux/synth.c:69-125.  The reference folds the top octave at
// compile time in double; the same double products evaluated at load time
// give the same IEEE results.
// ---------------------------------------------------------------------------
static uint32_t g_note_tab[12];
extern "C" const uint8_t midi_tab[128] = {
#define SMX_NOTE(o, n) (uint8_t)((((o) & 15) << 4) | ((n) & 15))
#define SMX_OCT(o)                                                                     \
    SMX_NOTE(o, 0), SMX_NOTE(o, 1), SMX_NOTE(o, 2), SMX_NOTE(o, 3), SMX_NOTE(o, 4),    \
    SMX_NOTE(o, 5), SMX_NOTE(o, 6), SMX_NOTE(o, 7), SMX_NOTE(o, 8), SMX_NOTE(o, 9),    \
    SMX_NOTE(o, 10), SMX_NOTE(o, 11)
    SMX_NOTE(10, 4), SMX_NOTE(10, 5), SMX_NOTE(10, 6), SMX_NOTE(10, 7),
    SMX_NOTE(10, 8), SMX_NOTE(10, 9), SMX_NOTE(10, 10), SMX_NOTE(10, 11),
    SMX_OCT(9), SMX_OCT(8), SMX_OCT(7), SMX_OCT(6), SMX_OCT(5),
    SMX_OCT(4), SMX_OCT(3), SMX_OCT(2), SMX_OCT(1), SMX_OCT(0),
#undef SMX_OCT
#undef SMX_NOTE
};
static std::once_flag g_tab_once;
static void init_note_tab()
{
    const double semitone_down = 0.9438743126816935;          // 2^(-1/12)
    double x = (12543.853951415975 / 48000.0) * 4294967296.0;  // MIDI 127 @48 kHz, 32-bit phasor
    for (int i = 11; i >= 0; i--) {
        g_note_tab[i] = (uint32_t)x;
        x = semitone_down * x;
    }
}

extern "C" phasor_t note_to_inc(int note)
{
    std::call_once(g_tab_once, init_note_tab);
    const int on = midi_tab[note & 127];
    return g_note_tab[on & 15] >> (on >> 4);
}

// ---------------------------------------------------------------------------
// misc
// ---------------------------------------------------------------------------
extern "C" const char *smx_last_error(void) { return smx::g_err; }
extern "C" int smx_version(void) { return 1; }
extern "C" int smx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------
// saw bank
// ---------------------------------------------------------------------------
struct smx_bank {
    uint32_t n = 0, n_pad = 0;
    int device = 0;
    uint32_t *d_inc = nullptr;
    uint32_t *d_state[2] = {nullptr, nullptr};   // ping-pong (saw_bank.hip)
    int cur = 0;
    // three bus buffers in rotation: [cur] holds the last block (and may be feeding an
    // all-reduce), [cur+1] was zeroed by the last launch for the next one, [cur+2] is
    // the one the next launch will zero.
    static constexpr int NBUS = 3;
    int32_t *d_bus[NBUS] = {nullptr, nullptr, nullptr};
    uint32_t bus_zeroed[NBUS] = {0, 0, 0};       // leading frames known to be zero
    int bus_cur = 0;
    uint32_t bus_cap = 0;
    int32_t *h_bus = nullptr;                    // pinned
    void *d_scratch = nullptr;                   // partial sums of saw_bank.hip's carry formulation
    hipStream_t stream = nullptr, comm_stream = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    hipEvent_t ev_kernel[NBUS] = {nullptr, nullptr, nullptr};   // kernel of bus[i] finished
    hipEvent_t ev_comm[NBUS] = {nullptr, nullptr, nullptr};     // all-reduce of bus[i] finished
    bool comm_pending[NBUS] = {false, false, false};
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    int note2voice[128];
    smx::FreeMap free_map;
};

static int bank_ensure_bus(smx_bank *b, uint32_t n)
{
    if (n <= b->bus_cap) return SMX_OK;
    SMX_HIP(hipStreamSynchronize(b->stream));
    if (b->comm_stream) SMX_HIP(hipStreamSynchronize(b->comm_stream));
    const uint32_t cap = smx::round_up(n < 4096 ? 4096 : n, 4096);
    for (int i = 0; i < smx_bank::NBUS; i++) {
        if (b->d_bus[i]) SMX_HIP(hipFree(b->d_bus[i]));
        b->d_bus[i] = nullptr;
        SMX_HIP(hipMalloc((void **)&b->d_bus[i], (size_t)cap * 4));
        SMX_HIP(hipMemset(b->d_bus[i], 0, (size_t)cap * 4));
        b->bus_zeroed[i] = cap;
        b->comm_pending[i] = false;
    }
    if (b->h_bus) SMX_HIP(hipHostFree(b->h_bus));
    b->h_bus = nullptr;
    SMX_HIP(hipHostMalloc((void **)&b->h_bus, (size_t)cap * 4, hipHostMallocDefault));
    if (b->d_scratch) SMX_HIP(hipFree(b->d_scratch));
    b->d_scratch = nullptr;
    if (b->n_pad >= (1u << 20)) {
        SMX_HIP(hipMalloc(&b->d_scratch, smx::saw_scratch_bytes(cap)));
        SMX_HIP(hipMemset(b->d_scratch, 0, smx::saw_scratch_bytes(cap)));   // slots are kept zero between launches
    }
    b->bus_cap = cap;
    return SMX_OK;
}

extern "C" smx_bank *smx_bank_create(uint32_t n_voices, int device)
{
    if (n_voices == 0 || n_voices > 0xFFFFF000u) { set_error("smx_bank_create: n_voices=%u (1..2^32-4096)", n_voices); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_bank_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_bank_create: device %d of %d", device, ndev); return nullptr; }
    smx_bank *b = new smx_bank();
    b->n = n_voices;
    b->n_pad = smx::round_up(n_voices, 1024);
    b->device = device;
    auto fail = [&](const char *what, hipError_t e) -> smx_bank * {
        set_error("smx_bank_create: %s: %s", what, hipGetErrorString(e));
        smx_bank_destroy(b);
        return nullptr;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    const size_t bytes = (size_t)b->n_pad * 4;
    if ((e = hipMalloc((void **)&b->d_inc, bytes)) != hipSuccess) return fail("hipMalloc inc", e);
    for (int i = 0; i < 2; i++)
        if ((e = hipMalloc((void **)&b->d_state[i], bytes)) != hipSuccess) return fail("hipMalloc state", e);
    if ((e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)) != hipSuccess) return fail("stream", e);
    if ((e = hipEventCreate(&b->ev_t0)) != hipSuccess) return fail("event", e);
    if ((e = hipEventCreate(&b->ev_t1)) != hipSuccess) return fail("event", e);
    for (int i = 0; i < smx_bank::NBUS; i++) {
        if ((e = hipEventCreateWithFlags(&b->ev_kernel[i], hipEventDisableTiming)) != hipSuccess) return fail("event", e);
        if ((e = hipEventCreateWithFlags(&b->ev_comm[i], hipEventDisableTiming)) != hipSuccess) return fail("event", e);
    }
    // synth_init: bzero (linux/synth.c:204-206); padding voices stay off forever
    if ((e = hipMemsetAsync(b->d_inc, 0, bytes, b->stream)) != hipSuccess) return fail("memset", e);
    for (int i = 0; i < 2; i++)
        if ((e = hipMemsetAsync(b->d_state[i], 0, bytes, b->stream)) != hipSuccess) return fail("memset", e);
    if ((e = hipStreamSynchronize(b->stream)) != hipSuccess) return fail("sync", e);
    memset(b->note2voice, 0, sizeof(b->note2voice));
    b->free_map.reset(b->n, true);
    if (bank_ensure_bus(b, 4096) != SMX_OK) { smx_bank_destroy(b); return nullptr; }
    return b;
}

extern "C" void smx_bank_destroy(smx_bank *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    if (b->comm_stream) (void)hipStreamSynchronize(b->comm_stream);
    if (b->comm) (void)ncclCommDestroy(b->comm);
    if (b->d_inc) (void)hipFree(b->d_inc);
    for (int i = 0; i < 2; i++)
        if (b->d_state[i]) (void)hipFree(b->d_state[i]);
    for (int i = 0; i < smx_bank::NBUS; i++) {
        if (b->d_bus[i]) (void)hipFree(b->d_bus[i]);
        if (b->ev_kernel[i]) (void)hipEventDestroy(b->ev_kernel[i]);
        if (b->ev_comm[i]) (void)hipEventDestroy(b->ev_comm[i]);
    }
    if (b->h_bus) (void)hipHostFree(b->h_bus);
    if (b->d_scratch) (void)hipFree(b->d_scratch);
    if (b->ev_t0) (void)hipEventDestroy(b->ev_t0);
    if (b->ev_t1) (void)hipEventDestroy(b->ev_t1);
    if (b->comm_stream) (void)hipStreamDestroy(b->comm_stream);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

extern "C" uint32_t smx_bank_voices(const smx_bank *b) { return b ? b->n : 0; }

extern "C" int smx_bank_load(smx_bank *b, const uint32_t *inc, const uint32_t *state)
{
    if (!b) { set_error("smx_bank_load: null bank"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    SMX_HIP(hipStreamSynchronize(b->stream));
    if (inc) {
        SMX_HIP(hipMemcpy(b->d_inc, inc, (size_t)b->n * 4, hipMemcpyHostToDevice));
        b->free_map.load(inc, b->n);
    }
    if (state)
        SMX_HIP(hipMemcpy(b->d_state[b->cur], state, (size_t)b->n * 4, hipMemcpyHostToDevice));
    return SMX_OK;
}

extern "C" int smx_bank_read(smx_bank *b, uint32_t *inc, uint32_t *state)
{
    if (!b) { set_error("smx_bank_read: null bank"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    SMX_HIP(hipStreamSynchronize(b->stream));
    if (inc) SMX_HIP(hipMemcpy(inc, b->d_inc, (size_t)b->n * 4, hipMemcpyDeviceToHost));
    if (state) SMX_HIP(hipMemcpy(state, b->d_state[b->cur], (size_t)b->n * 4, hipMemcpyDeviceToHost));
    return SMX_OK;
}

static int bank_set_inc(smx_bank *b, uint32_t v, uint32_t inc)
{
    SMX_HIP(hipSetDevice(b->device));
    // pageable 4-byte source: hipMemcpyAsync stages it before returning
    SMX_HIP(hipMemcpyAsync(b->d_inc + v, &inc, 4, hipMemcpyHostToDevice, b->stream));
    SMX_HIP(hipStreamSynchronize(b->stream));
    b->free_map.set_free(v, inc == 0);
    return SMX_OK;
}

// linux/synth.c:156-160 over N voices
extern "C" int smx_bank_note_on(smx_bank *b, int note)
{
    if (!b) { set_error("smx_bank_note_on: null bank"); return SMX_E_ARG; }
    if (note < 0) { set_error("smx_bank_note_on: note %d < 0", note); return SMX_E_ARG; }
    int64_t v = b->free_map.first_free();
    if (v < 0) v = 0;                                  // steal voice 0 (:150-153)
    b->note2voice[note % 128] = (int)v;
    return bank_set_inc(b, (uint32_t)v, note_to_inc(note % 128));
}

// linux/synth.c:161-165 over N voices
extern "C" int smx_bank_note_off(smx_bank *b, int note)
{
    if (!b) { set_error("smx_bank_note_off: null bank"); return SMX_E_ARG; }
    if (note < 0) { set_error("smx_bank_note_off: note %d < 0", note); return SMX_E_ARG; }
    const int v = b->note2voice[note % 128];
    b->note2voice[note % 128] = 0;
    return bank_set_inc(b, (uint32_t)v, 0);
}

// Wait (on the compute stream) until nothing in flight still uses bus buffer i.
static int bank_bus_release(smx_bank *b, int i)
{
    if (b->comm_pending[i]) {
        SMX_HIP(hipStreamWaitEvent(b->stream, b->ev_comm[i], 0));
        b->comm_pending[i] = false;
    }
    return SMX_OK;
}

// Rotate to the next bus buffer and make sure its first n frames are zero.
static int bank_bus_advance(smx_bank *b, uint32_t n, int *bi_out, int *bnext_out)
{
    const int bi = (b->bus_cur + 1) % smx_bank::NBUS;
    const int bnext = (bi + 1) % smx_bank::NBUS;
    int rv = bank_bus_release(b, bi);
    if (rv) return rv;
    rv = bank_bus_release(b, bnext);      // the launch is about to zero it
    if (rv) return rv;
    if (b->bus_zeroed[bi] < n) {
        SMX_HIP(hipMemsetAsync(b->d_bus[bi], 0, (size_t)n * 4, b->stream));
        b->bus_zeroed[bi] = n;
    }
    *bi_out = bi;
    *bnext_out = bnext;
    return SMX_OK;
}

extern "C" int smx_bank_run_async(smx_bank *b, int n)
{
    if (!b || n <= 0) { set_error("smx_bank_run_async: bad args"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    int rv = bank_ensure_bus(b, (uint32_t)n);
    if (rv) return rv;
    int bi, bnext;
    rv = bank_bus_advance(b, (uint32_t)n, &bi, &bnext);
    if (rv) return rv;
    rv = smx::launch_saw_bank(b->d_inc, b->d_state[b->cur], b->d_state[b->cur ^ 1], b->d_bus[bi],
                              b->d_bus[bnext], b->n_pad, (uint32_t)n, b->d_scratch, b->stream);
    if (rv) return rv;
    b->bus_zeroed[bi] = 0;                 // now holds this block's sums
    b->bus_zeroed[bnext] = (uint32_t)n;    // cleared by the launch
    b->cur ^= 1;
    b->bus_cur = bi;
    return SMX_OK;
}

extern "C" void *smx_bank_bus_dev(smx_bank *b) { return b ? b->d_bus[b->bus_cur] : nullptr; }

extern "C" int smx_bank_sync(smx_bank *b)
{
    if (!b) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(b->device));
    SMX_HIP(hipStreamSynchronize(b->stream));
    if (b->comm_stream) SMX_HIP(hipStreamSynchronize(b->comm_stream));
    return SMX_OK;
}

// linux/synth.c:180: (1.0 / 2^32) * (float)sum, product in double, result float
static inline float bus_to_float(int32_t sum)
{
    return (float)((1.0 / 4294967296.0) * (double)(float)sum);
}

extern "C" int smx_bank_fetch(smx_bank *b, float *vec, int32_t *bus, int n)
{
    if (!b || n <= 0 || (uint32_t)n > b->bus_cap) { set_error("smx_bank_fetch: bad args"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    const int bi = b->bus_cur;
    if (b->comm_pending[bi]) {
        SMX_HIP(hipStreamWaitEvent(b->stream, b->ev_comm[bi], 0));
        b->comm_pending[bi] = false;
    }
    SMX_HIP(hipMemcpyAsync(b->h_bus, b->d_bus[bi], (size_t)n * 4, hipMemcpyDeviceToHost, b->stream));
    SMX_HIP(hipStreamSynchronize(b->stream));
    if (bus) memcpy(bus, b->h_bus, (size_t)n * 4);
    if (vec) for (int i = 0; i < n; i++) vec[i] = bus_to_float(b->h_bus[i]);
    return SMX_OK;
}

extern "C" int smx_bank_run(smx_bank *b, float *vec, int32_t *bus, int n)
{
    int rv = smx_bank_run_async(b, n);
    if (rv) return rv;
    return smx_bank_fetch(b, vec, bus, n);
}

extern "C" int smx_bank_run_square(smx_bank *b, float *vec, int n)
{
    if (!b || n <= 0) { set_error("smx_bank_run_square: bad args"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    int rv = bank_ensure_bus(b, (uint32_t)n);
    if (rv) return rv;
    const int bi = (b->bus_cur + 1) % smx_bank::NBUS;
    rv = bank_bus_release(b, bi);
    if (rv) return rv;
    SMX_HIP(hipMemsetAsync(b->d_bus[bi], 0, (size_t)n * 4, b->stream));
    b->bus_zeroed[bi] = 0;
    rv = smx::launch_square_bank(b->d_inc, b->d_state[b->cur], b->d_state[b->cur ^ 1],
                                 (uint32_t *)b->d_bus[bi], b->n_pad, (uint32_t)n, b->stream);
    if (rv) return rv;
    b->cur ^= 1;
    b->bus_cur = bi;
    SMX_HIP(hipMemcpyAsync(b->h_bus, b->d_bus[bi], (size_t)n * 4, hipMemcpyDeviceToHost, b->stream));
    SMX_HIP(hipStreamSynchronize(b->stream));
    // linux/synth.c:194: (1.0 / 2^32) * (float)accu, accu unsigned
    if (vec)
        for (int i = 0; i < n; i++)
            vec[i] = (float)((1.0 / 4294967296.0) * (double)(float)(uint32_t)b->h_bus[i]);
    return SMX_OK;
}

extern "C" int smx_bank_timer_start(smx_bank *b)
{
    if (!b) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(b->device));
    SMX_HIP(hipEventRecord(b->ev_t0, b->stream));
    return SMX_OK;
}

extern "C" int smx_bank_timer_stop(smx_bank *b, float *ms)
{
    if (!b || !ms) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(b->device));
    SMX_HIP(hipEventRecord(b->ev_t1, b->stream));
    SMX_HIP(hipEventSynchronize(b->ev_t1));
    SMX_HIP(hipEventElapsedTime(ms, b->ev_t0, b->ev_t1));
    return SMX_OK;
}

// ---- multi-GPU ---------------------------------------------------------------
#define SMX_NCCL(expr)                                                         \
    do {                                                                       \
        ncclResult_t r_ = (expr);                                              \
        if (r_ != ncclSuccess) {                                               \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                      ncclGetErrorString(r_));                                 \
            return SMX_E_COMM;                                                 \
        }                                                                      \
    } while (0)

static_assert(sizeof(ncclUniqueId) == SMX_UNIQUE_ID_BYTES, "ncclUniqueId size");

extern "C" int smx_comm_unique_id(uint8_t id[SMX_UNIQUE_ID_BYTES])
{
    ncclUniqueId u;
    SMX_NCCL(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return SMX_OK;
}

extern "C" int smx_bank_comm_init(smx_bank *b, int rank, int nranks,
                                  const uint8_t id[SMX_UNIQUE_ID_BYTES])
{
    if (!b || nranks < 1 || rank < 0 || rank >= nranks) { set_error("smx_bank_comm_init: bad args"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    SMX_NCCL(ncclCommInitRank(&b->comm, nranks, u, rank));
    SMX_HIP(hipStreamCreateWithFlags(&b->comm_stream, hipStreamNonBlocking));
    b->rank = rank;
    b->nranks = nranks;
    return SMX_OK;
}

extern "C" int smx_bank_allreduce_async(smx_bank *b, int n)
{
    if (!b || n <= 0 || (uint32_t)n > b->bus_cap) { set_error("smx_bank_allreduce_async: bad args"); return SMX_E_ARG; }
    if (!b->comm) { set_error("smx_bank_allreduce_async: smx_bank_comm_init not called"); return SMX_E_STATE; }
    SMX_HIP(hipSetDevice(b->device));
    const int bi = b->bus_cur;
    SMX_HIP(hipEventRecord(b->ev_kernel[bi], b->stream));
    SMX_HIP(hipStreamWaitEvent(b->comm_stream, b->ev_kernel[bi], 0));
    // integer sum: associative, so the result is the same bits in any order
    SMX_NCCL(ncclAllReduce(b->d_bus[bi], b->d_bus[bi], (size_t)n, ncclInt32, ncclSum, b->comm,
                           b->comm_stream));
    SMX_HIP(hipEventRecord(b->ev_comm[bi], b->comm_stream));
    b->comm_pending[bi] = true;
    return SMX_OK;
}

// ---------------------------------------------------------------------------
// Linux drop-in: linux/synth.c:42-45, 145-165, 196-206
// ---------------------------------------------------------------------------
static smx_bank *g_dropin = nullptr;      // 64-voice scratch bank, JACK RT thread only
static std::mutex g_dropin_mu;

extern "C" int voice_alloc(struct synth *x)
{
    for (unsigned v = 0; v < 64; v++)
        if (x->voice[v].note_inc == 0) return (int)v;
    return 0;
}

extern "C" void synth_note_on(struct synth *x, int note)
{
    const int v = voice_alloc(x);
    x->note2voice[note % 128] = v;
    x->voice[v].note_inc = note_to_inc(note % 128);
}

extern "C" void synth_note_off(struct synth *x, int note)
{
    const int v = x->note2voice[note % 128];
    x->note2voice[note % 128] = 0;
    x->voice[v].note_inc = 0;
}

extern "C" void synth_init(struct synth *x) { memset(x, 0, sizeof(*x)); }

extern "C" void synth_midi_event(struct synth *x, const uint8_t *msg, size_t size)
{
    if (size != 3) return;
    if (msg[0] == 0x90) {                     // note on, channel 0 (linux/synth.c:246-256)
        if (msg[2] == 0) synth_note_off(x, msg[1]);
        else synth_note_on(x, msg[1]);
    } else if (msg[0] == 0x80) {              // note off, channel 0 (:257-261)
        synth_note_off(x, msg[1]);
    }                                         // CC 23..31 on 0xB0: accepted, no action (:240-245)
}

extern "C" void synth_run(struct synth *x, float *vec, int n)
{
    if (n <= 0) return;
    std::lock_guard<std::mutex> lock(g_dropin_mu);
    if (!g_dropin) {
        g_dropin = smx_bank_create(64, 0);
        if (!g_dropin) SMX_ASSERT_OK(SMX_E_NOGPU, "synth_run: smx_bank_create");
    }
    uint32_t inc[64], state[64];
    for (int v = 0; v < 64; v++) { inc[v] = x->voice[v].note_inc; state[v] = x->voice[v].note_state; }
    SMX_ASSERT_OK(smx_bank_load(g_dropin, inc, state), "synth_run: load");
    SMX_ASSERT_OK(smx_bank_run(g_dropin, vec, nullptr, n), "synth_run: run");
    SMX_ASSERT_OK(smx_bank_read(g_dropin, nullptr, state), "synth_run: read");
    for (int v = 0; v < 64; v++) x->voice[v].note_state = state[v];
}

// ---------------------------------------------------------------------------
// carry-out PDM bank: stm32f103/mod_pdm.c
// ---------------------------------------------------------------------------
struct smx_pdm {
    uint32_t n = 0, n_pad = 0;
    int device = 0;
    uint32_t *d_setpoint = nullptr, *d_accu = nullptr;
    uint32_t *d_dither = nullptr; uint32_t dither_cap = 0;
    uint32_t *d_bits = nullptr; size_t bits_cap = 0;     // bytes
    hipStream_t stream = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
};

extern "C" uint32_t pdm_safe_setpoint(uint32_t setpoint) { return setpoint; }  // mod_pdm.c:101-107

extern "C" smx_pdm *smx_pdm_create(uint32_t n_channels, int device)
{
    if (n_channels == 0 || n_channels > 0xFFFFF000u) { set_error("smx_pdm_create: n_channels=%u (1..2^32-4096)", n_channels); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_pdm_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_pdm_create: device %d of %d", device, ndev); return nullptr; }
    smx_pdm *p = new smx_pdm();
    p->n = n_channels;
    p->n_pad = smx::round_up(n_channels, 1024);
    p->device = device;
    const size_t bytes = (size_t)p->n_pad * 4;
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipMalloc((void **)&p->d_setpoint, bytes) == hipSuccess &&
              hipMalloc((void **)&p->d_accu, bytes) == hipSuccess &&
              hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&p->ev_t0) == hipSuccess && hipEventCreate(&p->ev_t1) == hipSuccess &&
              hipMemsetAsync(p->d_setpoint, 0, bytes, p->stream) == hipSuccess &&
              hipMemsetAsync(p->d_accu, 0, bytes, p->stream) == hipSuccess &&
              hipStreamSynchronize(p->stream) == hipSuccess;
    if (!ok) {
        set_error("smx_pdm_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_pdm_destroy(p);
        return nullptr;
    }
    return p;
}

extern "C" void smx_pdm_destroy(smx_pdm *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->d_setpoint) (void)hipFree(p->d_setpoint);
    if (p->d_accu) (void)hipFree(p->d_accu);
    if (p->d_dither) (void)hipFree(p->d_dither);
    if (p->d_bits) (void)hipFree(p->d_bits);
    if (p->ev_t0) (void)hipEventDestroy(p->ev_t0);
    if (p->ev_t1) (void)hipEventDestroy(p->ev_t1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

extern "C" int smx_pdm_load(smx_pdm *p, const uint32_t *setpoint, const uint32_t *accu)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipStreamSynchronize(p->stream));
    if (setpoint) SMX_HIP(hipMemcpy(p->d_setpoint, setpoint, (size_t)p->n * 4, hipMemcpyHostToDevice));
    if (accu) SMX_HIP(hipMemcpy(p->d_accu, accu, (size_t)p->n * 4, hipMemcpyHostToDevice));
    return SMX_OK;
}

extern "C" int smx_pdm_read(smx_pdm *p, uint32_t *setpoint, uint32_t *accu)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipStreamSynchronize(p->stream));
    if (setpoint) SMX_HIP(hipMemcpy(setpoint, p->d_setpoint, (size_t)p->n * 4, hipMemcpyDeviceToHost));
    if (accu) SMX_HIP(hipMemcpy(accu, p->d_accu, (size_t)p->n * 4, hipMemcpyDeviceToHost));
    return SMX_OK;
}

// pdm_init, mod_pdm.c:320-326
extern "C" int smx_pdm_init(smx_pdm *p)
{
    if (!p) return SMX_E_ARG;
    std::vector<uint32_t> sp(p->n, pdm_safe_setpoint(0x40000000u)), ac(p->n, 0u);
    sp[0] = 2000000000u;
    return smx_pdm_load(p, sp.data(), ac.data());
}

// SETPOINT, mod_synth.c:104-111
extern "C" int smx_pdm_set_setpoint(smx_pdm *p, uint32_t chan, uint32_t val)
{
    if (!p) return SMX_E_ARG;
    if (chan >= p->n) { set_error("smx_pdm_set_setpoint: chan %u >= %u", chan, p->n); return SMX_E_RANGE; }
    SMX_HIP(hipSetDevice(p->device));
    const uint32_t v = pdm_safe_setpoint(val);
    SMX_HIP(hipMemcpyAsync(p->d_setpoint + chan, &v, 4, hipMemcpyHostToDevice, p->stream));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

static int pdm_ensure(smx_pdm *p, uint32_t n_ticks)
{
    const size_t need = (size_t)n_ticks * (p->n_pad / 8);
    if (need > p->bits_cap) {
        SMX_HIP(hipStreamSynchronize(p->stream));
        if (p->d_bits) SMX_HIP(hipFree(p->d_bits));
        p->d_bits = nullptr; p->bits_cap = 0;
        SMX_HIP(hipMalloc((void **)&p->d_bits, need));
        p->bits_cap = need;
    }
    if (n_ticks > p->dither_cap) {
        SMX_HIP(hipStreamSynchronize(p->stream));
        if (p->d_dither) SMX_HIP(hipFree(p->d_dither));
        p->d_dither = nullptr; p->dither_cap = 0;
        SMX_HIP(hipMalloc((void **)&p->d_dither, (size_t)n_ticks * 4));
        p->dither_cap = n_ticks;
    }
    return SMX_OK;
}

extern "C" void *smx_pdm_dither_dev(smx_pdm *p, uint32_t n_ticks)
{
    if (!p || hipSetDevice(p->device) != hipSuccess || pdm_ensure(p, n_ticks) != SMX_OK) return nullptr;
    return p->d_dither;
}

extern "C" void *smx_pdm_bits_dev(smx_pdm *p) { return p ? p->d_bits : nullptr; }

extern "C" int smx_pdm_tick_n_async(smx_pdm *p, uint32_t n_ticks, int with_dither)
{
    if (!p) return SMX_E_ARG;
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pdm_ensure(p, n_ticks);
    if (rv) return rv;
    return smx::launch_pdm_bank(p->d_setpoint, p->d_accu, with_dither ? p->d_dither : nullptr,
                                p->d_bits, p->n_pad, p->n, n_ticks, p->stream);
}

extern "C" int smx_pdm_sync(smx_pdm *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

extern "C" int smx_pdm_tick_n(smx_pdm *p, uint32_t n_ticks, const uint32_t *dither, uint32_t *bits)
{
    if (!p) return SMX_E_ARG;
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pdm_ensure(p, n_ticks);
    if (rv) return rv;
    if (dither)
        SMX_HIP(hipMemcpyAsync(p->d_dither, dither, (size_t)n_ticks * 4, hipMemcpyHostToDevice, p->stream));
    rv = smx_pdm_tick_n_async(p, n_ticks, dither != nullptr);
    if (rv) return rv;
    if (bits) {
        const size_t words = (p->n + 31) / 32;
        SMX_HIP(hipMemcpy2DAsync(bits, words * 4, p->d_bits, (size_t)p->n_pad / 8, words * 4, n_ticks,
                                 hipMemcpyDeviceToHost, p->stream));
    }
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

extern "C" int smx_pdm_timer_start(smx_pdm *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipEventRecord(p->ev_t0, p->stream));
    return SMX_OK;
}

extern "C" int smx_pdm_timer_stop(smx_pdm *p, float *ms)
{
    if (!p || !ms) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipEventRecord(p->ev_t1, p->stream));
    SMX_HIP(hipEventSynchronize(p->ev_t1));
    SMX_HIP(hipEventElapsedTime(ms, p->ev_t0, p->ev_t1));
    return SMX_OK;
}

// mod_pdm.c:271-286: the reference's rrx register holds channel c at bit
// 32-nb+c; shifted right by (32-nb-4) that is bit 4+c.
extern "C" uint32_t smx_pdm_bsrr_word(uint32_t pulse_bits, uint32_t nb)
{
    if (nb == 0 || nb > 12) return 0;
    const uint32_t mask = ((1u << nb) - 1) << 4;
    const uint32_t set = (pulse_bits << 4) & mask;
    const uint32_t clr = (~set) & mask;
    return set | (clr << 16);
}

// ---------------------------------------------------------------------------
// poly voice bank (build-defined extension; BASELINE config 4)
// ---------------------------------------------------------------------------
struct smx_poly {
    uint32_t n = 0, n_pad = 0;
    int device = 0;
    smx::PolyArrays d{};               // device arrays
    int32_t *d_bus = nullptr;          // int32[2*64]
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
              hipMalloc((void **)&p->d_bus, 128 * 4) == hipSuccess &&
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
    if (p->d_bus) (void)hipFree(p->d_bus);
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
    SMX_HIP(hipMemsetAsync(p->d_bus, 0, (size_t)n * 8, p->stream));
    return smx::launch_poly_bank(p->d, p->d_bus, p->n_pad, (uint32_t)n, p->stream);
}

extern "C" int smx_poly_sync(smx_poly *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
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

// ---------------------------------------------------------------------------
// noise-shaped PWM bank: mod_pdm_pwm.c + pdm.h + mod_controlrate.c
// ---------------------------------------------------------------------------
struct smx_pwm {
    uint32_t n = 0, n_pad = 0;
    int order = 2, device = 0;
    uint32_t div_log = 12, out_shift = 24, div_count = 0;
    smx::PwmArrays d{};
    uint32_t *d_dither = nullptr; uint32_t dither_cap = 0;
    uint8_t *d_duty = nullptr; size_t duty_cap = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
};

static int pwm_slots(smx::PwmArrays &d, int order, void **slots[9])
{
    slots[0] = (void **)&d.setpoint; slots[1] = (void **)&d.pos0; slots[2] = (void **)&d.vel0;
    slots[3] = (void **)&d.pos1;     slots[4] = (void **)&d.vel1;
    for (int k = 0; k < order; k++) slots[5 + k] = (void **)&d.s[k];
    return 5 + order;
}

extern "C" smx_pwm *smx_pwm_create(uint32_t n_channels, int order, int device)
{
    if (n_channels == 0 || n_channels > 0xFFFFF000u || order < 1 || order > 4) {
        set_error("smx_pwm_create: n_channels=%u order=%d", n_channels, order);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_pwm_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_pwm_create: device %d of %d", device, ndev); return nullptr; }
    smx_pwm *p = new smx_pwm();
    p->n = n_channels;
    p->n_pad = smx::round_up(n_channels, 1024);
    p->order = order;
    p->device = device;
    const size_t bytes = (size_t)p->n_pad * 4;
    void **slots[9];
    const int ns = pwm_slots(p->d, order, slots);
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreate(&p->ev_t0) == hipSuccess && hipEventCreate(&p->ev_t1) == hipSuccess;
    for (int i = 0; ok && i < ns; i++)
        ok = hipMalloc(slots[i], bytes) == hipSuccess &&
             hipMemsetAsync(*slots[i], 0, bytes, p->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(p->stream) == hipSuccess;
    if (!ok) {
        set_error("smx_pwm_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_pwm_destroy(p);
        return nullptr;
    }
    return p;
}

extern "C" void smx_pwm_destroy(smx_pwm *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    void **slots[9];
    const int ns = pwm_slots(p->d, p->order, slots);
    for (int i = 0; i < ns; i++)
        if (*slots[i]) (void)hipFree(*slots[i]);
    if (p->d_dither) (void)hipFree(p->d_dither);
    if (p->d_duty) (void)hipFree(p->d_duty);
    if (p->ev_t0) (void)hipEventDestroy(p->ev_t0);
    if (p->ev_t1) (void)hipEventDestroy(p->ev_t1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

extern "C" int smx_pwm_config(smx_pwm *p, uint32_t control_div_log, uint32_t out_shift)
{
    if (!p || control_div_log == 0 || control_div_log > 31 || out_shift > 31) {
        set_error("smx_pwm_config: div_log=%u out_shift=%u", control_div_log, out_shift);
        return SMX_E_ARG;
    }
    p->div_log = control_div_log;
    p->out_shift = out_shift;
    p->div_count &= (1u << control_div_log) - 1;
    return SMX_OK;
}

static int pwm_copy(smx_pwm *p, const struct smx_pwm_arrays *a, bool to_device)
{
    if (!p || !a) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipStreamSynchronize(p->stream));
    void **slots[9];
    const int ns = pwm_slots(p->d, p->order, slots);
    void *host[9] = {a->setpoint, a->pos0, a->vel0, a->pos1, a->vel1, a->s[0], a->s[1], a->s[2], a->s[3]};
    for (int i = 0; i < ns; i++) {
        if (!host[i]) continue;
        if (to_device) SMX_HIP(hipMemcpy(*slots[i], host[i], (size_t)p->n * 4, hipMemcpyHostToDevice));
        else           SMX_HIP(hipMemcpy(host[i], *slots[i], (size_t)p->n * 4, hipMemcpyDeviceToHost));
    }
    return SMX_OK;
}

extern "C" int smx_pwm_load(smx_pwm *p, const struct smx_pwm_arrays *a) { return pwm_copy(p, a, true); }
extern "C" int smx_pwm_read(smx_pwm *p, const struct smx_pwm_arrays *a) { return pwm_copy(p, a, false); }

extern "C" int smx_pwm_set_div_count(smx_pwm *p, uint32_t c)
{
    if (!p || c >= (1u << p->div_log)) return SMX_E_ARG;
    p->div_count = c;
    return SMX_OK;
}
extern "C" uint32_t smx_pwm_div_count(const smx_pwm *p) { return p ? p->div_count : 0; }

// pdm_init, mod_pdm_pwm.c:147-160
extern "C" int smx_pwm_init(smx_pwm *p)
{
    if (!p) return SMX_E_ARG;
    std::vector<uint32_t> sp(p->n, pdm_safe_setpoint(0x40000000u)), z(p->n, 0u);
    sp[0] = 2000000000u;
    struct smx_pwm_arrays a = {sp.data(), z.data(), z.data(), z.data(), z.data(),
                               {z.data(), z.data(), z.data(), z.data()}};
    p->div_count = 0;
    return smx_pwm_load(p, &a);
}

extern "C" int smx_pwm_set_setpoint(smx_pwm *p, uint32_t chan, uint32_t val)
{
    if (!p) return SMX_E_ARG;
    if (chan >= p->n) { set_error("smx_pwm_set_setpoint: chan %u >= %u", chan, p->n); return SMX_E_RANGE; }
    SMX_HIP(hipSetDevice(p->device));
    const uint32_t v = pdm_safe_setpoint(val);
    SMX_HIP(hipMemcpyAsync(p->d.setpoint + chan, &v, 4, hipMemcpyHostToDevice, p->stream));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

static int pwm_ensure(smx_pwm *p, uint32_t n_ticks)
{
    const size_t need = (size_t)n_ticks * p->n_pad;
    if (need > p->duty_cap) {
        SMX_HIP(hipStreamSynchronize(p->stream));
        if (p->d_duty) SMX_HIP(hipFree(p->d_duty));
        p->d_duty = nullptr; p->duty_cap = 0;
        SMX_HIP(hipMalloc((void **)&p->d_duty, need));
        p->duty_cap = need;
    }
    if (n_ticks > p->dither_cap) {
        SMX_HIP(hipStreamSynchronize(p->stream));
        if (p->d_dither) SMX_HIP(hipFree(p->d_dither));
        p->d_dither = nullptr; p->dither_cap = 0;
        SMX_HIP(hipMalloc((void **)&p->d_dither, (size_t)n_ticks * 4));
        p->dither_cap = n_ticks;
    }
    return SMX_OK;
}

extern "C" void *smx_pwm_dither_dev(smx_pwm *p, uint32_t n_ticks)
{
    if (!p || hipSetDevice(p->device) != hipSuccess || pwm_ensure(p, n_ticks) != SMX_OK) return nullptr;
    return p->d_dither;
}

extern "C" int smx_pwm_tick_n_async(smx_pwm *p, uint32_t n_ticks, int with_dither)
{
    if (!p) return SMX_E_ARG;
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pwm_ensure(p, n_ticks);
    if (rv) return rv;
    rv = smx::launch_pwm_bank(p->d, p->order, with_dither ? p->d_dither : nullptr, p->d_duty, p->n_pad,
                              n_ticks, p->div_count, p->div_log, p->out_shift, p->stream);
    if (rv) return rv;
    p->div_count = (uint32_t)(((uint64_t)p->div_count + n_ticks) & ((1u << p->div_log) - 1));
    return SMX_OK;
}

extern "C" int smx_pwm_sync(smx_pwm *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

extern "C" int smx_pwm_tick_n(smx_pwm *p, uint32_t n_ticks, const uint32_t *dither, uint8_t *duty)
{
    if (!p) return SMX_E_ARG;
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(p->device));
    int rv = pwm_ensure(p, n_ticks);
    if (rv) return rv;
    if (dither)
        SMX_HIP(hipMemcpyAsync(p->d_dither, dither, (size_t)n_ticks * 4, hipMemcpyHostToDevice, p->stream));
    rv = smx_pwm_tick_n_async(p, n_ticks, dither != nullptr);
    if (rv) return rv;
    if (duty)
        SMX_HIP(hipMemcpy2DAsync(duty, p->n, p->d_duty, p->n_pad, p->n, n_ticks, hipMemcpyDeviceToHost,
                                 p->stream));
    SMX_HIP(hipStreamSynchronize(p->stream));
    return SMX_OK;
}

extern "C" int smx_pwm_timer_start(smx_pwm *p)
{
    if (!p) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipEventRecord(p->ev_t0, p->stream));
    return SMX_OK;
}

extern "C" int smx_pwm_timer_stop(smx_pwm *p, float *ms)
{
    if (!p || !ms) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->device));
    SMX_HIP(hipEventRecord(p->ev_t1, p->stream));
    SMX_HIP(hipEventSynchronize(p->ev_t1));
    SMX_HIP(hipEventElapsedTime(ms, p->ev_t0, p->ev_t1));
    return SMX_OK;
}

// ---------------------------------------------------------------------------
// oscillator bank: mod_pdm.c pwm_update + hard sync, mod_osc.c ISR, pmeas.h
// ---------------------------------------------------------------------------
struct smx_osc {
    uint32_t n = 0, n_pad = 0, log_max = 26;
    int device = 0;
    uint32_t *d_phase = nullptr, *d_speed = nullptr;
    smx::PmeasArrays pm{};
    void *d_tmp = nullptr; size_t tmp_cap = 0;      // sync/valid bits or timestamps
    void *d_tmp2 = nullptr; size_t tmp2_cap = 0;
    uint8_t *d_duty = nullptr; size_t duty_cap = 0;
    hipStream_t stream = nullptr;
};

static void pmeas_slots(smx::PmeasArrays &d, void **slots[9])
{
    slots[0] = (void **)&d.write; slots[1] = (void **)&d.avg0; slots[2] = (void **)&d.avg1;
    slots[3] = (void **)&d.num0;  slots[4] = (void **)&d.num1; slots[5] = (void **)&d.num;
    slots[6] = (void **)&d.accu;  slots[7] = (void **)&d.last_cc; slots[8] = (void **)&d.sub;
}

static int dev_reserve(void **ptr, size_t *cap, size_t need, hipStream_t stream)
{
    if (need <= *cap) return SMX_OK;
    SMX_HIP(hipStreamSynchronize(stream));
    if (*ptr) SMX_HIP(hipFree(*ptr));
    *ptr = nullptr; *cap = 0;
    SMX_HIP(hipMalloc(ptr, need));
    *cap = need;
    return SMX_OK;
}

extern "C" smx_osc *smx_osc_create(uint32_t n, int device)
{
    if (n == 0 || n > 0xFFFFF000u) { set_error("smx_osc_create: n=%u (1..2^32-4096)", n); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_osc_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_osc_create: device %d of %d", device, ndev); return nullptr; }
    smx_osc *o = new smx_osc();
    o->n = n;
    o->n_pad = smx::round_up(n, 1024);
    o->device = device;
    const size_t bytes = (size_t)o->n_pad * 4;
    void **slots[9];
    pmeas_slots(o->pm, slots);
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&o->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&o->d_phase, bytes) == hipSuccess &&
              hipMalloc((void **)&o->d_speed, bytes) == hipSuccess &&
              hipMemsetAsync(o->d_phase, 0, bytes, o->stream) == hipSuccess;
    for (int i = 0; ok && i < 9; i++)
        ok = hipMalloc(slots[i], bytes) == hipSuccess && hipMemsetAsync(*slots[i], 0, bytes, o->stream) == hipSuccess;
    if (ok) {
        std::vector<uint32_t> sp(o->n_pad, 256u * 13u);          // pwm_speed, mod_pdm.c:161
        ok = hipMemcpyAsync(o->d_speed, sp.data(), bytes, hipMemcpyHostToDevice, o->stream) == hipSuccess &&
             hipStreamSynchronize(o->stream) == hipSuccess;
    }
    if (!ok) {
        set_error("smx_osc_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_osc_destroy(o);
        return nullptr;
    }
    return o;
}

extern "C" void smx_osc_destroy(smx_osc *o)
{
    if (!o) return;
    (void)hipSetDevice(o->device);
    if (o->stream) (void)hipStreamSynchronize(o->stream);
    void **slots[9];
    pmeas_slots(o->pm, slots);
    for (int i = 0; i < 9; i++)
        if (*slots[i]) (void)hipFree(*slots[i]);
    if (o->d_phase) (void)hipFree(o->d_phase);
    if (o->d_speed) (void)hipFree(o->d_speed);
    if (o->d_tmp) (void)hipFree(o->d_tmp);
    if (o->d_tmp2) (void)hipFree(o->d_tmp2);
    if (o->d_duty) (void)hipFree(o->d_duty);
    if (o->stream) (void)hipStreamDestroy(o->stream);
    delete o;
}

extern "C" int smx_osc_set_log_max(smx_osc *o, uint32_t log_max)
{
    if (!o || log_max == 0 || log_max > 31) { set_error("smx_osc_set_log_max: %u (1..31)", log_max); return SMX_E_ARG; }
    o->log_max = log_max;
    return SMX_OK;
}

extern "C" int smx_osc_load_pwm(smx_osc *o, const uint32_t *phase, const uint32_t *speed)
{
    if (!o) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(o->device));
    SMX_HIP(hipStreamSynchronize(o->stream));
    if (phase) SMX_HIP(hipMemcpy(o->d_phase, phase, (size_t)o->n * 4, hipMemcpyHostToDevice));
    if (speed) SMX_HIP(hipMemcpy(o->d_speed, speed, (size_t)o->n * 4, hipMemcpyHostToDevice));
    return SMX_OK;
}

extern "C" int smx_osc_read_pwm(smx_osc *o, uint32_t *phase, uint32_t *speed)
{
    if (!o) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(o->device));
    SMX_HIP(hipStreamSynchronize(o->stream));
    if (phase) SMX_HIP(hipMemcpy(phase, o->d_phase, (size_t)o->n * 4, hipMemcpyDeviceToHost));
    if (speed) SMX_HIP(hipMemcpy(speed, o->d_speed, (size_t)o->n * 4, hipMemcpyDeviceToHost));
    return SMX_OK;
}

extern "C" int smx_osc_tick_n(smx_osc *o, uint32_t n_ticks, const uint32_t *sync_bits, uint8_t *duty)
{
    if (!o) return SMX_E_ARG;
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(o->device));
    int rv = dev_reserve((void **)&o->d_duty, &o->duty_cap, (size_t)n_ticks * o->n_pad, o->stream);
    if (rv) return rv;
    const uint32_t *d_sync = nullptr;
    if (sync_bits) {
        const size_t row = (size_t)o->n_pad / 8, words = (o->n + 31) / 32;
        rv = dev_reserve(&o->d_tmp, &o->tmp_cap, (size_t)n_ticks * row, o->stream);
        if (rv) return rv;
        SMX_HIP(hipMemsetAsync(o->d_tmp, 0, (size_t)n_ticks * row, o->stream));
        SMX_HIP(hipMemcpy2DAsync(o->d_tmp, row, sync_bits, words * 4, words * 4, n_ticks,
                                 hipMemcpyHostToDevice, o->stream));
        d_sync = (const uint32_t *)o->d_tmp;
    }
    rv = smx::launch_pwmosc(o->d_phase, o->d_speed, d_sync, o->d_duty, o->n_pad, n_ticks, o->stream);
    if (rv) return rv;
    if (duty)
        SMX_HIP(hipMemcpy2DAsync(duty, o->n, o->d_duty, o->n_pad, o->n, n_ticks, hipMemcpyDeviceToHost,
                                 o->stream));
    SMX_HIP(hipStreamSynchronize(o->stream));
    return SMX_OK;
}

extern "C" int smx_osc_events(smx_osc *o, uint32_t n_events, const uint32_t *cc, const uint32_t *valid_bits)
{
    if (!o || (n_events && !cc)) return SMX_E_ARG;
    if (n_events == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(o->device));
    int rv = dev_reserve(&o->d_tmp2, &o->tmp2_cap, (size_t)n_events * o->n * 4, o->stream);
    if (rv) return rv;
    SMX_HIP(hipMemcpyAsync(o->d_tmp2, cc, (size_t)n_events * o->n * 4, hipMemcpyHostToDevice, o->stream));
    const uint32_t *d_valid = nullptr;
    if (valid_bits) {
        const size_t bytes = (size_t)n_events * ((o->n + 31) / 32) * 4;
        rv = dev_reserve(&o->d_tmp, &o->tmp_cap, bytes, o->stream);
        if (rv) return rv;
        SMX_HIP(hipMemcpyAsync(o->d_tmp, valid_bits, bytes, hipMemcpyHostToDevice, o->stream));
        d_valid = (const uint32_t *)o->d_tmp;
    }
    rv = smx::launch_osc_events(o->pm, (const uint32_t *)o->d_tmp2, d_valid, o->n, n_events, o->log_max, o->stream);
    if (rv) return rv;
    SMX_HIP(hipStreamSynchronize(o->stream));
    return SMX_OK;
}

static int pmeas_copy(smx_osc *o, const struct smx_pmeas_arrays *a, bool to_device)
{
    if (!o || !a) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(o->device));
    SMX_HIP(hipStreamSynchronize(o->stream));
    void **slots[9];
    pmeas_slots(o->pm, slots);
    void *host[9] = {a->write, a->avg0, a->avg1, a->num0, a->num1, a->num, a->accu, a->last_cc, a->sub};
    for (int i = 0; i < 9; i++) {
        if (!host[i]) continue;
        if (to_device) SMX_HIP(hipMemcpy(*slots[i], host[i], (size_t)o->n * 4, hipMemcpyHostToDevice));
        else           SMX_HIP(hipMemcpy(host[i], *slots[i], (size_t)o->n * 4, hipMemcpyDeviceToHost));
    }
    return SMX_OK;
}

extern "C" int smx_osc_load_pmeas(smx_osc *o, const struct smx_pmeas_arrays *a) { return pmeas_copy(o, a, true); }
extern "C" int smx_osc_read_pmeas(smx_osc *o, const struct smx_pmeas_arrays *a) { return pmeas_copy(o, a, false); }

// ---------------------------------------------------------------------------
// firmware control surface, hosted: mod_synth.c:50-137, stm32f103/synth.c:27-42
// ---------------------------------------------------------------------------
struct smx_fw {
    smx_pwm *pwm = nullptr;
    smx_osc *osc = nullptr;
    int running = 0;
    uint32_t param[1] = {15u << 27};            // osc_setpoint, mod_synth.c:50-51
    std::vector<uint32_t> read_idx;             // pmeas_state.read per oscillator
    std::vector<uint8_t> wait;                  // measurement_wait: {len, bytes...}*, 64 bytes (mod_osc.c:43)
};

extern "C" smx_fw *smx_fw_create(uint32_t n_channels, uint32_t n_osc, int device)
{
    smx_fw *f = new smx_fw();
    f->pwm = smx_pwm_create(n_channels, 2, device);          // PDM_ORDER 2
    f->osc = n_osc ? smx_osc_create(n_osc, device) : nullptr;
    if (!f->pwm || (n_osc && !f->osc) || smx_pwm_init(f->pwm) != SMX_OK) {   // pdm_init
        smx_fw_destroy(f);
        return nullptr;
    }
    f->running = 1;                                          // pdm_start, mod_synth.c:67
    f->read_idx.assign(n_osc, 0);
    return f;
}

extern "C" void smx_fw_destroy(smx_fw *f)
{
    if (!f) return;
    smx_pwm_destroy(f->pwm);
    smx_osc_destroy(f->osc);
    delete f;
}

extern "C" smx_pwm *smx_fw_pwm(smx_fw *f) { return f ? f->pwm : nullptr; }
extern "C" smx_osc *smx_fw_osc(smx_fw *f) { return f ? f->osc : nullptr; }
extern "C" int smx_fw_running(const smx_fw *f) { return f ? f->running : 0; }
extern "C" uint32_t smx_fw_parameter(const smx_fw *f, uint32_t id) { return (f && id < 1) ? f->param[id] : 0; }

extern "C" int smx_fw_handle_tag_u32(smx_fw *f, const uint32_t *args, uint32_t nb_args,
                                     const uint8_t *bytes, uint32_t nb_bytes)
{
    if (!f) return -1;
    if (nb_args < 1) return -1;                              // mod_synth.c:91
    switch (args[0]) {
    case 100:                                                // MODE, :97-103
        if (nb_args < 2) return -1;
        f->running = args[1] ? 1 : 0;
        return 0;
    case 101:                                                // SETPOINT, :104-111
        if (nb_args < 3) return -1;
        return smx_pwm_set_setpoint(f->pwm, args[1], args[2]);   // -2 when chan >= PDM_NB_CHANNELS
    case 102:                                                // MEASURE, :112-127
        if (nb_args > 2) return -3;
        if (nb_args == 2 && f->osc) {
            // the firmware stores any value; the shift in pmeas_update is only defined for 1..31
            if (smx_osc_set_log_max(f->osc, args[1]) != SMX_OK) return -1;
        }
        if (nb_bytes) {
            // cbuf_put(len); cbuf_write(bytes): a 64-byte ring in the firmware
            if (nb_bytes > 255 || f->wait.size() + 1 + nb_bytes > 64) return 0;   // no room: dropped
            f->wait.push_back((uint8_t)nb_bytes);
            f->wait.insert(f->wait.end(), bytes, bytes + nb_bytes);
        }
        return 0;
    default:                                                 // parameter table, :129-135
        if (nb_args < 2 || args[0] >= 1) return -1;
        f->param[args[0]] = args[1];
        return 0;
    }
}

static inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

extern "C" int smx_fw_handle_packet(smx_fw *f, const uint8_t *buf, uint32_t len)
{
    if (!f || !buf || len < 2) return -1;
    const uint32_t tag = ((uint32_t)buf[0] << 8) | buf[1];
    if (tag != SMX_TAG_U32) {
        fprintf(stderr, "unknown tag 0x%x\n", tag);          // stm32f103/synth.c:39-40
        return 0;
    }
    if (len < 4) return -1;
    const uint32_t nb_from = buf[2], nb_args = buf[3];
    const uint64_t words = (uint64_t)nb_from + nb_args;
    if (4 + 4 * words > len) return -1;
    std::vector<uint32_t> args(nb_args);
    for (uint32_t i = 0; i < nb_args; i++) args[i] = be32(buf + 4 + 4 * (nb_from + i));
    const uint32_t off = 4 + 4 * (uint32_t)words;
    const int rv = smx_fw_handle_tag_u32(f, args.data(), nb_args, buf + off, len - off);
    if (rv) fprintf(stderr, "tag_u32_dispatch returned %d\n", rv);   // stm32f103/synth.c:36
    return rv;
}

extern "C" int smx_fw_tick_n(smx_fw *f, uint32_t n_ticks, const uint32_t *dither, uint8_t *duty)
{
    if (!f) return SMX_E_ARG;
    if (!f->running) return 0;
    const int rv = smx_pwm_tick_n(f->pwm, n_ticks, dither, duty);
    return rv ? rv : (int)n_ticks;
}

extern "C" int smx_fw_poll(smx_fw *f, uint32_t osc, uint32_t *avg, uint32_t *num,
                           uint8_t *cont, uint32_t cont_cap, uint32_t *cont_len)
{
    if (!f || !f->osc || osc >= f->read_idx.size()) return SMX_E_ARG;
    if (cont_len) *cont_len = 0;
    smx_osc *o = f->osc;
    SMX_HIP(hipSetDevice(o->device));
    SMX_HIP(hipStreamSynchronize(o->stream));
    uint32_t write = 0;
    SMX_HIP(hipMemcpy(&write, o->pm.write + osc, 4, hipMemcpyDeviceToHost));
    if (f->read_idx[osc] == write) return 0;                 // pmeas.h:32
    const uint32_t read = ++f->read_idx[osc];                // pmeas.h:35
    uint32_t a = 0, n = 0;
    SMX_HIP(hipMemcpy(&a, ((read & 1) ? o->pm.avg1 : o->pm.avg0) + osc, 4, hipMemcpyDeviceToHost));
    SMX_HIP(hipMemcpy(&n, ((read & 1) ? o->pm.num1 : o->pm.num0) + osc, 4, hipMemcpyDeviceToHost));
    if (avg) *avg = a;
    if (num) *num = n;
    if (!f->wait.empty()) {                                  // pmeas.h:44-58
        const uint32_t len = f->wait[0];
        if (len + 1 > f->wait.size()) {
            f->wait.clear();                                 // "bad measurement_wait size ... clearing"
        } else {
            if (cont && len <= cont_cap) memcpy(cont, f->wait.data() + 1, len);
            if (cont_len) *cont_len = len;
            f->wait.erase(f->wait.begin(), f->wait.begin() + 1 + len);
        }
    }
    return 1;
}

// ---------------------------------------------------------------------------
// cproc dataflow bank: generic/cproc.h, mod_bpmodular.c tick()
// ---------------------------------------------------------------------------
struct smx_cproc {
    uint32_t n = 0, n_pad = 0;
    int device = 0;
    smx::CprocProgram prog{};
    uint32_t *d_state = nullptr;
    void *d_in = nullptr; size_t in_cap = 0;
    void *d_g = nullptr; size_t g_cap = 0;
    void *d_out = nullptr; size_t out_cap = 0;
    hipStream_t stream = nullptr;
};

extern "C" smx_cproc *smx_cproc_create(uint32_t n_instances, const struct smx_cproc_node *nodes,
                                       uint32_t n_nodes, uint32_t n_inputs, int device)
{
    if (n_instances == 0 || n_instances > 0xFFFFF000u || !nodes || n_nodes == 0 || n_nodes > SMX_CPROC_MAX_NODES) {
        set_error("smx_cproc_create: n_instances=%u n_nodes=%u (1..%d)", n_instances, n_nodes, SMX_CPROC_MAX_NODES);
        return nullptr;
    }
    for (uint32_t k = 0; k < n_nodes; k++) {
        const uint32_t in = nodes[k].in;
        const bool ok_in = (in & 0x80000000u) ? (in & 0x7FFFFFFFu) < n_inputs : in < k;   // A-normal form
        if (!ok_in || (nodes[k].proc != SMX_PROC_ACC && nodes[k].proc != SMX_PROC_EDGE)) {
            set_error("smx_cproc_create: node %u: proc=%u in=0x%x", k, nodes[k].proc, in);
            return nullptr;
        }
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_cproc_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_cproc_create: device %d of %d", device, ndev); return nullptr; }
    smx_cproc *c = new smx_cproc();
    c->n = n_instances;
    c->n_pad = smx::round_up(n_instances, 256);
    c->device = device;
    c->prog.n_nodes = n_nodes;
    c->prog.n_inputs = n_inputs;
    for (uint32_t k = 0; k < n_nodes; k++) c->prog.nodes[k] = {nodes[k].proc, nodes[k].in, nodes[k].cond};
    const size_t bytes = (size_t)n_nodes * 2 * c->n_pad * 4;
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&c->d_state, bytes) == hipSuccess &&
              hipMemsetAsync(c->d_state, 0, bytes, c->stream) == hipSuccess &&
              hipStreamSynchronize(c->stream) == hipSuccess;
    if (!ok) {
        set_error("smx_cproc_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_cproc_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void smx_cproc_destroy(smx_cproc *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_g) (void)hipFree(c->d_g);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int smx_cproc_tick_n(smx_cproc *c, uint32_t n_ticks, const uint32_t *input, const uint32_t *g,
                                uint32_t out_node, uint32_t *out)
{
    if (!c || (c->prog.n_inputs && n_ticks && !input) || out_node >= c->prog.n_nodes) {
        set_error("smx_cproc_tick_n: bad args");
        return SMX_E_ARG;
    }
    if (n_ticks == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(c->device));
    const size_t row = (size_t)c->n_pad * 4, rows_in = (size_t)n_ticks * c->prog.n_inputs;
    int rv;
    if (rows_in) {
        if ((rv = dev_reserve(&c->d_in, &c->in_cap, rows_in * row, c->stream))) return rv;
        SMX_HIP(hipMemcpy2DAsync(c->d_in, row, input, (size_t)c->n * 4, (size_t)c->n * 4, rows_in,
                                 hipMemcpyHostToDevice, c->stream));
    }
    if (g) {
        if ((rv = dev_reserve(&c->d_g, &c->g_cap, (size_t)n_ticks * 4, c->stream))) return rv;
        SMX_HIP(hipMemcpyAsync(c->d_g, g, (size_t)n_ticks * 4, hipMemcpyHostToDevice, c->stream));
    }
    if (out && (rv = dev_reserve(&c->d_out, &c->out_cap, (size_t)n_ticks * row, c->stream))) return rv;
    rv = smx::launch_cproc(c->prog, c->d_state, (const uint32_t *)c->d_in, g ? (const uint32_t *)c->d_g : nullptr,
                           out ? (uint32_t *)c->d_out : nullptr, c->n_pad, n_ticks, out_node, c->stream);
    if (rv) return rv;
    if (out)
        SMX_HIP(hipMemcpy2DAsync(out, (size_t)c->n * 4, c->d_out, row, (size_t)c->n * 4, n_ticks,
                                 hipMemcpyDeviceToHost, c->stream));
    SMX_HIP(hipStreamSynchronize(c->stream));
    return SMX_OK;
}

static int cproc_state_copy(smx_cproc *c, uint32_t *host, bool to_device)
{
    if (!c || !host) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(c->device));
    SMX_HIP(hipStreamSynchronize(c->stream));
    const size_t rows = (size_t)c->prog.n_nodes * 2;
    if (to_device)
        SMX_HIP(hipMemcpy2D(c->d_state, (size_t)c->n_pad * 4, host, (size_t)c->n * 4, (size_t)c->n * 4, rows,
                            hipMemcpyHostToDevice));
    else
        SMX_HIP(hipMemcpy2D(host, (size_t)c->n * 4, c->d_state, (size_t)c->n_pad * 4, (size_t)c->n * 4, rows,
                            hipMemcpyDeviceToHost));
    return SMX_OK;
}
extern "C" int smx_cproc_read_state(smx_cproc *c, uint32_t *state) { return cproc_state_copy(c, state, false); }
extern "C" int smx_cproc_load_state(smx_cproc *c, const uint32_t *state) { return cproc_state_copy(c, (uint32_t *)state, true); }

// ---------------------------------------------------------------------------
// clock bank: linux/clock.c:58-62, 106-120
// ---------------------------------------------------------------------------
struct smx_clock {
    uint32_t n = 0, n_pad = 0;
    int device = 0;
    uint32_t *d_hperiod = nullptr, *d_phase = nullptr, *d_pol = nullptr;
    void *d_pbits = nullptr; size_t pbits_cap = 0;
    void *d_tbits = nullptr; size_t tbits_cap = 0;
    hipStream_t stream = nullptr;
};

extern "C" uint32_t smx_bpm_to_hperiod(uint32_t sr, uint32_t bpm) { return bpm ? (sr * 5) / (bpm * 4) : 0; }

extern "C" smx_clock *smx_clock_create(uint32_t n, int device)
{
    if (n == 0 || n > 0xFFFFF000u) { set_error("smx_clock_create: n=%u", n); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_clock_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_clock_create: device %d of %d", device, ndev); return nullptr; }
    smx_clock *c = new smx_clock();
    c->n = n;
    c->n_pad = smx::round_up(n, 1024);
    c->device = device;
    const size_t bytes = (size_t)c->n_pad * 4;
    std::vector<uint32_t> ones(c->n_pad, 1u);                      // clock_pol = 1, clock.c:62
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&c->d_hperiod, bytes) == hipSuccess &&
              hipMalloc((void **)&c->d_phase, bytes) == hipSuccess &&
              hipMalloc((void **)&c->d_pol, bytes) == hipSuccess &&
              hipMemset(c->d_hperiod, 0, bytes) == hipSuccess && hipMemset(c->d_phase, 0, bytes) == hipSuccess &&
              hipMemcpy(c->d_pol, ones.data(), bytes, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        set_error("smx_clock_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_clock_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void smx_clock_destroy(smx_clock *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_hperiod) (void)hipFree(c->d_hperiod);
    if (c->d_phase) (void)hipFree(c->d_phase);
    if (c->d_pol) (void)hipFree(c->d_pol);
    if (c->d_pbits) (void)hipFree(c->d_pbits);
    if (c->d_tbits) (void)hipFree(c->d_tbits);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int smx_clock_load(smx_clock *c, const uint32_t *hperiod, const int32_t *phase, const uint32_t *pol)
{
    if (!c) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(c->device));
    SMX_HIP(hipStreamSynchronize(c->stream));
    if (hperiod) SMX_HIP(hipMemcpy(c->d_hperiod, hperiod, (size_t)c->n * 4, hipMemcpyHostToDevice));
    if (phase) SMX_HIP(hipMemcpy(c->d_phase, phase, (size_t)c->n * 4, hipMemcpyHostToDevice));
    if (pol) SMX_HIP(hipMemcpy(c->d_pol, pol, (size_t)c->n * 4, hipMemcpyHostToDevice));
    return SMX_OK;
}

extern "C" int smx_clock_read(smx_clock *c, uint32_t *hperiod, int32_t *phase, uint32_t *pol)
{
    if (!c) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(c->device));
    SMX_HIP(hipStreamSynchronize(c->stream));
    if (hperiod) SMX_HIP(hipMemcpy(hperiod, c->d_hperiod, (size_t)c->n * 4, hipMemcpyDeviceToHost));
    if (phase) SMX_HIP(hipMemcpy(phase, c->d_phase, (size_t)c->n * 4, hipMemcpyDeviceToHost));
    if (pol) SMX_HIP(hipMemcpy(pol, c->d_pol, (size_t)c->n * 4, hipMemcpyDeviceToHost));
    return SMX_OK;
}

extern "C" int smx_clock_run(smx_clock *c, uint32_t n_frames, uint32_t *pol_bits, uint32_t *tick_bits)
{
    if (!c) return SMX_E_ARG;
    if (n_frames == 0) return SMX_OK;
    SMX_HIP(hipSetDevice(c->device));
    const size_t row = (size_t)c->n_pad / 8, words = (c->n + 31) / 32;
    int rv;
    if ((rv = dev_reserve(&c->d_pbits, &c->pbits_cap, (size_t)n_frames * row, c->stream))) return rv;
    if ((rv = dev_reserve(&c->d_tbits, &c->tbits_cap, (size_t)n_frames * row, c->stream))) return rv;
    rv = smx::launch_clock(c->d_hperiod, c->d_phase, c->d_pol, (uint32_t *)c->d_pbits, (uint32_t *)c->d_tbits,
                           c->n_pad, c->n, n_frames, c->stream);
    if (rv) return rv;
    if (pol_bits)
        SMX_HIP(hipMemcpy2DAsync(pol_bits, words * 4, c->d_pbits, row, words * 4, n_frames, hipMemcpyDeviceToHost, c->stream));
    if (tick_bits)
        SMX_HIP(hipMemcpy2DAsync(tick_bits, words * 4, c->d_tbits, row, words * 4, n_frames, hipMemcpyDeviceToHost, c->stream));
    SMX_HIP(hipStreamSynchronize(c->stream));
    return SMX_OK;
}
