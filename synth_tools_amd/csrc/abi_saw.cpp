// abi_saw.cpp -- part of the C-ABI of libsynth_mi355x.so (include/synth_mi355x.h): saw voice bank, note allocator over N voices, RCCL bus sum
// Host side of the drop-in boundary.  No CPU compute fallback exists: every compute entry
// point needs a HIP device and fails with SMX_E_NOGPU otherwise.
#include "abi_internal.h"
#include <rccl/rccl.h>
#include <time.h>
#include <unordered_map>

#define SMX_NCCL(expr)                                                         \
    do {                                                                       \
        ncclResult_t r_ = (expr);                                              \
        if (r_ != ncclSuccess) {                                               \
            set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                      ncclGetErrorString(r_));                                 \
            return SMX_E_COMM;                                                 \
        }                                                                      \
    } while (0)

namespace smx {

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
        if (all_free) {
            std::fill(levels_[0].begin(), levels_[0].end(), ~0ull);
            if (n & 63) levels_[0].back() = (1ull << (n & 63)) - 1ull;     // no voices beyond n
        }
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
    // the map's leaf words (bit v of word v/64 = voice v is free), for exchanging shards' maps between ranks
    const uint64_t *leaf_words() const { return levels_[0].data(); }
    size_t n_leaf_words() const { return levels_[0].size(); }
    void load_leaf_words(size_t first_word, const uint64_t *w, size_t nwords)
    {
        for (size_t i = 0; i < nwords; i++) levels_[0][first_word + i] = w[i];
        rebuild_summaries();
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

// ---------------------------------------------------------------------------
// saw bank
// ---------------------------------------------------------------------------
struct smx_bank {
    uint32_t n = 0, n_pad = 0;
    int device = 0;
    uint32_t *d_inc = nullptr;
    // Lazily materialised phases (saw_bank.hip): phase[v] = d_state0[v] + elapsed * d_inc[v].
    // Blocks only read; the host adds the block length to `elapsed`.
    uint32_t *d_state0 = nullptr;
    bool one_alloc = false;                      // d_state0 lies in d_inc's allocation (the default)
    uint32_t elapsed = 0;
    // A ring of bus buffers: [cur] holds the last block (and may be feeding an all-reduce),
    // [cur+1] was zeroed by the last launch for the next one, [cur+2] is the one the next
    // launch will zero; it was used NBUS-1 blocks ago.  The ring is deep so that the compute
    // stream has to wait for the comm stream only once per NBUS/2 blocks (see
    // bank_bus_release) instead of once per block.
    // The ring is ONE allocation with a tight stride (the longest block seen, rounded up to 64
    // frames): consecutive blocks lie side by side, so the sums of a whole group of blocks are
    // ONE all-reduce over a contiguous range instead of one collective per block (xGMI is
    // latency-bound at these sizes; the frames between a block's end and the stride are
    // summed along and ignored).
    static constexpr int NBUS = 32;
    int32_t *d_ring = nullptr;
    int32_t *d_bus[NBUS] = {};                   // d_ring + i * bus_cap
    uint32_t bus_zeroed[NBUS] = {};              // leading frames known to be zero
    int bus_cur = 0;
    uint32_t bus_cap = 0;                        // = stride of the ring, in frames
    uint32_t ring_alloc = 0;                     // frames per slot the ring's memory (and h_bus) is sized for (>= bus_cap)
    static constexpr uint32_t RING_MIN_FRAMES = 4096;
    uint32_t scratch_cap = 0;                    // frames d_scratch is sized for
    int32_t *h_bus = nullptr;                    // pinned
    // The synchronous block's way back to the host (round 3): coherent pinned words that the GPU writes itself --
    // the block's last (small) kernel publishes the bus and then a sequence number, the host polls that number
    // instead of queueing a copy and waiting for the stream (saw_publish_kernel / saw_dropin_kernel, saw_bank.hip).
    static constexpr uint32_t PUB_MAX = 4096;    // frames: longer blocks take the copy
    int32_t *h_pub = nullptr, *d_pub = nullptr;  // host / device address of the same PUB_MAX words
    uint32_t *h_pubflag = nullptr, *d_pubflag = nullptr;
    uint32_t pub_seq = 0;
    // pipelined block mode (smx_bank_set_block_mode): the bus of block k is copied to pinned
    // memory behind its kernel and handed out by the call that launches block k+1
    int block_mode = 0;
    int block_form = SMX_FORM_AUTO;              // smx_bank_set_block_form
    // AUTO: the device picks the form per launch and both forms are queued (the loser returns at
    // once, ~3 us).  The finalize kernel also writes {its pick, the number of long blocks finalized}
    // to these two pinned words; once the host has seen the same pick from FORM_STABLE consecutive
    // NEW long blocks, with no note event since, it queues only that form (both are exact on any
    // bank, so a stale pick would cost time, never bits -- and a note event un-pins at once).
    uint32_t *h_form = nullptr;
    uint32_t form_seen = 0xFFFFFFFFu, form_seq = 0;
    int form_stable = 0;
    // Every long block carries the host's own number for it (long_tag) and the device echoes that number with
    // its pick.  A pick counts only when its block was LAUNCHED after the last note event / reload
    // (tag > form_min_tag): finalizations that were still in flight when the increments changed describe the
    // old bank and must not pin a form for the new one.
    uint32_t long_tag = 0, form_min_tag = 0;
    static constexpr int FORM_STABLE = 4;
    int32_t *h_pipe[2] = {nullptr, nullptr};
    uint32_t pipe_cap = 0;
    hipEvent_t ev_pipe[2] = {nullptr, nullptr};
    int pipe_n[2] = {0, 0};                      // frames held by each slot (0: nothing yet)
    uint32_t pipe_k = 0;
    void *d_scratch = nullptr;                   // partial sums of saw_bank.hip's carry formulation
    smx::SawPending pend;                        // the slot fold the last launch left to its successor (smx_common.h)
    hipStream_t stream = nullptr, comm_stream = nullptr;
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
    hipEvent_t ev_kernel[NBUS] = {};             // kernel of bus[i] finished
    hipEvent_t ev_comm[NBUS] = {};               // all-reduce of bus[i] finished
    bool comm_pending[NBUS] = {};                // an all-reduce was issued on bus[i] and not waited for
    int ev_owner[NBUS] = {};                     // ev_comm[ev_owner[i]] completes after bus[i]'s all-reduce
    // All-reduces are requested per block but issued in groups (fewer, larger collectives): a
    // request is queued and the queue is flushed as ONE all-reduce over the contiguous slots when it holds comm_group
    // blocks, or as soon as somebody needs a result (fetch, sync, buffer reuse).
    int ar_queue[NBUS] = {};                     // bus indices with a requested, not yet issued sum
    int ar_frames[NBUS] = {};
    int ar_count = 0;
    int comm_group = 8;                          // blocks per collective (smx_bank_set_comm_group, 1 .. NBUS/2)
    int last_comm_ev = -1;                       // ev_comm[] index recorded behind the youngest collective on the comm stream
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    int comm_count = 0;                          // ncclCommCount: ranks the communicator really spans
    unsigned long long ar_launches = 0, ar_blocks = 0;   // collectives issued / block sums they carried
    int note2voice[128];
    // Sharded bank (smx_bank_shard): this bank holds voices [shard_first, shard_first + n) of a global bank of
    // shard_total voices.  Every rank runs the reference's allocator (linux/synth.c:145-165) over the WHOLE global
    // bank -- the same MIDI events in the same order give the same decisions everywhere -- and applies to its
    // device arrays only what falls into its own range: note2voice[] and free_map are then in global voice numbers.
    uint32_t shard_first = 0, shard_total = 0;   // shard_total == 0: not sharded (free_map covers n voices)
    smx::FreeMap free_map;
    // smx_bank_midi_events: (voice, increment) pairs of one batch, staged in pinned memory (two
    // slots, so the host can fill one while the copy of the other is in flight) and a device copy
    uint32_t *h_ev[2] = {nullptr, nullptr};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    bool ev_stage_busy[2] = {false, false};
    uint32_t ev_cap = 0;                         // pairs per slot
    uint32_t *d_ev = nullptr;
    int ev_k = 0;
    std::unordered_map<uint32_t, uint32_t> ev_net;   // voice -> index of its pair in the batch
};

static int bank_comm_flush(smx_bank *b);

// The increments changed (load / note event): what the host has seen of the device's picks is void.
static void bank_form_unpin(smx_bank *b)
{
    b->form_seen = 0xFFFFFFFFu;
    b->form_stable = 0;
    b->form_min_tag = b->long_tag;           // picks of the blocks launched so far are about the old increments
}

// Does the (global) voice number belong to this bank?  -> its local index
static inline bool bank_owns(const smx_bank *b, uint32_t v, uint32_t *local)
{
    if (!b->shard_total) { *local = v; return true; }
    if (v < b->shard_first || v - b->shard_first >= b->n) return false;
    *local = v - b->shard_first;
    return true;
}

// Whoever is about to read the current bus buffer (or to write the next one outside a saw launch) first runs the
// fold the last slot launch deferred; a no-op when nothing is owed.
static int bank_flush_fold(smx_bank *b) { return smx::launch_saw_flush(&b->pend, b->stream); }
// ... and may let that fold hand the bus to the host itself (a block of <= 64 frames ends in one workgroup of it)
static int bank_flush_fold_pub(smx_bank *b, const smx::SawPublish *pub, bool *published)
{
    return smx::launch_saw_flush(&b->pend, b->stream, pub, published);
}

static int bank_ensure_bus(smx_bank *b, uint32_t n)
{
    if (n <= b->bus_cap) return SMX_OK;
    {
        int rv = bank_flush_fold(b);                 // it refers to the buffers being replaced
        if (rv) return rv;
    }
    if (b->comm) {                                   // queued sums refer to the buffers being replaced
        int rv = bank_comm_flush(b);
        if (rv) return rv;
    }
    SMX_HIP(hipStreamSynchronize(b->stream));
    if (b->comm_stream) SMX_HIP(hipStreamSynchronize(b->comm_stream));
    const uint32_t cap = smx::round_up(n, 64);
    // The ring's MEMORY is sized for the usual JACK block lengths from the start (RING_MIN_FRAMES per slot), so a
    // first callback that is longer than anything seen before only re-strides it: no hipFree / hipMalloc /
    // hipHostMalloc on the real-time path.  The STRIDE stays tight (the longest block seen, rounded
    // up to 64 frames) so that a group of blocks is one short contiguous all-reduce.
    const uint32_t alloc = cap < smx_bank::RING_MIN_FRAMES ? smx_bank::RING_MIN_FRAMES : smx::round_up(cap, 4096);
    if (alloc > b->ring_alloc) {
        if (b->d_ring) SMX_HIP(hipFree(b->d_ring));
        b->d_ring = nullptr;
        SMX_HIP(hipMalloc((void **)&b->d_ring, (size_t)smx_bank::NBUS * alloc * 4));
        if (b->h_bus) SMX_HIP(hipHostFree(b->h_bus));
        b->h_bus = nullptr;
        SMX_HIP(hipHostMalloc((void **)&b->h_bus, (size_t)alloc * 4, hipHostMallocDefault));
        b->ring_alloc = alloc;
    }
    // (stream-ordered: hipMemset on the null stream is asynchronous to the host and b->stream, a non-blocking
    // stream, would not wait for it -- the first block's kernel could meet a buffer that is cleared under it)
    SMX_HIP(hipMemsetAsync(b->d_ring, 0, (size_t)smx_bank::NBUS * cap * 4, b->stream));
    for (int i = 0; i < smx_bank::NBUS; i++) {
        b->d_bus[i] = b->d_ring + (size_t)i * cap;
        b->bus_zeroed[i] = cap;
        b->comm_pending[i] = false;
    }
    // partial-sum slots of the slot / carry formulations: sized for 4096 frames at least, so
    // that the usual block lengths never reallocate them
    const uint32_t scap = smx::round_up(n < 4096 ? 4096 : n, 4096);
    if (b->n_pad >= (1u << 16) && scap > b->scratch_cap) {
        if (b->d_scratch) SMX_HIP(hipFree(b->d_scratch));
        b->d_scratch = nullptr;
        SMX_HIP(hipMalloc(&b->d_scratch, smx::saw_scratch_bytes(scap)));
        SMX_HIP(hipMemsetAsync(b->d_scratch, 0, smx::saw_scratch_bytes(scap), b->stream));   // slots are kept zero between launches
        b->scratch_cap = scap;
        static const bool no_defer = getenv("SMX_SAW_NO_DEFER") != nullptr;      // A/B switch: every launch folds its own slots
        b->pend = smx::SawPending{};
        b->pend.region_stride = no_defer ? 0 : smx::saw_scratch_region_bytes(scap);
        // a new header: the sum of the increments and the form pick are computed again
        int rv = smx::launch_saw_sum_inc(b->d_inc, b->n_pad, b->d_scratch, b->stream);
        if (rv) return rv;
        bank_form_unpin(b);
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
    // padding voices are off forever; big banks are padded to whole 4096-voice rows so that ragged sizes take
    // the 1024-thread tick kernel too (saw_bank.hip)
    b->n_pad = smx::round_up(n_voices, n_voices >= (1u << 20) ? 4096u : 1024u);
    b->device = device;
    auto fail = [&](const char *what, hipError_t e) -> smx_bank * {
        set_error("smx_bank_create: %s: %s", what, hipGetErrorString(e));
        smx_bank_destroy(b);
        return nullptr;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
    const size_t bytes = (size_t)b->n_pad * 4;
    {
        // Both arrays in ONE allocation, state0[] right behind inc[] (round 3).  The HBM-bound tick step is bimodal by
        // PROCESS with two allocations -- 76 us or 82 us on 64 Mi voices, depending on where the second one lands
        // relative to the first (round 2: "1 bank in 8 at 81 us"; round 3: three processes in a row at 82.4 us on one
        // box, then 76.2 in the next) -- and was 76.2-76.8 us in ten processes out of ten with one allocation, against
        // 76.2-77.7 with two on the same box (profiles/r03_one_alloc.txt).  SMX_BANK_TWO_ALLOCS=1: as before.
        static const bool two = getenv("SMX_BANK_TWO_ALLOCS") != nullptr;
        if (!two) {
            if ((e = hipMalloc((void **)&b->d_inc, 2 * bytes)) != hipSuccess) return fail("hipMalloc inc+state", e);
            b->d_state0 = b->d_inc + b->n_pad;
            b->one_alloc = true;
        } else {
            if ((e = hipMalloc((void **)&b->d_inc, bytes)) != hipSuccess) return fail("hipMalloc inc", e);
            if ((e = hipMalloc((void **)&b->d_state0, bytes)) != hipSuccess) return fail("hipMalloc state", e);
        }
    }
    if ((e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)) != hipSuccess) return fail("stream", e);
    if ((e = hipHostMalloc((void **)&b->h_form, 64, hipHostMallocDefault)) != hipSuccess) return fail("hipHostMalloc", e);
    b->h_form[0] = 0xFFFFFFFFu;
    b->h_form[1] = 0;
    if ((e = hipHostMalloc((void **)&b->h_pub, smx_bank::PUB_MAX * 4, hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess) return fail("hipHostMalloc (coherent)", e);
    if ((e = hipHostMalloc((void **)&b->h_pubflag, 64, hipHostMallocCoherent | hipHostMallocMapped)) != hipSuccess) return fail("hipHostMalloc (coherent)", e);
    if ((e = hipHostGetDevicePointer((void **)&b->d_pub, b->h_pub, 0)) != hipSuccess) return fail("hipHostGetDevicePointer", e);
    if ((e = hipHostGetDevicePointer((void **)&b->d_pubflag, b->h_pubflag, 0)) != hipSuccess) return fail("hipHostGetDevicePointer", e);
    b->h_pubflag[0] = 0;
    if ((e = hipEventCreate(&b->ev_t0)) != hipSuccess) return fail("event", e);
    if ((e = hipEventCreate(&b->ev_t1)) != hipSuccess) return fail("event", e);
    for (int i = 0; i < smx_bank::NBUS; i++) {
        if ((e = hipEventCreateWithFlags(&b->ev_kernel[i], hipEventDisableTiming)) != hipSuccess) return fail("event", e);
        if ((e = hipEventCreateWithFlags(&b->ev_comm[i], hipEventDisableTiming)) != hipSuccess) return fail("event", e);
    }
    // synth_init: bzero (linux/synth.c:204-206); padding voices stay off forever
    if ((e = hipMemsetAsync(b->d_inc, 0, bytes, b->stream)) != hipSuccess) return fail("memset", e);
    if ((e = hipMemsetAsync(b->d_state0, 0, bytes, b->stream)) != hipSuccess) return fail("memset", e);
    if ((e = hipStreamSynchronize(b->stream)) != hipSuccess) return fail("sync", e);
    memset(b->note2voice, 0, sizeof(b->note2voice));
    b->free_map.reset(b->n, true);
    if (bank_ensure_bus(b, 64) != SMX_OK) { smx_bank_destroy(b); return nullptr; }
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
    if (b->d_state0 && !b->one_alloc) (void)hipFree(b->d_state0);
    if (b->d_ring) (void)hipFree(b->d_ring);
    for (int i = 0; i < smx_bank::NBUS; i++) {
        if (b->ev_kernel[i]) (void)hipEventDestroy(b->ev_kernel[i]);
        if (b->ev_comm[i]) (void)hipEventDestroy(b->ev_comm[i]);
    }
    if (b->h_bus) (void)hipHostFree(b->h_bus);
    if (b->h_form) (void)hipHostFree(b->h_form);
    if (b->h_pub) (void)hipHostFree(b->h_pub);
    if (b->h_pubflag) (void)hipHostFree(b->h_pubflag);
    for (int i = 0; i < 2; i++) {
        if (b->h_pipe[i]) (void)hipHostFree(b->h_pipe[i]);
        if (b->ev_pipe[i]) (void)hipEventDestroy(b->ev_pipe[i]);
    }
    if (b->d_scratch) (void)hipFree(b->d_scratch);
    for (int i = 0; i < 2; i++) {
        if (b->h_ev[i]) (void)hipHostFree(b->h_ev[i]);
        if (b->ev_stage[i]) (void)hipEventDestroy(b->ev_stage[i]);
    }
    if (b->d_ev) (void)hipFree(b->d_ev);
    if (b->ev_t0) (void)hipEventDestroy(b->ev_t0);
    if (b->ev_t1) (void)hipEventDestroy(b->ev_t1);
    if (b->comm_stream) (void)hipStreamDestroy(b->comm_stream);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

extern "C" uint32_t smx_bank_voices(const smx_bank *b) { return b ? b->n : 0; }

// state0 += elapsed * inc for every voice, elapsed = 0: the stored phases are current again.
static int bank_materialize(smx_bank *b)
{
    if (b->elapsed) {
        int rv = smx::launch_saw_materialize(b->d_inc, b->d_state0, b->n_pad, b->elapsed, b->stream);
        if (rv) return rv;
        b->elapsed = 0;
    }
    return SMX_OK;
}

// A sharded bank got new increments: the ranks exchange which of their voices are free, so that every rank's
// copy of the global allocator sees the whole bank (ncclAllGather of the shards' free-bit words; n is a multiple
// of 64 on a sharded bank, so a shard is whole words).
static int bank_exchange_free_maps(smx_bank *b, const uint32_t *inc)
{
    const size_t w = b->n / 64;                               // words per shard
    std::vector<uint64_t> mine(w, 0ull);
    for (uint32_t v = 0; v < b->n; v++)
        if (inc[v] == 0) mine[v >> 6] |= 1ull << (v & 63);
    const int nr = b->comm ? b->nranks : 1;
    if ((size_t)nr * w != b->shard_total / 64 || b->shard_first != (uint32_t)(b->comm ? b->rank : 0) * b->n) {
        set_error("smx_bank_load: a sharded bank needs its communicator, equal shards (total = ranks x voices) and shard r at r x voices");
        return SMX_E_STATE;
    }
    std::vector<uint64_t> all((size_t)nr * w);
    if (nr == 1) {
        all = mine;
    } else {
        // one staging allocation [send | recv], released on every way out (a failed collective load must not
        // leave device memory behind on each retry); copies and the gather are ordered on the comm stream
        struct Staging {
            uint64_t *d = nullptr;
            ~Staging() { if (d) (void)hipFree(d); }
        } st;
        SMX_HIP(hipMalloc((void **)&st.d, ((size_t)nr + 1) * w * 8));
        uint64_t *d_send = st.d, *d_recv = st.d + w;
        SMX_HIP(hipMemcpyAsync(d_send, mine.data(), w * 8, hipMemcpyHostToDevice, b->comm_stream));
        SMX_NCCL(ncclAllGather(d_send, d_recv, w * 8, ncclUint8, b->comm, b->comm_stream));
        SMX_HIP(hipMemcpyAsync(all.data(), d_recv, (size_t)nr * w * 8, hipMemcpyDeviceToHost, b->comm_stream));
        SMX_HIP(hipStreamSynchronize(b->comm_stream));
    }
    b->free_map.load_leaf_words(0, all.data(), all.size());
    return SMX_OK;
}

// Declare this bank the shard [first_voice, first_voice + n) of a global bank of total_voices voices.
extern "C" int smx_bank_shard(smx_bank *b, uint32_t first_voice, uint32_t total_voices)
{
    if (!b || b->n == 0 || (b->n & 63) || total_voices < b->n || first_voice > total_voices - b->n || (first_voice % b->n) ||
        (total_voices % b->n)) {
        set_error("smx_bank_shard: first=%u total=%u for a bank of %u voices (equal shards, a multiple of 64 voices each)",
                  first_voice, total_voices, b ? b->n : 0);
        return SMX_E_ARG;
    }
    SMX_HIP(hipSetDevice(b->device));
    SMX_HIP(hipStreamSynchronize(b->stream));
    b->shard_first = first_voice;
    b->shard_total = total_voices;
    memset(b->note2voice, 0, sizeof(b->note2voice));
    // the allocator starts from what this shard holds now (a fresh bank: every voice free) and "unknown = free"
    // for the others; a later smx_bank_load(inc) exchanges the real maps
    std::vector<uint32_t> inc(b->n);
    SMX_HIP(hipMemcpy(inc.data(), b->d_inc, (size_t)b->n * 4, hipMemcpyDeviceToHost));
    b->free_map.reset(total_voices, true);
    for (uint32_t v = 0; v < b->n; v++)
        if (inc[v]) b->free_map.set_free(first_voice + v, false);
    return SMX_OK;
}

extern "C" int smx_bank_load(smx_bank *b, const uint32_t *inc, const uint32_t *state)
{
    if (!b) { set_error("smx_bank_load: null bank"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    int rv = bank_materialize(b);            // an array that is not replaced keeps its meaning
    if (rv) return rv;
    SMX_HIP(hipStreamSynchronize(b->stream));
    if (inc) {
        SMX_HIP(hipMemcpy(b->d_inc, inc, (size_t)b->n * 4, hipMemcpyHostToDevice));
        if (b->shard_total) {
            rv = bank_exchange_free_maps(b, inc);            // collective: every rank loads its shard
            if (rv) return rv;
        } else {
            b->free_map.load(inc, b->n);
        }
        // new increments: the statistic that picks the long-block form is computed again from them (sum and maximum
        // of the whole bank: launch_saw_sum_inc), what the host has seen of the old bank's picks is void
        if (b->d_scratch) {
            SMX_HIP(hipMemsetAsync(b->d_scratch, 0, smx::saw_scratch_header_bytes(), b->stream));
            rv = smx::launch_saw_sum_inc(b->d_inc, b->n_pad, b->d_scratch, b->stream);    // the header's sum of increments
            if (rv) return rv;
        }
        bank_form_unpin(b);
    }
    if (state)
        SMX_HIP(hipMemcpy(b->d_state0, state, (size_t)b->n * 4, hipMemcpyHostToDevice));
    return SMX_OK;
}

// load + run + fetch with ONE host synchronisation (the drop-in synth_run uses it per block)
extern "C" int smx_bank_load_run(smx_bank *b, const uint32_t *inc, const uint32_t *state, float *vec,
                                 int32_t *bus, int n)
{
    if (!b || !inc || !state || n <= 0) { set_error("smx_bank_load_run: bad args"); return SMX_E_ARG; }
    if (b->shard_total) { set_error("smx_bank_load_run: a sharded bank loads collectively (smx_bank_load)"); return SMX_E_STATE; }
    SMX_HIP(hipSetDevice(b->device));
    {
        int rv = bank_flush_fold(b);         // an owed fold belongs to the block before the new arrays
        if (rv) return rv;
    }
    // pageable sources: hipMemcpyAsync stages them before returning, so the caller's arrays may
    // change right after the call; both copies and the kernel are ordered on the bank's stream
    SMX_HIP(hipMemcpyAsync(b->d_inc, inc, (size_t)b->n * 4, hipMemcpyHostToDevice, b->stream));
    SMX_HIP(hipMemcpyAsync(b->d_state0, state, (size_t)b->n * 4, hipMemcpyHostToDevice, b->stream));
    b->elapsed = 0;
    b->free_map.load(inc, b->n);
    if (b->d_scratch) {
        // as in smx_bank_load: the long-block forms take the bank's sum of increments from the scratch header
        SMX_HIP(hipMemsetAsync(b->d_scratch, 0, smx::saw_scratch_header_bytes(), b->stream));
        int rv = smx::launch_saw_sum_inc(b->d_inc, b->n_pad, b->d_scratch, b->stream);
        if (rv) return rv;
    }
    bank_form_unpin(b);
    return smx_bank_run(b, vec, bus, n);
}

extern "C" int smx_bank_read(smx_bank *b, uint32_t *inc, uint32_t *state)
{
    if (!b) { set_error("smx_bank_read: null bank"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    if (state) {
        int rv = bank_materialize(b);
        if (rv) return rv;
    }
    SMX_HIP(hipStreamSynchronize(b->stream));
    if (inc) SMX_HIP(hipMemcpy(inc, b->d_inc, (size_t)b->n * 4, hipMemcpyDeviceToHost));
    if (state) SMX_HIP(hipMemcpy(state, b->d_state0, (size_t)b->n * 4, hipMemcpyDeviceToHost));
    return SMX_OK;
}

// One voice's increment changes: a one-lane kernel on the bank's stream, ordered before the
// next block, that also rebases the voice's stored phase so that state0 + elapsed*inc stays
// continuous ("note_on does not reset the phase", linux/synth.c:156-160).  No host sync per
// event: the JACK thread can apply a burst of MIDI events and launch the block behind them.
// (v is the allocator's voice number: global on a sharded bank, where only the owner touches the device.)
static int bank_set_inc(smx_bank *b, uint32_t v, uint32_t inc)
{
    b->free_map.set_free(v, inc == 0);
    uint32_t local;
    if (!bank_owns(b, v, &local)) return SMX_OK;
    SMX_HIP(hipSetDevice(b->device));
    int rv = smx::launch_saw_rebase(b->d_inc, b->d_state0, local, inc, b->elapsed, b->d_scratch, b->n_pad, b->stream);
    if (rv) return rv;
    bank_form_unpin(b);
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
// Make the compute stream wait until no all-reduce still uses bus buffer i.  The comm stream
// runs its all-reduces in order, so waiting for a YOUNGER one covers buffer i too: the wait
// is taken on the youngest all-reduce that is at least NBUS/2 blocks old, which retires half
// the ring at once -- one cross-stream barrier per NBUS/2 blocks in steady state.

// Issue every queued bus sum on the comm stream, ordered after all kernels enqueued so far.
// The queue holds consecutive ring slots (smx_bank_allreduce_async flushes before a gap or the
// ring's wrap), so the group is ONE all-reduce over [first slot, last slot + its frames): the
// frames between a block's end and the stride hold zeros or old sums, are summed along and
// never read.  (A non-consecutive queue cannot arise; it would fall back to one grouped launch
// of per-block calls.)  Integer sums: associative, the same bits in any order.
static int bank_comm_flush(smx_bank *b)
{
    if (b->ar_count == 0) return SMX_OK;
    {
        int rv = bank_flush_fold(b);                 // the youngest queued block may still owe its fold
        if (rv) return rv;
    }
    const int first = b->ar_queue[0], last = b->ar_queue[b->ar_count - 1];
    SMX_HIP(hipEventRecord(b->ev_kernel[last], b->stream));
    SMX_HIP(hipStreamWaitEvent(b->comm_stream, b->ev_kernel[last], 0));
    bool run = true;
    size_t frames = (size_t)b->ar_frames[0];
    for (int k = 1; k < b->ar_count; k++) {
        run = run && b->ar_queue[k] == b->ar_queue[k - 1] + 1;
        frames += (size_t)b->ar_frames[k];
    }
    const size_t count = (size_t)(last - first) * b->bus_cap + (size_t)b->ar_frames[b->ar_count - 1];
    // The stride only grows (the longest block ever seen): after one 4096-frame block a group of eight 1-frame
    // blocks would be 7 x 4096 + 1 words on xGMI instead of 8.  A range that is mostly gap goes as
    // one grouped launch of per-block sums instead (still one collective launch, a few bytes each).
    if (count > 4 * frames && count > 1024) run = false;
    if (run) {
        SMX_NCCL(ncclAllReduce(b->d_bus[first], b->d_bus[first], count, ncclInt32, ncclSum, b->comm, b->comm_stream));
    } else {
        SMX_NCCL(ncclGroupStart());
        for (int k = 0; k < b->ar_count; k++) {
            const int i = b->ar_queue[k];
            SMX_NCCL(ncclAllReduce(b->d_bus[i], b->d_bus[i], (size_t)b->ar_frames[k], ncclInt32, ncclSum, b->comm,
                                   b->comm_stream));
        }
        SMX_NCCL(ncclGroupEnd());
    }
    SMX_HIP(hipEventRecord(b->ev_comm[last], b->comm_stream));
    b->last_comm_ev = last;
    for (int k = 0; k < b->ar_count; k++) {
        b->comm_pending[b->ar_queue[k]] = true;
        b->ev_owner[b->ar_queue[k]] = last;
    }
    b->ar_launches++;
    b->ar_blocks += (unsigned long long)b->ar_count;
    b->ar_count = 0;
    return SMX_OK;
}

static bool bank_ar_queued(const smx_bank *b, int i)
{
    for (int k = 0; k < b->ar_count; k++)
        if (b->ar_queue[k] == i) return true;
    return false;
}

// Make the compute stream wait until no all-reduce still uses bus buffer i.  The comm stream
// runs its all-reduces in order, so waiting for a YOUNGER one covers buffer i too: the wait
// is taken on the youngest all-reduce that is at least NBUS/2 blocks old, which retires half
// the ring at once -- one cross-stream barrier per NBUS/2 blocks in steady state.
static int bank_bus_release(smx_bank *b, int i)
{
    if (bank_ar_queued(b, i)) {
        int rv = bank_comm_flush(b);
        if (rv) return rv;
    }
    if (!b->comm_pending[i]) return SMX_OK;
    int j = i;
    for (int k = smx_bank::NBUS / 2 - 1; k > 0; k--) {
        const int c = (i + k) % smx_bank::NBUS;
        if (b->comm_pending[c]) { j = c; break; }
    }
    SMX_HIP(hipStreamWaitEvent(b->stream, b->ev_comm[b->ev_owner[j]], 0));
    for (int c = i;; c = (c + 1) % smx_bank::NBUS) {     // everything from i up to j is now safe
        b->comm_pending[c] = false;
        if (c == j) break;
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

static int bank_run_async(smx_bank *b, int n, const smx::SawPublish *pub, bool *published);
extern "C" int smx_bank_run_async(smx_bank *b, int n) { return bank_run_async(b, n, nullptr, nullptr); }

// pub: the caller will fetch this block at once (the synchronous smx_bank_run): if the block's last kernel is a
// one-workgroup finalize, it publishes the bus itself (*published) and no publish kernel follows.
static int bank_run_async(smx_bank *b, int n, const smx::SawPublish *pub, bool *published)
{
    if (!b || n <= 0) { set_error("smx_bank_run_async: bad args"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    int rv = bank_ensure_bus(b, (uint32_t)n);
    if (rv) return rv;
    int bi, bnext;
    rv = bank_bus_advance(b, (uint32_t)n, &bi, &bnext);
    if (rv) return rv;
    int form = b->block_form;
    // "long" blocks: the ones that may take the carry formulations (saw_bank.hip: more than 32 frames, and 17..32
    // frames as one 32-frame chunk unless the stepping form is asked for)
    if (n > 16) b->long_tag++;                                           // this block's number (echoed with its pick)
    if (form == SMX_FORM_AUTO && n > 16 && b->h_form) {
        const volatile uint32_t *hf = b->h_form;                         // whatever has landed: no sync
        const uint32_t seq = hf[1], seen = hf[0];
        if (seq != b->form_seq) {                                        // a long block was finalized since the last look
            b->form_seq = seq;
            if ((int32_t)(seq - b->form_min_tag) > 0) {                  // ... and it was launched after the last note event
                if (seen == b->form_seen) b->form_stable++; else { b->form_seen = seen; b->form_stable = 0; }
            }
        }
        if (b->form_stable >= smx_bank::FORM_STABLE && b->form_seen <= 1u)
            form = b->form_seen ? SMX_FORM_EVENTS : SMX_FORM_STEPPING;
    }
    rv = smx::launch_saw_bank(b->d_inc, b->d_state0, b->d_bus[bi], b->d_bus[bnext], b->n_pad, (uint32_t)n,
                              b->elapsed, b->d_scratch, form, b->h_form, b->long_tag, b->stream, &b->pend, pub, published);
    if (rv) return rv;
    b->elapsed += (uint32_t)n;             // mod 2^32, like the phases
    b->bus_zeroed[bi] = 0;                 // now holds this block's sums
    b->bus_zeroed[bnext] = (uint32_t)n;    // cleared by the launch
    b->bus_cur = bi;
    return SMX_OK;
}

extern "C" void *smx_bank_bus_dev(smx_bank *b)
{
    if (!b || hipSetDevice(b->device) != hipSuccess || bank_flush_fold(b) != SMX_OK) return nullptr;
    return b->d_bus[b->bus_cur];
}

// process_midi dispatch (linux/synth.c:236-258) for one event, on the bank
extern "C" int smx_bank_midi_event(smx_bank *b, const uint8_t *msg, size_t size)
{
    if (!b || (size && !msg)) return SMX_E_ARG;
    if (size != 3) return SMX_OK;
    if (msg[0] == 0x90) return msg[2] == 0 ? smx_bank_note_off(b, msg[1]) : smx_bank_note_on(b, msg[1]);
    if (msg[0] == 0x80) return smx_bank_note_off(b, msg[1]);
    return SMX_OK;                                          // CC 23..31 on 0xB0: accepted, no action
}

// process_midi's event loop (linux/synth.c:246-258) for a whole block's events: the allocator
// runs on the host in event order, exactly as n_events calls of smx_bank_midi_event would, and
// the increments that result reach the GPU as ONE copy + ONE kernel (a burst of 1000 note-ons
// costs two queue entries instead of 1000 launches).
extern "C" int smx_bank_midi_events(smx_bank *b, const uint8_t *msgs3, size_t n_events)
{
    if (!b || (n_events && !msgs3)) { set_error("smx_bank_midi_events: bad args"); return SMX_E_ARG; }
    if (n_events == 0) return SMX_OK;
    if (n_events > 0x7FFFFFFFu) { set_error("smx_bank_midi_events: n_events=%zu", n_events); return SMX_E_RANGE; }
    SMX_HIP(hipSetDevice(b->device));
    if (b->ev_cap < n_events) {                              // grow the staging (never in steady state)
        SMX_HIP(hipStreamSynchronize(b->stream));
        uint32_t cap = b->ev_cap ? b->ev_cap : 1024u;
        while (cap < n_events) cap *= 2;
        for (int i = 0; i < 2; i++) {
            if (b->h_ev[i]) SMX_HIP(hipHostFree(b->h_ev[i]));
            b->h_ev[i] = nullptr;
            SMX_HIP(hipHostMalloc((void **)&b->h_ev[i], (size_t)cap * 8, hipHostMallocDefault));
            if (!b->ev_stage[i]) SMX_HIP(hipEventCreateWithFlags(&b->ev_stage[i], hipEventDisableTiming));
            b->ev_stage_busy[i] = false;
        }
        if (b->d_ev) SMX_HIP(hipFree(b->d_ev));
        b->d_ev = nullptr;
        SMX_HIP(hipMalloc((void **)&b->d_ev, (size_t)cap * 8));
        b->ev_cap = cap;
        b->ev_net.reserve(cap);
    }
    const int k = b->ev_k;
    if (b->ev_stage_busy[k]) {                               // the copy that last read this slot
        SMX_HIP(hipEventSynchronize(b->ev_stage[k]));
        b->ev_stage_busy[k] = false;
    }
    uint32_t *pairs = b->h_ev[k];
    uint32_t npairs = 0;
    b->ev_net.clear();
    auto put = [&](uint32_t v, uint32_t inc) {
        b->free_map.set_free(v, inc == 0);                   // the (global) allocator sees every event
        uint32_t local;
        if (!bank_owns(b, v, &local)) return;                // another rank's voice
        auto it = b->ev_net.find(local);
        if (it == b->ev_net.end()) { b->ev_net.emplace(local, npairs); pairs[2 * npairs] = local; pairs[2 * npairs + 1] = inc; npairs++; }
        else pairs[2 * it->second + 1] = inc;
    };
    for (size_t i = 0; i < n_events; i++) {
        const uint8_t *m = msgs3 + 3 * i;
        const bool on = m[0] == 0x90 && m[2] != 0;
        const bool off = (m[0] == 0x90 && m[2] == 0) || m[0] == 0x80;
        const int note = m[1] % 128;
        if (on) {                                            // linux/synth.c:156-160
            int64_t v = b->free_map.first_free();
            if (v < 0) v = 0;                                // steal voice 0 (:150-153)
            b->note2voice[note] = (int)v;
            put((uint32_t)v, note_to_inc(note));
        } else if (off) {                                    // linux/synth.c:161-165
            const int v = b->note2voice[note];
            b->note2voice[note] = 0;
            put((uint32_t)v, 0);
        }
    }
    if (npairs == 0) return SMX_OK;
    SMX_HIP(hipMemcpyAsync(b->d_ev, pairs, (size_t)npairs * 8, hipMemcpyHostToDevice, b->stream));
    SMX_HIP(hipEventRecord(b->ev_stage[k], b->stream));
    b->ev_stage_busy[k] = true;
    b->ev_k ^= 1;
    bank_form_unpin(b);
    return smx::launch_saw_rebase_batch(b->d_inc, b->d_state0, b->d_ev, npairs, b->elapsed, b->d_scratch, b->n_pad, b->stream);
}

extern "C" int smx_bank_sync(smx_bank *b)
{
    if (!b) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(b->device));
    int rv = bank_flush_fold(b);                 // after a sync the current bus buffer holds its sums
    if (rv) return rv;
    if (b->comm) {
        rv = bank_comm_flush(b);
        if (rv) return rv;
    }
    SMX_HIP(hipStreamSynchronize(b->stream));
    if (b->comm_stream) SMX_HIP(hipStreamSynchronize(b->comm_stream));
    return SMX_OK;
}

// SMX_NO_PUBLISH=1 (A/B switch): the bus comes back by hipMemcpyAsync + hipStreamSynchronize, as before round 3.
static bool publish_enabled()
{
    static const bool on = getenv("SMX_NO_PUBLISH") == nullptr;
    return on;
}

// Wait until the GPU has published block `seq` (the flag is written after the bus words, system-scope release).
// The wait is bounded: a kernel that never arrives (a lost device, a hung queue) must not hold a real-time thread
// forever -- after SMX_PUBLISH_TIMEOUT_MS (default 10 000) the call fails with SMX_E_NOGPU.
static int bank_wait_published(smx_bank *b, uint32_t seq)
{
    static const double limit_us = [] {
        const char *e = getenv("SMX_PUBLISH_TIMEOUT_MS");
        return (e ? atof(e) : 10000.0) * 1e3;
    }();
    auto now_us = [] {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
    };
    double t0 = -1.0;
    for (uint32_t spins = 1;; spins++) {
        if (__atomic_load_n(b->h_pubflag, __ATOMIC_ACQUIRE) == seq) return SMX_OK;
        __builtin_ia32_pause();
        if ((spins & 0x3FFu) == 0) {                 // look at the clock every 1024 polls
            const double t = now_us();
            if (t0 < 0) t0 = t;
            else if (t - t0 > limit_us) {
                const hipError_t q = hipStreamQuery(b->stream);
                set_error("the GPU did not publish block %u within %.0f ms (stream: %s)", seq, limit_us * 1e-3,
                          q == hipSuccess ? "idle" : hipGetErrorString(q));
                return SMX_E_NOGPU;
            }
        }
    }
}

namespace smx {
// The drop-in synth_run (abi_core.cpp): struct synth's 64 voices for one block, ONE launch, no upload, no copy back
// (saw_dropin_kernel); longer blocks and SMX_NO_PUBLISH take the bank path (load + run + fetch, one synchronisation).
int bank_dropin_run(smx_bank *b, const uint32_t *inc, const uint32_t *state, float *vec, int n)
{
    if (!b || b->n != 64 || !inc || !state || !vec || n <= 0) { set_error("bank_dropin_run: bad args"); return SMX_E_ARG; }
    if (!publish_enabled() || n > 1024) return smx_bank_load_run(b, inc, state, vec, nullptr, n);
    SMX_HIP(hipSetDevice(b->device));
    const uint32_t seq = ++b->pub_seq;
    int rv = launch_saw_dropin(inc, state, b->d_pub, b->d_pubflag, (uint32_t)n, seq, b->stream);
    if (rv) return rv;
    rv = bank_wait_published(b, seq);
    if (rv) return rv;
    for (int i = 0; i < n; i++) vec[i] = bus_to_float(b->h_pub[i]);
    return SMX_OK;
}
}  // namespace smx

// have_seq != 0: the block's own last kernel has already been told to publish under that sequence number.
static int bank_fetch(smx_bank *b, float *vec, int32_t *bus, int n, uint32_t have_seq)
{
    if (!b || n <= 0 || (uint32_t)n > b->bus_cap) { set_error("smx_bank_fetch: bad args"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    const int bi = b->bus_cur;
    const bool can_publish = publish_enabled() && (uint32_t)n <= smx_bank::PUB_MAX;
    uint32_t seq = have_seq;
    int rv;
    if (!seq && can_publish && !b->comm) {
        // a fold that the block still owes (direct form, >= 2^20 voices) is its last kernel: let it publish
        smx::SawPublish pub;
        pub.hbus = b->d_pub; pub.hflag = b->d_pubflag; pub.seq = b->pub_seq + 1u;
        bool published = false;
        rv = bank_flush_fold_pub(b, &pub, &published);
        if (rv) return rv;
        if (published) seq = ++b->pub_seq;
    } else {
        rv = bank_flush_fold(b);
        if (rv) return rv;
    }
    if (bank_ar_queued(b, bi)) {                   // somebody needs the sum now: issue the group
        rv = bank_comm_flush(b);
        if (rv) return rv;
    }
    if (b->comm_pending[bi]) {
        SMX_HIP(hipStreamWaitEvent(b->stream, b->ev_comm[b->ev_owner[bi]], 0));
        b->comm_pending[bi] = false;
    }
    const int32_t *src = b->h_bus;
    if (seq || can_publish) {
        // the stream's last kernel writes the bus to pinned host memory and then the sequence number: no copy
        // engine, no completion signal to wait for (22 -> 12 us per synchronous block on small banks)
        if (!seq) {
            seq = ++b->pub_seq;
            rv = smx::launch_saw_publish(b->d_bus[bi], b->d_pub, b->d_pubflag, (uint32_t)n, seq, b->stream);
            if (rv) return rv;
        }
        rv = bank_wait_published(b, seq);
        if (rv) return rv;
        src = b->h_pub;
    } else {
        SMX_HIP(hipMemcpyAsync(b->h_bus, b->d_bus[bi], (size_t)n * 4, hipMemcpyDeviceToHost, b->stream));
        SMX_HIP(hipStreamSynchronize(b->stream));
    }
    if (bus) memcpy(bus, src, (size_t)n * 4);
    if (vec) for (int i = 0; i < n; i++) vec[i] = bus_to_float(src[i]);
    return SMX_OK;
}

extern "C" int smx_bank_fetch(smx_bank *b, float *vec, int32_t *bus, int n) { return bank_fetch(b, vec, bus, n, 0); }

extern "C" int smx_bank_set_block_form(smx_bank *b, int form)
{
    if (!b || (form != SMX_FORM_AUTO && form != SMX_FORM_STEPPING && form != SMX_FORM_EVENTS)) {
        set_error("smx_bank_set_block_form: form %d", form);
        return SMX_E_ARG;
    }
    b->block_form = form;
    return SMX_OK;
}

// Which form would the next long block of an AUTO bank run?  (Synchronises; for tests and tools.)
extern "C" int smx_bank_next_block_form(smx_bank *b)
{
    if (!b) return SMX_E_ARG;
    if (b->block_form != SMX_FORM_AUTO) return b->block_form;
    if (!b->d_scratch) return SMX_FORM_STEPPING;
    SMX_HIP(hipSetDevice(b->device));
    SMX_HIP(hipStreamSynchronize(b->stream));
    uint32_t pick = 0;
    SMX_HIP(hipMemcpy(&pick, b->d_scratch, 4, hipMemcpyDeviceToHost));
    return pick ? SMX_FORM_EVENTS : SMX_FORM_STEPPING;
}

extern "C" int smx_bank_set_block_mode(smx_bank *b, int mode)
{
    if (!b || (mode != SMX_BLOCK_SYNC && mode != SMX_BLOCK_PIPELINED)) { set_error("smx_bank_set_block_mode: mode %d", mode); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    SMX_HIP(hipStreamSynchronize(b->stream));
    if (mode == SMX_BLOCK_PIPELINED && !b->ev_pipe[0]) {
        SMX_HIP(hipEventCreateWithFlags(&b->ev_pipe[0], hipEventDisableTiming));
        SMX_HIP(hipEventCreateWithFlags(&b->ev_pipe[1], hipEventDisableTiming));
    }
    b->block_mode = mode;
    b->pipe_n[0] = b->pipe_n[1] = 0;
    b->pipe_k = 0;
    return SMX_OK;
}

// Pipelined: launch block k, copy its bus to pinned slot k&1 behind the kernel, and return
// block k-1 from the other slot, which finished while the caller was away: the JACK thread
// never waits for a kernel it has just launched (one block of added latency).
static int bank_run_pipelined(smx_bank *b, float *vec, int32_t *bus, int n)
{
    if ((uint32_t)n > b->pipe_cap) {
        SMX_HIP(hipStreamSynchronize(b->stream));
        const uint32_t cap = smx::round_up((uint32_t)n < 4096u ? 4096u : (uint32_t)n, 4096);
        for (int i = 0; i < 2; i++) {
            if (b->h_pipe[i]) SMX_HIP(hipHostFree(b->h_pipe[i]));
            b->h_pipe[i] = nullptr;
            SMX_HIP(hipHostMalloc((void **)&b->h_pipe[i], (size_t)cap * 4, hipHostMallocDefault));
            b->pipe_n[i] = 0;
        }
        b->pipe_cap = cap;
    }
    int rv = smx_bank_run_async(b, n);
    if (rv) return rv;
    rv = bank_flush_fold(b);                     // the copy below reads this block's bus
    if (rv) return rv;
    const int cur = b->pipe_k & 1, prev = cur ^ 1;
    const int bi = b->bus_cur;
    if (b->comm) {
        // With a communicator the block that is handed out one call later is the SUM over all
        // ranks: the all-reduce of this block is issued at once on the comm stream and the copy
        // to pinned memory follows it THERE, so the compute stream goes straight on to the next
        // block -- the real-time thread waits neither for the kernel nor for the reduce.
        rv = smx_bank_allreduce_async(b, n);
        if (rv) return rv;
        rv = bank_comm_flush(b);
        if (rv) return rv;
        SMX_HIP(hipMemcpyAsync(b->h_pipe[cur], b->d_bus[bi], (size_t)n * 4, hipMemcpyDeviceToHost, b->comm_stream));
        SMX_HIP(hipEventRecord(b->ev_pipe[cur], b->comm_stream));
        // the buffer is busy until the copy has read it: ev_comm[bi] now completes after the copy
        SMX_HIP(hipEventRecord(b->ev_comm[b->ev_owner[bi]], b->comm_stream));
    } else {
        SMX_HIP(hipMemcpyAsync(b->h_pipe[cur], b->d_bus[bi], (size_t)n * 4, hipMemcpyDeviceToHost, b->stream));
        SMX_HIP(hipEventRecord(b->ev_pipe[cur], b->stream));
    }
    b->pipe_n[cur] = n;
    b->pipe_k++;
    const int have = b->pipe_n[prev];
    if (have) SMX_HIP(hipEventSynchronize(b->ev_pipe[prev]));      // normally long complete
    for (int i = 0; i < n; i++) {
        const int32_t v = i < have ? b->h_pipe[prev][i] : 0;       // the first call returns silence
        if (bus) bus[i] = v;
        if (vec) vec[i] = bus_to_float(v);
    }
    return SMX_OK;
}

extern "C" int smx_bank_run(smx_bank *b, float *vec, int32_t *bus, int n)
{
    if (!b || n <= 0) { set_error("smx_bank_run: bad args"); return SMX_E_ARG; }
    if (b->block_mode == SMX_BLOCK_PIPELINED) {
        SMX_HIP(hipSetDevice(b->device));
        return bank_run_pipelined(b, vec, bus, n);
    }
    if (b->comm) {
        // A sharded bank: synth_run returns the sum over all ranks.  The caller waits for this block anyway, so its
        // sum is issued on the COMPUTE stream, right behind the kernel: no hop to the comm stream and back (two
        // cross-stream event waits, ~10 us each on this stack: profiles/r03_bench_one_rank_communicator.json had the
        // synchronous block at 40-47 us against 18-25 without a communicator).  Sums that earlier asynchronous
        // blocks still have queued go first, in order (every rank does the same: the SPMD contract), and the compute
        // stream waits for the youngest collective of the comm stream, so that two collectives of one communicator
        // never run side by side.
        int rv = smx_bank_run_async(b, n);
        if (rv) return rv;
        rv = bank_flush_fold(b);
        if (rv) return rv;
        rv = bank_comm_flush(b);
        if (rv) return rv;
        if (b->last_comm_ev >= 0) SMX_HIP(hipStreamWaitEvent(b->stream, b->ev_comm[b->last_comm_ev], 0));
        const int bi = b->bus_cur;
        SMX_NCCL(ncclAllReduce(b->d_bus[bi], b->d_bus[bi], (size_t)n, ncclInt32, ncclSum, b->comm, b->stream));
        b->ar_launches++;
        b->ar_blocks++;
        return smx_bank_fetch(b, vec, bus, n);
    }
    // one rank: the block is fetched at once, so a finalize kernel that ends it may hand the bus over itself
    smx::SawPublish pub;
    bool published = false;
    if (publish_enabled() && n <= 64) { pub.hbus = b->d_pub; pub.hflag = b->d_pubflag; pub.seq = b->pub_seq + 1u; }
    int rv = bank_run_async(b, n, pub.hflag ? &pub : nullptr, &published);
    if (rv) return rv;
    uint32_t seq = 0;
    if (published) seq = ++b->pub_seq;
    return bank_fetch(b, vec, bus, n, seq);
}

extern "C" int smx_bank_run_square(smx_bank *b, float *vec, int n)
{
    if (!b || n <= 0) { set_error("smx_bank_run_square: bad args"); return SMX_E_ARG; }
    SMX_HIP(hipSetDevice(b->device));
    int rv = bank_ensure_bus(b, (uint32_t)n);
    if (rv) return rv;
    rv = bank_flush_fold(b);                     // an owed fold would zero the buffer this block writes
    if (rv) return rv;
    const int bi = (b->bus_cur + 1) % smx_bank::NBUS;
    rv = bank_bus_release(b, bi);
    if (rv) return rv;
    SMX_HIP(hipMemsetAsync(b->d_bus[bi], 0, (size_t)n * 4, b->stream));
    b->bus_zeroed[bi] = 0;
    rv = smx::launch_square_bank(b->d_inc, b->d_state0, (uint32_t *)b->d_bus[bi], b->n_pad, (uint32_t)n,
                                 b->elapsed, b->stream);
    if (rv) return rv;
    b->elapsed += (uint32_t)n;
    b->bus_cur = bi;
    if (b->comm) {
        // a sharded bank: the OR over all ranks' voices.  Every word is 0 or 0x80000000, so the unsigned maximum
        // is the OR (RCCL has no bitwise reduction); issued at once, after whatever was queued before it
        rv = bank_comm_flush(b);
        if (rv) return rv;
        SMX_HIP(hipEventRecord(b->ev_kernel[bi], b->stream));
        SMX_HIP(hipStreamWaitEvent(b->comm_stream, b->ev_kernel[bi], 0));
        SMX_NCCL(ncclAllReduce(b->d_bus[bi], b->d_bus[bi], (size_t)n, ncclUint32, ncclMax, b->comm, b->comm_stream));
        SMX_HIP(hipEventRecord(b->ev_comm[bi], b->comm_stream));
        b->last_comm_ev = bi;
        SMX_HIP(hipStreamWaitEvent(b->stream, b->ev_comm[bi], 0));
    }
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
    if (b->comm) { set_error("smx_bank_comm_init: the bank already has a communicator"); return SMX_E_STATE; }
    SMX_NCCL(ncclCommInitRank(&b->comm, nranks, u, rank));
    SMX_NCCL(ncclCommCount(b->comm, &b->comm_count));
    // (a high-priority stream was measured and makes the overlapped step 50 % slower)
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
    if (bank_ar_queued(b, bi) || b->comm_pending[bi]) return SMX_OK;      // already requested for this block
    // the queue stays a run of consecutive ring slots (one contiguous all-reduce): a gap (a block
    // without a request in between) or the ring's wrap closes the group first
    if (b->ar_count && b->ar_queue[b->ar_count - 1] + 1 != bi) {
        int rv = bank_comm_flush(b);
        if (rv) return rv;
    }
    b->ar_queue[b->ar_count] = bi;
    b->ar_frames[b->ar_count] = n;
    b->ar_count++;
    if (b->ar_count >= b->comm_group) return bank_comm_flush(b);
    return SMX_OK;
}

extern "C" int smx_bank_set_comm_group(smx_bank *b, int blocks)
{
    if (!b || blocks < 1 || blocks > smx_bank::NBUS / 2) { set_error("smx_bank_set_comm_group: %d (1..%d)", blocks, smx_bank::NBUS / 2); return SMX_E_ARG; }
    if (b->comm) {
        int rv = bank_comm_flush(b);
        if (rv) return rv;
    }
    b->comm_group = blocks;
    return SMX_OK;
}

// Measurement aid (no reference counterpart): what ONE bus sum of n_words int32 costs on this communicator, with no
// kernel in between -- the L of DESIGN 4's latency budget, measured instead of assumed.  Collective: every rank calls
// it with the same arguments.  us_sync: host time per all-reduce when each one is waited for (launch + collective +
// hipStreamSynchronize: what a synchronous smx_bank_run pays on top of its kernel); us_queued: per all-reduce when
// `reps` of them are queued back to back and waited for once (the comm stream's own rate: what throughput mode
// has to hide behind a group of kernels).
extern "C" int smx_bank_comm_probe(smx_bank *b, uint32_t n_words, uint32_t reps, float *us_sync, float *us_queued)
{
    if (!b || n_words == 0 || n_words > (1u << 20) || reps == 0 || reps > 100000u) {
        set_error("smx_bank_comm_probe: n_words=%u (1..2^20) reps=%u (1..100000)", n_words, reps);
        return SMX_E_ARG;
    }
    if (!b->comm) { set_error("smx_bank_comm_probe: smx_bank_comm_init not called"); return SMX_E_STATE; }
    SMX_HIP(hipSetDevice(b->device));
    int rv = smx_bank_sync(b);                      // issues what is queued; both streams idle
    if (rv) return rv;
    struct Scratch {
        int32_t *d = nullptr;
        ~Scratch() { if (d) (void)hipFree(d); }
    } sc;
    SMX_HIP(hipMalloc((void **)&sc.d, (size_t)n_words * 4));
    SMX_HIP(hipMemsetAsync(sc.d, 0, (size_t)n_words * 4, b->comm_stream));
    auto now_us = [] {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
    };
    for (int k = 0; k < 3; k++)                     // warm-up: the first collectives of a size set up their channels
        SMX_NCCL(ncclAllReduce(sc.d, sc.d, n_words, ncclInt32, ncclSum, b->comm, b->comm_stream));
    SMX_HIP(hipStreamSynchronize(b->comm_stream));
    double t0 = now_us();
    for (uint32_t k = 0; k < reps; k++) {
        SMX_NCCL(ncclAllReduce(sc.d, sc.d, n_words, ncclInt32, ncclSum, b->comm, b->comm_stream));
        SMX_HIP(hipStreamSynchronize(b->comm_stream));
    }
    if (us_sync) *us_sync = (float)((now_us() - t0) / reps);
    t0 = now_us();
    for (uint32_t k = 0; k < reps; k++)
        SMX_NCCL(ncclAllReduce(sc.d, sc.d, n_words, ncclInt32, ncclSum, b->comm, b->comm_stream));
    SMX_HIP(hipStreamSynchronize(b->comm_stream));
    if (us_queued) *us_queued = (float)((now_us() - t0) / reps);
    return SMX_OK;
}

extern "C" int smx_bank_comm_ranks(const smx_bank *b) { return (b && b->comm) ? b->comm_count : 0; }

extern "C" int smx_bank_comm_stats(const smx_bank *b, unsigned long long *collectives, unsigned long long *block_sums)
{
    if (!b) return SMX_E_ARG;
    if (collectives) *collectives = b->ar_launches;
    if (block_sums) *block_sums = b->ar_blocks;
    return SMX_OK;
}
