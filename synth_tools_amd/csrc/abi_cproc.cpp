// abi_cproc.cpp -- part of the C-ABI of libsynth_mi355x.so (include/synth_mi355x.h): cproc dataflow bank
// Host side of the drop-in boundary.  No CPU compute fallback exists: every compute entry
// point needs a HIP device and fails with SMX_E_NOGPU otherwise.
#include "abi_internal.h"
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
        if (!ok_in || (nodes[k].proc != SMX_PROC_ACC && nodes[k].proc != SMX_PROC_EDGE && nodes[k].proc != SMX_PROC_GPIN) ||
            (nodes[k].proc == SMX_PROC_GPIN && !(in & 0x80000000u))) {       // a gpin reads an input word
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
// Dynamic patcher: stm32f103/mod_bpmodular.c:36-45 (struct proc / struct inst), :72-78 (tick),
// :84-113 (apply), :218-222 (reset).  Instances are allocated one by one and connected by node
// index; the network runs in allocation order.  Here: N copies of the network, one per lane.
// A gpout has no state and computes nothing on the device: it is a patch node that names its source,
// so patch node numbers and kernel node numbers differ (map[]).
// ---------------------------------------------------------------------------
struct smx_patch {
    smx_cproc *c = nullptr;          // the bank that runs the network (node table grows with apply)
    uint32_t cap_nodes = 0;          // kernel nodes the device state is allocated for
    uint32_t words = 0;              // words the reference's bump allocator would have handed out
    uint32_t count = 0;              // patch nodes (alloc.count)
    uint32_t cls[SMX_CPROC_MAX_NODES * 2];
    uint32_t map[SMX_CPROC_MAX_NODES * 2];   // patch node -> kernel node (gpout: the kernel node of its source)
};
static constexpr uint32_t PATCH_ALLOC_WORDS = 1024;     // ALLOC_NB_WORDS, mod_bpmodular.c:27

// (fields of the state struct, inputs) of a processor class: cproc.h:134-155, hw_cproc_stm32f103.h:8-22
static bool patch_class(uint32_t cls, uint32_t *n_state, uint32_t *n_in)
{
    if (cls == SMX_PROC_ACC) { *n_state = 1; *n_in = 1; return true; }      // state {out}, input {in}
    if (cls == SMX_PROC_EDGE) { *n_state = 2; *n_in = 1; return true; }     // state {out, last}, input {in}
    if (cls == SMX_PROC_GPIN) { *n_state = 1; *n_in = 0; return true; }     // state {out}, config {port, pin}
    if (cls == SMX_PROC_GPOUT) { *n_state = 0; *n_in = 1; return true; }    // input {in}, config {port, pin}
    return false;
}

static int patch_reserve(smx_patch *p, uint32_t nodes)
{
    if (nodes <= p->cap_nodes) return SMX_OK;
    smx_cproc *c = p->c;
    uint32_t cap = p->cap_nodes ? p->cap_nodes : 4u;
    while (cap < nodes) cap *= 2;
    SMX_HIP(hipSetDevice(c->device));
    SMX_HIP(hipStreamSynchronize(c->stream));
    uint32_t *d = nullptr;
    const size_t row = (size_t)c->n_pad * 4;
    SMX_HIP(hipMalloc((void **)&d, (size_t)cap * 2 * row));
    // Cleared and copied ON THE BANK'S STREAM: the null stream's hipMemset returns before the fill has run and a
    // non-blocking stream does not wait for it (the ring of the saw bank once met its own clear).
    SMX_HIP(hipMemsetAsync(d, 0, (size_t)cap * 2 * row, c->stream));        // state initialises to zero (:98)
    if (c->d_state) {
        SMX_HIP(hipMemcpyAsync(d, c->d_state, (size_t)p->cap_nodes * 2 * row, hipMemcpyDeviceToDevice, c->stream));
        SMX_HIP(hipStreamSynchronize(c->stream));                           // the old rows are read before they are freed
        SMX_HIP(hipFree(c->d_state));
    }
    c->d_state = d;
    p->cap_nodes = cap;
    return SMX_OK;
}

extern "C" smx_patch *smx_patch_create(uint32_t n_instances, uint32_t n_inputs, int device)
{
    if (n_instances == 0 || n_instances > 0xFFFFF000u || n_inputs > 32) {
        set_error("smx_patch_create: n_instances=%u n_inputs=%u", n_instances, n_inputs);
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("smx_patch_create: no HIP device (this library has no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) { set_error("smx_patch_create: device %d of %d", device, ndev); return nullptr; }
    smx_patch *p = new smx_patch();
    p->c = new smx_cproc();
    p->c->n = n_instances;
    p->c->n_pad = smx::round_up(n_instances, 256);
    p->c->device = device;
    p->c->prog.n_inputs = n_inputs;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&p->c->stream, hipStreamNonBlocking) != hipSuccess || patch_reserve(p, 4) != SMX_OK) {
        set_error("smx_patch_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
        smx_patch_destroy(p);
        return nullptr;
    }
    return p;
}

extern "C" void smx_patch_destroy(smx_patch *p)
{
    if (!p) return;
    smx_cproc_destroy(p->c);
    delete p;
}

extern "C" uint32_t smx_patch_count(const smx_patch *p) { return p ? p->count : 0; }

// apply (mod_bpmodular.c:84-113, handle_apply :283-293): a new instance of class `cls` whose inputs are
// the `out` words of existing nodes.  Returns the node index, or SMX_PATCH_BAD_REF (unknown class or wrong
// number of inputs), SMX_PATCH_BAD_NODE (an input that is not an existing node), SMX_PATCH_ALLOC_FAIL.
extern "C" int smx_patch_apply(smx_patch *p, uint32_t cls, const uint32_t *in, uint32_t n_in, uint32_t config)
{
    if (!p) return SMX_E_ARG;
    uint32_t n_state = 0, want_in = 0;
    smx::CprocProgram &prog = p->c->prog;
    if (!patch_class(cls, &n_state, &want_in) || n_in != want_in || (n_in && !in) ||
        (cls == SMX_PROC_GPIN && config >= prog.n_inputs)) {
        set_error("smx_patch_apply: bad_ref (class %u, %u inputs, config %u)", cls, n_in, config);
        return SMX_PATCH_BAD_REF;
    }
    const uint32_t nb_words = 1 + n_state + n_in;                           // :88
    const bool on_device = cls != SMX_PROC_GPOUT;
    if (p->words + nb_words > PATCH_ALLOC_WORDS || p->count >= SMX_CPROC_MAX_NODES * 2 ||
        (on_device && prog.n_nodes >= SMX_CPROC_MAX_NODES)) {
        set_error("smx_patch_apply: alloc_fail");
        return SMX_PATCH_ALLOC_FAIL;
    }
    for (uint32_t i = 0; i < n_in; i++)
        // an input reads the first state word of its source (:107): a gpout has none
        if (in[i] >= p->count || p->cls[in[i]] == SMX_PROC_GPOUT) { set_error("smx_patch_apply: bad_node %u", in[i]); return SMX_PATCH_BAD_NODE; }
    const uint32_t node = p->count;
    if (on_device) {
        int rv = patch_reserve(p, prog.n_nodes + 1);
        if (rv) return rv;
        const uint32_t src = cls == SMX_PROC_GPIN ? SMX_CPROC_INPUT(config) : p->map[in[0]];
        prog.nodes[prog.n_nodes] = {cls, src, 0xFFFFFFFFu};                 // synchronous graph: always runs
        p->map[node] = prog.n_nodes++;
    } else {
        p->map[node] = p->map[in[0]];
    }
    p->cls[node] = cls;
    p->count = node + 1;
    p->words += nb_words;
    return (int)node;
}

// handle_reset (mod_bpmodular.c:218-222): balloci_clear.  Later instances start from zero state again.
extern "C" int smx_patch_reset(smx_patch *p)
{
    if (!p) return SMX_E_ARG;
    smx_cproc *c = p->c;
    SMX_HIP(hipSetDevice(c->device));
    SMX_HIP(hipMemsetAsync(c->d_state, 0, (size_t)p->cap_nodes * 2 * c->n_pad * 4, c->stream));
    SMX_HIP(hipStreamSynchronize(c->stream));
    c->prog.n_nodes = 0;
    p->words = 0;
    p->count = 0;
    return SMX_OK;
}

// handle_tick (mod_bpmodular.c:224-228) n_ticks times: every instance, allocation order.
extern "C" int smx_patch_tick(smx_patch *p, uint32_t n_ticks, const uint32_t *input, uint32_t gpout, uint32_t *out)
{
    if (!p) return SMX_E_ARG;
    if (out && (gpout >= p->count || p->cls[gpout] != SMX_PROC_GPOUT)) {
        set_error("smx_patch_tick: node %u is not a gpout", gpout);
        return SMX_PATCH_BAD_REF;
    }
    bool has_gpin = false;
    for (uint32_t k = 0; k < p->count; k++) has_gpin = has_gpin || p->cls[k] == SMX_PROC_GPIN;
    if (has_gpin && n_ticks && !input) { set_error("smx_patch_tick: the patch has gpin nodes: input needed"); return SMX_E_ARG; }
    if (n_ticks == 0 || p->c->prog.n_nodes == 0) return SMX_OK;
    // a gpout writes its input as it is when the gpout runs: its source ran earlier in the same tick
    // (allocation order), so that is the source's `out` after the tick
    return smx_cproc_tick_n(p->c, n_ticks, input, nullptr, out ? p->map[gpout] : 0, out);
}

// inst/<node>/state/<field>/get|set (mod_bpmodular.c:153-190) for every copy of the network:
// vals is host uint32[n_instances].  Field 0 is `out`.
static int patch_state_row(smx_patch *p, uint32_t node, uint32_t field, uint32_t **row)
{
    uint32_t n_state = 0, n_in = 0;
    if (!p || node >= p->count || !patch_class(p->cls[node], &n_state, &n_in) || field >= n_state) {
        set_error("smx_patch_state: bad_ref (node %u field %u)", node, field);
        return SMX_PATCH_BAD_REF;
    }
    *row = p->c->d_state + ((size_t)p->map[node] * 2 + field) * p->c->n_pad;
    return SMX_OK;
}
extern "C" int smx_patch_state_get(smx_patch *p, uint32_t node, uint32_t field, uint32_t *vals)
{
    uint32_t *row = nullptr;
    int rv = patch_state_row(p, node, field, &row);
    if (rv) return rv;
    if (!vals) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->c->device));
    SMX_HIP(hipStreamSynchronize(p->c->stream));
    SMX_HIP(hipMemcpy(vals, row, (size_t)p->c->n * 4, hipMemcpyDeviceToHost));
    return SMX_OK;
}
extern "C" int smx_patch_state_set(smx_patch *p, uint32_t node, uint32_t field, const uint32_t *vals)
{
    uint32_t *row = nullptr;
    int rv = patch_state_row(p, node, field, &row);
    if (rv) return rv;
    if (!vals) return SMX_E_ARG;
    SMX_HIP(hipSetDevice(p->c->device));
    SMX_HIP(hipStreamSynchronize(p->c->stream));
    SMX_HIP(hipMemcpy(row, vals, (size_t)p->c->n * 4, hipMemcpyHostToDevice));
    return SMX_OK;
}
