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

