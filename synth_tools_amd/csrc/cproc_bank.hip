// cproc_bank.hip -- N instances of one static dataflow chain in the cproc ABI, for
// gfx950 (MI355X).
//
// generic/cproc.h:72-81 defines a dataflow program as a sequence of PROC_COND
// bindings in A-normal form: each processor has zero-initialised private state with
// an `out` word, named inputs (external input words or earlier nodes' `out`), and
// runs only when its subgraph condition holds; stm32f103/mod_bpmodular.c:72-78 runs
// dynamically allocated instances in allocation (= topological) order.  All atoms
// are uint32 (`typedef uint32_t w`, cproc.h:128).  Processors restated from
// cproc.h:134-155: acc (out += in) and edge (out = in != last; last = in).
//
// Mapping: one lane per graph instance (N voices / N plugin boards running the same
// patch); the node table is wave-uniform (kernel argument), so the per-node
// dispatch is scalar control flow with no lane divergence.  Node state lives in LDS
// as state[node][field][lane] (conflict-free: consecutive lanes, consecutive
// banks) because it is indexed by a run-time node number.
#include "smx_common.h"

namespace {

// MAXN: compile-time bound of the node loop (4, 8, 16 or 32).  With the loop fully unrolled every
// prog.nodes[k] is a loop-invariant kernel-argument load that the compiler hoists out of the tick
// loop into scalar registers, and the node's own state address is an immediate LDS offset (the
// run-time node loop re-read the table from the kernel arguments for every node of every tick).
template <int MAXN>
__global__ __launch_bounds__(256)
void cproc_kernel(smx::CprocProgram prog, uint32_t *__restrict__ state,   // [node][2][n_pad]
                  const uint32_t *__restrict__ input,                      // [t][n_inputs][n_pad]
                  const uint32_t *__restrict__ g, uint32_t *__restrict__ out,  // [t][n_pad]
                  uint32_t n_pad, uint32_t nticks, uint32_t out_node,
                  uint32_t depth,                     // ticks of input staged ahead in LDS (0: none)
                  uint32_t pipe)                      // 1: two staging buffers of 8 rows, the next chunk's loads issued
                                                      //    before the current chunk's ticks (depth * n_inputs <= 8)
{
    extern __shared__ uint32_t lds[];                 // [n_nodes][2][256], then [depth][n_inputs][256] (pipe: [2][8][256])
    const uint32_t tid = threadIdx.x, inst = blockIdx.x * 256u + tid;
    for (uint32_t k = 0; k < prog.n_nodes; k++) {
        lds[(k * 2 + 0) * 256 + tid] = state[((size_t)k * 2 + 0) * n_pad + inst];
        lds[(k * 2 + 1) * 256 + tid] = state[((size_t)k * 2 + 1) * n_pad + inst];
    }
    // The inputs do not depend on the state, so the rows of the next `depth` ticks are requested
    // together (8 loads in flight per lane) and parked in the lane's own LDS column; a load inside
    // the tick recurrence costs its full latency every tick (1 Mi instances x 256 ticks of the
    // bp5 chain: 797 us with the load in the loop).
    uint32_t *inbuf = lds + prog.n_nodes * 2 * 256 + tid;
    const size_t in_stride = (size_t)prog.n_inputs * n_pad;
    auto run_tick = [&](uint32_t t, const uint32_t *row, size_t word_stride) {
        const uint32_t gt = g ? g[t] : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < MAXN; k++) {                        // allocation order
            if ((uint32_t)k >= prog.n_nodes) break;
            const smx::CprocNode nd = prog.nodes[k];
            if (!(gt & nd.cond)) continue;                      // PROC_COND
            const uint32_t in = (nd.in & 0x80000000u)
                ? row[(size_t)(nd.in & 0x7FFFFFFFu) * word_stride]
                : lds[(nd.in * 2) * 256 + tid];
            uint32_t *o = &lds[(k * 2 + 0) * 256 + tid];
            uint32_t *l = &lds[(k * 2 + 1) * 256 + tid];
            // acc (cproc.h:142-144): out += in;  gpin (hw_cproc_stm32f103.h:12-14): out = in;  edge (cproc.h:152-155):
            // out = (in != last), last = in.  One wave-uniform branch per node (the edge's `last`); acc against gpin is
            // a mask on the old `out`, not a branch: a three-way if/else here cost 17 % of the kernel (the
            // structurizer's chain of uniform branches: 1 Mi instances x 256 ticks of the bp5 chain 600 -> 705 us)
            const uint32_t keep = 0u - (uint32_t)(nd.proc == 1);
            uint32_t x = in;
            if (nd.proc == 2) {
                x = (in != *l) ? 1u : 0u;
                *l = in;
            }
            *o = (*o & keep) + x;
        }
        if (out) out[(size_t)t * n_pad + inst] = lds[(out_node * 2) * 256 + tid];
    };
    if (depth == 0) {
        for (uint32_t t = 0; t < nticks; t++) run_tick(t, input + (size_t)t * in_stride + inst, n_pad);
    } else if (pipe) {
        // Software pipeline: every workgroup runs the same load / tick / load / tick rhythm, so without it the whole
        // chip waits for memory and then computes, in turns (bp5 chain, 1 Mi instances x 256 ticks: 593 us against
        // 437 us for a one-node graph that moves the same rows).  The loads of chunk k+1 are issued into registers
        // BEFORE the ticks of chunk k and parked in the other LDS buffer after them.
        uint32_t v[8];
        auto issue = [&](uint32_t t0) {
            const uint32_t rows = min(depth, nticks - t0) * prog.n_inputs;
            const uint32_t *base = input + (size_t)t0 * in_stride + inst;
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = ((uint32_t)i < rows) ? base[(size_t)i * n_pad] : 0u;
        };
        auto park = [&](uint32_t buf) {
#pragma unroll
            for (int i = 0; i < 8; i++) inbuf[(buf * 8 + i) * 256] = v[i];
        };
        issue(0);
        park(0);
        for (uint32_t t0 = 0, k = 0; t0 < nticks; t0 += depth, k++) {
            const uint32_t nd_ticks = min(depth, nticks - t0);
            const bool more = t0 + depth < nticks;
            if (more) issue(t0 + depth);
            for (uint32_t d = 0; d < nd_ticks; d++)
                run_tick(t0 + d, inbuf + ((k & 1u) * 8 + d * prog.n_inputs) * 256, 256);
            if (more) park((k + 1) & 1u);
        }
    } else {
        for (uint32_t t0 = 0; t0 < nticks; t0 += depth) {
            const uint32_t nd_ticks = min(depth, nticks - t0);
            const uint32_t rows = nd_ticks * prog.n_inputs;     // [t][word] is contiguous in rows
            const uint32_t *base = input + (size_t)t0 * in_stride + inst;
            for (uint32_t e0 = 0; e0 < rows; e0 += 8) {
                uint32_t v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = (e0 + i < rows) ? base[(size_t)(e0 + i) * n_pad] : 0u;
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (e0 + i < rows) inbuf[(e0 + i) * 256] = v[i];
            }
            for (uint32_t d = 0; d < nd_ticks; d++) run_tick(t0 + d, inbuf + d * prog.n_inputs * 256, 256);
        }
    }
    for (uint32_t k = 0; k < prog.n_nodes; k++) {
        state[((size_t)k * 2 + 0) * n_pad + inst] = lds[(k * 2 + 0) * 256 + tid];
        state[((size_t)k * 2 + 1) * n_pad + inst] = lds[(k * 2 + 1) * 256 + tid];
    }
}

}  // namespace

namespace smx {

int launch_cproc(const CprocProgram &prog, uint32_t *d_state, const uint32_t *d_input,
                 const uint32_t *d_g, uint32_t *d_out, uint32_t n_pad, uint32_t nticks,
                 uint32_t out_node, hipStream_t stream)
{
    if (n_pad == 0 || (n_pad & 255) || prog.n_nodes == 0 || prog.n_nodes > SMX_CPROC_MAX_NODES ||
        out_node >= prog.n_nodes) {
        set_error("launch_cproc: n_pad=%u n_nodes=%u out_node=%u", n_pad, prog.n_nodes, out_node);
        return SMX_E_ARG;
    }
    if (nticks == 0) return SMX_OK;
    // LDS: 2 KB of state per node; what is left of 64 KB stages up to 8 ticks of input (1 KB per word)
    const uint32_t state_kb = prog.n_nodes * 2;
    uint32_t depth = prog.n_inputs ? (64u - state_kb) / prog.n_inputs : 0u;
    if (depth > 8) depth = 8;
    // up to 8 input words: chunks of 8 rows (8 / n_inputs ticks), double-buffered (16 KB of staging)
    static const bool no_pipe = getenv("SMX_CPROC_NO_PIPE") != nullptr;      // A/B switch
    uint32_t pipe = 0;
    if (!no_pipe && prog.n_inputs >= 1 && prog.n_inputs <= 8 && state_kb + 16u <= 64u) {
        pipe = 1;
        depth = 8u / prog.n_inputs;
    }
    const uint32_t lds_bytes = (state_kb + (pipe ? 16u : depth * prog.n_inputs)) * 1024u;
#define SMX_CPROC_LAUNCH(MAXN_)                                                                        \
    hipLaunchKernelGGL(cproc_kernel<MAXN_>, dim3(n_pad / 256), dim3(256), lds_bytes,                   \
                       stream, prog, d_state, d_input, d_g, d_out, n_pad, nticks, out_node, depth, pipe)
    if (prog.n_nodes <= 4)       SMX_CPROC_LAUNCH(4);
    else if (prog.n_nodes <= 8)  SMX_CPROC_LAUNCH(8);
    else if (prog.n_nodes <= 16) SMX_CPROC_LAUNCH(16);
    else                         SMX_CPROC_LAUNCH(32);
#undef SMX_CPROC_LAUNCH
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
