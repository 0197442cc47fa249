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

__global__ __launch_bounds__(256)
void cproc_kernel(smx::CprocProgram prog, uint32_t *__restrict__ state,   // [node][2][n_pad]
                  const uint32_t *__restrict__ input,                      // [t][n_inputs][n_pad]
                  const uint32_t *__restrict__ g, uint32_t *__restrict__ out,  // [t][n_pad]
                  uint32_t n_pad, uint32_t nticks, uint32_t out_node)
{
    extern __shared__ uint32_t lds[];                 // [n_nodes][2][256]
    const uint32_t tid = threadIdx.x, inst = blockIdx.x * 256u + tid;
    for (uint32_t k = 0; k < prog.n_nodes; k++) {
        lds[(k * 2 + 0) * 256 + tid] = state[((size_t)k * 2 + 0) * n_pad + inst];
        lds[(k * 2 + 1) * 256 + tid] = state[((size_t)k * 2 + 1) * n_pad + inst];
    }
    for (uint32_t t = 0; t < nticks; t++) {
        const uint32_t gt = g ? g[t] : 0xFFFFFFFFu;
        for (uint32_t k = 0; k < prog.n_nodes; k++) {           // allocation order
            const smx::CprocNode nd = prog.nodes[k];
            if (!(gt & nd.cond)) continue;                      // PROC_COND
            const uint32_t in = (nd.in & 0x80000000u)
                ? input[((size_t)t * prog.n_inputs + (nd.in & 0x7FFFFFFFu)) * n_pad + inst]
                : lds[(nd.in * 2) * 256 + tid];
            uint32_t *o = &lds[(k * 2 + 0) * 256 + tid];
            uint32_t *l = &lds[(k * 2 + 1) * 256 + tid];
            if (nd.proc == 1) {                                 // acc, cproc.h:142-144
                *o += in;
            } else {                                            // edge, cproc.h:152-155
                *o = (in != *l);
                *l = in;
            }
        }
        if (out) out[(size_t)t * n_pad + inst] = lds[(out_node * 2) * 256 + tid];
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
    hipLaunchKernelGGL(cproc_kernel, dim3(n_pad / 256), dim3(256), prog.n_nodes * 2 * 256 * 4, stream,
                       prog, d_state, d_input, d_g, d_out, n_pad, nticks, out_node);
    SMX_HIP(hipGetLastError());
    return SMX_OK;
}

}  // namespace smx
