"""GPU parity: cproc dataflow bank (generic/cproc.h acc/edge atoms, PROC_COND chains,
allocation-order tick) against the CPU oracle, bit-exact."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _oracle_run(orc, nodes, n, n_inputs, state, inp, g, out_node):
    arr = (oracle.CprocNode * len(nodes))(*[oracle.CprocNode(*nd) for nd in nodes])
    nt = inp.size // (n_inputs * n)
    out = np.zeros((nt, n), np.uint32)
    orc.orc_cproc_run(arr, len(nodes), n, n_inputs, state.reshape(-1), np.ascontiguousarray(inp).reshape(-1),
                      None if g is None else np.ascontiguousarray(g, np.uint32).ctypes.data, nt, out_node, out.reshape(-1))
    return out


def test_reference_graphs(smx, orc):
    """linux/test_cproc.c:13-17 (edge -> acc) and stm32f103/bp5_plugin.c:4-9 (edge -> acc -> acc),
    every PROC under condition g & 1."""
    from synth_tools_amd import PROC_ACC, PROC_EDGE, cproc_input
    rng = np.random.default_rng(3)
    for nodes in ([(PROC_EDGE, cproc_input(0), 1), (PROC_ACC, 0, 1)],
                  [(PROC_EDGE, cproc_input(0), 1), (PROC_ACC, 0, 1), (PROC_ACC, 1, 1)]):
        n = 300
        bank = smx.CprocBank(n, nodes, 1)
        state = np.zeros((len(nodes), 2, n), np.uint32)
        for call in range(3):
            nt = 50
            inp = rng.integers(0, 3, (nt, 1, n)).astype(np.uint32)
            g = rng.integers(0, 4, nt).astype(np.uint32)         # some ticks skip the subgraph
            got = bank.tick_n(inp, g)
            want = _oracle_run(orc, nodes, n, 1, state, inp, g, len(nodes) - 1)
            assert np.array_equal(got, want)
        assert np.array_equal(bank.read_state(), state)
        bank.close()
    # hand-checked: the chain counts input changes
    bank = smx.CprocBank(1, [(PROC_EDGE, cproc_input(0), 1), (PROC_ACC, 0, 1)], 1)
    out = bank.tick_n(np.array([0, 0, 5, 5, 5, 2, 2, 0], np.uint32).reshape(8, 1, 1))
    assert out[:, 0].tolist() == [0, 0, 1, 1, 1, 2, 2, 3]
    bank.close()


@pytest.mark.parametrize("n_nodes", [2, 4, 5, 8, 9, 12, 32])     # every unroll bucket (4, 8, 16, 32 nodes)
@pytest.mark.parametrize("n", [1, 255, 257, 5000])
def test_random_chains(smx, orc, n, n_nodes):
    from synth_tools_amd import PROC_ACC, PROC_EDGE, cproc_input
    rng = np.random.default_rng(n * 100 + n_nodes)
    n_inputs = 3
    nodes = []
    for k in range(n_nodes):
        src = cproc_input(int(rng.integers(0, n_inputs))) if k == 0 or rng.random() < 0.3 else int(rng.integers(0, k))
        nodes.append((int(rng.choice([PROC_ACC, PROC_EDGE])), src, int(rng.integers(1, 8))))
    bank = smx.CprocBank(n, nodes, n_inputs)
    state = np.zeros((n_nodes, 2, n), np.uint32)
    for call in range(3):
        nt = 33
        inp = rng.integers(0, 2**32, (nt, n_inputs, n), dtype=np.uint64).astype(np.uint32)
        inp[rng.random(inp.shape) < 0.5] = 7                     # repeated values: edges both ways
        g = None if call == 1 else rng.integers(0, 8, nt).astype(np.uint32)
        out_node = int(rng.integers(0, n_nodes))
        got = bank.tick_n(inp, g, out_node)
        want = _oracle_run(orc, nodes, n, n_inputs, state, inp, g, out_node)
        assert np.array_equal(got, want)
    assert np.array_equal(bank.read_state(), state)
    bank.close()


def test_rejects_non_anf(smx):
    from synth_tools_amd import PROC_ACC, cproc_input
    with pytest.raises(smx.SmxError):
        smx.CprocBank(4, [(PROC_ACC, 1, 1), (PROC_ACC, 0, 1)], 1)      # forward reference
    with pytest.raises(smx.SmxError):
        smx.CprocBank(4, [(PROC_ACC, cproc_input(2), 1)], 1)           # no such input
    with pytest.raises(smx.SmxError):
        smx.CprocBank(4, [(9, cproc_input(0), 1)], 1)                  # unknown processor
