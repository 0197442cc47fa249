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


def test_dynamic_patcher(smx, orc):
    """mod_bpmodular.c: `class/<cls>/apply` allocates instances one by one and connects them by node index
    (:84-113), `patch/tick` runs them in allocation order (:72-78), `inst/<n>/state/<k>/get|set` (:153-190),
    `patch/reset` (:218-222); answers bad_ref / bad_node / alloc_fail.  Classes: acc, edge (cproc.h), gpin,
    gpout (hw_cproc_stm32f103.h; the pin replaced by an input / output word)."""
    from synth_tools_amd import (PROC_ACC, PROC_EDGE, PROC_GPIN, PROC_GPOUT, PATCH_BAD_REF, PATCH_BAD_NODE,
                                 PATCH_ALLOC_FAIL, cproc_input)
    n, n_inputs = 700, 3
    rng = np.random.default_rng(11)
    p = smx.Patch(n, n_inputs)
    assert p.count() == 0
    assert p.apply(PROC_ACC, [0]) == PATCH_BAD_NODE                 # no node 0 yet (node_to_inst == NULL, :102-106)
    assert p.apply(7, [0]) == PATCH_BAD_REF                         # no such class (:285)
    assert p.apply(PROC_ACC, []) == PATCH_BAD_REF                   # nb_args != input.nb_fields (:287)
    assert p.apply(PROC_GPIN, [], config=n_inputs) == PATCH_BAD_REF
    for round_ in range(2):
        # a random patch: sources first, then processors reading any earlier node, gpouts anywhere
        classes, srcs, kernel_nodes, kmap, gpouts = [], [], [], {}, []
        for k in range(40):
            if k < 2 or rng.random() < 0.15:
                cls, src, cfg = PROC_GPIN, [], int(rng.integers(0, n_inputs))
            else:
                readable = [j for j, c in enumerate(classes) if c != PROC_GPOUT]
                cls = int(rng.choice([PROC_ACC, PROC_EDGE, PROC_EDGE, PROC_GPOUT]))
                src, cfg = [int(rng.choice(readable))], 0
            got = p.apply(cls, src, cfg)
            if cls != PROC_GPOUT and len(kernel_nodes) == 32:
                assert got == PATCH_ALLOC_FAIL
                continue
            assert got == len(classes), (k, got)
            if cls == PROC_GPOUT:
                kmap[got] = kmap[src[0]]
                gpouts.append(got)
            else:
                kmap[got] = len(kernel_nodes)
                kernel_nodes.append((cls, cproc_input(cfg) if cls == PROC_GPIN else kmap[src[0]], 0xFFFFFFFF))
            classes.append(cls)
            srcs.append(src)
        assert p.count() == len(classes) and gpouts
        # a gpout has no state to read; as in the reference the allocation is tried first (:89-94, then :102-106)
        assert p.apply(PROC_ACC, [gpouts[0]]) == (PATCH_ALLOC_FAIL if len(kernel_nodes) == 32 else PATCH_BAD_NODE)
        state = np.zeros((len(kernel_nodes), 2, n), np.uint32)
        # poke some state from outside (inst/<n>/state/<k>/set)
        for node in rng.choice(len(classes), 5, replace=False):
            node = int(node)
            if classes[node] == PROC_GPOUT:
                assert p.state_set(node, 0, 1) == PATCH_BAD_REF
                continue
            vals = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
            assert p.state_set(node, 0, vals) == 0
            state[kmap[node], 0] = vals
        assert p.state_set(0, 1, 5) == PATCH_BAD_REF                # a gpin has one state word
        assert isinstance(p.state_get(len(classes), 0), int)        # no such node
        for call in range(3):
            nt = 20
            inp = rng.integers(0, 4, (nt, n_inputs, n)).astype(np.uint32)
            gp = int(rng.choice(gpouts))
            got = p.tick(nt, inp, gp)
            want = _oracle_run(orc, kernel_nodes, n, n_inputs, state, inp, None, kmap[gp])
            assert np.array_equal(got, want)
        for node, cls in enumerate(classes):
            if cls == PROC_GPOUT:
                continue
            assert np.array_equal(p.state_get(node, 0), state[kmap[node], 0])
            if cls == PROC_EDGE:
                assert np.array_equal(p.state_get(node, 1), state[kmap[node], 1])
        p.reset()
        assert p.count() == 0
    # the bump allocator: 1024 words (:27); an edge takes 1 + 2 + 1 = 4 of them, a gpout 1 + 0 + 1 = 2
    assert p.apply(PROC_GPIN, [], 0) == 0
    for k in range(1, 64):
        assert p.apply(PROC_GPOUT, [0]) == k
    assert p.apply(PROC_GPOUT, [0]) == PATCH_ALLOC_FAIL             # node table full (a bound of this build)
    p.close()


@pytest.mark.parametrize("n_inputs", [1, 2, 5, 8, 9, 20])
def test_input_staging_paths(smx, orc, n_inputs):
    """Up to 8 input words per tick the kernel runs its double-buffered pipeline (chunks of 8 / n_inputs ticks); more
    words take the single staging buffer.  Ragged tick counts, several calls, every input word used."""
    from synth_tools_amd import PROC_ACC, PROC_EDGE, cproc_input
    rng = np.random.default_rng(700 + n_inputs)
    nodes = [(PROC_EDGE if w % 2 else PROC_ACC, cproc_input(w), 1 + w % 3) for w in range(n_inputs)]
    nodes += [(PROC_ACC, int(rng.integers(0, n_inputs)), 3), (PROC_EDGE, n_inputs, 1)]
    nodes = nodes[:32]
    n = 777
    bank = smx.CprocBank(n, nodes, n_inputs)
    state = np.zeros((len(nodes), 2, n), np.uint32)
    for nt in (1, 7, 8, 9, 33):
        inp = rng.integers(0, 5, (nt, n_inputs, n)).astype(np.uint32)
        g = rng.integers(0, 8, nt).astype(np.uint32)
        got = bank.tick_n(inp, g)
        want = _oracle_run(orc, nodes, n, n_inputs, state, inp, g, len(nodes) - 1)
        assert np.array_equal(got, want), (n_inputs, nt)
    assert np.array_equal(bank.read_state(), state)
    bank.close()


def test_dynamic_patcher_grows_between_ticks(smx, orc):
    """Instances applied BETWEEN ticks (mod_bpmodular.c:84-113 after :72-78 has run): the network grows one node at a
    time while it is running, earlier instances keep their state (the reference's bump allocator never moves them; here
    the device state is re-reserved and copied when the node table grows), a new node starts from zero state, and state
    poked from outside in between survives the next growth."""
    from synth_tools_amd import PROC_ACC, PROC_EDGE, PROC_GPIN, PROC_GPOUT, cproc_input
    n, n_inputs = 1300, 2
    rng = np.random.default_rng(0x9A7C)
    p = smx.Patch(n, n_inputs)
    classes, kernel_nodes, kmap, gpouts = [], [], {}, []
    state = np.zeros((0, 2, n), np.uint32)
    for step in range(30):
        if step < 2 or rng.random() < 0.2:
            cls, src, cfg = PROC_GPIN, [], int(rng.integers(0, n_inputs))
        else:
            readable = [j for j, c in enumerate(classes) if c != PROC_GPOUT]
            cls = int(rng.choice([PROC_ACC, PROC_EDGE, PROC_GPOUT]))
            src, cfg = [int(rng.choice(readable))], 0
        got = p.apply(cls, src, cfg)
        assert got == len(classes), (step, got)
        if cls == PROC_GPOUT:
            kmap[got] = kmap[src[0]]
            gpouts.append(got)
        else:
            kmap[got] = len(kernel_nodes)
            kernel_nodes.append((cls, cproc_input(cfg) if cls == PROC_GPIN else kmap[src[0]], 0xFFFFFFFF))
            state = np.concatenate([state, np.zeros((1, 2, n), np.uint32)])      # a new instance starts from zero (cproc.h:65-66)
        classes.append(cls)
        if step % 5 == 3:                                    # poke an earlier instance between growths
            node = int(rng.choice([j for j, c in enumerate(classes) if c in (PROC_ACC, PROC_EDGE)] or [0]))
            if classes[node] in (PROC_ACC, PROC_EDGE):
                vals = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
                assert p.state_set(node, 0, vals) == 0
                state[kmap[node], 0] = vals
        nt = int(rng.integers(1, 9))
        inp = rng.integers(0, 3, (nt, n_inputs, n)).astype(np.uint32)
        if gpouts:                                           # the last sink's words of these ticks
            gp = gpouts[-1]
            res = p.tick(nt, inp, gp)
            want = _oracle_run(orc, kernel_nodes, n, n_inputs, state, inp, None, kmap[gp])
            assert np.array_equal(res, want), step
        else:                                                # no sink yet: the ticks run all the same
            p.tick(nt, inp)
            _oracle_run(orc, kernel_nodes, n, n_inputs, state, inp, None, kmap[got])
    for node, cls in enumerate(classes):
        if cls == PROC_GPOUT:
            continue
        assert np.array_equal(p.state_get(node, 0), state[kmap[node], 0]), node
        if cls == PROC_EDGE:
            assert np.array_equal(p.state_get(node, 1), state[kmap[node], 1]), node
    p.close()
