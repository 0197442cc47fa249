"""CPU: the C-ABI library loads and exports every symbol include/synth_mi355x.h
declares; host-only entry points (tables, allocator, packing) behave like the
reference; compute entry points refuse to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import synth_tools_amd as sta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "synth_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    funcs = set(re.findall(r"\b([a-z_][a-z0-9_]*)\s*\(", text))
    funcs -= {"defined", "sizeof"}
    data = set(re.findall(r"extern\s+const\s+\w+\s+(\w+)\s*\[", text))
    return funcs, data


def test_every_declared_symbol_is_exported():
    L = C.CDLL(sta.LIB_PATH)
    funcs, data = _declared_symbols()
    assert len(funcs) >= 40
    for name in sorted(funcs | data):
        assert hasattr(L, name), "libsynth_mi355x.so does not export %s" % name
    bound = {n for n, _, _ in sta.ABI} | set(sta.ABI_DATA)
    assert funcs | data == bound, "binding table and header disagree: %s" % sorted((funcs | data) ^ bound)


def test_nothing_but_the_declared_abi_is_exported():
    """The dynamic symbol table is the header, no more (csrc/exports.map): the C++ launchers
    between the ABI files and the kernels are not part of the boundary."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", sta.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TBDRW"}
    funcs, data = _declared_symbols()
    assert exported == funcs | data, sorted(exported ^ (funcs | data))


def test_note_tables_match_oracle(orc):
    L = sta.lib()
    for n in range(-3, 260):
        assert L.note_to_inc(n) == orc.orc_note_to_inc(n)
    tab = (C.c_uint8 * 128).in_dll(L, "midi_tab")
    assert [tab[i] for i in range(128)] == [orc.orc_midi_tab(i) for i in range(128)]


def test_struct_synth_layout():
    assert C.sizeof(sta.Synth) == 1024          # linux/synth.c:35-38
    assert sta.Synth.voice.offset == 512


def test_host_side_note_logic_matches_oracle(orc):
    """synth_init/note_on/note_off/voice_alloc/midi dispatch touch only the caller's
    struct (linux/synth.c:145-165, 236-258): checked on the CPU against the oracle."""
    L = sta.lib()
    x = sta.Synth()
    L.synth_init(C.byref(x))
    n2v = np.zeros(128, np.int32)
    inc = np.zeros(64, np.uint32)
    rng = np.random.default_rng(5)
    for _ in range(600):
        kind = int(rng.integers(0, 4))
        note = int(rng.integers(0, 256))
        vel = int(rng.integers(0, 3)) * 60
        msg = np.array([[0x90, note, vel], [0x80, note, vel], [0xB0, 25, vel], [0x91, note, vel]][kind], np.uint8)
        L.synth_midi_event(C.byref(x), msg, 3)
        orc.orc_midi_event(n2v, inc, 64, msg, 3)
        assert [x.note2voice[i] for i in range(128)] == n2v.tolist()
        assert [x.voice[v].note_inc for v in range(64)] == inc.tolist()
    assert L.voice_alloc(C.byref(x)) == orc.orc_voice_alloc(inc, 64)


def test_bsrr_word_matches_oracle(orc):
    from synth_tools_amd import synthetic
    L = sta.lib()
    for nb in (1, 2, 3, 12):
        sp, accu = synthetic.pdm_bank(nb, 9)
        a2 = accu.copy()
        for t in range(100):
            bits = np.zeros(1, np.uint32)
            orc.orc_pdm_tick(sp, accu, nb, 0, bits)
            assert L.smx_pdm_bsrr_word(int(bits[0]), nb) == orc.orc_pdm_bsrr(sp, a2, nb, 0)
    assert L.pdm_safe_setpoint(0x12345678) == 0x12345678     # identity, mod_pdm.c:101-107


def test_no_cpu_fallback():
    """Without a GPU the bank constructors must fail loudly, never compute on the CPU."""
    L = sta.lib()
    if L.smx_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(sta.SmxError, match="no HIP device"):
        sta.SawBank(64)
    with pytest.raises(sta.SmxError, match="no HIP device"):
        sta.PdmBank(2)
