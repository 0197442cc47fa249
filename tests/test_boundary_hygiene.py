"""CPU: boundary hygiene.  The header is valid ISO C; the product never touches the oracle;
there is no CPU compute fallback hiding in the package; argument errors come back as codes."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import synth_tools_amd as sta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "synth_mi355x.h"\n'
                   'int main(void) { struct synth s; struct smx_cproc_node n = {SMX_PROC_ACC, SMX_CPROC_INPUT(0), 1u};\n'
                   '  (void)s; (void)n; return sizeof(struct synth) == 1024 ? 0 : 1; }\n')
    for std in ("c99", "c11"):
        subprocess.check_call(["gcc", "-std=" + std, "-Wall", "-Wextra", "-Werror", "-pedantic",
                               "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)])
    # and from C++
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-x", "c++", "-fsyntax-only", str(src)])


def test_product_never_references_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load oracle/."""
    offenders = []
    for base in ("synth_tools_amd", "host", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dp or "__pycache__" in dp:
                continue
            for f in files:
                if f.endswith((".so", ".o", ".elf", ".pyc")):
                    continue
                text = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"\boracle\b|liboracle|orc_[a-z]", text):
                    offenders.append(os.path.join(dp, f))
    # DESIGN-level mentions in comments are allowed only where they name the definition's home
    allowed = {os.path.join(ROOT, "synth_tools_amd", "csrc", "poly_bank.hip"),
               os.path.join(ROOT, "synth_tools_amd", "__init__.py"),
               os.path.join(ROOT, "synth_tools_amd", "synthetic.py"),
               os.path.join(ROOT, "include", "synth_mi355x.h")}
    assert set(offenders) <= allowed, offenders
    for f in allowed & set(offenders):
        text = open(f).read()
        assert "import oracle" not in text and "dlopen" not in text and "liboracle" not in text
    bench = open(os.path.join(ROOT, "bench.py")).read()
    # bench.py loads the oracle only inside its cpu_baseline functions
    import ast
    tree = ast.parse(bench)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            uses = any(isinstance(n, ast.Import) and any(a.name == "oracle" for a in n.names) for n in ast.walk(node))
            assert uses == node.name.startswith("cpu_baseline"), node.name
    assert not any(isinstance(n, ast.Import) and any(a.name == "oracle" for a in n.names) for n in tree.body)


def test_no_numpy_compute_in_the_package():
    """The Python package is a binding: no synthesis arithmetic in it (the only numpy math is
    input generation in synthetic.py)."""
    text = open(os.path.join(ROOT, "synth_tools_amd", "__init__.py")).read()
    for needle in (">> 4", "np.cumsum", "np.add.reduce", "note_state +=", "for v in range"):
        assert needle not in text


def test_argument_errors_are_codes_not_crashes():
    L = sta.lib()
    assert L.smx_bank_load(None, None, None) == -1
    assert L.smx_bank_run(None, None, None, 64) == -1
    assert L.smx_bank_note_on(None, 60) == -1
    assert L.smx_pdm_tick_n(None, 10, None, None) == -1
    assert L.smx_pwm_tick_n(None, 10, None, None) == -1
    assert L.smx_fw_handle_tag_u32(None, None, 0, None, 0) == -1
    assert L.smx_fw_handle_packet(None, None, 0) == -1
    assert L.smx_bank_voices(None) == 0 and L.smx_fw_running(None) == 0
    L.smx_bank_destroy(None); L.smx_pdm_destroy(None); L.smx_pwm_destroy(None)
    L.smx_poly_destroy(None); L.smx_osc_destroy(None); L.smx_cproc_destroy(None)
    L.smx_clock_destroy(None); L.smx_fw_destroy(None)
    assert L.smx_bank_create(0, 0) is None and b"n_voices" in L.smx_last_error()
    assert L.smx_pwm_create(4, 7, 0) is None and b"order" in L.smx_last_error()
    assert L.smx_pdm_bsrr_word(3, 0) == 0 and L.smx_pdm_bsrr_word(3, 13) == 0
    assert L.smx_bpm_to_hperiod(48000, 0) == 0
