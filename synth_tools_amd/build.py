"""Build libsynth_mi355x.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsynth_mi355x.so")
SOURCES = ["saw_bank.hip", "pdm_bank.hip", "poly_bank.hip", "pwm_bank.hip", "osc_bank.hip", "cproc_bank.hip",
           "abi_core.cpp", "abi_saw.cpp", "abi_pdm.cpp", "abi_pwm.cpp", "abi_poly.cpp", "abi_osc.cpp",
           "abi_cproc.cpp", "abi_fw.cpp"]
HEADERS = ["smx_common.h", "abi_internal.h", "exports.map", os.path.join("..", "..", "include", "synth_mi355x.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-result"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, "build", s + ".o")
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + \
          ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib",
           "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
    subprocess.check_call(cmd)
    return LIB


def build_hosts():
    """Plain-C host programs over the library (host/*.elf)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(os.path.dirname(HERE), "host")])


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    build_hosts()
