"""ctypes loader for the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  The product (synth_tools_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "synth_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


class PwmBank(C.Structure):
    _fields_ = [("n", C.c_uint32),
                ("setpoint", C.c_void_p), ("pos0", C.c_void_p), ("vel0", C.c_void_p),
                ("pos1", C.c_void_p), ("vel1", C.c_void_p),
                ("s", C.c_void_p * 4), ("order", C.c_uint32),
                ("div_count", C.c_uint32), ("div_log", C.c_uint32), ("out_shift", C.c_uint32)]


class Pmeas(C.Structure):
    _fields_ = [("log_max", C.c_uint32), ("write", C.c_uint32), ("read", C.c_uint32),
                ("avg", C.c_uint32 * 2), ("num_pub", C.c_uint32 * 2),
                ("num", C.c_uint32), ("accu", C.c_uint32), ("last_cc", C.c_uint32),
                ("sub", C.c_uint32)]


class CprocNode(C.Structure):
    _fields_ = [("proc", C.c_uint32), ("inp", C.c_uint32), ("cond", C.c_uint32)]


class PolyBank(C.Structure):
    _fields_ = [("n", C.c_uint32)] + [(k, C.c_void_p) for k in
                ("inc", "phase", "y", "a", "level", "stage", "gate", "ar", "dr", "sl", "rr", "pan")]


def load():
    lib = C.CDLL(build())
    lib.orc_note_tab.argtypes = [_u32p]
    lib.orc_midi_tab.argtypes = [C.c_int]; lib.orc_midi_tab.restype = C.c_uint8
    lib.orc_note_to_inc.argtypes = [C.c_int]; lib.orc_note_to_inc.restype = C.c_uint32
    lib.orc_voice_alloc.argtypes = [_u32p, C.c_uint32]; lib.orc_voice_alloc.restype = C.c_int
    lib.orc_note_on.argtypes = [_i32p, _u32p, C.c_uint32, C.c_int]
    lib.orc_note_off.argtypes = [_i32p, _u32p, C.c_uint32, C.c_int]
    lib.orc_sum_tick_saw.argtypes = [_u32p, _u32p, C.c_uint32]; lib.orc_sum_tick_saw.restype = C.c_int32
    lib.orc_bus_to_float.argtypes = [C.c_int32]; lib.orc_bus_to_float.restype = C.c_float
    lib.orc_sum_tick_square.argtypes = [_u32p, _u32p, C.c_uint32]; lib.orc_sum_tick_square.restype = C.c_float
    lib.orc_synth_run.argtypes = [_u32p, _u32p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_midi_event.argtypes = [_i32p, _u32p, C.c_uint32, _u8p, C.c_size_t]
    lib.orc_pdm_tick.argtypes = [_u32p, _u32p, C.c_uint32, C.c_uint32, _u32p]
    lib.orc_pdm_run.argtypes = [_u32p, _u32p, C.c_uint32, C.c_void_p, C.c_uint32, _u32p]
    lib.orc_pdm_bsrr.argtypes = [_u32p, _u32p, C.c_uint32, C.c_uint32]; lib.orc_pdm_bsrr.restype = C.c_uint32
    lib.orc_pwm_update.argtypes = [_u32p, C.c_uint32]; lib.orc_pwm_update.restype = C.c_uint32
    lib.orc_pdm1_update.argtypes = [_u32p, C.c_uint32, C.c_uint32]; lib.orc_pdm1_update.restype = C.c_uint32
    for k in (2, 3, 4):
        f = getattr(lib, "orc_pdm%d_update" % k)
        f.argtypes = [_u32p, C.c_uint32, C.c_uint32, C.c_uint32]; f.restype = C.c_uint32
    lib.orc_pwm_bank_run.argtypes = [C.POINTER(PwmBank), C.c_void_p, C.c_uint32, C.c_void_p]
    lib.orc_pmeas_update.argtypes = [C.POINTER(Pmeas), C.c_uint32]
    lib.orc_osc_event.argtypes = [C.POINTER(Pmeas), C.c_uint32]
    lib.orc_pwmosc_run.argtypes = [_u32p, _u32p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.orc_osc_bank_events.argtypes = [C.POINTER(Pmeas), C.c_uint32, _u32p, C.c_void_p, C.c_uint32]
    lib.orc_bpm_to_hperiod.argtypes = [C.c_uint32, C.c_uint32]; lib.orc_bpm_to_hperiod.restype = C.c_uint32
    lib.orc_clock_run.argtypes = [_u32p, _i32p, _u32p, C.c_uint32, C.c_uint32, _u32p, _u32p]
    lib.orc_acc_update.argtypes = [_u32p, C.c_uint32]
    lib.orc_edge_update.argtypes = [_u32p, _u32p, C.c_uint32]
    lib.orc_cproc_run.argtypes = [C.POINTER(CprocNode), C.c_uint32, C.c_uint32, C.c_uint32, _u32p, _u32p,
                                  C.c_void_p, C.c_uint32, C.c_uint32, _u32p]
    lib.orc_poly_run.argtypes = [C.POINTER(PolyBank), _i32p, C.c_int]
    return lib


def load_ref_pdm():
    """The reference's own pdm.h, compiled into oracle/_ref (None if absent)."""
    so = os.path.join(_HERE, "_ref", "libref_pdm.so")
    if os.path.isdir("/root/reference"):
        build()
    if not os.path.exists(so):
        return None
    lib = C.CDLL(so)
    lib.ref_pdm1_update.argtypes = [_u32p, C.c_uint32, C.c_uint32]; lib.ref_pdm1_update.restype = C.c_uint32
    for k in (2, 3, 4):
        f = getattr(lib, "ref_pdm%d_update" % k)
        f.argtypes = [_u32p, C.c_uint32, C.c_uint32, C.c_uint32]; f.restype = C.c_uint32
    return lib


class RefVoice(C.Structure):                 # linux/synth.c:33-36
    _fields_ = [("note_inc", C.c_uint32), ("note_state", C.c_uint32)]


class RefSynth(C.Structure):                 # linux/synth.c:37-40 (1024 bytes)
    _fields_ = [("note2voice", C.c_int * 128), ("voice", RefVoice * 64)]

    def arrays(self):
        """(note2voice int32[128], inc u32[64], state u32[64]) copies."""
        v = np.frombuffer(bytes(self.voice), np.uint32).reshape(64, 2)
        return np.array(self.note2voice[:], np.int32), v[:, 0].copy(), v[:, 1].copy()


def load_ref_synth():
    """The SYNTH section of the reference's linux/synth.c (:27-208), compiled verbatim into
    oracle/_ref/libref_synth.so by oracle/Makefile (None if absent).  note_to_inc() writes a
    LOG line to stderr per call (linux/synth.c:123)."""
    so = os.path.join(_HERE, "_ref", "libref_synth.so")
    if os.path.isdir("/root/reference"):
        build()
    if not os.path.exists(so):
        return None
    lib = C.CDLL(so)
    P = C.POINTER(RefSynth)
    lib.synth_init.argtypes = [P]
    lib.synth_note_on.argtypes = [P, C.c_int]
    lib.synth_note_off.argtypes = [P, C.c_int]
    lib.synth_run.argtypes = [P, _f32p, C.c_int]
    lib.sum_tick_saw.argtypes = [P]; lib.sum_tick_saw.restype = C.c_float
    lib.sum_tick_square.argtypes = [P]; lib.sum_tick_square.restype = C.c_float
    lib.voice_alloc.argtypes = [P]; lib.voice_alloc.restype = C.c_int
    lib.note_to_inc.argtypes = [C.c_int]; lib.note_to_inc.restype = C.c_uint32
    lib.ref_midi_tab = (C.c_uint8 * 128).in_dll(lib, "midi_tab")
    return lib


class RefPmeas:
    """struct pmeas_state of the reference's pmeas.h (:10-28) as a raw buffer; field offsets
    come from the compiled reference itself (ref_pmeas_layout)."""
    FIELDS = ("log_max", "write", "read", "avg0", "num0", "avg1", "num1", "num", "accu", "last_cc")

    def __init__(self, lib, log_max):
        lay = np.zeros(11, np.uint32)
        lib.ref_pmeas_layout(lay)
        self._off = dict(zip(self.FIELDS, (int(x) for x in lay[:10])))
        self.buf = np.zeros(int(lay[10]) // 4 + 1, np.uint32)
        self.lib = lib
        self.set("log_max", log_max)

    def get(self, k):
        return int(self.buf[self._off[k] // 4])

    def set(self, k, v):
        self.buf[self._off[k] // 4] = v

    def update(self, cc):
        self.lib.ref_pmeas_update(self.buf.ctypes.data, int(cc) & 0xFFFFFFFF)

    def snapshot(self):
        return [self.get(k) for k in self.FIELDS]


def load_ref_pmeas():
    """pmeas_update of the reference's stm32f103/pmeas.h (:64-108), compiled verbatim into
    oracle/_ref/libref_pmeas.so (None if absent)."""
    so = os.path.join(_HERE, "_ref", "libref_pmeas.so")
    if os.path.isdir("/root/reference"):
        build()
    if not os.path.exists(so):
        return None
    lib = C.CDLL(so)
    lib.ref_pmeas_update.argtypes = [C.c_void_p, C.c_uint32]
    lib.ref_pmeas_layout.argtypes = [_u32p]
    return lib


def load_ref_pwmosc():
    """OSC_HARD_SYNC / pwm_phase / pwm_speed / pwm_update of the reference's stm32f103/mod_pdm.c (:159-175),
    compiled verbatim into oracle/_ref/libref_pwmosc.so (None if absent).  The reference keeps ONE oscillator
    in globals: ref_pwm_set() loads it."""
    so = os.path.join(_HERE, "_ref", "libref_pwmosc.so")
    if os.path.isdir("/root/reference"):
        build()
    if not os.path.exists(so):
        return None
    lib = C.CDLL(so)
    lib.ref_pwm_update.restype = C.c_uint32
    lib.ref_pwm_get_phase.restype = C.c_uint32
    lib.ref_pwm_get_speed.restype = C.c_uint32
    lib.ref_pwm_control_div.restype = C.c_uint32
    lib.ref_pwm_set.argtypes = [C.c_uint32, C.c_uint32]
    lib.ref_pwmosc_run.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


class quiet_stderr:
    """Silences fd 2 (the reference's note_to_inc LOGs every call, linux/synth.c:123)."""
    def __enter__(self):
        import sys
        sys.stderr.flush()
        self._saved = os.dup(2)
        self._null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self._null, 2)

    def __exit__(self, *a):
        os.dup2(self._saved, 2)
        os.close(self._saved)
        os.close(self._null)


# ---- convenience wrappers ---------------------------------------------------
def synth_run(lib, inc, state, nframes, want_vec=True):
    """Runs orc_synth_run in place on state; returns (bus int32[n], vec f32[n])."""
    bus = np.zeros(nframes, np.int32)
    vec = np.zeros(nframes, np.float32)
    lib.orc_synth_run(inc, state, len(inc), vec.ctypes.data if want_vec else None,
                      bus.ctypes.data, nframes)
    return bus, vec


def pdm_run(lib, setpoint, accu, nticks, dither=None):
    words = (len(setpoint) + 31) // 32
    bits = np.zeros(nticks * words, np.uint32)
    d = None if dither is None else np.ascontiguousarray(dither, np.uint32)
    lib.orc_pdm_run(setpoint, accu, len(setpoint), None if d is None else d.ctypes.data, nticks, bits)
    return bits.reshape(nticks, words)
