"""synth_tools_amd -- ctypes binding of libsynth_mi355x.so (include/synth_mi355x.h).

This is a thin test/bench driver over the C-ABI; the product is the shared
library.  There is no Python or CPU implementation of any compute path here:
if the HIP library is missing, import fails; if no GPU is visible, every
compute call raises.

Load order: a process that also uses PyTorch-ROCm must `import torch` BEFORE this package
(as bench.py does).  The torch wheel bundles its own HIP runtime; loaded second, it leaves
this library's runtime without a device ("no HIP device").
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMX_LIB") or os.path.join(_HERE, "libsynth_mi355x.so")   # SMX_LIB: A/B builds (tools/)

_u32 = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32 = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

UNIQUE_ID_BYTES = 128


class SmxError(RuntimeError):
    pass


class Voice(C.Structure):            # linux/synth.c:31-34
    _fields_ = [("note_inc", C.c_uint32), ("note_state", C.c_uint32)]


class Synth(C.Structure):            # linux/synth.c:35-38
    _fields_ = [("note2voice", C.c_int * 128), ("voice", Voice * 64)]


POLY_FIELDS = ("inc", "phase", "y", "a", "level", "stage", "gate", "ar", "dr", "sl", "rr", "pan")
POLY_FLOAT = ("y", "a")


class PolyArrays(C.Structure):       # struct smx_poly_arrays
    _fields_ = [(k, C.c_void_p) for k in POLY_FIELDS]


PMEAS_FIELDS = ("write", "avg0", "avg1", "num0", "num1", "num", "accu", "last_cc", "sub")


class PmeasArrays(C.Structure):      # struct smx_pmeas_arrays
    _fields_ = [(k, C.c_void_p) for k in PMEAS_FIELDS]


class CprocNode(C.Structure):        # struct smx_cproc_node
    _fields_ = [("proc", C.c_uint32), ("inp", C.c_uint32), ("cond", C.c_uint32)]


PROC_ACC, PROC_EDGE, PROC_GPIN, PROC_GPOUT = 1, 2, 3, 4


def cproc_input(k):
    return 0x80000000 | k


class PwmArrays(C.Structure):        # struct smx_pwm_arrays
    _fields_ = [(k, C.c_void_p) for k in ("setpoint", "pos0", "vel0", "pos1", "vel1")] + [("s", C.c_void_p * 4)]


# Every symbol include/synth_mi355x.h declares: (name, restype, argtypes)
_P = C.c_void_p
ABI = [
    ("smx_last_error", C.c_char_p, []),
    ("smx_device_count", C.c_int, []),
    ("smx_version", C.c_int, []),
    ("smx_device_synchronize", C.c_int, [C.c_int]),
    ("synth_note_on", None, [C.POINTER(Synth), C.c_int]),
    ("synth_note_off", None, [C.POINTER(Synth), C.c_int]),
    ("synth_init", None, [C.POINTER(Synth)]),
    ("synth_run", None, [C.POINTER(Synth), _f32, C.c_int]),
    ("sum_tick_saw", C.c_float, [C.POINTER(Synth)]),
    ("sum_tick_square", C.c_float, [C.POINTER(Synth)]),
    ("note_to_inc", C.c_uint32, [C.c_int]),
    ("voice_alloc", C.c_int, [C.POINTER(Synth)]),
    ("synth_midi_event", None, [C.POINTER(Synth), _u8, C.c_size_t]),
    ("smx_bank_create", _P, [C.c_uint32, C.c_int]),
    ("smx_bank_destroy", None, [_P]),
    ("smx_bank_voices", C.c_uint32, [_P]),
    ("smx_bank_load", C.c_int, [_P, _P, _P]),
    ("smx_bank_read", C.c_int, [_P, _P, _P]),
    ("smx_bank_load_run", C.c_int, [_P, _P, _P, _P, _P, C.c_int]),
    ("smx_bank_note_on", C.c_int, [_P, C.c_int]),
    ("smx_bank_note_off", C.c_int, [_P, C.c_int]),
    ("smx_bank_midi_event", C.c_int, [_P, _u8, C.c_size_t]),
    ("smx_bank_midi_events", C.c_int, [_P, _u8, C.c_size_t]),
    ("smx_bank_run", C.c_int, [_P, _P, _P, C.c_int]),
    ("smx_bank_set_block_mode", C.c_int, [_P, C.c_int]),
    ("smx_bank_set_block_form", C.c_int, [_P, C.c_int]),
    ("smx_bank_next_block_form", C.c_int, [_P]),
    ("smx_bank_run_async", C.c_int, [_P, C.c_int]),
    ("smx_bank_bus_dev", _P, [_P]),
    ("smx_bank_sync", C.c_int, [_P]),
    ("smx_bank_run_square", C.c_int, [_P, _P, C.c_int]),
    ("smx_bank_timer_start", C.c_int, [_P]),
    ("smx_bank_timer_stop", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("smx_comm_unique_id", C.c_int, [_u8]),
    ("smx_bank_comm_init", C.c_int, [_P, C.c_int, C.c_int, _u8]),
    ("smx_bank_allreduce_async", C.c_int, [_P, C.c_int]),
    ("smx_bank_shard", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("smx_bank_comm_ranks", C.c_int, [_P]),
    ("smx_bank_set_comm_group", C.c_int, [_P, C.c_int]),
    ("smx_bank_comm_stats", C.c_int, [_P, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    ("smx_bank_comm_probe", C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("smx_bank_fetch", C.c_int, [_P, _P, _P, C.c_int]),
    ("smx_pdm_create", _P, [C.c_uint32, C.c_int]),
    ("smx_pdm_destroy", None, [_P]),
    ("smx_pdm_init", C.c_int, [_P]),
    ("smx_pdm_set_setpoint", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("pdm_safe_setpoint", C.c_uint32, [C.c_uint32]),
    ("smx_pdm_load", C.c_int, [_P, _P, _P]),
    ("smx_pdm_read", C.c_int, [_P, _P, _P]),
    ("smx_pdm_tick_n", C.c_int, [_P, C.c_uint32, _P, _P]),
    ("smx_pdm_tick_n_async", C.c_int, [_P, C.c_uint32, C.c_int]),
    ("smx_pdm_tick_n_streams", C.c_int, [_P, C.c_uint32, _P, _P]),
    ("smx_pdm_tick_n_streams_async", C.c_int, [_P, C.c_uint32, C.c_int]),
    ("smx_pdm_bits_dev", _P, [_P]),
    ("smx_pdm_dither_dev", _P, [_P, C.c_uint32]),
    ("smx_pdm_sync", C.c_int, [_P]),
    ("smx_pdm_timer_start", C.c_int, [_P]),
    ("smx_pdm_timer_stop", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("smx_pdm_bsrr_word", C.c_uint32, [C.c_uint32, C.c_uint32]),
    ("smx_poly_create", _P, [C.c_uint32, C.c_int]),
    ("smx_poly_destroy", None, [_P]),
    ("smx_poly_load", C.c_int, [_P, C.POINTER(PolyArrays)]),
    ("smx_poly_read", C.c_int, [_P, C.POINTER(PolyArrays)]),
    ("smx_poly_run", C.c_int, [_P, _P, _P, C.c_int]),
    ("smx_poly_run_async", C.c_int, [_P, C.c_int]),
    ("smx_poly_sync", C.c_int, [_P]),
    ("smx_poly_timer_start", C.c_int, [_P]),
    ("smx_poly_timer_stop", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("smx_pwm_create", _P, [C.c_uint32, C.c_int, C.c_int]),
    ("smx_pwm_destroy", None, [_P]),
    ("smx_pwm_config", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("smx_pwm_init", C.c_int, [_P]),
    ("smx_pwm_set_setpoint", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("smx_pwm_load", C.c_int, [_P, C.POINTER(PwmArrays)]),
    ("smx_pwm_read", C.c_int, [_P, C.POINTER(PwmArrays)]),
    ("smx_pwm_set_div_count", C.c_int, [_P, C.c_uint32]),
    ("smx_pwm_div_count", C.c_uint32, [_P]),
    ("smx_pwm_controlrate", C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("smx_pwm_controlrate_poll", C.c_int, [_P]),
    ("smx_pwm_tick_n", C.c_int, [_P, C.c_uint32, _P, _P]),
    ("smx_pwm_tick_n_async", C.c_int, [_P, C.c_uint32, C.c_int]),
    ("smx_pwm_dither_dev", _P, [_P, C.c_uint32]),
    ("smx_pwm_sync", C.c_int, [_P]),
    ("smx_pwm_timer_start", C.c_int, [_P]),
    ("smx_pwm_timer_stop", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("smx_osc_create", _P, [C.c_uint32, C.c_int]),
    ("smx_osc_destroy", None, [_P]),
    ("smx_osc_set_log_max", C.c_int, [_P, C.c_uint32]),
    ("smx_osc_load_pwm", C.c_int, [_P, _P, _P]),
    ("smx_osc_read_pwm", C.c_int, [_P, _P, _P]),
    ("smx_osc_tick_n", C.c_int, [_P, C.c_uint32, _P, _P]),
    ("smx_osc_events", C.c_int, [_P, C.c_uint32, _P, _P]),
    ("smx_osc_load_pmeas", C.c_int, [_P, C.POINTER(PmeasArrays)]),
    ("smx_osc_read_pmeas", C.c_int, [_P, C.POINTER(PmeasArrays)]),
    ("smx_bpm_to_hperiod", C.c_uint32, [C.c_uint32, C.c_uint32]),
    ("smx_clock_create", _P, [C.c_uint32, C.c_int]),
    ("smx_clock_destroy", None, [_P]),
    ("smx_clock_load", C.c_int, [_P, _P, _P, _P]),
    ("smx_clock_read", C.c_int, [_P, _P, _P, _P]),
    ("smx_clock_run", C.c_int, [_P, C.c_uint32, _P, _P]),
    ("smx_modpdm_create", _P, [C.c_uint32, C.c_uint32, C.c_int]),
    ("smx_modpdm_destroy", None, [_P]),
    ("smx_modpdm_pdm", _P, [_P]),
    ("smx_modpdm_osc", _P, [_P]),
    ("smx_modpdm_tick_n", C.c_int, [_P, C.c_uint32, _P, _P, _P, _P, C.POINTER(C.c_uint32)]),
    ("smx_modpdm_control_div_count", C.c_uint32, [_P]),
    ("smx_modpdm_controlrate", C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("smx_cproc_create", _P, [C.c_uint32, C.POINTER(CprocNode), C.c_uint32, C.c_uint32, C.c_int]),
    ("smx_cproc_destroy", None, [_P]),
    ("smx_cproc_tick_n", C.c_int, [_P, C.c_uint32, _P, _P, C.c_uint32, _P]),
    ("smx_cproc_read_state", C.c_int, [_P, _P]),
    ("smx_cproc_load_state", C.c_int, [_P, _P]),
    ("smx_patch_create", _P, [C.c_uint32, C.c_uint32, C.c_int]),
    ("smx_patch_destroy", None, [_P]),
    ("smx_patch_apply", C.c_int, [_P, C.c_uint32, _P, C.c_uint32, C.c_uint32]),
    ("smx_patch_count", C.c_uint32, [_P]),
    ("smx_patch_reset", C.c_int, [_P]),
    ("smx_patch_tick", C.c_int, [_P, C.c_uint32, _P, C.c_uint32, _P]),
    ("smx_patch_state_get", C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    ("smx_patch_state_set", C.c_int, [_P, C.c_uint32, C.c_uint32, _P]),
    ("smx_fw_create", _P, [C.c_uint32, C.c_uint32, C.c_int]),
    ("smx_fw_destroy", None, [_P]),
    ("smx_fw_pwm", _P, [_P]),
    ("smx_fw_osc", _P, [_P]),
    ("smx_fw_handle_tag_u32", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32]),
    ("smx_fw_handle_packet", C.c_int, [_P, _P, C.c_uint32]),
    ("smx_fw_running", C.c_int, [_P]),
    ("smx_fw_parameter", C.c_uint32, [_P, C.c_uint32]),
    ("smx_fw_tick_n", C.c_int, [_P, C.c_uint32, _P, _P]),
    ("smx_fw_poll", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _P, C.c_uint32,
                              C.POINTER(C.c_uint32)]),
]
ABI_DATA = ["midi_tab"]

_lib = None


def lib():
    """Load the HIP library (never a fallback: raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SmxError("%s is missing: run `python -m synth_tools_amd.build` "
                           "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in ABI:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _check(rv, what):
    if rv != 0:
        raise SmxError("%s failed (%d): %s" % (what, rv, lib().smx_last_error().decode()))


def _ptr(a):
    return None if a is None else a.ctypes.data


class SawBank:
    """N-voice saw bank (linux/synth.c:27-208 widened), state resident in HBM."""

    def __init__(self, n_voices, device=0):
        self._h = lib().smx_bank_create(n_voices, device)
        if not self._h:
            raise SmxError("smx_bank_create: " + lib().smx_last_error().decode())
        self.n = n_voices

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.smx_bank_destroy(self._h)
        self._h = None

    __del__ = close

    def load(self, inc=None, state=None):
        inc = None if inc is None else np.ascontiguousarray(inc, np.uint32)
        state = None if state is None else np.ascontiguousarray(state, np.uint32)
        for a in (inc, state):
            assert a is None or a.shape == (self.n,)
        _check(lib().smx_bank_load(self._h, _ptr(inc), _ptr(state)), "smx_bank_load")

    def read(self):
        inc = np.empty(self.n, np.uint32)
        state = np.empty(self.n, np.uint32)
        _check(lib().smx_bank_read(self._h, _ptr(inc), _ptr(state)), "smx_bank_read")
        return inc, state

    def note_on(self, note):
        _check(lib().smx_bank_note_on(self._h, note), "smx_bank_note_on")

    def note_off(self, note):
        _check(lib().smx_bank_note_off(self._h, note), "smx_bank_note_off")

    def set_block_mode(self, pipelined):
        _check(lib().smx_bank_set_block_mode(self._h, 1 if pipelined else 0), "smx_bank_set_block_mode")

    def set_block_form(self, form):
        """0 auto, 1 stepping, 2 wrap events (long blocks of big banks; same bits either way)."""
        _check(lib().smx_bank_set_block_form(self._h, form), "smx_bank_set_block_form")

    def next_block_form(self):
        return lib().smx_bank_next_block_form(self._h)

    def midi_event(self, msg):
        m = np.ascontiguousarray(msg, np.uint8)
        _check(lib().smx_bank_midi_event(self._h, m, len(m)), "smx_bank_midi_event")

    def midi_events(self, msgs):
        """All 3-byte MIDI events of a block at once: msgs is (n, 3) uint8."""
        m = np.ascontiguousarray(msgs, np.uint8).reshape(-1, 3)
        _check(lib().smx_bank_midi_events(self._h, m.reshape(-1), len(m)), "smx_bank_midi_events")

    def run(self, n):
        """synth_run for n frames -> (bus int32[n], vec float32[n])."""
        vec = np.empty(n, np.float32)
        bus = np.empty(n, np.int32)
        _check(lib().smx_bank_run(self._h, _ptr(vec), _ptr(bus), n), "smx_bank_run")
        return bus, vec

    def load_run(self, inc, state, n):
        """smx_bank_load_run: new increments and phases, then synth_run for n frames, one synchronisation."""
        inc = np.ascontiguousarray(inc, np.uint32)
        state = np.ascontiguousarray(state, np.uint32)
        assert inc.shape == (self.n,) and state.shape == (self.n,)
        vec = np.empty(n, np.float32)
        bus = np.empty(n, np.int32)
        _check(lib().smx_bank_load_run(self._h, _ptr(inc), _ptr(state), _ptr(vec), _ptr(bus), n), "smx_bank_load_run")
        return bus, vec

    def run_square(self, n):
        vec = np.empty(n, np.float32)
        _check(lib().smx_bank_run_square(self._h, _ptr(vec), n), "smx_bank_run_square")
        return vec

    def run_async(self, n):
        _check(lib().smx_bank_run_async(self._h, n), "smx_bank_run_async")

    def sync(self):
        _check(lib().smx_bank_sync(self._h), "smx_bank_sync")

    def fetch(self, n):
        vec = np.empty(n, np.float32)
        bus = np.empty(n, np.int32)
        _check(lib().smx_bank_fetch(self._h, _ptr(vec), _ptr(bus), n), "smx_bank_fetch")
        return bus, vec

    def timer_start(self):
        _check(lib().smx_bank_timer_start(self._h), "smx_bank_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        _check(lib().smx_bank_timer_stop(self._h, C.byref(ms)), "smx_bank_timer_stop")
        return ms.value

    def comm_init(self, rank, nranks, unique_id):
        uid = np.ascontiguousarray(unique_id, np.uint8)
        assert uid.shape == (UNIQUE_ID_BYTES,)
        _check(lib().smx_bank_comm_init(self._h, rank, nranks, uid), "smx_bank_comm_init")

    def allreduce_async(self, n):
        _check(lib().smx_bank_allreduce_async(self._h, n), "smx_bank_allreduce_async")

    def shard(self, first_voice, total_voices):
        _check(lib().smx_bank_shard(self._h, first_voice, total_voices), "smx_bank_shard")

    def comm_ranks(self):
        return lib().smx_bank_comm_ranks(self._h)

    def set_comm_group(self, blocks):
        _check(lib().smx_bank_set_comm_group(self._h, blocks), "smx_bank_set_comm_group")

    def comm_probe(self, n_words, reps=50):
        """(us per all-reduce when each is waited for, us per all-reduce queued back to back); collective."""
        a, b = C.c_float(), C.c_float()
        _check(lib().smx_bank_comm_probe(self._h, n_words, reps, C.byref(a), C.byref(b)), "smx_bank_comm_probe")
        return a.value, b.value

    def comm_stats(self):
        """(collectives issued, block sums they carried)."""
        a, b = C.c_ulonglong(), C.c_ulonglong()
        _check(lib().smx_bank_comm_stats(self._h, C.byref(a), C.byref(b)), "smx_bank_comm_stats")
        return a.value, b.value


def comm_unique_id():
    uid = np.zeros(UNIQUE_ID_BYTES, np.uint8)
    _check(lib().smx_comm_unique_id(uid), "smx_comm_unique_id")
    return uid


class PdmBank:
    """N-channel carry-out PDM bank (stm32f103/mod_pdm.c:198-286)."""

    def __init__(self, n_channels, device=0):
        self._h = lib().smx_pdm_create(n_channels, device)
        if not self._h:
            raise SmxError("smx_pdm_create: " + lib().smx_last_error().decode())
        self.n = n_channels
        self.words = (n_channels + 31) // 32

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.smx_pdm_destroy(self._h)
        self._h = None

    __del__ = close

    def init(self):
        _check(lib().smx_pdm_init(self._h), "smx_pdm_init")

    def set_setpoint(self, chan, val):
        return lib().smx_pdm_set_setpoint(self._h, chan, val)

    def load(self, setpoint=None, accu=None):
        setpoint = None if setpoint is None else np.ascontiguousarray(setpoint, np.uint32)
        accu = None if accu is None else np.ascontiguousarray(accu, np.uint32)
        _check(lib().smx_pdm_load(self._h, _ptr(setpoint), _ptr(accu)), "smx_pdm_load")

    def read(self):
        sp = np.empty(self.n, np.uint32)
        ac = np.empty(self.n, np.uint32)
        _check(lib().smx_pdm_read(self._h, _ptr(sp), _ptr(ac)), "smx_pdm_read")
        return sp, ac

    def tick_n(self, n_ticks, dither=None, want_bits=True):
        d = None if dither is None else np.ascontiguousarray(dither, np.uint32)
        bits = np.empty((n_ticks, self.words), np.uint32) if want_bits else None
        _check(lib().smx_pdm_tick_n(self._h, n_ticks, _ptr(d), _ptr(bits)), "smx_pdm_tick_n")
        return bits

    def tick_n_async(self, n_ticks, with_dither=False):
        _check(lib().smx_pdm_tick_n_async(self._h, n_ticks, int(with_dither)), "smx_pdm_tick_n_async")

    def tick_n_streams(self, n_ticks, dither=None, want=True):
        """-> uint32[n_ticks/32, n]: bit j of word [k, c] = pulse of channel c at tick 32k+j."""
        d = None if dither is None else np.ascontiguousarray(dither, np.uint32)
        out = np.empty((n_ticks // 32, self.n), np.uint32) if want else None
        _check(lib().smx_pdm_tick_n_streams(self._h, n_ticks, _ptr(d), _ptr(out)), "smx_pdm_tick_n_streams")
        return out

    def tick_n_streams_async(self, n_ticks, with_dither=False):
        _check(lib().smx_pdm_tick_n_streams_async(self._h, n_ticks, int(with_dither)), "smx_pdm_tick_n_streams_async")

    def sync(self):
        _check(lib().smx_pdm_sync(self._h), "smx_pdm_sync")

    def timer_start(self):
        _check(lib().smx_pdm_timer_start(self._h), "smx_pdm_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        _check(lib().smx_pdm_timer_stop(self._h, C.byref(ms)), "smx_pdm_timer_stop")
        return ms.value


class PolyBank:
    """Poly voice bank: saw -> 1-pole LPF -> ADSR -> stereo int32 bus (build-defined
    extension, BASELINE config 4; definition: oracle orc_poly_run / DESIGN.md)."""

    def __init__(self, n_voices, device=0):
        self._h = lib().smx_poly_create(n_voices, device)
        if not self._h:
            raise SmxError("smx_poly_create: " + lib().smx_last_error().decode())
        self.n = n_voices

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.smx_poly_destroy(self._h)
        self._h = None

    __del__ = close

    def _arrays(self, d):
        keep = {}
        for k in POLY_FIELDS:
            if k in d and d[k] is not None:
                keep[k] = np.ascontiguousarray(d[k], np.float32 if k in POLY_FLOAT else np.uint32)
                assert keep[k].shape == (self.n,)
        return PolyArrays(**{k: v.ctypes.data for k, v in keep.items()}), keep

    def load(self, **arrays):
        st, keep = self._arrays(arrays)
        _check(lib().smx_poly_load(self._h, C.byref(st)), "smx_poly_load")

    def read(self, fields=POLY_FIELDS):
        out = {k: np.empty(self.n, np.float32 if k in POLY_FLOAT else np.uint32) for k in fields}
        st = PolyArrays(**{k: v.ctypes.data for k, v in out.items()})
        _check(lib().smx_poly_read(self._h, C.byref(st)), "smx_poly_read")
        return out

    def run(self, n):
        """-> (bus_lr int32[n,2], vec_lr float32[n,2])"""
        vec = np.empty((n, 2), np.float32)
        bus = np.empty((n, 2), np.int32)
        _check(lib().smx_poly_run(self._h, _ptr(vec), _ptr(bus), n), "smx_poly_run")
        return bus, vec

    def run_async(self, n):
        _check(lib().smx_poly_run_async(self._h, n), "smx_poly_run_async")

    def sync(self):
        _check(lib().smx_poly_sync(self._h), "smx_poly_sync")

    def timer_start(self):
        _check(lib().smx_poly_timer_start(self._h), "smx_poly_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        _check(lib().smx_poly_timer_stop(self._h, C.byref(ms)), "smx_poly_timer_stop")
        return ms.value


PWM_FIELDS = ("setpoint", "pos0", "vel0", "pos1", "vel1", "s1", "s2", "s3", "s4")


class PwmBank:
    """N-channel noise-shaped PWM bank with control-rate glide
    (stm32f103/mod_pdm_pwm.c, pdm.h, mod_controlrate.c)."""

    def __init__(self, n_channels, order=2, device=0, control_div_log=12, out_shift=24):
        self._h = lib().smx_pwm_create(n_channels, order, device)
        if not self._h:
            raise SmxError("smx_pwm_create: " + lib().smx_last_error().decode())
        self.n, self.order = n_channels, order
        _check(lib().smx_pwm_config(self._h, control_div_log, out_shift), "smx_pwm_config")

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.smx_pwm_destroy(self._h)
        self._h = None

    __del__ = close

    def _struct(self, d):
        s = (C.c_void_p * 4)(*[d["s%d" % k].ctypes.data if d.get("s%d" % k) is not None else None
                               for k in (1, 2, 3, 4)])
        return PwmArrays(s=s, **{k: (d[k].ctypes.data if d.get(k) is not None else None)
                                 for k in ("setpoint", "pos0", "vel0", "pos1", "vel1")})

    def init(self):
        _check(lib().smx_pwm_init(self._h), "smx_pwm_init")

    def set_setpoint(self, chan, val):
        return lib().smx_pwm_set_setpoint(self._h, chan, val)

    def load(self, **arrays):
        keep = {k: np.ascontiguousarray(v).view(np.uint32) for k, v in arrays.items() if v is not None}
        for v in keep.values():
            assert v.shape == (self.n,)
        st = self._struct(keep)
        _check(lib().smx_pwm_load(self._h, C.byref(st)), "smx_pwm_load")

    def read(self):
        out = {k: np.empty(self.n, np.uint32) for k in PWM_FIELDS[:5 + self.order]}
        st = self._struct(out)
        _check(lib().smx_pwm_read(self._h, C.byref(st)), "smx_pwm_read")
        return out

    @property
    def div_count(self):
        return lib().smx_pwm_div_count(self._h)

    @div_count.setter
    def div_count(self, c):
        _check(lib().smx_pwm_set_div_count(self._h, c), "smx_pwm_set_div_count")

    def controlrate(self):
        """struct controlrate (mod_controlrate.c:21-26): (isr_count, beat_pulse, beat_handled)."""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().smx_pwm_controlrate(self._h, C.byref(a), C.byref(b), C.byref(c)), "smx_pwm_controlrate")
        return a.value, b.value, c.value

    def controlrate_poll(self):
        return lib().smx_pwm_controlrate_poll(self._h)

    def tick_n(self, n_ticks, dither=None, want_duty=True):
        d = None if dither is None else np.ascontiguousarray(dither, np.uint32)
        duty = np.empty((n_ticks, self.n), np.uint8) if want_duty else None
        _check(lib().smx_pwm_tick_n(self._h, n_ticks, _ptr(d), _ptr(duty)), "smx_pwm_tick_n")
        return duty

    def tick_n_async(self, n_ticks, with_dither=False):
        _check(lib().smx_pwm_tick_n_async(self._h, n_ticks, int(with_dither)), "smx_pwm_tick_n_async")

    def sync(self):
        _check(lib().smx_pwm_sync(self._h), "smx_pwm_sync")

    def timer_start(self):
        _check(lib().smx_pwm_timer_start(self._h), "smx_pwm_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        _check(lib().smx_pwm_timer_stop(self._h, C.byref(ms)), "smx_pwm_timer_stop")
        return ms.value


class OscBank:
    """Oscillator bank: hard-synced PWM phase accumulators (mod_pdm.c:159-175) and the
    osc event ISR with period measurement (mod_osc.c:47-74, pmeas.h:64-100)."""

    def __init__(self, n, device=0):
        self._h = lib().smx_osc_create(n, device)
        if not self._h:
            raise SmxError("smx_osc_create: " + lib().smx_last_error().decode())
        self.n = n
        self.words = (n + 31) // 32

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.smx_osc_destroy(self._h)
        self._h = None

    __del__ = close

    def set_log_max(self, log_max):
        return lib().smx_osc_set_log_max(self._h, log_max)

    def load_pwm(self, phase=None, speed=None):
        phase = None if phase is None else np.ascontiguousarray(phase, np.uint32)
        speed = None if speed is None else np.ascontiguousarray(speed, np.uint32)
        _check(lib().smx_osc_load_pwm(self._h, _ptr(phase), _ptr(speed)), "smx_osc_load_pwm")

    def read_pwm(self):
        ph = np.empty(self.n, np.uint32)
        sp = np.empty(self.n, np.uint32)
        _check(lib().smx_osc_read_pwm(self._h, _ptr(ph), _ptr(sp)), "smx_osc_read_pwm")
        return ph, sp

    def tick_n(self, n_ticks, sync_bits=None, want_duty=True):
        sb = None if sync_bits is None else np.ascontiguousarray(sync_bits, np.uint32)
        assert sb is None or sb.size == n_ticks * self.words
        duty = np.empty((n_ticks, self.n), np.uint8) if want_duty else None
        _check(lib().smx_osc_tick_n(self._h, n_ticks, _ptr(sb), _ptr(duty)), "smx_osc_tick_n")
        return duty

    def events(self, cc, valid_bits=None):
        cc = np.ascontiguousarray(cc, np.uint32)
        ne = cc.size // self.n
        vb = None if valid_bits is None else np.ascontiguousarray(valid_bits, np.uint32)
        assert cc.size == ne * self.n and (vb is None or vb.size == ne * self.words)
        _check(lib().smx_osc_events(self._h, ne, _ptr(cc), _ptr(vb)), "smx_osc_events")

    def load_pmeas(self, **arrays):
        keep = {k: np.ascontiguousarray(v, np.uint32) for k, v in arrays.items()}
        st = PmeasArrays(**{k: v.ctypes.data for k, v in keep.items()})
        _check(lib().smx_osc_load_pmeas(self._h, C.byref(st)), "smx_osc_load_pmeas")

    def read_pmeas(self):
        out = {k: np.empty(self.n, np.uint32) for k in PMEAS_FIELDS}
        st = PmeasArrays(**{k: v.ctypes.data for k, v in out.items()})
        _check(lib().smx_osc_read_pmeas(self._h, C.byref(st)), "smx_osc_read_pmeas")
        return out


PATCH_BAD_REF, PATCH_BAD_NODE, PATCH_ALLOC_FAIL = -11, -12, -13


class ModPdm:
    """mod_pdm.c as one module (its timer ISR, stm32f103/mod_pdm.c:177-194): a PDM bank and an oscillator bank ticking
    in lockstep plus the control-rate divider.  `pdm` / `osc` are views of the module's own banks (borrowed handles)."""

    def __init__(self, n_channels, n_osc=1, device=0):
        self._h = lib().smx_modpdm_create(n_channels, n_osc, device)
        if not self._h:
            raise SmxError("smx_modpdm_create: " + lib().smx_last_error().decode())
        self.pdm = PdmBank.__new__(PdmBank)
        self.pdm._h, self.pdm.n, self.pdm.words = lib().smx_modpdm_pdm(self._h), n_channels, (n_channels + 31) // 32
        self.pdm.close = lambda: None                      # the module owns it
        self.osc = OscBank.__new__(OscBank)
        self.osc._h, self.osc.n, self.osc.words = lib().smx_modpdm_osc(self._h), n_osc, (n_osc + 31) // 32
        self.osc.close = lambda: None

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            self.pdm._h = self.osc._h = None
            _lib.smx_modpdm_destroy(self._h)
        self._h = None

    __del__ = close

    def tick_n(self, n_ticks, dither=None, sync_bits=None):
        """-> (pulse bits uint32[n_ticks, words], duty uint8[n_ticks, n_osc], control_trigger() calls)"""
        d = None if dither is None else np.ascontiguousarray(dither, np.uint32)
        sb = None if sync_bits is None else np.ascontiguousarray(sync_bits, np.uint32)
        bits = np.empty((n_ticks, self.pdm.words), np.uint32)
        duty = np.empty((n_ticks, self.osc.n), np.uint8)
        trig = C.c_uint32()
        _check(lib().smx_modpdm_tick_n(self._h, n_ticks, _ptr(d), _ptr(sb), _ptr(bits), _ptr(duty), C.byref(trig)),
               "smx_modpdm_tick_n")
        return bits, duty, trig.value

    @property
    def control_div_count(self):
        return lib().smx_modpdm_control_div_count(self._h)

    def controlrate(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(lib().smx_modpdm_controlrate(self._h, C.byref(a), C.byref(b), C.byref(c)), "smx_modpdm_controlrate")
        return a.value, b.value, c.value


class Patch:
    """Dynamic patcher (stm32f103/mod_bpmodular.c): instances allocated one by one, connected by node
    index, run in allocation order; n copies of the network."""

    def __init__(self, n_instances, n_inputs=0, device=0):
        self._h = lib().smx_patch_create(n_instances, n_inputs, device)
        if not self._h:
            raise SmxError("smx_patch_create: " + lib().smx_last_error().decode())
        self.n, self.n_inputs = n_instances, n_inputs

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.smx_patch_destroy(self._h)
        self._h = None

    __del__ = close

    def apply(self, cls, inputs=(), config=0):
        """-> node index, or a negative PATCH_* code."""
        a = np.ascontiguousarray(inputs, np.uint32)
        return lib().smx_patch_apply(self._h, cls, _ptr(a) if len(a) else None, len(a), config)

    def count(self):
        return lib().smx_patch_count(self._h)

    def reset(self):
        _check(lib().smx_patch_reset(self._h), "smx_patch_reset")

    def tick(self, n=1, inputs=None, gpout=None):
        """inputs: uint32[n, n_inputs, n_instances]; -> uint32[n, n_instances] written by node `gpout`, or None."""
        inp = None if inputs is None else np.ascontiguousarray(inputs, np.uint32)
        assert inp is None or inp.shape == (n, self.n_inputs, self.n)
        out = None if gpout is None else np.empty((n, self.n), np.uint32)
        _check(lib().smx_patch_tick(self._h, n, _ptr(inp), 0 if gpout is None else gpout, _ptr(out)), "smx_patch_tick")
        return out

    def state_get(self, node, field):
        v = np.empty(self.n, np.uint32)
        rv = lib().smx_patch_state_get(self._h, node, field, _ptr(v))
        return v if rv == 0 else rv

    def state_set(self, node, field, vals):
        v = np.ascontiguousarray(np.broadcast_to(np.asarray(vals, np.uint32), (self.n,)), np.uint32)
        return lib().smx_patch_state_set(self._h, node, field, _ptr(v))


def tag_u32_packet(args, payload=b"", frm=()):
    """TAG_U32 frame body: tag:16, nb_from:8, nb_args:8, from[], args[] (big-endian), payload
    (the layout of the reference's own example, stm32f103/mod_synth.c:98)."""
    import struct
    return struct.pack(">HBB", 0xFFF5, len(frm), len(args)) + b"".join(struct.pack(">I", x & 0xFFFFFFFF) for x in tuple(frm) + tuple(args)) + bytes(payload)


class Firmware:
    """Hosted firmware control surface (stm32f103/mod_synth.c:50-137)."""

    def __init__(self, n_channels=3, n_oscillators=1, device=0):
        self._h = lib().smx_fw_create(n_channels, n_oscillators, device)
        if not self._h:
            raise SmxError("smx_fw_create: " + lib().smx_last_error().decode())
        self.n = n_channels
        # borrowed views of the module banks
        self.pwm = PwmBank.__new__(PwmBank)
        self.pwm._h, self.pwm.n, self.pwm.order = lib().smx_fw_pwm(self._h), n_channels, 2
        self.osc = None
        if n_oscillators:
            self.osc = OscBank.__new__(OscBank)
            self.osc._h, self.osc.n, self.osc.words = lib().smx_fw_osc(self._h), n_oscillators, (n_oscillators + 31) // 32

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            self.pwm._h = None               # borrowed handles die with the firmware object
            if self.osc:
                self.osc._h = None
            _lib.smx_fw_destroy(self._h)
        self._h = None

    __del__ = close

    def handle_tag_u32(self, args, payload=b""):
        a = np.ascontiguousarray(args, np.uint32)
        b = np.frombuffer(bytes(payload), np.uint8) if payload else None
        return lib().smx_fw_handle_tag_u32(self._h, _ptr(a) if len(a) else None, len(a), _ptr(b), 0 if b is None else len(b))

    def handle_packet(self, data):
        b = np.frombuffer(bytes(data), np.uint8)
        return lib().smx_fw_handle_packet(self._h, _ptr(b) if len(b) else None, len(b))

    @property
    def running(self):
        return bool(lib().smx_fw_running(self._h))

    def parameter(self, i):
        return lib().smx_fw_parameter(self._h, i)

    def tick_n(self, n_ticks, dither=None):
        d = None if dither is None else np.ascontiguousarray(dither, np.uint32)
        duty = np.empty((n_ticks, self.n), np.uint8)
        ran = lib().smx_fw_tick_n(self._h, n_ticks, _ptr(d), _ptr(duty))
        if ran < 0:
            _check(ran, "smx_fw_tick_n")
        return duty[:ran]

    def poll(self, osc=0):
        avg, num, ln = C.c_uint32(), C.c_uint32(), C.c_uint32()
        cont = np.zeros(64, np.uint8)
        rv = lib().smx_fw_poll(self._h, osc, C.byref(avg), C.byref(num), _ptr(cont), 64, C.byref(ln))
        if rv < 0:
            _check(rv, "smx_fw_poll")
        return None if rv == 0 else (avg.value, num.value, bytes(cont[:ln.value]))


class CprocBank:
    """N instances of one static cproc chain (generic/cproc.h PROC_COND bindings)."""

    def __init__(self, n_instances, nodes, n_inputs, device=0):
        arr = (CprocNode * len(nodes))(*[CprocNode(*nd) for nd in nodes])
        self._h = lib().smx_cproc_create(n_instances, arr, len(nodes), n_inputs, device)
        if not self._h:
            raise SmxError("smx_cproc_create: " + lib().smx_last_error().decode())
        self.n, self.n_nodes, self.n_inputs = n_instances, len(nodes), n_inputs

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.smx_cproc_destroy(self._h)
        self._h = None

    __del__ = close

    def tick_n(self, input, g=None, out_node=None):
        inp = np.ascontiguousarray(input, np.uint32)
        nt = inp.size // (self.n_inputs * self.n)
        gg = None if g is None else np.ascontiguousarray(g, np.uint32)
        out = np.empty((nt, self.n), np.uint32)
        _check(lib().smx_cproc_tick_n(self._h, nt, _ptr(inp), _ptr(gg),
                                      self.n_nodes - 1 if out_node is None else out_node, _ptr(out)),
               "smx_cproc_tick_n")
        return out

    def read_state(self):
        st = np.empty((self.n_nodes, 2, self.n), np.uint32)
        _check(lib().smx_cproc_read_state(self._h, _ptr(st)), "smx_cproc_read_state")
        return st


class ClockBank:
    """N integer-divider square clocks / MIDI clock generators (linux/clock.c:106-120)."""

    def __init__(self, n, device=0):
        self._h = lib().smx_clock_create(n, device)
        if not self._h:
            raise SmxError("smx_clock_create: " + lib().smx_last_error().decode())
        self.n, self.words = n, (n + 31) // 32

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.smx_clock_destroy(self._h)
        self._h = None

    __del__ = close

    def load(self, hperiod=None, phase=None, pol=None):
        hp = None if hperiod is None else np.ascontiguousarray(hperiod, np.uint32)
        ph = None if phase is None else np.ascontiguousarray(phase, np.int32)
        po = None if pol is None else np.ascontiguousarray(pol, np.uint32)
        _check(lib().smx_clock_load(self._h, _ptr(hp), _ptr(ph), _ptr(po)), "smx_clock_load")

    def read(self):
        hp, ph, po = np.empty(self.n, np.uint32), np.empty(self.n, np.int32), np.empty(self.n, np.uint32)
        _check(lib().smx_clock_read(self._h, _ptr(hp), _ptr(ph), _ptr(po)), "smx_clock_read")
        return hp, ph, po

    def run(self, n_frames):
        pb = np.empty((n_frames, self.words), np.uint32)
        tb = np.empty((n_frames, self.words), np.uint32)
        _check(lib().smx_clock_run(self._h, n_frames, _ptr(pb), _ptr(tb)), "smx_clock_run")
        return pb, tb
