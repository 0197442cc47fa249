"""GPU: the firmware control surface (mod_synth.c:89-137) and the two plain-C host
programs, end to end through pipes, against the oracle."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle
import synth_tools_amd as sta
from synth_tools_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYNTH_ELF = os.path.join(ROOT, "host", "synth.dynamic.host.elf")
FW_ELF = os.path.join(ROOT, "host", "fw.dynamic.host.elf")


def _oracle_pwm(n):
    arrs = {k: np.zeros(n, np.uint32) for k in ("setpoint", "pos0", "vel0", "pos1", "vel1", "s1", "s2", "s3", "s4")}
    arrs["setpoint"][:] = 0x40000000
    arrs["setpoint"][0] = 2000000000
    b = oracle.PwmBank(n=n, order=2, div_count=0, div_log=12, out_shift=24,
                       s=(C.c_void_p * 4)(*[arrs["s%d" % k].ctypes.data for k in (1, 2, 3, 4)]),
                       **{k: arrs[k].ctypes.data for k in ("setpoint", "pos0", "vel0", "pos1", "vel1")})
    return b, arrs


def test_tag_u32_commands(smx, orc):
    fw = smx.Firmware(3, 1)
    assert fw.running                                        # pdm_start at init, mod_synth.c:67
    assert fw.parameter(0) == 15 << 27                       # osc_setpoint default, mod_synth.c:51
    # argument errors, mod_synth.c:91,99,106-107,113
    assert fw.handle_tag_u32([]) == -1
    assert fw.handle_tag_u32([100]) == -1
    assert fw.handle_tag_u32([101, 1]) == -1
    assert fw.handle_tag_u32([101, 3, 5]) == -2
    assert fw.handle_tag_u32([102, 1, 2]) == -3
    assert fw.handle_tag_u32([7, 1]) == -1                   # no such parameter
    # the reference's own example packet: MODE 1 (mod_synth.c:98)
    assert fw.handle_packet(bytes.fromhex("fff50002" "00000064" "00000001")) == 0 and fw.running
    assert fw.handle_packet(smx.tag_u32_packet([100, 0])) == 0 and not fw.running
    assert len(fw.tick_n(100)) == 0                          # stopped: the ISR does not run
    assert fw.handle_tag_u32([100, 1]) == 0
    assert fw.handle_packet(smx.tag_u32_packet([0, 0x12345678])) == 0 and fw.parameter(0) == 0x12345678
    assert fw.handle_packet(b"\xff\xfe\x00\x00") == 0        # unknown tag: logged, ignored (synth.c:39-40)
    assert fw.handle_packet(b"\xff\xf5\x00\x05\x00") == -1   # truncated
    # SETPOINT then the ISR against the oracle
    ob, keep = _oracle_pwm(3)
    assert fw.handle_packet(smx.tag_u32_packet([101, 2, 0x90000000])) == 0
    keep["setpoint"][2] = 0x90000000
    d = synthetic.dither_stream(5000, 3, 0x3FF)
    got = fw.tick_n(5000, d)
    want = np.zeros((5000, 3), np.uint8)
    orc.orc_pwm_bank_run(C.byref(ob), d.ctypes.data, 5000, want.ctypes.data)
    assert np.array_equal(got, want)
    fw.close()


def test_measure_and_poll(smx, orc):
    fw = smx.Firmware(3, 1)
    assert fw.poll() is None
    assert fw.handle_tag_u32([102, 14], b"pid-A") == 0       # log_max 14 + a continuation
    assert fw.handle_tag_u32([102], b"pid-B") == 0
    cc = (np.arange(1, 41, dtype=np.uint32) * 1000).reshape(40, 1)
    fw.osc.events(cc)
    p = oracle.Pmeas(log_max=14)
    for t in cc[:, 0]:
        orc.orc_osc_event(C.byref(p), int(t))
    assert p.write == 2
    first = fw.poll()
    second = fw.poll()
    assert first == (p.avg[1], p.num_pub[1], b"pid-A")       # read=1 -> meas[1]
    assert second == (p.avg[0], p.num_pub[0], b"pid-B")
    assert fw.poll() is None
    fw.close()


def _events_file(path, events):
    with open(path, "wb") as f:
        for blk, msg in events:
            f.write(struct.pack("<IB3s", blk, len(msg), bytes(msg) + b"\0" * (3 - len(msg))))


@pytest.mark.parametrize("voices", [64, 5000])
def test_synth_host_fake_jack(tmp_path, orc, voices):
    """linux/synth.c's process() loop, 40 blocks of 64 frames, scripted MIDI, float output
    compared with the oracle; then the stdin-EOF exit convention (exit status 1)."""
    events = [(0, [0x90, 60, 100]), (0, [0x90, 64, 100]), (3, [0x90, 67, 90]), (10, [0x80, 64, 0]),
              (11, [0x90, 60, 0]), (12, [0x80, 99, 0]), (12, [0xB0, 25, 3]), (12, [0x91, 50, 100]),
              (20, [0x90, 72, 1])] + [(25, [0x90, n, 100]) for n in range(30, 100)]
    ev, out = tmp_path / "ev.bin", tmp_path / "out.f32"
    _events_file(ev, events)
    env = dict(os.environ, SYNTH_VOICES=str(voices))
    r = subprocess.run([SYNTH_ELF, "--fake-jack", "40", "64", str(ev), str(out)], env=env,
                       stdin=subprocess.DEVNULL, capture_output=True, timeout=120)
    assert r.returncode == 1, r.stderr.decode()              # EOF on stdin -> exit(1), linux/synth.c:305-310
    got = np.fromfile(out, np.float32)
    n2v = np.zeros(128, np.int32)
    inc = np.zeros(voices, np.uint32)
    st = np.zeros(voices, np.uint32)
    want = []
    for blk in range(40):
        for b, msg in events:
            if b == blk:
                orc.orc_midi_event(n2v, inc, voices, np.array(msg, np.uint8), 3)
        want.append(oracle.synth_run(orc, inc, st, 64)[1])
    assert np.array_equal(got.view(np.uint32), np.concatenate(want).view(np.uint32))


def test_synth_host_sounding_bank(tmp_path, orc):
    """SYNTH_FILL=1: the host program starts with every voice of the bank sounding (the bank the
    real-time latency log is measured on); a few note events on top (voice 0 is stolen: the bank is
    full), output against the oracle."""
    voices = 3000
    events = [(2, [0x90, 40, 100]), (5, [0x80, 40, 0]), (6, [0x90, 100, 100])]
    ev, out = tmp_path / "ev.bin", tmp_path / "out.f32"
    _events_file(ev, events)
    env = dict(os.environ, SYNTH_VOICES=str(voices), SYNTH_FILL="1")
    r = subprocess.run([SYNTH_ELF, "--fake-jack", "12", "64", str(ev), str(out)], env=env,
                       stdin=subprocess.DEVNULL, capture_output=True, timeout=120)
    assert r.returncode == 1, r.stderr.decode()
    got = np.fromfile(out, np.float32)
    v = np.arange(voices, dtype=np.uint64)
    h = (v * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    inc = np.array([orc.orc_note_to_inc(21 + int((x >> np.uint64(12)) % np.uint64(88))) for x in h], np.uint32)
    st = ((h * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    n2v = np.zeros(128, np.int32)
    want = []
    for blk in range(12):
        for b, msg in events:
            if b == blk:
                orc.orc_midi_event(n2v, inc, voices, np.array(msg, np.uint8), 3)
        want.append(oracle.synth_run(orc, inc, st, 64)[1])
    assert np.array_equal(got.view(np.uint32), np.concatenate(want).view(np.uint32))


def test_fw_host_port_protocol(orc):
    """{packet,4} on stdin/stdout like an Erlang port (erl/jack_client.erl:63-68)."""
    p = subprocess.Popen([FW_ELF, "3", "1"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)

    def send(body):
        p.stdin.write(struct.pack(">I", len(body)) + body)
        p.stdin.flush()

    def ticks(n):
        send(struct.pack(">HHI", 0xFFFB, 1, n))
        (ln,) = struct.unpack(">I", p.stdout.read(4))
        body = p.stdout.read(ln)
        assert body[:4] == b"\xff\xfb\x00\x01"
        return np.frombuffer(body[4:], np.uint8).reshape(-1, 3)

    ob, keep = _oracle_pwm(3)
    want = np.zeros((6000, 3), np.uint8)
    send(sta.tag_u32_packet([101, 1, 0xB0000000]))
    keep["setpoint"][1] = 0xB0000000
    got = ticks(6000)
    orc.orc_pwm_bank_run(C.byref(ob), None, 6000, want.ctypes.data)
    assert np.array_equal(got, want)
    send(sta.tag_u32_packet([100, 0]))                       # MODE 0: pdm_stop
    assert len(ticks(10)) == 0
    send(sta.tag_u32_packet([100, 1]))
    got = ticks(100)
    orc.orc_pwm_bank_run(C.byref(ob), None, 100, want.ctypes.data)
    assert np.array_equal(got, want[:100])
    p.stdin.close()                                          # port_close -> EOF -> exit(1)
    assert p.wait(timeout=30) == 1


def test_c_abi_test_program():
    """tests/c/test_synth_abi.c: plain-C, ASSERT-based, against the shared library."""
    exe = os.path.join(ROOT, "host", "test_synth_abi.dynamic.host.elf")
    r = subprocess.run([exe], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()
    assert r.stderr.decode().strip().endswith("test_synth_abi.c")


def test_c1_one_voice_750_blocks(tmp_path, orc):
    """BASELINE config 1: 1 voice, note 69, 64-frame blocks, 750 blocks (one second
    at 48 kHz) through the JACK process() callback of the host program, bit-exact vs the oracle."""
    ev, out = tmp_path / "ev.bin", tmp_path / "out.f32"
    _events_file(ev, [(0, [0x90, 69, 100])])
    r = subprocess.run([SYNTH_ELF, "--fake-jack", "750", "64", str(ev), str(out)],
                       stdin=subprocess.DEVNULL, capture_output=True, timeout=300)
    assert r.returncode == 1, r.stderr.decode()
    got = np.fromfile(out, np.float32)
    n2v = np.zeros(128, np.int32)
    inc = np.zeros(64, np.uint32)
    st = np.zeros(64, np.uint32)
    orc.orc_note_on(n2v, inc, 64, 69)
    want = oracle.synth_run(orc, inc, st, 750 * 64)[1]
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # the saw wraps 440 times per second: the phase after one second is 48000 * inc(69)
    assert st[0] == (48000 * 39370533) & 0xFFFFFFFF


def test_create_destroy_cycles_do_not_leak(smx):
    """200 create/run/destroy cycles of every bank type leave device memory where it was."""
    hip = C.CDLL("libamdhip64.so")            # the runtime the library already loaded

    def free_bytes():
        free, total = C.c_size_t(), C.c_size_t()
        assert hip.hipDeviceSynchronize() == 0
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    smx.SawBank(64).close()                   # make sure the device context exists
    free0 = free_bytes()
    nodes = [(smx.PROC_EDGE, smx.cproc_input(0), 1), (smx.PROC_ACC, 0, 1)]
    for _ in range(200):
        b = smx.SawBank(5000); b.run(64); b.close()
        p = smx.PdmBank(3000); p.tick_n(70, want_bits=False); p.close()
        w = smx.PwmBank(2000); w.tick_n(10, want_duty=False); w.close()
        q = smx.PolyBank(1000); q.run(8); q.close()
        o = smx.OscBank(500); o.tick_n(5, want_duty=False); o.close()
        c = smx.ClockBank(100); c.run(5); c.close()
        g = smx.CprocBank(300, nodes, 1); g.tick_n(np.zeros((2, 1, 300), np.uint32)); g.close()
        f = smx.Firmware(3, 1); f.tick_n(5); f.close()
    free1 = free_bytes()
    assert free0 - free1 < 64 << 20, "device memory shrank by %d MiB" % ((free0 - free1) >> 20)
