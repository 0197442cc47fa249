"""Scratch: oscillator / clock / cproc bank kernels under rocprofv3 (their ABI is synchronous with
host buffers, so kernel time comes from the kernel trace):
  cd /tmp && rocprofv3 --kernel-trace --stats -d gpurun_out/osc -- python3 tools/explore_osc.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth_tools_amd as sta
from synth_tools_amd import synthetic

n, nt = 1 << 20, 1024
r = synthetic.splitmix64(11, n)
o = sta.OscBank(n)
o.load_pwm(phase=(r & np.uint64(0xFFFFFF)).astype(np.uint32), speed=((r >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.uint32))
for _ in range(3):
    o.tick_n(nt, None, want_duty=False)
sync = (synthetic.splitmix64(12, nt * o.words) & synthetic.splitmix64(13, nt * o.words) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
for _ in range(3):
    o.tick_n(nt, sync, want_duty=False)
ne = 64
cc = np.cumsum(np.random.default_rng(1).integers(1000, 200000, (ne, n), dtype=np.uint32), axis=0, dtype=np.uint32)
for _ in range(3):
    o.events(cc)
o.close()
c = sta.ClockBank(n)
c.load(hperiod=(1000 + (r & np.uint64(0xFFFF))).astype(np.uint32))
for _ in range(3):
    c.run(1024)
c.close()
from synth_tools_amd import PROC_ACC, PROC_EDGE, cproc_input
nodes = [(PROC_EDGE, cproc_input(0), 1), (PROC_ACC, 0, 1), (PROC_ACC, 1, 1)]      # stm32f103/bp5_plugin.c:4-9
cb = sta.CprocBank(n, nodes, 1)
inp = (synthetic.splitmix64(21, 256 * n) & np.uint64(3)).astype(np.uint32).reshape(256, 1, n)
for _ in range(3):
    cb.tick_n(inp)
cb.close()
print("done", flush=True)
