"""bp5 chain (edge, acc, acc: stm32f103/bp5_plugin.c:4-9) on 1 Mi instances x 256 ticks: wall time of smx_cproc_tick_n
with device-resident rows is not available (the ABI is synchronous with host buffers), so run this under
  rocprofv3 --kernel-trace --stats   and read cproc_kernel's average."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth_tools_amd as sta
from synth_tools_amd import synthetic, PROC_ACC, PROC_EDGE, cproc_input
n = 1 << 20
nodes = [(PROC_EDGE, cproc_input(0), 1), (PROC_ACC, 0, 1), (PROC_ACC, 1, 1)]
cb = sta.CprocBank(n, nodes, 1)
inp = (synthetic.splitmix64(21, 256 * n) & np.uint64(3)).astype(np.uint32).reshape(256, 1, n)
for _ in range(6):
    cb.tick_n(inp)
cb.close()
print("done", flush=True)
