"""cproc bank, 1 Mi instances x 256 ticks: the bp5 chain (edge, acc, acc: stm32f103/bp5_plugin.c:4-9) and a one-node
graph (gpin: out = the input word) -- the same rows in and out with next to no work in between.  The ABI is synchronous
with host buffers, so run this under  rocprofv3 --kernel-trace --stats  and read cproc_kernel's launches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import synth_tools_amd as sta
from synth_tools_amd import synthetic, PROC_ACC, PROC_EDGE, cproc_input
PROC_GPIN = 3
n = 1 << 20
inp = (synthetic.splitmix64(21, 256 * n) & np.uint64(3)).astype(np.uint32).reshape(256, 1, n)
for nodes in ([(PROC_EDGE, cproc_input(0), 1), (PROC_ACC, 0, 1), (PROC_ACC, 1, 1)], [(PROC_GPIN, cproc_input(0), 1)]):
    cb = sta.CprocBank(n, nodes, 1)
    for _ in range(4):
        cb.tick_n(inp)
    cb.close()
print("done", flush=True)
