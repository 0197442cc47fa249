#!/bin/bash
# Scratch: same-box A/B of how the direct kernels accumulate.  The libraries compared were built by hand from
# saw_bank.hip with acc_add4's PLAIN forced off (libsmx_acc0.so: the compiler's v_add3_u32) and on
# (libsmx_acc1.so: plain v_add_u32) for every instantiation; the numbers are in saw_bank.hip and DESIGN.md 3.1.
for rep in 1 2; do
for lib in "" "$PWD/tools/ubench/libsmx_acc0.so" "$PWD/tools/ubench/libsmx_acc1.so"; do
  export SMX_LIB=$lib
  echo "== lib=${lib:-default (pair sums)}"
  SMX_SAW_NO_CARRY=1 LGS=26 NFS=8,16,32,64 python tools/explore_saw11.py
  NFS=16,64,256 LGS=16,18,20,22 python tools/explore_small.py
  NFS=4096 LGS=16 python tools/explore_small.py
done
done
