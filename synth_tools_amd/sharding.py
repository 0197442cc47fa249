"""Host-side voice sharding for the multi-GPU path.

Voices never interact except in the final integer sum (linux/synth.c:172-179), so a
bank of n_total voices is cut into contiguous ranges, one per rank; each rank mixes its
range to an int32 bus and the buses are added.  The sum is a wrapping 32-bit integer
sum, hence associative: any rank order gives the same bits.

The production path does the sum inside libsynth_mi355x.so with RCCL
(smx_bank_allreduce_async).  `allreduce_bus` is the same exchange through
torch.distributed (RCCL when the group backend is "nccl", gloo on CPU) for hosts that
already own a process group -- and it is what the world_size-2 CPU tests drive.
"""
import numpy as np


def shard_range(n_total, rank, world):
    """[lo, hi) of the voices rank owns: contiguous, sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank %d of %d" % (rank, world))
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def voice_owner(voice, n_total, world):
    """Rank that owns a global voice index (note-on routing is host-side)."""
    base, rem = divmod(n_total, world)
    split = rem * (base + 1)
    if voice < split:
        return voice // (base + 1)
    return rem + (voice - split) // base


def allreduce_bus(bus, group=None):
    """In-place wrapping int32 sum of a bus over the ranks of a torch process group."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(bus) if isinstance(bus, np.ndarray) else bus
    if t.dtype != torch.int32:
        raise TypeError("the bus is int32 (reduce integers, convert to float afterwards)")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return bus
