"""Frame-sharded multi-GPU solves, one process per GPU: frames sharded by slot range, ONE small all-reduce of
the packed sums of a step - [reduced camera system | cost | model decrease | failed blocks], 100 .. 400 doubles - per
optimizer step, Gauss-Newton and Levenberg-Marquardt alike.  Mode E needs no collective at all.  The production transport is
the library's own ncclAllReduce (Problem.set_rccl_comm); a single process reaches several GPUs through ccal_multi_*
(engine.MultiContext / MultiProblem).  This module is the CALLBACK transport the tests drive with gloo:

torch.distributed is plumbing only: the hook below wraps the library's device buffer as a tensor
(zero copy) and calls all_reduce on the library's own HIP stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np


class _DevBuf:
    """CUDA-array-interface view of `count` doubles at a raw device pointer."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def make_allreduce_hook(group=None, device=None):
    """Return fn(ptr, count, stream) -> int for Problem.set_allreduce (device buffer, `device` = torch
    device) or for a host buffer (`device` None; used by the CPU gloo tests with the oracle)."""
    import torch
    import torch.distributed as dist

    def hook(ptr: int, count: int, stream: int) -> int:
        if device is None:
            arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(count,))
            t = torch.from_numpy(arr)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)       # in place on the caller's buffer
            return 0
        t = torch.as_tensor(_DevBuf(ptr, count), device=device)
        if stream:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=device)):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return 0

    return hook


def slot_range(n_slots: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous frame-slot range of one rank (all cameras' observations of a slot stay together)."""
    return n_slots * rank // world, n_slots * (rank + 1) // world


def gather_poses(local_poses: np.ndarray, n_slots: int, group=None) -> np.ndarray:
    """All ranks' pose blocks concatenated in slot order (host side, after the solve)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return local_poses
    parts = [None] * world
    dist.all_gather_object(parts, np.ascontiguousarray(local_poses), group=group)
    out = np.concatenate(parts, axis=0)
    assert out.shape[0] == n_slots
    return out
