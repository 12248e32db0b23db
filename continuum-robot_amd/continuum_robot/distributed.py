"""Multi-GPU layout of an ensemble: independent beams are sharded over ranks, nothing is exchanged
per step, terminal states are all-gathered once.

One process per GPU, ``torch.distributed`` with backend "nccl" (= RCCL over xGMI on ROCm); the same
code runs on "gloo" for the CPU tests of the sharding logic.  The reference's only parallelism is
multiprocessing.Pool.map over independent simulations (examples/beam_comparison_fluid.py:82-83);
this is its counterpart.
"""
from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of beams owned by ``rank``: [lo, hi).  Earlier ranks take the remainder."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def impulse_amplitudes(n_total: int, lo: int, hi: int, base_amp: float = 0.1) -> np.ndarray:
    """Per-beam tip-impulse amplitudes of the synthetic ensembles:
    a_b = base_amp * (1 + b / B) with b the GLOBAL beam index."""
    return base_amp * (1.0 + np.arange(lo, hi) / n_total)


def gather_terminal_states(local: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather equally sized per-rank tensors along dim 0, in rank order.

    One collective per rollout: with nccl this is a single RCCL all-gather (direct over the xGMI
    mesh); message = local.numel() * itemsize bytes per rank.
    """
    if not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out
