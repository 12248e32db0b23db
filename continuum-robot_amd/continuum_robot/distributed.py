"""Multi-GPU layout of an ensemble: independent beams are sharded over ranks, nothing is exchanged
per step, terminal states are all-gathered once.

One process per GPU, ``torch.distributed`` with backend "nccl" (= RCCL over xGMI on ROCm); the same
code runs on "gloo" for the CPU tests of the sharding logic.  The reference's only parallelism is
multiprocessing.Pool.map over independent simulations (examples/beam_comparison_fluid.py:82-83);
this is its counterpart.
"""
from typing import List, Optional, Sequence, Tuple

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


def shard_sizes(n_total: int, world_size: int) -> List[int]:
    """Number of beams every rank owns under ``shard_range`` (no communication needed to know it)."""
    return [hi - lo for lo, hi in (shard_range(n_total, world_size, r) for r in range(world_size))]


def impulse_amplitudes(n_total: int, lo: int, hi: int, base_amp: float = 0.1) -> np.ndarray:
    """Per-beam tip-impulse amplitudes of the synthetic ensembles:
    a_b = base_amp * (1 + b / B) with b the GLOBAL beam index."""
    return base_amp * (1.0 + np.arange(lo, hi) / n_total)


def gather_terminal_states(local: torch.Tensor, group=None, sizes: Optional[Sequence[int]] = None) -> torch.Tensor:
    """All-gather per-rank tensors along dim 0, in rank order; shards may be RAGGED (``shard_range`` hands out
    unequal ones whenever the ensemble size is not a multiple of the world size).

    ``sizes``: rows owned by every rank when the caller knows them (``shard_sizes``); otherwise they are
    exchanged first (one tiny all-gather).  Equal shards: one ``all_gather_into_tensor`` (with nccl a single
    RCCL all-gather, direct over the xGMI mesh; message = local.numel() * itemsize bytes per rank).  Ragged
    shards: every rank pads its block to the largest shard, the same single collective runs on the padded
    blocks, and the padding rows are cut out of the result.
    """
    if not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    if sizes is None:
        mine = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        every = torch.empty((world,), dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(every, mine, group=group)
        sizes = [int(v) for v in every.cpu().tolist()]
    else:
        sizes = [int(v) for v in sizes]
        if len(sizes) != world:
            raise ValueError(f"sizes has {len(sizes)} entries for a world of {world}")
        if sizes[dist.get_rank(group)] != local.shape[0]:
            raise ValueError(f"rank {dist.get_rank(group)} holds {local.shape[0]} rows, sizes says {sizes[dist.get_rank(group)]}")
    tail = tuple(local.shape[1:])
    biggest = max(sizes)
    if min(sizes) == biggest:
        out = torch.empty((world * biggest,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    padded = local
    if local.shape[0] < biggest:
        padded = torch.zeros((biggest,) + tail, dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    out = torch.empty((world * biggest,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * biggest:r * biggest + sizes[r]] for r in range(world)], dim=0)


def rollout_and_gather(chunks, advance, group=None):
    """Step an ensemble that was split into ``chunks`` (a list of BeamEnsemble-like objects: consecutive, equally sized
    blocks of this rank's beams) and all-gather every chunk's terminal states WHILE the next chunk is being stepped.

    ``advance(ens)`` enqueues the whole rollout of one chunk on the current stream.  After each chunk its states are
    converted to the reference's reduced ordering and handed to an asynchronous ``all_gather_into_tensor``
    (``async_op=True``: with nccl the RCCL kernels run on the process group's own stream, ordered after the chunk's
    stepper by an event) -- so on N > 1 GPUs only the LAST chunk's exchange is exposed: a rollout of a few dozen steps
    is as short as the all-gather of its result (20 steps of 4096 x 256: 0.6 ms; 8 x 50 MB over xGMI: ~0.5 ms).
    Every rank must hold the same number of rows per chunk.  Returns the list of gathered tensors, one per chunk,
    each ``[world * rows_per_chunk, ...]`` in rank order (``assemble_chunks`` restores the global beam order);
    without an initialised process group the chunks' own states are returned.
    """
    outs, works, keep = [], [], []
    on = dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    for ens in chunks:
        advance(ens)
        if on:
            # the layout conversion writes this rank's block straight into its slot of the gathered tensor and the collective
            # runs in place on it (RCCL recognises the send buffer as that slot: no staging copy of the shard on either side)
            rows, width, like = ens.n_beams, 2 * ens.n, ens.state
            out = torch.empty((world * rows, width), dtype=like.dtype, device=like.device)
            mine = out[rank * rows:(rank + 1) * rows]
            ens.unpack_state(out=mine)
            works.append(dist.all_gather_into_tensor(out, mine, group=group, async_op=True))
            outs.append(out)
        else:
            outs.append(ens.unpack_state().contiguous())
    for w in works:
        w.wait()                  # the current stream now waits for every exchange
    return outs


def assemble_chunks(outs, world: int) -> torch.Tensor:
    """``rollout_and_gather``'s per-chunk results -> one tensor in GLOBAL beam order (rank-major, then chunk, then row)."""
    if len(outs) == 1:
        return outs[0]
    per = [o.reshape((world, o.shape[0] // world) + tuple(o.shape[1:])) for o in outs]
    return torch.cat(per, dim=1).reshape((-1,) + tuple(outs[0].shape[1:]))
