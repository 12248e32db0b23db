"""N > 1 path on CPU: world_size-2 gloo processes exercise the sharding and the one collective
(all-gather of terminal states) exactly as bench.py / a rollout driver uses them."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def test_shard_ranges_cover_the_ensemble():
    from continuum_robot.distributed import impulse_amplitudes, shard_range

    for total, world in ((4096, 8), (10, 3), (7, 8), (1, 1)):
        ranges = [shard_range(total, world, r) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        sizes = [hi - lo for lo, hi in ranges]
        assert max(sizes) - min(sizes) <= 1
        amps = np.concatenate([impulse_amplitudes(total, lo, hi) for lo, hi in ranges])
        assert np.allclose(amps, 0.1 * (1 + np.arange(total) / total))
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, total, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from continuum_robot.distributed import gather_terminal_states, impulse_amplitudes, shard_range

        lo, hi = shard_range(total, world, rank)
        amps = impulse_amplitudes(total, lo, hi)
        # stand-in for a rank's terminal states [B_local, 2, n_node, 4]: encodes (global beam, amplitude)
        local = torch.zeros((hi - lo, 2, 3, 4), dtype=torch.float64)
        local[:, 0, 0, 0] = torch.arange(lo, hi, dtype=torch.float64)
        local[:, 1, 2, 1] = torch.as_tensor(amps)
        full = gather_terminal_states(local)
        assert full.shape == (total, 2, 3, 4)
        assert torch.equal(full[:, 0, 0, 0], torch.arange(total, dtype=torch.float64))
        assert torch.allclose(full[:, 1, 2, 1], torch.as_tensor(0.1 * (1 + np.arange(total) / total)))
        # barrier + max-over-ranks timing, as bench.py does
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == world
        open(os.path.join(tmpdir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_allgather_of_terminal_states(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, total = 2, 8
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
