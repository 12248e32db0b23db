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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,total", [(2, 8),    # equal shards: one all_gather_into_tensor
                                         (2, 7),    # ragged: 4 + 3 (padded to the largest shard, padding cut out)
                                         (3, 10),   # ragged: 4 + 3 + 3
                                         (3, 2)])   # a rank that owns NO beam (0-row shard)
def test_gloo_allgather_of_terminal_states(tmp_path, world, total):
    mp.spawn(_worker, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _worker_known_sizes(rank, world, port, total, tmpdir):
    """The bench's form: shard sizes known from shard_sizes(), no size exchange; a wrong size is refused."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from continuum_robot.distributed import gather_terminal_states, shard_range, shard_sizes

        lo, hi = shard_range(total, world, rank)
        sizes = shard_sizes(total, world)
        assert sizes[rank] == hi - lo and sum(sizes) == total
        local = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1).repeat(1, 5)
        full = gather_terminal_states(local, sizes=sizes)
        assert torch.equal(full[:, 3], torch.arange(total, dtype=torch.float64))
        with pytest.raises(ValueError):
            gather_terminal_states(local, sizes=[s + 1 for s in sizes])
        with pytest.raises(ValueError):
            gather_terminal_states(local, sizes=sizes[:-1])
        open(os.path.join(tmpdir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_gloo_allgather_with_known_ragged_sizes(tmp_path):
    world, total = 3, 11
    mp.spawn(_worker_known_sizes, args=(world, _free_port(), total, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


STUB_WORKER = """
import json, os, sys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["CRB_BENCH_LAUNCHED"] == "1"
open(os.path.join(os.environ["STUB_DIR"], f"rank{rank}"), "w").write(" ".join(sys.argv[1:]))
if os.environ.get("STUB_FAIL") == str(rank):
    sys.exit(7)
if os.environ.get("STUB_FAIL") is not None and rank != int(os.environ["STUB_FAIL"]):
    import time
    time.sleep(30)          # must be terminated by the launcher, not waited for
print(json.dumps({"rank": rank, "n_gpus": world}))   # rank 0's stdout is the launcher's stdout
"""


def test_bench_launcher_spawns_one_child_per_gpu(tmp_path):
    """`python bench.py --gpus N` without WORLD_SIZE in the environment is the launcher: N children with the
    torch.distributed environment, rank 0's stdout passed through, nothing GPU-related in the parent (the stub
    worker stands in for the rank code, which needs a GPU)."""
    import json
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = tmp_path / "stub_worker.py"
    stub.write_text(STUB_WORKER)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(CRB_BENCH_WORKER=str(stub), STUB_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "5", "--scaling", "strong"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"rank": 0, "n_gpus": 3}   # ONE line: rank 0's
    for k in range(3):
        assert (tmp_path / f"rank{k}").read_text() == "--gpus 3 --steps 5 --scaling strong"
    assert '"rank": 1' in r.stderr and '"rank": 2' in r.stderr                      # the others go to stderr
    # CRB_BENCH_FORCE_DIST=1 takes the launcher path with ONE rank
    env1 = dict(env, CRB_BENCH_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], env=env1, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout) == {"rank": 0, "n_gpus": 1}
    # a failing rank: its status is the launcher's, the sleeping ranks are terminated instead of awaited
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3"], env=dict(env, STUB_FAIL="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 7 and time.time() - t0 < 25
    assert "rank 1 exited with status 7" in r.stderr


def test_bench_initial_states_do_not_depend_on_the_world_size():
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from continuum_robot.distributed import shard_range

    whole = bench.initial_states(0, 1500, 12)
    assert whole.shape == (1500, 24) and np.all(whole[:, 0:12:3] == 0) and np.all(whole[:, 12::3] == 0)
    parts = [bench.initial_states(*shard_range(1500, 4, r), 12) for r in range(4)]
    assert np.array_equal(np.concatenate(parts), whole)
    a = bench.parse(["--gpus", "8", "--config", "config5", "--scaling", "strong"])
    assert a.steps == 1000 and a.scaling == "strong" and a.dtype == "f64"
    assert bench.parse(["--config", "config4"]).scaling == "strong" and bench.parse([]).scaling == "weak"



class _FakeChunk:
    """stands in for a BeamEnsemble chunk on the CPU: 'stepping' adds 1 to its rows"""

    def __init__(self, rows):
        self.x = rows.clone()

    # (the part of BeamEnsemble's interface that rollout_and_gather uses)
    @property
    def state(self):
        return self.x

    @property
    def n_beams(self):
        return self.x.shape[0]

    @property
    def n(self):
        return self.x.shape[1] // 2

    def unpack_state(self, out=None):
        if out is None:
            return self.x
        out.copy_(self.x)
        return out


def _worker_chunked(rank, world, port, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from continuum_robot.distributed import assemble_chunks, rollout_and_gather, shard_range

        total, n_chunks = 24, 3
        lo, hi = shard_range(total, world, rank)
        rows = torch.arange(lo, hi, dtype=torch.float64).reshape(-1, 1).repeat(1, 4)
        per = (hi - lo) // n_chunks
        chunks = [_FakeChunk(rows[c * per:(c + 1) * per]) for c in range(n_chunks)]
        stepped = []

        def advance(ens):
            ens.x = ens.x + 1.0
            stepped.append(ens)

        outs = rollout_and_gather(chunks, advance)
        assert len(outs) == n_chunks and stepped == chunks
        full = assemble_chunks(outs, world)
        assert full.shape == (total, 4)
        assert torch.equal(full[:, 2], torch.arange(total, dtype=torch.float64) + 1.0)   # global beam order, every chunk stepped
        open(os.path.join(tmpdir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_chunked_rollout_with_overlapped_allgather(tmp_path):
    """rollout_and_gather: per-chunk asynchronous all-gathers (the exchange of chunk c runs while chunk c + 1 is
    stepped) and the reassembly into global beam order, on two gloo ranks."""
    world = 2
    mp.spawn(_worker_chunked, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
    # without a process group the chunks come back as they are
    from continuum_robot.distributed import assemble_chunks, rollout_and_gather

    chunks = [_FakeChunk(torch.full((2, 3), float(c))) for c in range(3)]
    outs = rollout_and_gather(chunks, lambda e: None)
    assert torch.equal(assemble_chunks(outs, 1)[:, 0], torch.tensor([0.0, 0, 1, 1, 2, 2]))
