#!/usr/bin/env python3
"""bench.py -- beam-element-steps/s of the fused RK4 beam stepper on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config configX] [--scaling weak|strong]

One "step" = one classical RK4 time step (dt = 2e-5 s) of the whole batch.  Workload at N = 1 is
BASELINE.json's metric configuration (configs[2]): 4096 beams x 256 nonlinear Euler-Bernoulli
elements + fluid drag (Nitinol constants of examples/example_utilities.py:25-34, FIXED at node 0,
zero initial state, per-beam tip impulse 0.1*(1 + b/B) N for t < 0.01 s), fp64.

N > 1: one process per GPU over RCCL.  Started by a launcher (python -m torch.distributed.run ...:
RANK/LOCAL_RANK/WORLD_SIZE in the environment) this file is rank RANK of WORLD_SIZE; started plainly as
`python bench.py --gpus N` it IS the launcher: the parent spawns N children of itself before anything touches
a GPU (no re-exec of a process that has initialised HIP), passes rank 0's JSON line through and exits with the
first non-zero child status.  Beams are independent units: contiguous shards, no per-step communication, one
all-gather of the terminal states inside the timed region.  `--scaling weak` keeps the per-GPU ensemble fixed
(config3: 4096 beams per GPU), `--scaling strong` keeps BASELINE's totals (config3/4: 4096 beams, config5: 16384
beams) and shards them, raggedly if need be.

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit, ms_per_step,
`roofline` (algorithmic HBM bytes = 96 B per element-step; config5: fp64 MFMA flop of the
feedback GEMM) and `cpu_baseline` (the C oracle -- a port of the reference path -- on this box's host cores,
all cores and one core, bounded samples).  The timed region is K steps bracketed by barrier + synchronize; when
that region is shorter than 50 ms it is repeated (state reset untimed in between) and the MEDIAN is reported.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "continuum-robot_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_F64_PEAK_TF = 78.6        # MI355X fp64 matrix peak (public spec; measured issue rate 65-71 TF)
MFMA_F32_PEAK_TF = 157.3       # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
VALU_CLOCK_HZ = 2.4e9          # nominal engine clock (the chip holds ~2.2 GHz under this fp64 load)
BYTES_PER_ELEM_STEP = {"f64": 96.0, "f32": 48.0}
MIN_TIMED_S = 0.05             # shorter timed regions are repeated and the median reported

CONFIGS = {
    # BASELINE configs[0], the reference's own CPU-runnable case: ONE 10-element linear cantilever under gravity
    # integrated for 1 s (examples/beam_comparison_gravity.py; the reference hands it to solve_ivp(LSODA): 207 s,
    # BASELINE.md §2).  Here: crb_step_implicit, h = 1e-4 s, 10000 steps = 1 s in ONE launch.  A parity-test case and a
    # wall-time line, not the metric configuration (one beam cannot fill a GPU).
    "config1": dict(beams=1, total=1, elems=10, kind="linear", drag=False, gravity=True, x0=False, implicit=True,
                    dt=1e-4, steps=10000, scaling="weak",
                    label="1 beam x 10 elem, linear + gravity, implicit midpoint h = 1e-4 s (1 s per 10000 steps), fp64"),
    # name: beams per GPU (weak) / total (strong), elements, element type, forces, random x0, defaults
    "config3": dict(beams=4096, total=4096, elems=256, kind="nonlinear", drag=True, gravity=False, x0=False,
                    scaling="weak",
                    label="4096 beams x 256 elem, nonlinear Euler-Bernoulli + fluid drag, fp64"),
    # BASELINE config 4: the same ensemble in fp32, batch-sharded across the node's GPUs (fixed total); 200 steps by
    # default, beyond which single precision drifts past 1e-5 from the fp64 oracle
    "config4": dict(beams=4096, total=4096, elems=256, kind="nonlinear", drag=True, gravity=False, x0=False, dtype="f32",
                    steps=200, scaling="strong",
                    label="4096 beams x 256 elem, nonlinear Euler-Bernoulli + fluid drag, fp32"),
    "config2": dict(beams=1024, total=1024, elems=64, kind="linear", drag=False, gravity=True, x0=True, scaling="weak",
                    label="1024 beams x 64 elem, linear + gravity, fp64"),
    # LQR rollout ensemble (BASELINE config 5: 16384 beams, 2048 per GPU on 8 GPUs): state feedback u = K(0 - x) at
    # every RK4 stage, the whole rollout ONE persistent launch (csrc/crb_loop.h).  dt = 5e-6: the closed loop has
    # |lambda|max = 3.2e5 1/s, RK4 is unstable at the open-loop dt = 2e-5.
    "config5": dict(beams=2048, total=16384, elems=128, kind="linear", drag=False, gravity=True, x0=True, lqr=True,
                    dt=5e-6, amp=10.0, scaling="weak",
                    label="LQR rollout ensemble x 128 elem, linear + gravity + LQR feedback per stage, fp64"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed RK4 steps (default 1000; 200 for config4)")
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="config3", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"], help="default: the config's (f64; config4: f32)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: the config's beams on EVERY GPU; strong: BASELINE's total (config3/4 4096, config5 16384) "
                         "sharded over the GPUs.  Default: the config's (config4: strong, the others weak)")
    ap.add_argument("--launch-steps", type=int, default=None,
                    help="RK4 steps fused per launch (0 = all of --steps); warmup uses launches of the same size, so "
                         "every stepper launch of a run is identical and rocprof's per-kernel average is the launch time")
    ap.add_argument("--gather-chunks", type=int, default=0,
                    help="N > 1: split each rank's shard into this many chunks and all-gather chunk c asynchronously while "
                         "chunk c+1 is stepped (0 = 4 when the shard allows it, 1 = one blocking exchange at the end)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed repeats of the K-step rollout (0 = as many as a 50 ms timed region needs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-all", action="store_true",
                    help="after timing, compare EVERY beam of rank 0 with the oracle (open-loop configs; tens of CPU-seconds)")
    ap.add_argument("--hetero", action="store_true",
                    help="heterogeneous variant (SURVEY 8(d)): per-beam E, rho, r scaled by U(0.9, 1.1), seed 4321")
    args = ap.parse_args(argv)
    cfg = CONFIGS[args.config]
    if args.dtype is None:
        args.dtype = cfg.get("dtype", "f64")
    if args.steps is None:
        args.steps = cfg.get("steps", 1000)
    if args.scaling is None:
        args.scaling = cfg["scaling"]
    if args.launch_steps is None:
        args.launch_steps = 0 if cfg.get("implicit") else 100
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        ap.error("--gpus >= 1, --steps >= 1, --warmup >= 0")
    return args


# ------------------------------------------------------------------ launcher (parent process: never touches a GPU)
def launch(n_ranks, argv):
    """Start ``n_ranks`` children of this script, one per GPU, with the torch.distributed environment; rank 0's
    stdout (the JSON line) is this process's stdout, the other ranks' stdout goes to stderr.  Returns the exit
    status: 0, or the first non-zero child status (the remaining children are then terminated by PID)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    worker = os.environ.get("CRB_BENCH_WORKER") or os.path.abspath(__file__)   # (tests substitute a stub worker)
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CRB_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, worker] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    status = 0
    live = set(range(n_ranks))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and status == 0:
                status = rc
                print(f"[bench launcher] rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr)
                for o in live:
                    procs[o].terminate()
        time.sleep(0.05)
    return status


# ------------------------------------------------------------------ host-side pieces
def lqr_gain(ens):
    """LQR gain of lqr_control.py:46-84 for the ensemble's (linear) beam: Q = diag(100 I, 10 I), R = I."""
    import numpy as np

    from continuum_robot.control import LinearQuadraticRegulator

    K, M = ens.plan.stiffness(), ens.plan.mass()
    n = K.shape[0]
    Q = np.eye(2 * n)
    Q[:n, :n] *= 100
    Q[n:, n:] *= 10
    return LinearQuadraticRegulator(K, M, Q, np.eye(n)).compute_gain_matrix()


def initial_states(lo, hi, n):
    """SURVEY §8(d) random initial states of the linear configs, generated per 512-beam block of the GLOBAL
    ensemble so that beam b starts from the same state whatever the number of ranks."""
    import numpy as np

    rows = []
    for blk in range(lo // 512, (hi - 1) // 512 + 1):
        rng = np.random.default_rng([1234, blk])
        x = np.concatenate([rng.normal(0, 1e-5, (512, n)), rng.normal(0, 1e-3, (512, n))], axis=1)
        a, b = max(lo, blk * 512) - blk * 512, min(hi, (blk + 1) * 512) - blk * 512
        rows.append(x[a:b])
    x0 = np.concatenate(rows, axis=0)
    x0[:, 0:n:3] = 0.0
    x0[:, n::3] = 0.0
    return x0


def cpu_baseline(cols, kw, n_elem, target_s=10.0):
    """Time the C oracle (port of the reference path) on this box's host cores: all cores the process may use, and
    ONE core (SURVEY §8(d) asks for both), bounded samples of the same workload."""
    import numpy as np

    from tests.helpers import oracle_beam

    ob = oracle_beam(cols, **kw)
    try:
        avail = len(os.sched_getaffinity(0))   # the cores this process may run on
    except AttributeError:
        avail = os.cpu_count() or 1
    # every core the box grants: the affinity mask, capped by the cgroup's CPU quota (CRB_BENCH_CPU_THREADS caps it further)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    # (threads beyond the quota are throttled, not refused: 256 threads on a 64-core share ran the sample at 2.4e7 against
    #  3.7e7 element-steps/s with 64 -- the baseline uses what the box actually grants)
    usable = avail if not quota else max(1, min(avail, int(round(quota))))
    cores = max(1, min(usable, int(os.environ.get("CRB_BENCH_CPU_THREADS", str(usable)))))
    t0 = time.perf_counter()
    ob.rk4_impulse(np.zeros(2 * ob.n), 2e-5, 20, 0.1)
    per_beam_step = (time.perf_counter() - t0) / 20
    steps = 200

    def sample(threads, budget_s):
        beams = int(max(threads, min(4096, budget_s * threads / (per_beam_step * steps))))
        beams = (beams // threads) * threads or threads
        amps = 0.1 * (1.0 + np.arange(beams) / beams)
        t0 = time.perf_counter()
        _, used = ob.rk4_impulse_batch(np.zeros((beams, 2 * ob.n)), 2e-5, steps, amps, n_threads=threads)
        wall = time.perf_counter() - t0
        return {"value": beams * n_elem * steps / wall, "unit": "beam-element-steps/s", "cores": int(used),
                "sample": f"{beams} beams x {n_elem} elem x {steps} RK4 steps, C oracle (oracle/crb_oracle.c), "
                          f"{'OpenMP over beams' if threads > 1 else 'one thread'}, {wall:.1f} s wall"}

    out = sample(cores, target_s)
    out["kind"] = "port"
    out["host_cpu_count"] = os.cpu_count()
    out["cores_available"] = avail
    out["cgroup_cpu_quota"] = quota
    out["cores_note"] = ("OpenMP threads = every core in the process's affinity mask" if not quota or quota >= avail else
                         f"OpenMP threads = the container's CPU quota ({quota:g} cores of the {avail} in the affinity mask: a 1-GPU box "
                         f"of the pool is a share of the host; threads beyond the quota are throttled)")
    out["one_core"] = sample(1, 0.6 * target_s)
    out["reference_python_1core"] = 3796.0   # BASELINE.md §2, measured in the survey container (reference proper)
    return out


def source_hash(files):
    """sha256 (first 16 hex digits) over the named kernel sources: profiles/traffic.json entries carry the hash of the sources
    they were measured on, and a counter figure is only printed for the code it belongs to."""
    import hashlib

    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def worker(args):
    import numpy as np
    import torch

    # stdout carries exactly ONE line (the JSON result): libraries that chat on fd 1 (RCCL prints its
    # version banner there) are redirected to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    cfg = CONFIGS[args.config]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the stepper has no CPU path")
    # CRB_BENCH_REHEARSAL=1: every rank on GPU 0 and the exchange over gloo -- the N > 1 code path (shards, chunked rollout
    # with the asynchronous all-gather, max-reduce) run on a ONE-GPU box; its number measures nothing (the ranks
    # share the GPU, the exchange goes through the host) and the line says so
    rehearsal = os.environ.get("CRB_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    # CRB_BENCH_FORCE_DIST=1: take the RCCL path (init, barrier, all-gather, max-reduce) even with one rank
    if world > 1 or os.environ.get("CRB_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import ctypes as C

    from continuum_robot import _native as nat
    from continuum_robot.batched import BeamEnsemble
    from continuum_robot.distributed import gather_terminal_states, impulse_amplitudes, shard_range, shard_sizes
    from continuum_robot.models.force_params import ForceParams
    from tests.helpers import block_errs, nitinol_columns, oracle_beam, rel_err, rollout_conditioning

    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    ne = cfg["elems"]
    B_total = cfg["beams"] * world if args.scaling == "weak" else cfg["total"]
    lo, hi = shard_range(B_total, world, rank)
    B = hi - lo
    sizes = shard_sizes(B_total, world)
    if B < 1:
        raise SystemExit(f"rank {rank} owns no beam of {B_total}")
    cols = nitinol_columns(ne, cfg["kind"])
    fp = ForceParams(fluid_density=1000.0 if cfg["drag"] else 0.0, enable_fluid_effects=cfg["drag"],
                     enable_gravity_effects=cfg["gravity"])
    okw = dict(fluid_density=fp.fluid_density, enable_fluid=cfg["drag"], enable_gravity=cfg["gravity"])
    params = cols
    if args.hetero:
        rng_h = np.random.default_rng(4321 + rank)
        params = []
        for _ in range(B):
            sE, sr, srho = rng_h.uniform(0.9, 1.1, 3)
            c = dict(cols)
            c["elastic_modulus"] = cols["elastic_modulus"] * sE
            c["density"] = cols["density"] * srho
            c["cross_area"] = cols["cross_area"] * sr**2
            c["moment_inertia"] = cols["moment_inertia"] * sr**4
            c["wetted_area"] = cols["wetted_area"] * sr
            params.append(c)
    # N > 1: the shard is stepped in chunks so that the exchange of chunk c overlaps the stepping of chunk c + 1
    # (continuum_robot.distributed.rollout_and_gather); every chunk is an ensemble of its own.  One rank: one ensemble.
    n_chunks = 1
    if dist and not cfg.get("lqr") and not cfg.get("implicit") and not args.hetero and min(sizes) == max(sizes):
        want = args.gather_chunks
        if want == 0:
            # chunk only when the exchange is a visible share of the rollout (a rule every rank evaluates identically):
            # one shard travels to each peer over its own xGMI link (~60 GB/s effective assumed, + 50 us of launch
            # latency) against the stepping time at the nominal single-GPU rate.  20 steps: 0.9 ms of exchange next to
            # 0.6 ms of stepping -> 4 chunks; 1000 steps: 3 % -> one blocking exchange (chunking costs ~3 % by itself)
            shard_bytes = B * 2 * 3 * ne * (8 if args.dtype == "f64" else 4)
            est_exchange = shard_bytes / 60e9 + 50e-6
            est_stepping = B * ne * args.steps / (3.7e10 if args.dtype == "f64" else 7.5e10)
            want = 4 if est_exchange > 0.15 * est_stepping else 1
        while want > 1 and (B % want or B // want < 512):
            want -= 1
        n_chunks = max(1, want)
    Bc = B // n_chunks
    t_plan = time.perf_counter()
    enss = [BeamEnsemble(params if not args.hetero else params[c * Bc:(c + 1) * Bc], Bc, force_params=fp, dtype=dtype,
                         device=f"cuda:{local_rank}") for c in range(n_chunks)]
    ens = enss[0]
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - t_plan) * 1e3

    amps = torch.as_tensor(impulse_amplitudes(B_total, lo, hi, cfg.get("amp", 0.1)), dtype=dtype, device=ens.device)
    amps_c = [amps[c * Bc:(c + 1) * Bc].contiguous() for c in range(n_chunks)]
    gain = None
    if cfg.get("lqr"):
        # the gain is replicated: rank 0 solves the Riccati equation (tens of CPU-seconds at 768 states), the
        # others receive it
        t_gain = time.perf_counter()
        if rank == 0:
            gain = torch.as_tensor(lqr_gain(ens), dtype=dtype, device=ens.device)
        else:
            gain = torch.empty((ens.n, 2 * ens.n), dtype=dtype, device=ens.device)
        if dist:
            dist.broadcast(gain, src=0)
        print(f"[bench] LQR gain {tuple(gain.shape)} ready after {time.perf_counter() - t_gain:.1f} s", file=sys.stderr)
    x0n = None
    x0_c = [None] * n_chunks
    if cfg["x0"]:
        x0n = initial_states(lo, hi, ens.n)
        x0_c = [enss[c].pack_state(x0n[c * Bc:(c + 1) * Bc]) for c in range(n_chunks)]

    def reset():
        for e, x0 in zip(enss, x0_c):
            if x0 is None:
                e.zero_state()
            else:
                e.state.copy_(x0)
                e.time = 0.0

    dt = cfg.get("dt", 2e-5)
    per_launch = args.launch_steps if args.launch_steps > 0 else args.steps
    per_launch = min(per_launch, args.steps)

    def advance_one(c, k):
        e = enss[c]
        if cfg.get("implicit"):
            e.step_implicit(k, dt, n_iter=2, impulse_amp=amps_c[c])
        elif gain is None:
            e.step(k, dt, impulse_amp=amps_c[c])
        else:
            e.step_feedback(k, dt, gain, impulse_amp=amps_c[c])

    def advance(k):
        for c in range(n_chunks):
            advance_one(c, k)

    # ---- warmup (untimed), then restore the initial state so the timed K steps are the
    # parity-checked trajectory (the shipped nonlinear element is only stable to ~1000 steps)
    reset()
    # untimed device spin-up before the W warmup steps: a fresh box starts with idle clocks and cold
    # code objects, which 100 steps (4 ms) do not cover (observed once: a first run at half speed)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:
        reset()
        advance(per_launch)
        torch.cuda.synchronize()
    reset()
    w_done = 0
    while w_done < args.warmup:
        k = min(per_launch, args.warmup - w_done)
        advance(k)
        w_done += k
    if dist:
        # warm the collective path too (RCCL builds its communicator / channels lazily on first use:
        # tens of ms that do not belong to the timed steps)
        from continuum_robot.distributed import assemble_chunks, rollout_and_gather

        if n_chunks > 1:
            rollout_and_gather(enss, lambda e: None)
        else:
            gather_terminal_states(ens.unpack_state(), sizes=sizes)
        warm = torch.zeros(1, dtype=torch.float64, device=ens.device)
        dist.all_reduce(warm, op=dist.ReduceOp.MAX)
        dist.barrier()

    events = []

    def timed_rollout():
        """EXACTLY --steps steps bracketed by barrier + synchronize on both sides; returns (max-over-ranks wall, gathered)."""
        reset()
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()

        def rollout(c):   # all launches of one chunk (the whole shard when it is not chunked)
            done = 0
            while done < args.steps:
                k = min(per_launch, args.steps - done)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()   # (the stepper launches on torch's current stream: BeamEnsemble._stream)
                advance_one(c, k)
                e1.record()
                events.append((e0, e1, k))
                done += k

        # the one exchange: RCCL all-gather of the terminal states in the reference's reduced ordering
        # ([B, 2n], no padding lanes: 50 MB per rank for config 3).  One rank has nobody to exchange with: the
        # states stay where they are (the layout conversion belongs to the exchange and is not run either).
        # With chunks the exchange of chunk c runs on RCCL's stream while chunk c + 1 is stepped.
        if n_chunks > 1:
            index = {id(e): c for c, e in enumerate(enss)}
            gathered = rollout_and_gather(enss, lambda e: rollout(index[id(e)]))
        else:
            rollout(0)
            gathered = gather_terminal_states(ens.unpack_state(), sizes=sizes) if dist else None
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        wall = time.perf_counter() - t_start
        if dist:
            tmax = torch.tensor([wall], dtype=torch.float64, device=ens.device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            wall = float(tmax.item())
        return wall, gathered

    walls = []
    wall, gathered = timed_rollout()
    walls.append(wall)
    # a region shorter than 50 ms is not a measurement to rank kernels by: repeat it (every rank derives the same
    # count from the max-reduced first wall; the first repeat is the slowest, hence the 25 % margin) and report the median
    repeats = args.repeats if args.repeats > 0 else max(1, min(1000, int(np.ceil(1.25 * MIN_TIMED_S / max(wall, 1e-6)))))
    for _ in range(repeats - 1):
        wall, gathered = timed_rollout()
        walls.append(wall)
    wall = float(np.median(walls))

    kernel_ms = [e0.elapsed_time(e1) for e0, e1, _ in events]
    full = [ms for (ms, (_, _, k)) in zip(kernel_ms, events) if k == per_launch] or kernel_ms
    avg_launch_s = float(np.mean(full)) * 1e-3

    # ---- the exchange alone (after the clock stopped; N > 1): how much of the timed region one blocking all-gather of this
    # rank's terminal states costs by itself -- beams never interact, so this is the only thing a multi-GPU run adds, and
    # at a few dozen steps it is as long as the stepping (median of 5, max over ranks)
    exchange_ms = None
    if dist:
        red_all = torch.cat([e.unpack_state() for e in enss], dim=0) if n_chunks > 1 else ens.unpack_state()
        ex = []
        for _ in range(5):
            torch.cuda.synchronize()
            dist.barrier()
            t_ex = time.perf_counter()
            gather_terminal_states(red_all, sizes=sizes)
            torch.cuda.synchronize()
            ex.append(time.perf_counter() - t_ex)
        tex = torch.tensor([float(np.median(ex))], dtype=torch.float64, device=ens.device)
        dist.all_reduce(tex, op=dist.ReduceOp.MAX)
        exchange_ms = float(tex.item()) * 1e3
        del red_all

    # ---- sanity / parity of what was just timed (after the clock stopped)
    state = torch.cat([e.unpack_state() for e in enss], dim=0) if n_chunks > 1 else ens.unpack_state()
    finite = bool(torch.isfinite(state).all())
    if isinstance(gathered, list):     # chunked exchange: back to global beam order, and it must equal what this rank holds
        gathered = assemble_chunks(gathered, world)
        assert torch.equal(gathered[lo:hi], state), "chunked all-gather does not reproduce this rank's states"
    check = {"finite": finite, "gathered_beams": int(gathered.shape[0]) if gathered is not None else int(state.shape[0])}
    if rank == 0:
        b = B - 1
        ob = oracle_beam(params[b] if args.hetero else cols, **okw)
        x0b = np.zeros(2 * ob.n) if x0n is None else x0n[b]
        if cfg.get("implicit"):
            ref = ob.implicit(x0b, dt, args.steps, n_iter=2, amp=float(amps[b].item()))
            check["simulated_seconds"] = args.steps * dt
            check["reference_lsoda_wall_s_per_simulated_s"] = 207.0   # BASELINE.md §2 (survey container, 1 core)
            g8 = os.path.join(ROOT, "tests", "golden", "g8_lsoda.npz")
            if os.path.exists(g8) and abs(args.steps * dt - 0.1) < 1e-9:
                z = np.load(g8)
                check["tip_w_lsoda_tight_reference_rhs"] = float(z["lin10_grav/x_tight"][-1][ob.n - 2])
        elif gain is None:
            ref = ob.rk4_impulse(x0b, dt, args.steps, float(amps[b].item()))
        else:
            ref = ob.rk4_feedback(x0b, dt, args.steps, gain.double().cpu().numpy(), amp=float(amps[b].item()))
        got_b = state[b].double().cpu().numpy()
        check["rel_err_vs_oracle_last_beam"] = rel_err(got_b, ref)
        # per DOF block (u, w, phi and their rates), each relative to its own largest entry (tests/helpers.block_errs)
        check["block_err_vs_oracle_last_beam"] = block_errs(got_b, ref, ens.free_index)
        if gain is None and cfg["kind"] == "nonlinear" and args.dtype == "f64":
            # how ill-conditioned the trajectory itself is: the oracle's own per-block response to a 4-ulp change of the
            # impulse amplitude (the axial blocks of long nonlinear chains are exponentially unstable beyond ~600 steps)
            check["oracle_block_sensitivity_4ulp"] = rollout_conditioning(ob, x0b, dt, args.steps, float(amps[b].item()))
        check["tip_w_last_beam"] = float(state[b, ens.n - 2].item())
        if args.verify_all and gain is None and not args.hetero:
            X0 = np.zeros((B, 2 * ob.n)) if x0n is None else x0n
            ref_all, _ = ob.rk4_impulse_batch(X0, dt, args.steps, amps.double().cpu().numpy())
            got_all = state.double().cpu().numpy()
            errs = np.linalg.norm(got_all - ref_all, axis=1) / np.maximum(np.linalg.norm(ref_all, axis=1), 1e-300)
            check["rel_err_vs_oracle_all_beams_max"] = float(errs.max())
            check["block_err_vs_oracle_all_beams_max"] = block_errs(got_all, ref_all, ens.free_index)
            check["beams_compared"] = int(B)

    # ---- roofline of the dominant kernel
    if gain is None:
        # `frac` is the task contract's figure: ALGORITHMIC bytes (96 / 48 B per element-step) over the launch time
        # against the HBM peak.  It is a nominal figure: the state lives in registers for the whole launch, real HBM traffic is
        # `traffic` (`hbm_real_frac` of the peak), and what binds is named in `bound` with its own utilisation.
        algo_bytes_launch = BYTES_PER_ELEM_STEP[args.dtype] * Bc * ne * per_launch   # (per launch: one chunk's beams)
        achieved = algo_bytes_launch / avg_launch_s / 1e9
        roofline = {"bound": "valu_fp64" if args.dtype == "f64" else "lds_issue",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "frac_is": "algorithmic bytes / launch time / HBM peak (contract accounting; the limiter is `bound`)",
                    "traffic": None, "hbm_real_frac": None,
                    "kernel": "crb_implicit_lean_kernel" if cfg.get("implicit") else "crb_step_lean_kernel",
                    "avg_launch_ms": avg_launch_s * 1e3, "algorithmic_bytes_per_launch": algo_bytes_launch,
                    "valu_issue_frac": None}
    else:
        # config 5: the closed loop is ONE persistent launch (crb_loop_kernel): four fp64-MFMA products U = (R - X) K^T per step
        # inside it.  The roofline is the matrix work of the launch over its duration against the fp64 matrix peak.
        n = ens.n
        gemm_flop = 2.0 * B * (2 * n) * n
        peak_tf = MFMA_F64_PEAK_TF if args.dtype == "f64" else MFMA_F32_PEAK_TF
        step_s = avg_launch_s / per_launch
        flop_launch = 4 * gemm_flop * per_launch
        ach_tf = flop_launch / avg_launch_s / 1e12
        persistent = int(ens.feedback_path() == "persistent")
        roofline = {"bound": "mfma_f64" if args.dtype == "f64" else "mfma_f32", "achieved": ach_tf, "peak": peak_tf,
                    "unit": "TFLOP/s", "frac": ach_tf / peak_tf, "traffic": None, "hbm_real_frac": None,
                    "kernel": "crb_loop_kernel" if persistent else "crb_feedback_ws_kernel + crb_stage_lean_kernel",
                    "avg_launch_ms": avg_launch_s * 1e3, "steps_per_launch": per_launch,
                    "algorithmic_flop_per_launch": flop_launch,
                    "launches_per_step": (1.0 / per_launch) if persistent else 8,
                    "us_per_step": step_s * 1e6,
                    "matrix_time_floor_us_per_step": 4 * gemm_flop / (peak_tf * 1e12) * 1e6,
                    "feedback_status": ens.feedback_status()}

    if rank == 0:
        value = B_total * ne * args.steps / wall
        out = {
            "metric": "beam-element-steps/s",
            "value": value,
            "unit": "beam-element-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "repeats": repeats,
            "timed_region_ms": {"median": wall * 1e3, "min": min(walls) * 1e3, "max": max(walls) * 1e3,
                                "total": sum(walls) * 1e3},
            "config": {"workload": cfg["label"] + (" [heterogeneous: per-beam coefficients]" if args.hetero else ""),
                       "beams_per_gpu": sizes, "beams_total": B_total, "elements": ne,
                       "dt": dt, "steps_per_launch": per_launch, "parallelism": f"beam-shard x{world}",
                       "ranks_seen": int(dist.get_world_size()) if dist else 1,
                       "launched_by": "bench.py launcher" if os.environ.get("CRB_BENCH_LAUNCHED") else
                                      ("external launcher" if "WORLD_SIZE" in os.environ else "single process"),
                       "collective": ("all_gather_into_tensor(terminal states [B,2n]" +
                                      (", padded to the largest shard)" if min(sizes) != max(sizes) else ")") +
                                      (f" in {n_chunks} chunks, chunk c overlapped with the stepping of chunk c+1" if n_chunks > 1 else ""))
                                     if dist else "none",
                       "gather_chunks": n_chunks,
                       **({"exchange_alone_ms": exchange_ms} if exchange_ms is not None else {}),
                       "plan_ms": plan_ms,
                       **({"rehearsal": "all ranks on GPU 0, exchange over gloo: NOT a measurement"} if rehearsal else {})},
            "roofline": roofline,
            "check": check,
        }
        if cfg.get("implicit") and world == 1 and not args.hetero:
            # the example's own call on this configuration (examples/example_utilities.py:153-159): 1 s, method="LSODA",
            # scipy's default tolerances, t_eval every millisecond -- the step size controlled by rtol / atol, once inside
            # the kernel (ONE launch, crb_solve_controlled) and once by the host loop over fixed-step launches.  Wall times
            # outside the timed region; the first device run also factorises the table ladder (reported apart).
            t_eval = np.arange(0.0, 1.0005, 0.001)
            walls_ivp = {}
            for ctrl, reps in (("device", 2), ("host", 1)):
                for rep in range(reps):
                    e1 = BeamEnsemble(params, Bc, force_params=fp, dtype=dtype, device=f"cuda:{local_rank}") if rep == 0 else e1
                    e1.zero_state()
                    torch.cuda.synchronize()
                    t_a = time.perf_counter()
                    sol = e1.solve_ivp((0.0, 1.0005), t_eval, method="LSODA", impulse_amp=amps_c[0], controller=ctrl)
                    torch.cuda.synchronize()
                    walls_ivp[(ctrl, rep)] = time.perf_counter() - t_a
                if ctrl == "device":
                    steps_dev, tip = int(np.sum(sol.substeps_per_beam[0])), float(sol.y[0, e1.n - 2, -1].item())
            out["solve_ivp_1s_default_tolerances"] = {
                "wall_s": walls_ivp[("device", 1)], "wall_s_first_call_with_table_ladder": walls_ivp[("device", 0)],
                "wall_s_host_loop_controller": walls_ivp[("host", 0)], "launches": 1, "fine_steps_accepted": steps_dev,
                # (every accepted fine solution of 2m steps comes with a coarse one of m steps: 1.5 steps taken per accepted step, more after a rejection)
                "us_per_implicit_step_taken": walls_ivp[("device", 1)] / (1.5 * steps_dev) * 1e6,
                "tip_w_1s": tip, "tip_w_1s_lsoda_tight": -0.41624141,
                "reference_wall_s": 207.0, "reference_source": "BASELINE.md section 2 (scipy LSODA over the reference RHS, one core)"}
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file) and world == 1 and n_chunks == 1:
            try:
                tj = json.load(open(traffic_file))
                # HBM traffic of a launch is one read + one write of the state, whatever the number of fused steps: keyed by
                # config and dtype only (PMC passes of profiles/r03_profile.sh).  An entry names the kernel sources it was
                # measured on and their hash: if the sources have changed since, the figure is not this build's and is withheld.
                key = f"{args.config}:{args.dtype}" + (":hetero" if args.hetero else "")
                ent = tj.get(key)
                if ent and ent.get("sources") and source_hash(ent["sources"]) != ent.get("sources_sha256_16"):
                    print(f"[bench] WARNING: profiles/traffic.json[{key}] was measured on other kernel sources "
                          f"({ent.get('sources_sha256_16')} != {source_hash(ent['sources'])}): traffic withheld -- re-run profiles/r03_profile.sh",
                          file=sys.stderr)
                    out["roofline"]["traffic_stale"] = True
                    ent = None
                if ent:
                    out["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = ent.get("source")
                    out["roofline"]["hbm_real_frac"] = ent["hbm_bytes_per_launch"] / avg_launch_s / (HBM_PEAK_GBS * 1e9)
                    if "launch_steps" in ent and ent["launch_steps"] != per_launch:
                        # (the persistent closed-loop launch moves bytes per STEP through L2 / MALL; its HBM figure belongs to the
                        #  launch length it was measured at)
                        out["roofline"]["traffic"] = None
                        out["roofline"]["hbm_real_frac"] = None
                        out["roofline"]["traffic_note"] = f"measured at {ent['launch_steps']} steps per launch; this run: {per_launch}"
                    if "phases_us_per_stage" in ent:
                        out["roofline"]["phases_us_per_stage"] = ent["phases_us_per_stage"]
                        out["roofline"]["phases_source"] = ent.get("phases_source")
                    if "valu_instr_per_elem_step" in ent and gain is None:
                        # the limiter that actually binds: vector-ALU issue slots (one wave instruction per
                        # SIMD every 4 cycles, 1024 SIMDs, nominal 2.4 GHz), instruction count from the PMC run
                        # (fp64 plans: every vector instruction of that kernel holds the port for 4 cycles, measured.
                        #  fp32 arithmetic issues faster than that -- the same formula gives 1.19 for config 4 -- so
                        #  the fp32 line carries the instruction rate instead of a fraction)
                        lane_instr = ent["valu_instr_per_elem_step"] * Bc * ne * per_launch / avg_launch_s
                        if args.dtype == "f64":
                            out["roofline"]["valu_issue_frac"] = lane_instr / (1024 * 64 * VALU_CLOCK_HZ / 4)
                        else:
                            out["roofline"]["valu_wave_instr_per_s_per_simd"] = lane_instr / 64 / 1024
                        out["roofline"]["valu_source"] = ent.get("valu_source")
            except Exception as e:  # a malformed side file must not cost the run its result line
                print(f"[bench] profiles/traffic.json ignored: {e}", file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cols, okw, ne)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist:
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("CRB_BENCH_FORCE_DIST") == "1"):
        # plain `python bench.py --gpus N`: this process is the launcher.  Nothing above imported torch or touched HIP.
        return launch(args.gpus, argv)
    worker(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
