#!/usr/bin/env python3
"""bench.py -- beam-element-steps/s of the fused RK4 beam stepper on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one classical RK4 time step (dt = 2e-5 s) of the whole batch.  Workload at N = 1 is
BASELINE.json's metric configuration (configs[2]): 4096 beams x 256 nonlinear Euler-Bernoulli
elements + fluid drag (Nitinol constants of examples/example_utilities.py:25-34, FIXED at node 0,
zero initial state, per-beam tip impulse 0.1*(1 + b/B) N for t < 0.01 s), fp64.  With N > 1 every
rank owns 4096 beams of the 4096*N ensemble (independent units, weak scaling, no per-step
communication) and the terminal states are all-gathered over RCCL inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit, ms_per_step,
`roofline` (algorithmic HBM bytes = 96 B per element-step) and `cpu_baseline`
(the C oracle -- a port of the reference path -- on this box's host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "continuum-robot_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
VALU_CLOCK_HZ = 2.4e9          # nominal engine clock (the chip holds ~2.2 GHz under this fp64 load)
BYTES_PER_ELEM_STEP = {torch.float64: 96.0, torch.float32: 48.0}

CONFIGS = {
    # name: (beams per GPU, elements, element type, force kwargs, random x0, default steps)
    "config3": dict(beams=4096, elems=256, kind="nonlinear", drag=True, gravity=False, x0=False,
                    label="4096 beams x 256 elem, nonlinear Euler-Bernoulli + fluid drag, fp64"),
    # BASELINE config 4: the same ensemble in fp32 (batch-sharded across the node's GPUs); 200 steps by default,
    # beyond which single precision drifts past 1e-5 from the fp64 oracle
    "config4": dict(beams=4096, elems=256, kind="nonlinear", drag=True, gravity=False, x0=False, dtype="f32", steps=200,
                    label="4096 beams x 256 elem, nonlinear Euler-Bernoulli + fluid drag, fp32"),
    "config2": dict(beams=1024, elems=64, kind="linear", drag=False, gravity=True, x0=True,
                    label="1024 beams x 64 elem, linear + gravity, fp64"),
    # LQR rollout ensemble (BASELINE config 5: 2048 beams/GPU): state feedback u = K(0 - x) at every RK4
    # stage (stage-split path: one GEMM + one stage kernel per stage).  dt = 5e-6: the closed loop has
    # |lambda|max = 3.2e5 1/s, RK4 is unstable at the open-loop dt = 2e-5.
    "config5": dict(beams=2048, elems=128, kind="linear", drag=False, gravity=True, x0=True, lqr=True, dt=5e-6,
                    amp=10.0, label="2048 beams/GPU x 128 elem, linear + gravity + LQR feedback per stage, fp64"),
}


def lqr_gain(ens):
    """LQR gain of lqr_control.py:46-84 for the ensemble's (linear) beam: Q = diag(100 I, 10 I), R = I."""
    from continuum_robot.control import LinearQuadraticRegulator

    K, M = ens.plan.stiffness(), ens.plan.mass()
    n = K.shape[0]
    Q = np.eye(2 * n)
    Q[:n, :n] *= 100
    Q[n:, n:] *= 10
    return LinearQuadraticRegulator(K, M, Q, np.eye(n)).compute_gain_matrix()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed RK4 steps (default 1000; 200 for config4)")
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="config3", choices=sorted(CONFIGS))
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"], help="default: the config's (f64; config4: f32)")
    ap.add_argument("--launch-steps", type=int, default=100,
                    help="RK4 steps fused per launch (0 = all of --steps); warmup uses launches of the same size, so "
                         "every stepper launch of a run is identical and rocprof's per-kernel average is the launch time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify-all", action="store_true",
                    help="after timing, compare EVERY beam of rank 0 with the oracle (open-loop configs; tens of CPU-seconds)")
    ap.add_argument("--hetero", action="store_true",
                    help="heterogeneous variant (SURVEY 8(d)): per-beam E, rho, r scaled by U(0.9, 1.1), seed 4321")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    if args.dtype is None:
        args.dtype = cfg.get("dtype", "f64")
    if args.steps is None:
        args.steps = cfg.get("steps", 1000)
    return args


def cpu_baseline(cols, kw, n_elem, target_s=15.0):
    """Time the C oracle (port of the reference path) on this box's host cores, bounded sample."""
    from tests.helpers import oracle_beam

    ob = oracle_beam(cols, **kw)
    # host cores this process may use (a 1-GPU box's share is 16 of the host's cores)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("CRB_BENCH_CPU_THREADS", "16"))))
    # calibrate on one beam, then size the sample for ~target_s of wall time on all cores
    t0 = time.perf_counter()
    ob.rk4_impulse(np.zeros(2 * ob.n), 2e-5, 20, 0.1)
    per_beam_step = (time.perf_counter() - t0) / 20
    steps = 200
    beams = int(max(cores, min(4096, target_s * cores / (per_beam_step * steps))))
    beams = (beams // cores) * cores or cores
    amps = 0.1 * (1.0 + np.arange(beams) / beams)
    t0 = time.perf_counter()
    _, used = ob.rk4_impulse_batch(np.zeros((beams, 2 * ob.n)), 2e-5, steps, amps, n_threads=cores)
    wall = time.perf_counter() - t0
    return {"value": beams * n_elem * steps / wall, "unit": "beam-element-steps/s", "cores": int(used), "kind": "port",
            "sample": f"{beams} beams x {n_elem} elem x {steps} RK4 steps, C oracle (oracle/crb_oracle.c), "
                      f"OpenMP over beams, {wall:.1f} s wall",
            "reference_python_1core": 3796.0}  # BASELINE.md §2, measured in the survey container


def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON result): libraries that chat on fd 1 (RCCL prints its
    # version banner there) are redirected to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    cfg = CONFIGS[args.config]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the stepper has no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    # CRB_BENCH_FORCE_DIST=1: take the RCCL path (init, barrier, all-gather, max-reduce) even with one rank
    if world > 1 or os.environ.get("CRB_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from continuum_robot.batched import BeamEnsemble
    from continuum_robot.distributed import gather_terminal_states, impulse_amplitudes, shard_range
    from continuum_robot.models.force_params import ForceParams
    from tests.helpers import nitinol_columns, oracle_beam, rel_err

    dtype = torch.float64 if args.dtype == "f64" else torch.float32
    B, ne = cfg["beams"], cfg["elems"]
    B_total = B * world
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
    t_plan = time.perf_counter()
    ens = BeamEnsemble(params, B, force_params=fp, dtype=dtype, device=f"cuda:{local_rank}")
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - t_plan) * 1e3

    lo, hi = shard_range(B_total, world, rank)
    assert hi - lo == B
    amps = torch.as_tensor(impulse_amplitudes(B_total, lo, hi, cfg.get("amp", 0.1)), dtype=dtype, device=ens.device)
    gain = None
    if cfg.get("lqr"):
        t_gain = time.perf_counter()
        gain = torch.as_tensor(lqr_gain(ens), dtype=dtype, device=ens.device)
        print(f"[bench] LQR gain {tuple(gain.shape)} solved in {time.perf_counter() - t_gain:.1f} s", file=sys.stderr)
    x0 = None
    if cfg["x0"]:
        rng = np.random.default_rng(1234 + rank)
        n = ens.n
        x0n = np.concatenate([rng.normal(0, 1e-5, (B, n)), rng.normal(0, 1e-3, (B, n))], axis=1)
        x0n[:, 0:n:3] = 0.0
        x0n[:, n::3] = 0.0
        x0 = ens.pack_state(x0n)

    def reset():
        if x0 is None:
            ens.zero_state()
        else:
            ens.state.copy_(x0)
            ens.time = 0.0

    dt = cfg.get("dt", 2e-5)
    per_launch = args.launch_steps if args.launch_steps > 0 else args.steps

    def advance(k):
        if gain is None:
            ens.step(k, dt, impulse_amp=amps)
        else:
            ens.step_feedback(k, dt, gain, impulse_amp=amps)

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
        gather_terminal_states(ens.unpack_state())
        warm = torch.zeros(1, dtype=torch.float64, device=ens.device)
        dist.all_reduce(warm, op=dist.ReduceOp.MAX)
        dist.barrier()
    reset()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()

    events = []
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    done = 0
    while done < args.steps:
        k = min(per_launch, args.steps - done)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        advance(k)
        e1.record()
        events.append((e0, e1, k))
        done += k
    # the one exchange: RCCL all-gather of the terminal states in the reference's reduced ordering
    # ([B, 2n], no padding lanes: 50 MB per rank for config 3); no-op at N = 1
    gathered = gather_terminal_states(ens.unpack_state())
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    wall = time.perf_counter() - t_start
    if dist:
        tmax = torch.tensor([wall], dtype=torch.float64, device=ens.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())

    kernel_ms = [e0.elapsed_time(e1) for e0, e1, _ in events]
    full = [ms for (ms, (_, _, k)) in zip(kernel_ms, events) if k == per_launch] or kernel_ms
    avg_launch_s = float(np.mean(full)) * 1e-3
    algo_bytes_launch = BYTES_PER_ELEM_STEP[dtype] * B * ne * min(per_launch, args.steps)
    achieved = algo_bytes_launch / avg_launch_s / 1e9

    # ---- sanity / parity of what was just timed (after the clock stopped)
    state = ens.unpack_state()
    finite = bool(torch.isfinite(state).all())
    check = {"finite": finite, "gathered_beams": int(gathered.shape[0])}
    if rank == 0:
        b = B - 1
        ob = oracle_beam(params[b] if args.hetero else cols, **okw)
        x0b = np.zeros(2 * ob.n) if x0 is None else x0n[b]
        if gain is None:
            ref = ob.rk4_impulse(x0b, dt, args.steps, float(amps[b].item()))
        else:
            ref = ob.rk4_feedback(x0b, dt, args.steps, gain.cpu().numpy(), amp=float(amps[b].item()))
        check["rel_err_vs_oracle_last_beam"] = rel_err(state[b].double().cpu().numpy(), ref)
        check["tip_w_last_beam"] = float(state[b, ens.n - 2].item())
        if args.verify_all and gain is None and not args.hetero:
            X0 = np.zeros((B, 2 * ob.n)) if x0 is None else x0n
            ref_all, _ = ob.rk4_impulse_batch(X0, dt, args.steps, amps.double().cpu().numpy())
            got_all = state.double().cpu().numpy()
            errs = np.linalg.norm(got_all - ref_all, axis=1) / np.maximum(np.linalg.norm(ref_all, axis=1), 1e-300)
            check["rel_err_vs_oracle_all_beams_max"] = float(errs.max())
            check["beams_compared"] = int(B)

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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": cfg["label"] + (" [heterogeneous: per-beam coefficients]" if args.hetero else ""), "beams_per_gpu": B, "beams_total": B_total, "elements": ne,
                       "dt": dt, "steps_per_launch": min(per_launch, args.steps), "parallelism": f"beam-shard x{world}",
                       "collective": "all_gather_into_tensor(terminal states [B,2n])" if world > 1 else "none",
                       "plan_ms": plan_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": ("crb_stage_lean_kernel + crb_feedback_kernel" if gain is not None
                                    else "crb_step_lean_kernel"),
                         "avg_launch_ms": avg_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": algo_bytes_launch,
                         "valu_issue_frac": None},
            "check": check,
        }
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                tj = json.load(open(traffic_file))
                key = f"{args.config}:{args.dtype}:{min(per_launch, args.steps)}"
                if key in tj:
                    out["roofline"]["traffic"] = tj[key]["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = tj[key].get("source")
                    if "valu_instr_per_elem_step" in tj[key]:
                        # the limiter that actually binds: vector-ALU issue slots (one wave instruction per
                        # SIMD every 4 cycles, 1024 SIMDs, nominal 2.4 GHz), instruction count from the PMC run
                        lane_instr = tj[key]["valu_instr_per_elem_step"] * B * ne * min(per_launch, args.steps) / avg_launch_s
                        out["roofline"]["valu_issue_frac"] = lane_instr / (1024 * 64 * VALU_CLOCK_HZ / 4)
                        out["roofline"]["valu_source"] = tj[key].get("valu_source")
            except Exception:
                pass
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cols, okw, ne)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
