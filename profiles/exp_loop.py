"""Development check + timing of the persistent closed-loop stepper (csrc/crb_loop.h) against the oracle and the
stage-split path.  python profiles/exp_loop.py [check|time] ..."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "continuum-robot_amd")):
    sys.path.insert(0, p)
from continuum_robot.batched import BeamEnsemble  # noqa: E402
from continuum_robot.models.force_params import ForceParams  # noqa: E402
from tests.helpers import block_errs, nitinol_columns, oracle_beam  # noqa: E402


def ens_of(n_e, B, kind="linear", grav=True, drag=False):
    fp = ForceParams(fluid_density=1000.0 if drag else 0.0, enable_fluid_effects=drag, enable_gravity_effects=grav)
    return BeamEnsemble(nitinol_columns(n_e, kind), B, force_params=fp, device="cuda:0")


def check(n_e=128, B=70, steps=12, kind="linear", grav=True, drag=False, with_ref=False, groups=None):
    rng = np.random.default_rng(n_e + B)
    os.environ["CRB_LOOP"] = "1"
    if groups:
        os.environ["CRB_LOOP_MAX_GROUPS"] = str(groups)
    else:
        os.environ.pop("CRB_LOOP_MAX_GROUPS", None)
    ens = ens_of(n_e, B, kind, grav, drag)
    n = ens.n
    gain = rng.normal(0.0, 2e-2, (n, 2 * n))
    ref = rng.normal(0.0, 1e-4, (B, 2 * n)) if with_ref else None
    x0 = rng.normal(0.0, 1e-4, (B, 2 * n))
    amps = 0.05 * (1.0 + np.arange(B) / B)
    dt = 2e-5
    ens.set_state(x0)
    ens.step_feedback(steps, dt, gain, reference=ref, impulse_amp=amps)
    st = ens.feedback_status()
    got = ens.unpack_state().cpu().numpy()
    os.environ["CRB_LOOP"] = "0"
    e2 = ens_of(n_e, B, kind, grav, drag)
    e2.set_state(x0)
    e2.step_feedback(steps, dt, gain, reference=ref, impulse_amp=amps)
    old = e2.unpack_state().cpu().numpy()
    ob = oracle_beam(nitinol_columns(n_e, kind), fluid_density=1000.0 if drag else 0.0, enable_fluid=drag, enable_gravity=grav)
    worst, worst_old = 0.0, 0.0
    for b in sorted(set([0, 1, B // 2, 63 % B, 64 % B, B - 1])):
        want = ob.rk4_feedback(x0[b], dt, steps, gain, reference=None if ref is None else ref[b], amp=amps[b])
        worst = max(worst, max(block_errs(got[b], want, ens.free_index).values()))
        worst_old = max(worst_old, max(block_errs(old[b], want, ens.free_index).values()))
    vs_old = max(block_errs(got, old, ens.free_index).values())
    print(f"check n_e={n_e} B={B} steps={steps} {kind} grav={grav} drag={drag} ref={with_ref} groups={groups}: status {st}, "
          f"finite {np.isfinite(got).all()}, worst block err vs oracle {worst:.2e} (stage-split {worst_old:.2e}), vs stage-split {vs_old:.2e}",
          flush=True)
    return st == 0 and worst < 1e-9


def timeit(n_e=128, B=2048, steps=100, reps=5):
    rng = np.random.default_rng(7)
    out = {}
    for mode in ("1", "0"):
        os.environ["CRB_LOOP"] = mode
        ens = ens_of(n_e, B)
        n = ens.n
        gain = rng.normal(0.0, 2e-2, (n, 2 * n))
        x0 = rng.normal(0.0, 1e-5, (B, 2 * n))
        amps = 10.0 * (1.0 + np.arange(B) / B)
        ens.set_state(x0)
        ens.step_feedback(steps, 5e-6, gain, impulse_amp=amps)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            ens.set_state(x0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ens.step_feedback(steps, 5e-6, gain, impulse_amp=amps)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        us = min(ts) / steps * 1e6
        out[mode] = us
        print(f"time n_e={n_e} B={B} CRB_LOOP={mode} fences={os.environ.get('CRB_LOOP_FENCES', '0')}: {us:.1f} us/step "
              f"({B * n_e / us * 1e6:.3e} element-steps/s), status {ens.feedback_status()}", flush=True)
    return out


def prof(n_e=128, B=2048, steps=50):
    """phase breakdown of a CRB_LOOP_PROF=1 build (make fast EXTRA=-DCRB_LOOP_PROF=1, CRB_LIB_PATH=..._fast.so)"""
    os.environ["CRB_LOOP"] = "1"
    rng = np.random.default_rng(7)
    ens = ens_of(n_e, B)
    n = ens.n
    gain = rng.normal(0.0, 2e-2, (n, 2 * n))
    ens.set_state(rng.normal(0.0, 1e-5, (B, 2 * n)))
    for _ in range(2):
        ens.step_feedback(steps, 5e-6, gain, impulse_amp=np.ones(B))
        torch.cuda.synchronize()
    words = ens._feedback_work[:128].cpu().numpy().view(np.uint64)
    tot = words[1:11].astype(np.float64)
    nwg = ((B + 63) // 64) * (8 if n_e > 64 else 4)
    per_stage = tot / nwg / (4 * steps) * 0.01   # us (100 MHz ticks)
    names = ["gemm", "reduce+store U", "hand-off A", "stage(rest)", "hand-off B", "loop overhead", "round: issue", "round: rhs", "round: update+stores", "-"]
    print("prof per stage (us): " + ", ".join(f"{k} {v:.2f}" for k, v in zip(names, per_stage)) + f"; sum {per_stage.sum():.2f}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what == "prof":
        prof()
        sys.exit(0)
    ok = True
    if what in ("check", "all"):
        ok &= check(128, 70, 12)
        ok &= check(128, 70, 12, with_ref=True)
        ok &= check(128, 200, 8, groups=2)
        ok &= check(100, 64, 8, kind="nonlinear", grav=False, drag=True)
        ok &= check(64, 130, 8)
        ok &= check(40, 65, 8, with_ref=True, grav=False)
    if what in ("time", "all"):
        timeit(128, 2048, 100)
        os.environ["CRB_LOOP_FENCES"] = "1"
        timeit(128, 2048, 100)
        os.environ.pop("CRB_LOOP_FENCES")
        timeit(128, 16384, 20, reps=3)
        timeit(64, 2048, 100)
    print("OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)
