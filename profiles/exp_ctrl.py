"""Experiment: the in-kernel step-size controller (crb_solve_controlled) against the host-loop controller and golden G8 / G6.
usage: python profiles/exp_ctrl.py"""
import os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "continuum-robot_amd"))
sys.path.insert(0, ROOT)
from tests.helpers import beam_columns, force_kwargs, oracle_beam      # noqa: E402
from tests.test_gpu_parity import ensemble                              # noqa: E402

g8 = np.load(os.path.join(ROOT, "tests/golden/g8_lsoda.npz"))
g6 = np.load(os.path.join(ROOT, "tests/golden/g6_lqr_loop.npz"))


def blocks(n):
    idx = np.arange(n)
    return {"u": idx[0::3], "w": idx[1::3], "phi": idx[2::3], "du": n + idx[0::3], "dw": n + idx[1::3], "dphi": n + idx[2::3]}


def run(name, T, times, tight, dflt, B=2, **tol):
    cols, kw = beam_columns(g8, name), force_kwargs(g8, name)
    t_eval = np.arange(0.0, T + 0.0005, 0.001)
    for ctrl in ("device", "device-packed", "host"):
        ens = ensemble(cols, B, kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sol = ens.solve_ivp((0.0, T + 0.0005), t_eval, method="LSODA", impulse_amp=np.full(B, 0.1), controller=ctrl, **tol)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        y = sol.y.cpu().numpy()
        n = ens.n
        print(f"{name} T={T} {ctrl}: wall {wall:.3f} s, substeps min/max {min(sol.substeps)}/{max(sol.substeps)} sum {sum(sol.substeps)}", tol)
        for ti, t in enumerate(times):
            band = 1e-6 + 1e-3 * np.abs(tight[ti])
            ours = np.abs(y[0][:, int(round(t / 0.001))] - tight[ti]) / band
            lsoda = np.abs(dflt[ti] - tight[ti]) / band
            msg = " ".join(f"{k}:{ours[ix].max():.2f}/{lsoda[ix].max():.2f}" for k, ix in blocks(n).items())
            print(f"   t={t:.2f} rms {np.sqrt((ours**2).mean()):.1f}/{np.sqrt((lsoda**2).mean()):.1f}  {msg}")


for name in ("lin10_grav", "lin6_fluid", "mixed6_fluid"):
    run(name, float(g8[name + "/times"][-1]), g8[name + "/times"], g8[name + "/x_tight"], g8[name + "/x_default_tol"])
run("lin10_grav", 0.1, g8["lin10_grav/times"], g8["lin10_grav/x_tight"], g8["lin10_grav/x_default_tol"], rtol=1e-6, atol=1e-9)
run("lin10_grav", 1.0, g8["lin10_grav_1s/times"], g8["lin10_grav_1s/x_tight"], g8["lin10_grav_1s/x_default_tol"])
run("lin10_grav", 1.0, g8["lin10_grav_1s/times"], g8["lin10_grav_1s/x_tight"], g8["lin10_grav_1s/x_default_tol"], B=4096)

# closed loop
from scipy.integrate import solve_ivp
for name, T in (("lqr6", 0.03), ("lqr24", 0.012)):
    cols, kw = beam_columns(g6, name), force_kwargs(g6, name)
    K, amp = g6[f"{name}/gain"], float(g6[f"{name}/amp"])
    ob = oracle_beam(cols, **kw)
    n = ob.n

    def closed_loop(t, x):
        u = -K @ x
        if t < 0.01:
            u[n - 2] += amp
        return ob.rhs(x, u)

    t_eval = np.arange(0.0, T + 0.0005, 0.001)
    ref = solve_ivp(closed_loop, (0.0, t_eval[-1]), np.zeros(2 * n), method="LSODA", t_eval=t_eval, rtol=1e-8, atol=1e-10)
    for ctrl in ("device", "host"):
        ens = ensemble(cols, 3, kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sol = ens.solve_ivp((0.0, t_eval[-1]), t_eval, method="LSODA", rtol=1e-8, atol=1e-10, impulse_amp=np.full(3, amp), gain=K, controller=ctrl)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        y = sol.y.cpu().numpy()
        tight = np.abs(y[0] - ref.y) / (1e-10 + 1e-8 * np.abs(ref.y))
        print(f"{name} {ctrl}: wall {wall:.3f} s substeps {min(sol.substeps)}..{max(sol.substeps)}; default band {np.max(np.abs(y[0] - ref.y) / (1e-6 + 1e-3 * np.abs(ref.y))):.3g}; "
              f"tight pos {tight[:n].max():.3g} vel {tight[n:].max():.3g}; equal beams {np.array_equal(y[0], y[2])}")
