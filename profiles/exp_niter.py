"""Experiment: modified-Newton iterations per implicit step UNDER the in-kernel controller.  The controller's steps are small
(3 .. 6 us at the default tolerances), so the previous step's iterate is already converged after one iteration; is that also
true at loose tolerances and for nonlinear elements?  Error of every run against a tight run of ours (rtol 1e-6, 3 iterations),
in units of the run's own tolerance band.  usage: python profiles/exp_niter.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
from tests.helpers import nitinol_columns
from tests.test_gpu_parity import ensemble

cases = [("10 linear + gravity", nitinol_columns(10, "linear"), dict(enable_gravity=True), 0.2),
         ("6 nonlinear + fluid", nitinol_columns(6, "nonlinear"), dict(fluid_density=1000.0, enable_fluid=True), 0.1),
         ("10 nonlinear + gravity", nitinol_columns(10, "nonlinear"), dict(enable_gravity=True), 0.1),
         ("10 mixed + fluid + gravity", nitinol_columns(10, ["nonlinear" if i % 3 else "linear" for i in range(10)]), dict(enable_gravity=True, fluid_density=1000.0, enable_fluid=True), 0.1),
         ("64 linear + gravity", nitinol_columns(64, "linear"), dict(enable_gravity=True), 0.03)]
for label, cols, kw, T in cases:
    n_int = int(round(T / 1e-3))
    amps = np.array([0.1, 1.0])
    ref = ensemble(cols, 2, kw)
    snaps, _, _ = ref.solve_controlled(n_int, 1e-3, rtol=1e-6, atol=1e-9, impulse_amp=amps, n_iter=3, t0=0.0, max_rungs=18)
    yr = ref.unpack_snapshots(snaps).cpu().numpy()
    for rtol, atol in ((1e-3, 1e-6), (1e-2, 1e-5), (1e-1, 1e-4)):
        row = []
        for n_iter in (1, 2):
            ens = ensemble(cols, 2, kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            try:
                snaps, st, used = ens.solve_controlled(n_int, 1e-3, rtol=rtol, atol=atol, impulse_amp=amps, n_iter=n_iter, t0=0.0)
            except RuntimeError as e:
                row.append(f"n_iter={n_iter}: FAILED"); continue
            torch.cuda.synchronize(); wall = time.perf_counter() - t0
            y = ens.unpack_snapshots(snaps).cpu().numpy()
            n = ens.n
            e = np.abs(y - yr) / (atol + rtol * np.abs(yr))
            row.append(f"n_iter={n_iter}: {wall:.3f} s, {int(used.sum())} steps, pos {e[:, :, :n].max():.2f} vel {e[:, :, n:].max():.1f} bands")
        print(f"{label:28s} rtol {rtol:g}: " + " | ".join(row), flush=True)
