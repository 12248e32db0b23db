"""Latency of the single-beam closures (the call pattern of scipy.solve_ivp over the drop-in classes) and the wall
time of the reference's own example integration through them: BASELINE config 1 (10 linear elements + gravity,
examples/beam_comparison_gravity.py) with solve_ivp(method="LSODA"), as examples/example_utilities.py:153-159 does."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, pandas as pd
from scipy.integrate import solve_ivp
from continuum_robot.models.dynamic_beam_model import DynamicEulerBernoulliBeam
from continuum_robot.models.force_params import ForceParams
from tests.helpers import nitinol_columns, COLS

cols = nitinol_columns(10, "linear")
f = tempfile.NamedTemporaryFile(mode="w", delete=False, suffix=".csv")
pd.DataFrame({c: cols[c] for c in COLS}).to_csv(f, index=False)
f.close()
beam = DynamicEulerBernoulliBeam(f.name, force_params=ForceParams(enable_gravity_effects=True))
os.unlink(f.name)
beam.create_system_func()
beam.create_input_func()
n = beam.beam_model.M.shape[0]
x = np.random.default_rng(0).normal(0, 1e-4, 2 * n)
u = np.zeros(n); u[-2] = 0.1
for name, dyn in (("get_composed_dynamic_system (the literal two-call composition: two launches per call)", beam.get_composed_dynamic_system()),
                  ("get_dynamic_system (default closures: one launch per call)", beam.get_dynamic_system())):
    for _ in range(50):
        dyn(0.0, x, u)
    t0 = time.perf_counter()
    for _ in range(2000):
        dyn(0.0, x, u)
    print(f"{name}: {(time.perf_counter() - t0) / 2000 * 1e6:.1f} us per call", flush=True)

    def uf(t):
        v = np.zeros(n)
        if t < 0.01:
            v[-2] = 0.1
        return v

    t_end = float(os.environ.get("T_END", "0.05"))
    t0 = time.perf_counter()
    sol = solve_ivp(lambda t, y: dyn(t, y, uf(t)), (0.0, t_end), np.zeros(2 * n), method="LSODA",
                    t_eval=np.arange(0.0, t_end, 0.001))
    wall = time.perf_counter() - t0
    print(f"   LSODA (default tolerances) to t = {t_end} s: {wall:.2f} s wall = {wall / t_end:.1f} s per simulated second, "
          f"nfev {sol.nfev}, tip w = {sol.y[n - 2, -1]:.8e}   [reference: 207 s per simulated second]", flush=True)
