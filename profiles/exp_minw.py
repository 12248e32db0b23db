"""Experiment: RK4 step time of gravity ensembles (lean step kernels with register spills at two waves per SIMD)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "continuum-robot_amd"))
from tests.helpers import nitinol_columns
from tests.test_gpu_parity import ensemble

shapes = [(1024, 64, "linear", dict(enable_gravity=True)), (4096, 64, "linear", dict(enable_gravity=True)),
          (4096, 32, "linear", dict(enable_gravity=True)),
          (4096, 128, "linear", dict(enable_gravity=True)), (4096, 256, "linear", dict(enable_gravity=True)),
          (4096, 256, "nonlinear", dict(enable_gravity=True, fluid_density=1000.0, enable_fluid=True)),
          (2048, 128, "linear", dict(enable_gravity=True)), (4096, 256, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True))]
for B, ne, kind, kw in shapes:
    ens = ensemble(nitinol_columns(ne, kind), B, kw)
    amps = np.full(B, 0.1)
    ens.step(100, 2e-5, impulse_amp=amps)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        ens.step(200, 2e-5, impulse_amp=amps)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{B:5d} x {ne:3d} {kind:9s} {sorted(kw)}: {best / 200 * 1e6:8.2f} us/step  {B * ne * 200 / best:.3e} el-steps/s", flush=True)

print("rk45 (10 ms, rtol 1e-6) and implicit (h = 1e-4, 100 steps)")
for B, ne, kind, kw in [(1024, 64, "linear", dict()), (4096, 64, "linear", dict()), (4096, 32, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True)),
                        (1024, 16, "linear", dict())]:
    ens = ensemble(nitinol_columns(ne, kind), B, kw)
    amps = np.full(B, 0.1)
    best = 1e9
    for rep in range(3):
        ens.zero_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = ens.solve_rk45(2e-3, rtol=1e-6, atol=1e-9, impulse_amp=amps, t0=0.0)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"rk45 {B:5d} x {ne:3d} {kind:9s}: {best * 1e3:8.2f} ms, {int(st['accepted'][0])} steps", flush=True)
for B, ne, kind, kw in [(1024, 64, "linear", dict(enable_gravity=True)), (4096, 64, "linear", dict(enable_gravity=True)),
                        (4096, 40, "linear", dict(enable_gravity=True))]:
    ens = ensemble(nitinol_columns(ne, kind), B, kw)
    amps = np.full(B, 0.1)
    ens.step_implicit(100, 1e-4, impulse_amp=amps)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        ens.step_implicit(100, 1e-4, impulse_amp=amps)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"implicit {B:5d} x {ne:3d} {kind:9s} grav: {best / 100 * 1e6:8.2f} us/step", flush=True)
