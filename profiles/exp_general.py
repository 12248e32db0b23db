"""Experiment: the GENERAL stepper (CRB_DISABLE_LEAN=1) at two / one wave per SIMD for 5+ reduction levels."""
import os, sys, time
os.environ["CRB_DISABLE_LEAN"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "continuum-robot_amd"))
from tests.helpers import nitinol_columns
from tests.test_gpu_parity import ensemble
for B, ne, kind, kw in [(4096, 256, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True)), (1024, 256, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True)),
                        (4096, 128, "linear", dict(enable_gravity=True)), (512, 64, "linear", dict(enable_gravity=True)), (4096, 64, "linear", dict(enable_gravity=True))]:
    ens = ensemble(nitinol_columns(ne, kind), B, kw)
    amps = np.full(B, 0.1)
    ens.step(20, 2e-5, impulse_amp=amps); torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter(); ens.step(100, 2e-5, impulse_amp=amps); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"general {B} x {ne} {kind}: {best / 100 * 1e6:8.2f} us/step levels {int(ens.plan.layout.pcr_levels)}", flush=True)
