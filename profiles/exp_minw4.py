"""Experiment: 4-level single-wave gravity kernels (beams of 9 .. 16 nodes, packed 4 .. 7 to a wave) at two / one wave per SIMD."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "continuum-robot_amd"))
from tests.helpers import nitinol_columns
from tests.test_gpu_parity import ensemble
for B, ne in [(1, 10), (4096, 10), (65536, 10), (32768, 15)]:
    ens = ensemble(nitinol_columns(ne, "linear"), B, dict(enable_gravity=True))
    amps = np.full(B, 0.1)
    for name, fn in (("rk4", lambda: ens.step(200, 2e-5, impulse_amp=amps)), ("implicit", lambda: ens.step_implicit(100, 1e-4, impulse_amp=amps))):
        fn(); torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{B:6d} x {ne} {name}: {best / (200 if name == 'rk4' else 100) * 1e6:8.2f} us/step levels {int(ens.plan.layout.pcr_levels)}", flush=True)
