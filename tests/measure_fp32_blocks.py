"""Measure the per-DOF-block drift of the fp32 plan against the fp64 oracle over the whole config-4 ensemble
(sets FP32_TOL of tests/test_gpu_parity.py)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (lives under tests/: it uses the oracle as its checker)
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
from continuum_robot.batched import BeamEnsemble
from continuum_robot.models.force_params import ForceParams
from tests.helpers import nitinol_columns, oracle_beam, block_errs

cols = nitinol_columns(256, "nonlinear")
fp = ForceParams(fluid_density=1000.0, enable_fluid_effects=True, enable_gravity_effects=False)
B = 4096
amps = 0.1 * (1 + np.arange(B) / B)
ob = oracle_beam(cols, fluid_density=1000.0, enable_fluid=True)
for steps in (100, 200):
    ens = BeamEnsemble(cols, B, force_params=fp, dtype=torch.float32)
    ens.step(steps, 2e-5, impulse_amp=amps)
    got = ens.unpack_state().double().cpu().numpy()
    ref, _ = ob.rk4_impulse_batch(np.zeros((B, 2 * ob.n)), 2e-5, steps, amps)
    print(steps, {k: "%.2e" % v for k, v in block_errs(got, ref, ens.free_index).items()}, flush=True)
