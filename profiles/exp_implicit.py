"""Throughput of crb_step_implicit on ensembles (element-steps/s and simulated seconds per wall second)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
from continuum_robot.batched import BeamEnsemble
from continuum_robot.models.force_params import ForceParams
from tests.helpers import nitinol_columns

for ne, B, kind, kw, h in ((256, 4096, "linear", dict(enable_gravity_effects=True), 1e-4),   # (the examples' h: A's reduction stops at 5 levels)
                           (256, 4096, "nonlinear", dict(fluid_density=1000.0, enable_fluid_effects=True), 1e-4),
                           (128, 4096, "linear", dict(enable_gravity_effects=True), 1e-4),
                           (256, 4096, "nonlinear", dict(fluid_density=1000.0, enable_fluid_effects=True), 2e-4),   # 6 levels
                           (256, 4096, "linear", dict(enable_gravity_effects=True), 3e-4),                         # 6 - 7 levels
                           (256, 4096, "linear", dict(enable_gravity_effects=True), 1e-3),
                           (256, 4096, "nonlinear", dict(fluid_density=1000.0, enable_fluid_effects=True), 2e-4),
                           (64, 4096, "linear", dict(enable_gravity_effects=True), 1e-3),
                           (10, 65536, "linear", dict(enable_gravity_effects=True), 1e-3)):
    ens = BeamEnsemble(nitinol_columns(ne, kind), B, force_params=ForceParams(**kw))
    amps = torch.full((B,), 0.1, dtype=torch.float64, device=ens.device)
    steps = 200
    for n_iter in (1, 2):
        ens.zero_state(); ens.step_implicit(20, h, n_iter=n_iter, impulse_amp=amps); torch.cuda.synchronize()
        ens.zero_state(); t0 = time.perf_counter()
        ens.step_implicit(steps, h, n_iter=n_iter, impulse_amp=amps); torch.cuda.synchronize()
        w = time.perf_counter() - t0
        print(f"{B} x {ne} {kind} h={h:g} n_iter={n_iter}: {w / steps * 1e6:.1f} us/step, {B * ne * steps / w:.3e} element-steps/s, "
              f"{B * steps * h / w:.3e} beam-seconds per wall second, finite={bool(torch.isfinite(ens.state).all())}", flush=True)
