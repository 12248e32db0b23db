"""measurement helper: small rods (the reference's LQR example is 6 elements), open-loop stepper against the
closed-loop one (crb_step_rk4_feedback: fused for beams that live in one wave, CRB_FUSED_FEEDBACK=0 = the
stage-split path with one GEMM + one stage kernel per stage), microseconds per RK4 step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd"), os.path.join(ROOT, "examples")]
import numpy as np, torch
from _common import rod
from continuum_robot.batched import BeamEnsemble
from continuum_robot.models.force_params import ForceParams

for ne in (6, 16, 27):
    for B in (64, 2048):
        ens = BeamEnsemble(rod(ne, "linear"), B, force_params=ForceParams(enable_gravity_effects=True))
        amps = np.full(B, 0.1)
        n = ens.n
        g = torch.zeros((n, 2 * n), dtype=torch.float64, device="cuda")
        row = []
        for label, fn in (("open loop", lambda k: ens.step(k, 5e-6, impulse_amp=amps)),
                          ("feedback", lambda k: ens.step_feedback(k, 5e-6, g, impulse_amp=amps))):
            fn(100); torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(1000); torch.cuda.synchronize()
            row.append(f"{label} {(time.perf_counter() - t0) / 1000 * 1e6:.2f} us/step")
        print(f"{B} rods x {ne} elements:", ", ".join(row), "(stage-split)" if os.environ.get("CRB_FUSED_FEEDBACK") == "0" else "")
