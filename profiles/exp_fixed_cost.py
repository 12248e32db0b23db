"""measurement helper: where the fixed cost of a lean-stepper launch comes from.  Launch time (HIP events, median of 30)
of the config-3 stepper for B beams x n fused steps: t = f(B) + n * s(B).  B = 512 is one beam per workgroup (no walk),
4096 is eight beams per workgroup: a per-launch constant shows as f(512) = f(4096), a per-beam one as f growing with B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
from continuum_robot.batched import BeamEnsemble
from continuum_robot.models.force_params import ForceParams
from tests.helpers import nitinol_columns

fp = ForceParams(fluid_density=1000.0, enable_fluid_effects=True, enable_gravity_effects=False)
cols = nitinol_columns(256, "nonlinear")
for B in (256, 512, 1024, 2048, 4096):
    ens = BeamEnsemble(cols, B, force_params=fp)
    amps = torch.as_tensor(0.1 * (1.0 + np.arange(B) / B), device="cuda")
    row = []
    for n in (1, 2, 5, 20, 50):
        ts = []
        for rep in range(40):
            ens.zero_state()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ens.step(n, 2e-5, impulse_amp=amps)
            e1.record()
            torch.cuda.synchronize()
            if rep >= 10:
                ts.append(e0.elapsed_time(e1) * 1e3)
        row.append((n, float(np.median(ts))))
    (n0, t0), (n1, t1) = row[2], row[4]
    s = (t1 - t0) / (n1 - n0)
    print(f"B={B:5d}: " + "  ".join(f"n={n}: {t:7.1f} us" for n, t in row) + f"   per step {s:.2f} us, fixed {t0 - n0 * s:.1f} us")
