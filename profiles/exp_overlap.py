"""Experiment: do the feedback GEMM (MFMA-bound) and the stage kernel (HBM-bound) of two half-ensembles overlap when
their chains are issued on two HIP streams?  (config 5: 2048 beams x 128 elements per GPU)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
import bench
from continuum_robot.batched import BeamEnsemble
from continuum_robot.models.force_params import ForceParams
from tests.helpers import nitinol_columns

cols = nitinol_columns(128, "linear")
fp = ForceParams(enable_gravity_effects=True)
full = BeamEnsemble(cols, 2048, force_params=fp)
gain = torch.as_tensor(bench.lqr_gain(full), dtype=torch.float64, device=full.device) if os.environ.get("REAL_GAIN") else \
    torch.randn((full.n, 2 * full.n), dtype=torch.float64, device=full.device) * 1e-3
steps, dt = 100, 5e-6


def run(enss, streams, reps=5):
    best = 1e9
    for _ in range(reps):
        for e in enss:
            e.zero_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e, s in zip(enss, streams):
            with torch.cuda.stream(s):
                e.step_feedback(steps, dt, gain, impulse_amp=torch.full((e.n_beams,), 10.0, dtype=torch.float64, device=e.device))
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / steps * 1e6


main = torch.cuda.current_stream()
print("one ensemble of 2048, one stream: %.1f us/step" % run([full], [main]))
for parts in (2, 4):
    enss = [BeamEnsemble(cols, 2048 // parts, force_params=fp) for _ in range(parts)]
    ss = [torch.cuda.Stream() for _ in range(parts)]
    print(f"{parts} ensembles of {2048 // parts}, one stream (sequential): %.1f us/step" % run(enss, [main] * parts))
    print(f"{parts} ensembles of {2048 // parts}, {parts} streams: %.1f us/step" % run(enss, ss))
