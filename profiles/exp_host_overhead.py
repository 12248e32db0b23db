"""measurement helper: host-side cost of one BeamEnsemble.step() call (Python + ctypes + launch), on an ensemble whose
kernel is shorter than the call: calls per second of a back-to-back loop without synchronisation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
from continuum_robot.batched import BeamEnsemble
from tests.helpers import nitinol_columns

ens = BeamEnsemble(nitinol_columns(4, "linear"), 8)
amps = torch.full((8,), 0.1, dtype=torch.float64, device="cuda")
for label, kw in (("no input", {}), ("impulse (device tensor)", dict(impulse_amp=amps)), ("impulse (numpy)", dict(impulse_amp=np.full(8, 0.1)))):
    for _ in range(200):
        ens.step(1, 1e-6, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5000):
        ens.step(1, 1e-6, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label}: {(t1 - t0) / 5000 * 1e6:.2f} us per call (host side), {(time.perf_counter() - t0) / 5000 * 1e6:.2f} us incl. drain")
