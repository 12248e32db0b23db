"""Measurement helper: host-side cost of one BeamEnsemble.step call (launches queued, no synchronisation): 7.6 us without an input,
9.2 us with device-resident impulse amplitudes (hipLaunchKernel itself is about half of it), 26 us when the amplitudes come as a
numpy array (one H2D copy per call).  usage: python profiles/exp_host_call.py"""
import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
from continuum_robot.batched import BeamEnsemble
from continuum_robot.models.force_params import ForceParams
from tests.helpers import nitinol_columns
ens = BeamEnsemble(nitinol_columns(10, "linear"), 1, force_params=ForceParams())
amps = torch.as_tensor([0.1], dtype=torch.float64, device="cuda")
for label, kw in (("no input", dict()), ("device amps", dict(impulse_amp=amps)), ("numpy amps", dict(impulse_amp=np.array([0.1])))):
    ens.step(1, 2e-5, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000): ens.step(1, 2e-5, **kw)
    host = (time.perf_counter() - t0) / 3000
    torch.cuda.synchronize()
    print(f"{label}: {host * 1e6:.1f} us per call (host side, launches queued)")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(2000): ens.step(1, 2e-5, impulse_amp=amps)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
