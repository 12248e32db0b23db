"""Experiment (round 3): launch time of the config-3 stepper for 1 / 20 / 100 fused steps -- what a beam switch costs a short launch.
A/B against a library built from other sources with CRB_LIB_PATH.  The prefetch of the next beam's state (taken) measured
62.6 / 626 / 2750 us before and 66.4 / 609 / 2749 us after at B = 4096.  usage: [CRB_LIB_PATH=...] python profiles/exp_beam_switch.py"""
import os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
from continuum_robot.batched import BeamEnsemble
from continuum_robot.models.force_params import ForceParams
from tests.helpers import nitinol_columns
fp = ForceParams(fluid_density=1000.0, enable_fluid_effects=True, enable_gravity_effects=False)
cols = nitinol_columns(256, "nonlinear")
for B in (2048, 4096):
    ens = BeamEnsemble(cols, B, force_params=fp)
    amps = torch.as_tensor(0.1 * (1.0 + np.arange(B) / B), device="cuda")
    for n in (1, 20, 100):
        ts = []
        for rep in range(40):
            ens.zero_state()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ens.step(n, 2e-5, impulse_amp=amps); e1.record()
            torch.cuda.synchronize()
            if rep >= 10: ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"B={B} n={n}: median {np.median(ts):8.1f} us  min {np.min(ts):8.1f}", flush=True)
    x = ens.unpack_state().cpu().numpy()
    print("checksum", float(np.abs(x).sum()))
