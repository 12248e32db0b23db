"""Experiment (round 3): step time of the fused small-beam closed loop (gain in LDS) for 6 / 10 / 20 elements x 64 beams, next to the
open-loop lean stepper; DESIGN.md section 7 ("Fused small-beam closed loop, round 3") lists what was measured with it and which
variant was taken (the product K e on the matrix cores with the gain as the resident A operand, for gains of 21 .. 32 rows).
usage: python profiles/exp_fbstamp.py"""
import os, sys, ctypes as C, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd"), os.path.join(ROOT, "examples")]
import numpy as np, torch
from continuum_robot.batched import BeamEnsemble
from continuum_robot.models.force_params import ForceParams
from continuum_robot import _native as nat
from tests.helpers import nitinol_columns
for ne in (6, 10, 20):
    ens = BeamEnsemble(nitinol_columns(ne, "linear"), 64, force_params=ForceParams(enable_gravity_effects=True))
    rng = np.random.default_rng(0)
    K = 10.0 * rng.standard_normal((ens.n, 2 * ens.n))
    amps = np.full(64, 1.0)
    ens.step_feedback(50, 1e-7, K, impulse_amp=amps); torch.cuda.synchronize()
    t0 = time.perf_counter(); ens.step_feedback(2000, 1e-7, K, impulse_amp=amps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    buf = (C.c_ulonglong * 64)()
    try:
        nat.load().crb_fbdbg_read(buf)
        v = np.array(buf[:5], dtype=np.int64)
        print(f"n_e={ne} path {ens.feedback_path()} {dt / 2000 * 1e6:.2f} us/step; stage stamps (cycles): e-write {v[1]-v[0]}, product {v[2]-v[1]}, read {v[3]-v[2]}, rhs {v[4]-v[3]}")
    except AttributeError:
        print(f"n_e={ne} path {ens.feedback_path()} {dt / 2000 * 1e6:.2f} us/step")
    t0 = time.perf_counter(); ens.step(2000, 1e-7, impulse_amp=amps); torch.cuda.synchronize()
    print(f"      open-loop lean stepper: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us/step")
