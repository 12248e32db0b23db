"""Experiment: closed-loop step time of SMALL ensembles (128 / 64 elements) through the persistent stepper (CRB_LOOP=1) and the
stage-split launches (CRB_LOOP=0).  usage: CRB_LOOP=0|1 python profiles/exp_small_loop.py"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "continuum-robot_amd"))
from tests.helpers import nitinol_columns
from tests.test_gpu_parity import ensemble

for ne in (128, 64):
    for B in (16, 64, 128, 256, 512, 1024):
        ens = ensemble(nitinol_columns(ne, "linear"), B, dict(enable_gravity=True))
        rng = np.random.default_rng(0)
        K = 1e2 * rng.standard_normal((ens.n, 2 * ens.n))
        K[:, ens.n:] *= 1e-3
        ens.set_state(1e-4 * rng.standard_normal((B, 2 * ens.n)))
        ens.step_feedback(20, 1e-7, K)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            ens.step_feedback(100, 1e-7, K)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"CRB_LOOP={os.environ.get('CRB_LOOP')} n_e={ne} B={B:5d}: {best / 100 * 1e6:8.1f} us/step path {ens.feedback_path()}", flush=True)
