"""kernel-tuning helper: times crb_feedback_force (the LQR feedback GEMM) at the config-5 shape
(2048 beams x 128 elements: [2048 x 768] x [768 x 384]) and checks it against torch.matmul.
usage: CRB_LIB_PATH=... python profiles/exp_gemm.py [B] [n_elem]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
from continuum_robot import _native as nat
from continuum_robot.batched import BeamEnsemble
from tests.helpers import nitinol_columns

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cols = nitinol_columns(ne, "linear")
ens = BeamEnsemble(cols, B)
n = ens.n
rng = np.random.default_rng(0)
K = torch.tensor(rng.normal(size=(n, 2 * n)), device="cuda")
x = rng.normal(size=(B, 2 * n))
ens.set_state(x)
u = torch.zeros((B, ens.n_node, 4), dtype=torch.float64, device="cuda")
lib, st = ens._lib, ens._stream()
call = lambda: nat.check(lib.crb_feedback_force(ens.plan.h, ens._ptr(ens.state), ens._ptr(K), None, ens._ptr(u), st))
call(); torch.cuda.synchronize()
want = (-torch.tensor(x, device="cuda")) @ K.t()
got = ens.unpack_vec(u)
err = float((got - want).abs().max() / want.abs().max())
for _ in range(20): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): call()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 200
print(os.environ.get("CRB_LIB_PATH", "default").split("/")[-1], f"{us:.1f} us  {2*B*2*n*n/us*1e-6:.1f} TF  err {err:.1e}")
