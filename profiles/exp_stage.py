"""measurement helper: the lean stage kernel alone (crb_rk4_stage, stage 1) at the same node count in three shapes --
one / two / four waves per beam, i.e. a wave's slots are consecutive nodes / every 2nd / every 4th node: how much of the
kernel's time is the access pattern of its 32-byte node records."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
from continuum_robot import _native as nat
from continuum_robot.batched import BeamEnsemble
from continuum_robot.models.force_params import ForceParams
from tests.helpers import nitinol_columns

for B, ne in ((4096, 64), (2048, 128), (1024, 256)):
    ens = BeamEnsemble(nitinol_columns(ne, "linear"), B, force_params=ForceParams(enable_gravity_effects=True))
    x = ens.state
    xs, acc, nxt = torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)
    u = torch.zeros((B, ens.n_node, 4), dtype=torch.float64, device="cuda")
    xs.copy_(x)
    lib, st = ens._lib, ens._stream()
    call = lambda: nat.check(lib.crb_rk4_stage(ens.plan.h, ens._ptr(x), ens._ptr(xs), ens._ptr(acc), ens._ptr(nxt), ens._ptr(u), 1, 0.0, 5e-6, None, st))
    for _ in range(20): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 200
    nodes = B * (ne + 1)
    print(f"{B} x {ne}: {us:.1f} us per stage launch, {nodes * 352 / us * 1e-6:.2f} TB/s of records, {us / nodes * 1e6:.1f} ps per node")
