"""Closed-loop LQR rollouts of one rod under many disturbance impulses, as an ensemble.

The reference designs a full-state LQR for the linear rod and integrates ONE disturbed closed loop
(examples/lqr_control.py:46-84 design, :87-130 simulation with u = K (0 - x) evaluated inside the RHS, impulse of
10 N for 0.01 s at the tip).  Here the same controller is applied to B copies of the rod, each hit by its own impulse
amplitude, with the feedback evaluated in every Runge-Kutta stage on the GPU (`BeamEnsemble.step_feedback`).

    python examples/lqr_ensemble.py [--elements 6] [--beams 64] [--t-final 0.05] [--lsoda]

--lsoda adds the reference's own call, `solve_ivp(system_with_inputs, method="LSODA", rtol=1e-8, atol=1e-10, t_eval=...)`
(lqr_control.py:113-125), for the whole ensemble: `BeamEnsemble.solve_ivp(..., gain=K)` chooses the step by these tolerances.
"""
import argparse
import time

import numpy as np
import torch

from _common import rod

from continuum_robot.batched import BeamEnsemble
from continuum_robot.control import LinearQuadraticRegulator
from continuum_robot.models.force_params import ForceParams


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--elements", type=int, default=6)
    ap.add_argument("--beams", type=int, default=64)
    ap.add_argument("--t-final", type=float, default=0.05)
    ap.add_argument("--dt", type=float, default=5e-6, help="explicit step (the closed loop is stiffer than the rod)")
    ap.add_argument("--lsoda", action="store_true", help="also integrate the closed loop under tolerance control")
    args = ap.parse_args(argv)

    ens = BeamEnsemble(rod(args.elements, "linear"), args.beams, force_params=ForceParams(enable_gravity_effects=True))
    K, M = ens.plan.stiffness(), ens.plan.mass()
    n = K.shape[0]
    Q = np.eye(2 * n)
    Q[:n, :n] *= 100.0                     # Q / R of lqr_control.py:61-66
    Q[n:, n:] *= 10.0
    gain = LinearQuadraticRegulator(K, M, Q, np.eye(n)).compute_gain_matrix()
    gain_dev = torch.as_tensor(gain, dtype=torch.float64, device=ens.device)
    amps = 10.0 * (1.0 + np.arange(args.beams) / args.beams)
    steps = int(round(args.t_final / args.dt))

    rows = []
    for label, g in (("open loop", torch.zeros_like(gain_dev)), ("LQR", gain_dev)):
        ens.zero_state()
        t0 = time.perf_counter()
        ens.step_feedback(steps, args.dt, g, impulse_amp=amps)
        tip = ens.tip_displacement().cpu().numpy()
        rows.append((label, tip, time.perf_counter() - t0))
    if args.lsoda:
        ens.zero_state()
        t0 = time.perf_counter()
        t_eval = np.arange(0.0, args.t_final + 0.5e-3, 1e-3)            # DT of lqr_control.py:31
        sol = ens.solve_ivp((0.0, float(t_eval[-1])), t_eval, method="LSODA", rtol=1e-8, atol=1e-10, impulse_amp=amps, gain=gain_dev)
        rows.append((f"LQR, controlled ({max(sol.substeps)} steps/ms)", sol.y[:, n - 2, -1].cpu().numpy(), time.perf_counter() - t0))
    print(f"{args.beams} rods x {args.elements} elements, {steps} RK4 steps of {args.dt:g} s, impulses {amps[0]:.1f} .. {amps[-1]:.1f} N")
    for label, tip, wall in rows:
        print(f"{label:<32} tip w at t = {args.t_final:g} s: {tip.min():+.4e} .. {tip.max():+.4e} m   ({wall * 1e3:.1f} ms)")
    return rows


if __name__ == "__main__":
    main()
