"""Linear / nonlinear / mixed rods, with and without fluid drag and gravity, simulated as ONE ensemble on the GPU.

The reference runs these comparisons as a task list mapped over a process pool, one `solve_ivp(..., "LSODA")` per
task (examples/beam_comparison_fluid.py:52-83, beam_comparison_gravity.py:53-83, example_utilities.py:116-170:
tip impulse of 0.1 N for t < 0.01 s, output every DT).  Here every task is one beam of a `BeamEnsemble` with its own
element types and its own ForceParams, and the whole list is integrated by one kernel launch per call.

    python examples/beam_comparison_ensemble.py [--elements 10] [--t-final 0.05] [--method LSODA|RK45|RK4]

Note: the shipped nonlinear element grows without bound along the axis after ~2e-2 s of explicit stepping
(SURVEY.md appendix B-1); the default horizon stays below that.  Linear rods run to any horizon.
"""
import argparse
import time

import numpy as np

from _common import rod

from continuum_robot.batched import BeamEnsemble
from continuum_robot.models.force_params import ForceParams


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--elements", type=int, default=10)
    ap.add_argument("--t-final", type=float, default=0.02)
    ap.add_argument("--dt-out", type=float, default=1e-3, help="output interval (DT of the reference's examples)")
    ap.add_argument("--method", default="LSODA")
    ap.add_argument("--substeps", type=int, default=0,
                    help="0 (default): the step size follows solve_ivp's tolerances, per rod, inside the kernel -- the reference's "
                         "call; N > 0: N fixed implicit steps per output interval, no control")
    ap.add_argument("--rtol", type=float, default=1e-3)
    ap.add_argument("--atol", type=float, default=1e-6)
    args = ap.parse_args(argv)

    none = ForceParams()
    fluid = ForceParams(fluid_density=1000.0, enable_fluid_effects=True)
    gravity = ForceParams(enable_gravity_effects=True)
    tasks = [(kind, label, fp) for kind in ("linear", "nonlinear", "mixed")
             for label, fp in (("dry", none), ("fluid", fluid), ("gravity", gravity))]
    ens = BeamEnsemble.from_dataframes([rod(args.elements, kind) for kind, _, _ in tasks],
                                       force_params=[fp for _, _, fp in tasks])
    t_eval = np.arange(0.0, args.t_final + 0.5 * args.dt_out, args.dt_out)
    t0 = time.perf_counter()
    sol = ens.solve_ivp((0.0, float(t_eval[-1])), t_eval, method=args.method, substeps=args.substeps if args.substeps > 0 else "auto",
                        rtol=args.rtol, atol=args.atol, impulse_amp=np.full(len(tasks), 0.1))
    y = sol.y.cpu().numpy()                      # [B, 2n, n_t]: y[b] is what the reference's sol.y holds for task b
    wall = time.perf_counter() - t0
    tip = ens.reduced_index(args.elements, "w")  # the tip's transverse displacement
    print(f"{len(tasks)} rods x {args.elements} elements, {t_eval.size} output times to t = {t_eval[-1]:.3f} s "
          f"({sol.method}{', ' + sol.controller + ' controller' if hasattr(sol, 'controller') else ''}): {wall * 1e3:.1f} ms")
    if hasattr(sol, "substeps_per_beam"):
        print("implicit steps taken per rod:", sol.substeps_per_beam.sum(axis=1).tolist())
    print(f"{'elements':<10} {'forces':<8} {'tip w(t_final) [m]':>20} {'max |tip w| [m]':>18}")
    for b, (kind, label, _) in enumerate(tasks):
        w = y[b, tip]
        print(f"{kind:<10} {label:<8} {w[-1]:>20.6e} {np.abs(w).max():>18.6e}")
    return y


if __name__ == "__main__":
    main()
