"""Worker of tests/test_dropin_api.py::test_reference_pool_pattern: what /root/reference/examples/example_utilities.py:116-170
(`simulate_single_beam`) does inside `multiprocessing.Pool.map` -- build the beam IN the worker from a CSV, integrate its
closure with scipy, hand back a picklable result -- written against the drop-in package."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]


def simulate(task):
    import numpy as np
    from scipy.integrate import solve_ivp

    from continuum_robot.models.dynamic_beam_model import DynamicEulerBernoulliBeam
    from continuum_robot.models.force_params import ForceParams

    name, csv_file, fluid, t_final = task
    fp = ForceParams(fluid_density=1000.0, enable_fluid_effects=True) if fluid else ForceParams()
    beam = DynamicEulerBernoulliBeam(csv_file, force_params=fp)
    beam.create_system_func()
    beam.create_input_func()
    system = beam.get_dynamic_system()
    n = beam.beam_model.M.shape[0]

    def u(t):                       # the examples' tip impulse (example_utilities.py:144-148)
        f = np.zeros(n)
        if t < 0.01:
            f[-2] = 0.1
        return f

    t0 = time.perf_counter()
    sol = solve_ivp(lambda t, x: system(t, x, u), (0.0, t_final), np.zeros(2 * n), method="LSODA",
                    t_eval=np.linspace(0.0, t_final, 5))
    return name, sol, time.perf_counter() - t0, {"nfev": int(sol.nfev), "pid": os.getpid()}
