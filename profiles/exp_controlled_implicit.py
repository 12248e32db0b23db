import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "continuum-robot_amd")]
import numpy as np, torch
from tests.helpers import *
from tests.test_gpu_parity import beam_columns, force_kwargs, ensemble
z = dict(np.load(os.path.join(ROOT, "tests/golden/g8_lsoda.npz"), allow_pickle=False))
for name in ("lin10_grav", "lin6_fluid", "mixed6_fluid"):
    cols, kw = beam_columns(z, name), force_kwargs(z, name)
    times = z[name + "/times"]; tight = z[name + "/x_tight"]; dflt = z[name + "/x_default_tol"]
    amp = float(z[name + "/amp"])
    T = float(times[-1])
    print(name, "times", times, "amp", amp)
    for label, extra in (("fixed10", dict(substeps=10)), ("auto-all", dict(substeps="auto")), ("auto-pos", dict(substeps="auto", control="positions")),
                         ("auto-all-tight", dict(substeps="auto", rtol=1e-6, atol=1e-9))):
        ens = ensemble(cols, 2, kw)
        n = ens.n
        t_eval = np.arange(0.0, T + 0.0005, 0.001)
        t0 = time.perf_counter()
        sol = ens.solve_ivp((0.0, T + 0.0005), t_eval, method="LSODA", impulse_amp=np.full(2, amp), **extra)
        torch.cuda.synchronize(); wall = time.perf_counter() - t0
        y = sol.y.cpu().numpy()[0]
        for ti, t in enumerate(times):
            k = int(round(t / 0.001))
            ref = tight[ti]
            sc = 1e-6 + 1e-3 * np.abs(ref)
            e = np.abs(y[:, k] - ref) / sc
            ed = np.abs(dflt[ti] - ref) / sc
            print(f"  {label:15s} t={t:.3f} max scaled err q {e[:n].max():9.3g} v {e[n:].max():9.3g} rms {np.sqrt((e**2).mean()):9.3g} | LSODA-default vs tight: q {ed[:n].max():9.3g} v {ed[n:].max():9.3g} rms {np.sqrt((ed**2).mean()):9.3g}  wall {wall:.2f}s  substeps {getattr(sol,'substeps',[10])[:3]}..max {max(getattr(sol,'substeps',[10]))}")
