"""Shared helpers of the test-suite: golden-vector access and beam construction."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COLS = ["length", "elastic_modulus", "moment_inertia", "density", "cross_area", "type", "boundary_condition",
        "wetted_area", "drag_coef"]


class Golden:
    """Lazy access to tests/golden/*.npz (numpy.load, allow_pickle=False)."""

    def __init__(self):
        self._files = {}

    def __getitem__(self, name):
        if name not in self._files:
            self._files[name] = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        return self._files[name]


def beam_columns(z, prefix):
    """The CSV columns of a beam stored under ``prefix`` in a golden file."""
    return {c: z[f"{prefix}/{c}"] for c in COLS}


def force_kwargs(z, prefix):
    return dict(fluid_density=float(z[f"{prefix}/fluid_density"]), enable_fluid=bool(z[f"{prefix}/enable_fluid"]),
                gravity=z[f"{prefix}/gravity"], enable_gravity=bool(z[f"{prefix}/enable_gravity"]))


def oracle_beam(cols, node_bc=None, **force_kw):
    from oracle import OracleBeam

    kw = dict(length=cols["length"], elastic_modulus=cols["elastic_modulus"], moment_inertia=cols["moment_inertia"],
              density=cols["density"], cross_area=cols["cross_area"], type=cols["type"],
              wetted_area=cols["wetted_area"], drag_coef=cols["drag_coef"])
    if node_bc is not None:
        kw["node_bc"] = node_bc
    else:
        kw["boundary_condition"] = cols["boundary_condition"]
    kw.update(force_kw)
    return OracleBeam(**kw)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def nitinol_columns(n, kind="linear", bcs=None):
    """Synthetic beam of SURVEY §8(d): Nitinol constants (values are data of
    /root/reference/examples/example_utilities.py:25-34), FIXED at node 0."""
    r, L = 0.005, 0.25
    return {
        "length": np.full(n, L),
        "elastic_modulus": np.full(n, 75e9),
        "moment_inertia": np.full(n, np.pi * r**4 / 4),
        "density": np.full(n, 6450.0),
        "cross_area": np.full(n, np.pi * r**2),
        "type": np.array([kind] * n) if isinstance(kind, str) else np.array(kind),
        "boundary_condition": np.array(bcs if bcs is not None else ["FIXED"] + ["NONE"] * (n - 1)),
        "wetted_area": np.full(n, 2 * np.pi * r * L),
        "drag_coef": np.full(n, 0.82),
    }
