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


BLOCKS = ("u", "w", "phi", "du_dt", "dw_dt", "dphi_dt")
# A block (one DOF kind of one plane: positions or rates) is measured relative to ITS OWN largest reference
# entry, per beam.  Floor: a block whose largest entry is below BLOCK_FLOOR x the largest entry of its plane is
# physically zero at the working precision of that plane (e.g. the axial DOFs of a linear beam without axial
# load are exactly 0) and is measured against that floor instead.
BLOCK_FLOOR = 1e-9


def block_errs(got, ref, free_index, floor=BLOCK_FLOOR):
    """Per-DOF-block relative errors of reduced states [..., 2n] (or reduced vectors [..., n]).

    ``free_index`` is the plan's reduced -> full index map (full = 3*node + dof); block of reduced entry r is
    ``free_index[r] % 3`` (u, w, phi), second half of a state = the rates.  For every beam (row) and block:
    max|got - ref| / max(max|ref| over the block, floor * max|ref| over the block's plane); the result is the
    maximum over beams, as a dict keyed by BLOCKS.  The single normwise figure (rel_err) hides the small DOFs:
    in the golden rollouts |u| ~ 1e-9 against |v| ~ 1e-2."""
    a = np.atleast_2d(np.asarray(got, dtype=np.float64))
    b = np.atleast_2d(np.asarray(ref, dtype=np.float64))
    assert a.shape == b.shape, (a.shape, b.shape)
    dof = np.asarray(free_index) % 3
    n = dof.size
    planes = a.shape[-1] // n
    assert a.shape[-1] == planes * n and planes in (1, 2)
    out = {}
    for pl in range(planes):
        sa, sb = a[:, pl * n:(pl + 1) * n], b[:, pl * n:(pl + 1) * n]
        plane_max = np.max(np.abs(sb), axis=1)
        for d in range(3):
            sel = dof == d
            if not sel.any():
                continue
            den = np.maximum(np.maximum(np.max(np.abs(sb[:, sel]), axis=1), floor * plane_max), 1e-300)
            with np.errstate(invalid="ignore"):
                e = np.max(np.abs(sa[:, sel] - sb[:, sel]), axis=1) / den
            out[BLOCKS[3 * pl + d]] = float(np.max(e)) if np.all(np.isfinite(e)) else float("inf")
    return out


AXIAL_BLOCKS = ("u", "du_dt")
COND_MIN_STEPS = 600      # below this horizon every block of every golden rollout keeps its fixed bound (measured <= 2e-15)
COND_CEILING = 1e-4       # no conditioning argument admits more than this; beyond it parity is unpinned, not met
ADMITTED = []             # (what, block, error, allowed) of every block that passed through the conditioning clause


def assert_blocks(got, ref, free_index, tol, bar=1e-6, what="", cond=None, cond_factor=16.0, steps=None):
    """Every DOF block within ``tol`` (the measured bound, written at the call site; never looser than
    north_star's 1e-6 ``bar``).  Returns the per-block errors.

    ``cond`` (from ``rollout_conditioning``) + ``steps``: the conditioning clause, for the AXIAL blocks (u, du/dt) of
    rollouts longer than COND_MIN_STEPS steps ONLY.  ``cond`` is the per-block forward error of the ORACLE ITSELF under
    a 4-ulp relative perturbation of its input.  The shipped f1 of the reference (segments.py:178-208)
    makes the axial DOFs of long nonlinear chains exponentially unstable: rounding differences grow x1000 per 200 steps
    beyond step 600, and the C oracle itself is 3.4e-6 away from the reference's own du/dt block at 1000 steps
    (nl256_drag, tests/golden) -- no implementation that rounds differently can meet a fixed bound there.  Such a block
    is held to ``cond_factor`` x that sensitivity (the oracle's own response to a 64-ulp input change), capped at
    COND_CEILING, and is recorded in ADMITTED.  w, phi and their rates never take this clause, whatever is passed."""
    errs = block_errs(got, ref, free_index)
    assert tol <= bar * (1 + 1e-12)
    for k, e in errs.items():
        allowed = tol
        if cond and k in AXIAL_BLOCKS and steps is not None and steps > COND_MIN_STEPS:
            allowed = min(max(tol, cond_factor * cond.get(k, 0.0)), COND_CEILING)
        assert e <= allowed, (what, k, e, allowed, errs, cond)
        if e > tol:
            ADMITTED.append((what, k, e, allowed))
            print(f"[assert_blocks] {what}: block {k} admitted through the conditioning clause: {e:.2e} <= {allowed:.2e} "
                  f"(fixed bound {tol:.0e}; parity of this block is unpinned at this horizon)")
    return errs


def rollout_conditioning(ob, x0, dt, steps, amp, **kw):
    """Per-block forward sensitivity of the oracle's impulse rollout to a 4-ulp relative perturbation of the
    impulse amplitude (see assert_blocks)."""
    a = ob.rk4_impulse(x0, dt, steps, amp, **kw)
    b = ob.rk4_impulse(x0, dt, steps, amp * (1.0 + 2.0 ** -50), **kw)
    return block_errs(b, a, ob.red2full())


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
