#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference package is imported read-only from /root/reference/src; nothing is
written there.  Every fixture is data: beam definitions (the CSV columns), seeded
input states and the reference's outputs on them.  SURVEY.md §8(c) lists what is
pinned (G1..G5).  The fixed-step RK4 loop below is OURS (the reference has no
integrator, every caller hands its RHS to scipy.solve_ivp); it is the loop the
oracle and the HIP stepper restate:

    k1 = f(t, x, u(t)) ; k2 = f(t+dt/2, x+dt/2*k1, u(t+dt/2))
    k3 = f(t+dt/2, x+dt/2*k2, u(t+dt/2)) ; k4 = f(t+dt, x+dt*k3, u(t+dt))
    x <- x + dt/6*(k1 + 2*k2 + 2*k3 + k4) ;  t <- t + dt      (time ACCUMULATES by addition)

The accumulate-by-addition clock is the convention under which SURVEY.md §8(c)'s anchors
were measured (it decides on which stage the `t < 0.01` impulse of
examples/example_utilities.py:144-148 switches off); the script prints the anchors.
"""
import os
import sys
import tempfile
import time

import numpy as np

REF_SRC = "/root/reference/src"
sys.dont_write_bytecode = True
sys.path.insert(0, REF_SRC)

import pandas as pd  # noqa: E402
from continuum_robot.models.abstractions import (  # noqa: E402
    BoundaryConditionType,
    Properties,
)
from continuum_robot.models.dynamic_beam_model import DynamicEulerBernoulliBeam  # noqa: E402
from continuum_robot.models.euler_bernoulli_beam import EulerBernoulliBeam  # noqa: E402
from continuum_robot.models.force_params import ForceParams  # noqa: E402
from continuum_robot.models.segments import LinearSegment, NonlinearSegment  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
COLS = [
    "length",
    "elastic_modulus",
    "moment_inertia",
    "density",
    "cross_area",
    "type",
    "boundary_condition",
    "wetted_area",
    "drag_coef",
]


# --------------------------------------------------------------------------- beams
def nitinol(n, kind, bcs=None):
    """Synthetic beam with the Nitinol constants of examples/example_utilities.py:25-34."""
    r = 0.005
    L = 0.25
    kinds = [kind] * n if isinstance(kind, str) else list(kind)
    bcs = bcs or (["FIXED"] + ["NONE"] * (n - 1))
    return pd.DataFrame(
        {
            "length": [L] * n,
            "elastic_modulus": [75e9] * n,
            "moment_inertia": [np.pi * r**4 / 4] * n,
            "density": [6450.0] * n,
            "cross_area": [np.pi * r**2] * n,
            "type": kinds,
            "boundary_condition": bcs,
            "wetted_area": [2 * np.pi * r * L] * n,
            "drag_coef": [0.82] * n,
        }
    )


def test4(kind, bcs=None):
    """The 4-segment beam of tests/test_dynamic_beam.py:22-41 (values are data)."""
    n = 4
    bcs = bcs or ["FIXED", "NONE", "NONE", "NONE"]
    return pd.DataFrame(
        {
            "length": [0.25] * n,
            "elastic_modulus": [75e9] * n,
            "moment_inertia": [4.91e-10] * n,
            "density": [6450.0] * n,
            "cross_area": [7.85e-5] * n,
            "type": [kind] * n,
            "boundary_condition": bcs,
            "wetted_area": [0.001] * n,
            "drag_coef": [0.5] * n,
        }
    )


def mixed5(bcs=None):
    """The 5-segment mixed beam of tests/test_advanced_composition.py:15-20."""
    n = 5
    bcs = bcs or ["FIXED", "NONE", "NONE", "NONE", "NONE"]
    return pd.DataFrame(
        {
            "length": [0.2] * n,
            "elastic_modulus": [200e9] * n,
            "moment_inertia": [1e-8] * n,
            "density": [8000.0] * n,
            "cross_area": [1e-4] * n,
            "type": ["linear", "linear", "nonlinear", "nonlinear", "nonlinear"],
            "boundary_condition": bcs,
            "wetted_area": [1e-4] * n,
            "drag_coef": [1.2] * n,
        }
    )


def hetero7(bcs=None):
    """Heterogeneous 7-segment beam (ours): every column varies along the beam."""
    n = 7
    rng = np.random.default_rng(77)
    bcs = bcs or ["FIXED"] + ["NONE"] * (n - 1)
    return pd.DataFrame(
        {
            "length": 0.15 + 0.2 * rng.random(n),
            "elastic_modulus": 75e9 * (0.8 + 0.4 * rng.random(n)),
            "moment_inertia": 4.9e-10 * (0.7 + 0.6 * rng.random(n)),
            "density": 6450.0 * (0.8 + 0.4 * rng.random(n)),
            "cross_area": 7.85e-5 * (0.8 + 0.4 * rng.random(n)),
            "type": ["nonlinear", "linear", "linear", "nonlinear", "nonlinear", "linear", "nonlinear"],
            "boundary_condition": bcs,
            "wetted_area": 0.001 * (0.5 + rng.random(n)),
            "drag_coef": 0.3 + rng.random(n),
        }
    )


def df_arrays(prefix, df):
    out = {}
    for c in COLS:
        v = df[c].to_numpy()
        out[f"{prefix}/{c}"] = v.astype(str) if c in ("type", "boundary_condition") else v.astype(np.float64)
    return out


def write_csv(df):
    f = tempfile.NamedTemporaryFile(mode="w", delete=False, suffix=".csv")
    df[COLS].to_csv(f, index=False)
    f.close()
    return f.name


# --------------------------------------------------------------------------- G1
def g1_elements():
    out = {}
    r = 0.005
    mats = np.array(
        [
            [1.0, 200e9, 1e-6, 7850.0, 1e-4],
            [0.25, 75e9, np.pi * r**4 / 4, 6450.0, np.pi * r**2],
            [0.2, 200e9, 1e-8, 8000.0, 1e-4],
        ]
    )
    rng = np.random.default_rng(101)
    states = np.vstack([[0.01, 0.001, 0.1, 0.02, 0.002, 0.2], rng.normal(0.0, 1e-2, (16, 6))])
    K, M, F = [], [], []
    for L, E, I, rho, A in mats:
        lin = LinearSegment(Properties(L, E, I, rho, A, 0, "linear"))
        nl = NonlinearSegment(Properties(L, E, I, rho, A, 0, "nonlinear"))
        K.append(lin.get_stiffness_func())
        M.append(lin.get_mass_matrix())
        assert np.array_equal(M[-1], nl.get_mass_matrix())
        fn = nl.get_stiffness_func()
        F.append(np.array([fn(s) for s in states]))
    out["materials"] = mats  # columns: L, E, I, rho, A
    out["states"] = states
    out["K_e"] = np.array(K)
    out["M_e"] = np.array(M)
    out["f_nl"] = np.array(F)  # [material, state, 6] in the reference's output order
    np.savez_compressed(os.path.join(HERE, "g1_elements.npz"), **out)
    print("G1 anchor f_nl[0,0] =", repr(out["f_nl"][0, 0]))


# --------------------------------------------------------------------------- G2
BC_SETS = {
    "none": {},
    "fixed0": {0: "fixed"},
    "pinned0": {0: "pinned"},
    "fixed0_pinned2": {0: "fixed", 2: "pinned"},
    "pinned0_pinnedN": {0: "pinned", -1: "pinned"},
}


def g2_assembly():
    out = {}
    beams = {"test4_lin": test4("linear"), "test4_nl": test4("nonlinear"), "mixed5": mixed5(), "hetero7": hetero7()}
    rng = np.random.default_rng(202)
    for bname, df in beams.items():
        out.update(df_arrays(f"{bname}", df))
        n_nodes = len(df) + 1
        for sname, bcs in BC_SETS.items():
            key = f"{bname}/{sname}"
            beam = EulerBernoulliBeam(df)
            conds = {
                (k if k >= 0 else n_nodes + k): BoundaryConditionType(v) for k, v in bcs.items()
            }
            node_bc = np.zeros(n_nodes, dtype=np.int32)
            for k, v in conds.items():
                node_bc[k] = 1 if v == BoundaryConditionType.FIXED else 2
            if conds:
                beam.apply_boundary_conditions(conds)
            M = beam.get_mass_matrix()
            n = M.shape[0]
            out[f"{key}/node_bc"] = node_bc
            out[f"{key}/M"] = M
            out[f"{key}/constrained"] = np.array(sorted(beam.get_constrained_dofs()), dtype=np.int32)
            names = {"u": 0, "w": 1, "phi": 2}
            out[f"{key}/dof_param"] = np.array([names[beam.dof_to_node_param[i][0]] for i in range(n)], dtype=np.int32)
            out[f"{key}/dof_node"] = np.array([beam.dof_to_node_param[i][1] for i in range(n)], dtype=np.int32)
            if not beam.is_hybrid() and bname == "test4_lin":
                out[f"{key}/K"] = beam.get_stiffness_matrix()
            q = rng.normal(0.0, 1e-2, (3, n))
            out[f"{key}/q"] = q
            out[f"{key}/k_q"] = np.array([beam.get_stiffness_function()(qq) for qq in q])
    np.savez_compressed(os.path.join(HERE, "g2_assembly.npz"), **out)


# --------------------------------------------------------------------------- G3/G4
FORCE_SETS = {
    "none": dict(),
    "drag": dict(fluid_density=1000.0, enable_fluid_effects=True),
    "grav": dict(enable_gravity_effects=True),
    "both": dict(fluid_density=1000.0, enable_fluid_effects=True, enable_gravity_effects=True),
    "grav_xy": dict(gravity_vector=[3.0, -9.81, 0.0], enable_gravity_effects=True),
    "both_xy": dict(
        fluid_density=870.0, enable_fluid_effects=True, gravity_vector=[3.0, -9.81, 0.5], enable_gravity_effects=True
    ),
}

DYN_BEAMS = {
    "test4_lin": lambda: test4("linear"),
    "test4_nl": lambda: test4("nonlinear"),
    "mixed5": lambda: mixed5(),
    "hetero7": lambda: hetero7(),
    "test4_nl_pinned0": lambda: test4("nonlinear", ["PINNED", "NONE", "NONE", "NONE"]),
    "mixed5_fixed0_pinned2": lambda: mixed5(["FIXED", "NONE", "PINNED", "NONE", "NONE"]),
    "hetero7_free": lambda: hetero7(["NONE"] * 7),
    "hetero7_pinned0_fixed3": lambda: hetero7(["PINNED", "NONE", "NONE", "FIXED", "NONE", "NONE", "NONE"]),
}


def fp_arrays(prefix, kw):
    fp = ForceParams(**kw)
    return {
        f"{prefix}/fluid_density": np.float64(fp.fluid_density),
        f"{prefix}/enable_fluid": np.int32(fp.enable_fluid_effects),
        f"{prefix}/gravity": fp.get_gravity_vector(),
        f"{prefix}/enable_gravity": np.int32(fp.enable_gravity_effects),
    }


def g34_forces_rhs():
    out = {}
    rng = np.random.default_rng(303)
    for bname, mk in DYN_BEAMS.items():
        df = mk()
        path = write_csv(df)
        # store the columns as the reference PARSED them (pandas' default read_csv float
        # parser is not round-trip exact, so these differ from df by a few ulp)
        out.update(df_arrays(bname, pd.read_csv(path)))
        try:
            for fname, kw in FORCE_SETS.items():
                key = f"{bname}/{fname}"
                out.update(fp_arrays(key, kw))
                beam = DynamicEulerBernoulliBeam(path, force_params=ForceParams(**kw))
                beam.create_system_func()
                beam.create_input_func()
                n = beam.beam_model.M.shape[0]
                X = rng.normal(0.0, 1e-2, (3, 2 * n))
                X[0, n:] *= 50.0  # larger velocities: drag matters
                out[f"{key}/x"] = X
                out[f"{key}/M_inv"] = beam.M_inv.toarray() if fname == "none" else np.zeros(0)
                out[f"{key}/constrained"] = np.array(sorted(beam.constrained_dofs), dtype=np.int32)
                forces = beam.force_registry.get_registered_forces()
                agg = beam.force_registry.create_aggregated_function()
                out[f"{key}/f_total"] = np.array([agg(x, 0.0) for x in X])
                for f in forces:
                    nm = type(f).__name__
                    out[f"{key}/{nm}"] = np.array([f.compute_forces(x, 0.0) for x in X])
                    if nm == "FluidDragForce":
                        fc = f.fluid_coefficients
                        out[f"{key}/drag_w_vel_indices"] = np.array(fc["w_vel_indices"], dtype=np.int32)
                        out[f"{key}/drag_w_pos_indices"] = np.array(fc["w_pos_indices"], dtype=np.int32)
                        out[f"{key}/drag_factors"] = np.array(fc["drag_factors"], dtype=np.float64)
                dyn = beam.get_dynamic_system()
                u_tip = np.zeros(n)
                u_tip[-2] = 0.1
                U = np.vstack([np.zeros(n), np.ones(n), u_tip, rng.normal(0.0, 1.0, n)])
                out[f"{key}/u"] = U
                out[f"{key}/xdot"] = np.array([[dyn(0.0, x, u) for u in U] for x in X])
        finally:
            os.unlink(path)
    np.savez_compressed(os.path.join(HERE, "g34_forces_rhs.npz"), **out)


# --------------------------------------------------------------------------- G5
def rk4(dyn, u_of_t, x0, t0, dt, n_steps, checkpoints=()):
    x = x0.copy()
    snaps = {}
    t = t0
    for n in range(n_steps):
        th = t + 0.5 * dt
        t1 = t + dt
        k1 = dyn(t, x, u_of_t(t))
        k2 = dyn(th, x + (0.5 * dt) * k1, u_of_t(th))
        k3 = dyn(th, x + (0.5 * dt) * k2, u_of_t(th))
        k4 = dyn(t1, x + dt * k3, u_of_t(t1))
        x = x + (dt / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
        t = t + dt
        if (n + 1) in checkpoints:
            snaps[n + 1] = x.copy()
    return x, snaps


def g5_rollouts():
    out = {}
    dt = 2e-5
    jobs = [
        # name, df, force kw, amp, checkpoints, random x0?
        ("lin10_grav", nitinol(10, "linear"), dict(enable_gravity_effects=True), 0.1, (1000, 5000), False),
        ("lin64_grav", nitinol(64, "linear"), dict(enable_gravity_effects=True), 0.1, (200, 1000), False),
        ("lin64_grav_x0", nitinol(64, "linear"), dict(enable_gravity_effects=True), 0.137, (200, 1000), True),
        ("nl64_drag", nitinol(64, "nonlinear"), dict(fluid_density=1000.0, enable_fluid_effects=True), 0.1, (200, 1000), False),
        ("nl256_drag", nitinol(256, "nonlinear"), dict(fluid_density=1000.0, enable_fluid_effects=True), 0.1, (200, 1000), False),
        ("nl256_drag_a2", nitinol(256, "nonlinear"), dict(fluid_density=1000.0, enable_fluid_effects=True), 0.2, (200,), False),
        ("mixed5_both", mixed5(), FORCE_SETS["both_xy"], 0.5, (300,), False),
        ("hetero7_both", hetero7(), FORCE_SETS["both"], 0.3, (300,), False),
        ("hetero7_p0f3_grav", hetero7(["PINNED", "NONE", "NONE", "FIXED", "NONE", "NONE", "NONE"]), FORCE_SETS["grav_xy"], 0.3, (300,), False),
    ]
    for name, df, kw, amp, cps, rand_x0 in jobs:
        t_start = time.time()
        out.update(fp_arrays(name, kw))
        path = write_csv(df)
        try:
            beam = DynamicEulerBernoulliBeam(path, force_params=ForceParams(**kw))
        finally:
            os.unlink(path)
        out.update(df_arrays(name, beam.params))  # as parsed by the reference (see g34)
        beam.create_system_func()
        beam.create_input_func()
        dyn = beam.get_dynamic_system()
        n = beam.beam_model.M.shape[0]
        x0 = np.zeros(2 * n)
        if rand_x0:
            # sigma_q=1e-5, sigma_v=1e-3 on free w/phi DOFs, axial 0
            rng = np.random.default_rng(1234)
            q = rng.normal(0.0, 1e-5, n)
            v = rng.normal(0.0, 1e-3, n)
            for i in range(n):
                if beam.beam_model.dof_to_node_param[i][0] == "u":
                    q[i] = 0.0
                    v[i] = 0.0
            x0 = np.concatenate([q, v])

        def u_of_t(t, n=n, amp=amp):
            u = np.zeros(n)
            if t < 0.01:
                u[-2] = amp
            return u

        xT, snaps = rk4(dyn, u_of_t, x0, 0.0, dt, max(cps), cps)
        out[f"{name}/x0"] = x0
        out[f"{name}/amp"] = np.float64(amp)
        out[f"{name}/dt"] = np.float64(dt)
        out[f"{name}/duration"] = np.float64(0.01)
        out[f"{name}/checkpoints"] = np.array(cps, dtype=np.int32)
        for c in cps:
            out[f"{name}/x_{c}"] = snaps[c]
            print(f"G5 {name}: tip w @{c} = {snaps[c][n - 2]!r}")
        print(f"   ({time.time() - t_start:.1f} s)")
    np.savez_compressed(os.path.join(HERE, "g5_rollouts.npz"), **out)


# --------------------------------------------------------------------------- G6
def g6_lqr_loop():
    """Closed-loop rollout of examples/lqr_control.py:87-130 with fixed-step RK4: the controller
    u = K (r - x) (control/full_state_linear.py:81, r = 0) is evaluated inside the RHS at every stage,
    plus the tip impulse of lqr_control.py:33-41.  `control` (python-control) is not installed, so the
    gain comes from scipy's CARE solver on the A/B of linear_quadratic_regulator.py:84-146 built from
    the reference's own K and M; the gain is stored, so the fixture pins the LOOP, not the solver."""
    from scipy.linalg import solve_continuous_are

    out = {}
    # The closed loop has |lambda|max = 3.24e5 1/s (velocity feedback through the tiny rotational
    # inertias), so explicit RK4 needs dt <= 2.78/|lambda| = 8.6e-6 s; dt = 2e-5 blows up.
    dt = 5e-6
    for name, n_seg, steps in (("lqr6", 6, 3000), ("lqr24", 24, 600)):
        df = nitinol(n_seg, "linear")
        kw = dict(enable_gravity_effects=True)
        out.update(fp_arrays(name, kw))
        path = write_csv(df)
        try:
            beam = DynamicEulerBernoulliBeam(path, force_params=ForceParams(**kw))
        finally:
            os.unlink(path)
        out.update(df_arrays(name, beam.params))
        beam.create_system_func()
        beam.create_input_func()
        dyn = beam.get_dynamic_system()
        Kb, Mb = beam.beam_model.get_stiffness_matrix(), beam.beam_model.get_mass_matrix()
        n = Kb.shape[0]
        Minv = np.linalg.inv(Mb)
        A = np.zeros((2 * n, 2 * n))
        A[:n, n:] = np.eye(n)
        A[n:, :n] = -Minv @ Kb
        Bm = np.zeros((2 * n, n))
        Bm[n:, :] = Minv
        Q = np.eye(2 * n)
        Q[:n, :n] *= 100  # lqr_control.py:61-66
        Q[n:, n:] *= 10
        R = np.eye(n)
        S = solve_continuous_are(A, Bm, Q, R)
        gain = np.linalg.solve(R, Bm.T @ S)
        assert np.all(np.real(np.linalg.eigvals(A - Bm @ gain)) < 0)
        amp = 10.0
        x = np.zeros(2 * n)
        ref = np.zeros(2 * n)

        def total_input(t, xx):
            u = gain @ (ref - xx)
            if t < 0.01:
                u[-2] += amp
            return u

        t = 0.0
        for _ in range(steps):
            th, t1 = t + 0.5 * dt, t + dt
            k1 = dyn(t, x, total_input(t, x))
            x2 = x + (0.5 * dt) * k1
            k2 = dyn(th, x2, total_input(th, x2))
            x3 = x + (0.5 * dt) * k2
            k3 = dyn(th, x3, total_input(th, x3))
            x4 = x + dt * k3
            k4 = dyn(t1, x4, total_input(t1, x4))
            x = x + (dt / 6.0) * (k1 + 2.0 * k2 + 2.0 * k3 + k4)
            t = t + dt
        out[f"{name}/gain"] = gain
        out[f"{name}/K"] = Kb
        out[f"{name}/M"] = Mb
        out[f"{name}/amp"] = np.float64(amp)
        out[f"{name}/dt"] = np.float64(dt)
        out[f"{name}/steps"] = np.int32(steps)
        out[f"{name}/x_final"] = x
        print(f"G6 {name}: tip w @{steps} = {x[n - 2]!r}, max|gain| = {np.abs(gain).max():.3f}")
    np.savez_compressed(os.path.join(HERE, "g6_lqr_loop.npz"), **out)


# --------------------------------------------------------------------------- G7
def g7_rk45():
    """scipy.integrate.solve_ivp(method="RK45") over the reference RHS, the call pattern of the reference's
    own tests (tests/test_dynamic_beam.py:218-220: default rtol/atol; test_functional_composition.py:539-546:
    rtol=1e-6), with the examples' tip impulse.  Pins terminal state, accepted steps and nfev."""
    from scipy.integrate import solve_ivp

    out = {}
    jobs = [
        ("lin40_grav", nitinol(40, "linear"), dict(enable_gravity_effects=True), 0.1, 0.002, 0.004, 1e-3, 1e-6),
        ("nl64_drag", nitinol(64, "nonlinear"), dict(fluid_density=1000.0, enable_fluid_effects=True), 0.1, 0.001, 0.002,
         1e-6, 1e-9),
    ]
    for name, df, kw, amp, duration, t_end, rtol, atol in jobs:
        t_start = time.time()
        out.update(fp_arrays(name, kw))
        path = write_csv(df)
        try:
            beam = DynamicEulerBernoulliBeam(path, force_params=ForceParams(**kw))
        finally:
            os.unlink(path)
        out.update(df_arrays(name, beam.params))
        beam.create_system_func()
        beam.create_input_func()
        dyn = beam.get_dynamic_system()
        n = beam.beam_model.M.shape[0]

        def u_of_t(t, n=n, amp=amp, duration=duration):
            u = np.zeros(n)
            if t < duration:
                u[-2] = amp
            return u

        sol = solve_ivp(lambda t, x: dyn(t, x, u_of_t(t)), (0.0, t_end), np.zeros(2 * n), method="RK45", rtol=rtol,
                        atol=atol)
        assert sol.success
        for k, v in (("amp", amp), ("duration", duration), ("t_end", t_end), ("rtol", rtol), ("atol", atol)):
            out[f"{name}/{k}"] = np.float64(v)
        out[f"{name}/x_final"] = sol.y[:, -1]
        out[f"{name}/nfev"] = np.int32(sol.nfev)
        out[f"{name}/accepted"] = np.int32(len(sol.t) - 1)
        out[f"{name}/t_steps"] = sol.t
        # the same run sampled like the examples do (t_eval=np.arange(...), example_utilities.py:158)
        n_eval = 37
        eval_dt = t_end / n_eval
        sol_e = solve_ivp(lambda t, x: dyn(t, x, u_of_t(t)), (0.0, t_end), np.zeros(2 * n), method="RK45", rtol=rtol,
                          atol=atol, t_eval=0.0 + np.arange(n_eval) * eval_dt)
        assert sol_e.success and sol_e.nfev == sol.nfev
        out[f"{name}/eval_dt"] = np.float64(eval_dt)
        out[f"{name}/tip_w_eval"] = sol_e.y[n - 2]
        out[f"{name}/tip_dw_eval"] = sol_e.y[2 * n - 2]
        print(f"G7 {name}: {len(sol.t) - 1} steps, nfev {sol.nfev}, tip w = {sol.y[n - 2, -1]!r} ({time.time() - t_start:.0f} s)")
    np.savez_compressed(os.path.join(HERE, "g7_rk45.npz"), **out)


def g8_lsoda():
    """scipy.integrate.solve_ivp(method="LSODA") over the reference RHS: the integration the reference's examples run
    (examples/example_utilities.py:153-159: LSODA, default rtol 1e-3 / atol 1e-6, tip impulse 0.1 N for t < 0.01 s),
    once at TIGHT tolerances (rtol 1e-10, atol 1e-13: the ODE's solution, what an implicit stepper must converge to)
    and once at the example's own default tolerances (the band the example itself lands in).  Config 1 of
    BASELINE.json (10 linear elements + gravity) and the 6-segment beams of the parallel examples."""
    from scipy.integrate import solve_ivp

    out = {}
    half = 3
    jobs = [
        ("lin10_grav", nitinol(10, "linear"), dict(enable_gravity_effects=True), [0.02, 0.05, 0.1]),
        ("lin6_fluid", nitinol(6, "linear"), dict(fluid_density=1000.0, enable_fluid_effects=True), [0.02, 0.05]),
        ("mixed6_fluid", nitinol(6, ["linear"] * half + ["nonlinear"] * (6 - half)),
         dict(fluid_density=1000.0, enable_fluid_effects=True), [0.02, 0.05]),
    ]
    for name, df, kw, times in jobs:
        t_start = time.time()
        out.update(fp_arrays(name, kw))
        path = write_csv(df)
        try:
            beam = DynamicEulerBernoulliBeam(path, force_params=ForceParams(**kw))
        finally:
            os.unlink(path)
        out.update(df_arrays(name, beam.params))
        beam.create_system_func()
        beam.create_input_func()
        dyn = beam.get_dynamic_system()
        n = beam.beam_model.M.shape[0]
        amp, duration = 0.1, 0.01

        def u_of_t(t, n=n):
            u = np.zeros(n)
            if t < duration:
                u[-2] = amp
            return u

        def run(rtol, atol):
            # (integrate piecewise over the impulse switch-off so that tight tolerances are meaningful there)
            x, t0, ys, nfev = np.zeros(2 * n), 0.0, [], 0
            for t1 in sorted(set([duration] + list(times))):
                sol = solve_ivp(lambda t, x: dyn(t, x, u_of_t(0.5 * (t0 + t1))), (t0, t1), x, method="LSODA", rtol=rtol, atol=atol)
                assert sol.success
                x, t0, nfev = sol.y[:, -1], t1, nfev + sol.nfev
                if t1 in times:
                    ys.append(x.copy())
            return np.array(ys), nfev

        tight, nfev_t = run(1e-10, 1e-13)
        loose, nfev_l = run(1e-3, 1e-6)
        out[f"{name}/amp"], out[f"{name}/duration"] = np.float64(amp), np.float64(duration)
        out[f"{name}/times"] = np.array(times)
        out[f"{name}/x_tight"] = tight
        out[f"{name}/x_default_tol"] = loose
        out[f"{name}/nfev_tight"], out[f"{name}/nfev_default_tol"] = np.int64(nfev_t), np.int64(nfev_l)
        print(f"G8 {name}: tip w(t_end) tight {tight[-1][n - 2]!r} / default tol {loose[-1][n - 2]!r}; nfev {nfev_t} / {nfev_l} "
              f"({time.time() - t_start:.0f} s)", flush=True)
    np.savez_compressed(os.path.join(HERE, "g8_lsoda.npz"), **out)


def g8_long():
    """BASELINE config 1 over the example's FULL horizon: LSODA over the reference RHS to t = 1 s (10 linear elements +
    gravity, tip impulse 0.1 N for t < 0.01 s; examples/example_utilities.py:153-159), sampled every 0.1 s, at
    rtol 1e-8 / atol 1e-11 and at the example's default tolerances.  Merged into g8_lsoda.npz (keys lin10_grav_1s/*).
    ~5 minutes."""
    from scipy.integrate import solve_ivp

    df = nitinol(10, "linear")
    path = write_csv(df)
    try:
        beam = DynamicEulerBernoulliBeam(path, force_params=ForceParams(enable_gravity_effects=True))
    finally:
        os.unlink(path)
    beam.create_system_func()
    beam.create_input_func()
    dyn = beam.get_dynamic_system()
    n = beam.beam_model.M.shape[0]
    times = np.round(np.arange(0.1, 1.0001, 0.1), 10)
    out = {}
    for tag, rtol, atol in (("default_tol", 1e-3, 1e-6), ("tight", 1e-8, 1e-11)):
        t_start = time.time()
        x, t0, ys, nfev = np.zeros(2 * n), 0.0, [], 0
        for t1 in [0.01] + list(times):        # piecewise over the impulse switch-off
            u = np.zeros(n)
            if 0.5 * (t0 + t1) < 0.01:
                u[-2] = 0.1
            sol = solve_ivp(lambda t, y: dyn(t, y, u), (t0, t1), x, method="LSODA", rtol=rtol, atol=atol)
            assert sol.success
            x, t0, nfev = sol.y[:, -1], t1, nfev + sol.nfev
            if t1 >= 0.1 - 1e-12:
                ys.append(x.copy())
        out[f"lin10_grav_1s/x_{tag}"] = np.array(ys)
        out[f"lin10_grav_1s/nfev_{tag}"] = np.int64(nfev)
        print(f"G8 long {tag}: tip w(1 s) = {x[n - 2]!r}, nfev {nfev} ({time.time() - t_start:.0f} s)", flush=True)
    out["lin10_grav_1s/times"] = times
    f = os.path.join(HERE, "g8_lsoda.npz")
    old = dict(np.load(f, allow_pickle=False))
    old.update(out)
    np.savez_compressed(f, **old)


# --------------------------------------------------------------------------- G9
def g9_reference_suite():
    """Numbers behind the reference's own GPU-dependent tests (tests/README.md maps every one of the 34 to its restated test
    in tests/test_reference_suite_restated.py): the beams those tests build (their CSV rows are data), SEEDED states in
    place of their unseeded np.random draws, and what the reference returns on them -- system / input / dynamic-system
    vectors, registry aggregates, fluid_coefficients, the segment and beam stiffness functions, and the end states of the
    solve_ivp calls the tests make (RK45 at scipy's defaults over [0, 0.1] s with u = sin(t); rtol 1e-6 over 10 ms)."""
    from scipy.integrate import solve_ivp

    from continuum_robot.models.abstractions import AbstractForce, AbstractInputHandler

    rng = np.random.default_rng(99)
    out = {}

    def frame(rows, cols=COLS):
        return pd.DataFrame({c: [r[i] for r in rows] for i, c in enumerate(cols)})

    def store(prefix, df):
        for c in df.columns:
            v = df[c].to_numpy()
            out[f"{prefix}/{c}"] = v.astype(str) if c in ("type", "boundary_condition") else v.astype(np.float64)

    def csv_of(df):
        f = tempfile.NamedTemporaryFile(mode="w", delete=False, suffix=".csv")
        df.to_csv(f, index=False)
        f.close()
        return f.name

    bc4 = ["FIXED", "NONE", "NONE", "NONE"]
    beams = {
        "t4lin": frame([(0.25, 75e9, 4.91e-10, 6450, 7.85e-5, "linear", bc, 0.001, 0.5) for bc in bc4]),
        "t4nl": frame([(0.25, 75e9, 4.91e-10, 6450, 7.85e-5, "nonlinear", bc, 0.001, 0.5) for bc in bc4]),
        "t4fluid": frame([(0.25, 75e9, 4.91e-10, 6450, 7.85e-5, "linear", bc, 0.001, 1.2) for bc in bc4]),
        "fc4": frame([(0.25, 200e9, 1e-8, 8000, 1e-4, "linear", bc, 1e-4, 1.2) for bc in bc4]),
        "cx5": frame([(0.2, 200e9, 1e-8, 8000, 1e-4, k, bc, 1e-4, 1.2)
                      for k, bc in zip(["linear", "linear", "nonlinear", "nonlinear", "nonlinear"], ["FIXED"] + ["NONE"] * 4)]),
        "mix2": frame([(1.0, 200e9, 1e-6, 7850, 1e-4, k, "NONE") for k in ("linear", "nonlinear")], COLS[:7]),
        "hyb3": frame([(1.0, 200e9, 1e-6, 7850, 1e-4, k, bc) for k, bc in zip(("linear", "nonlinear", "linear"), ("FIXED", "NONE", "NONE"))],
                      COLS[:7]),
    }
    files = {}
    for name, df in beams.items():
        store(name, df)
        files[name] = csv_of(df)

    class TipSpringDamper(AbstractForce):   # the state-dependent force those tests register: spring + damper on the last w DOF
        def __init__(self, k, c=10.0):
            self.k, self.c, self.enabled = k, c, True

        def compute_forces(self, x, t):
            n = len(x) // 2
            f = np.zeros(n)
            f[n - 2] = -self.k * x[n - 2] - self.c * x[n + n - 2]
            return f

        def is_enabled(self):
            return self.enabled

    class Gain(AbstractInputHandler):
        def __init__(self, g):
            self.g = g

        def compute_input(self, x, u, t):
            return u * self.g

        def is_enabled(self):
            return True

    def dyn(name, **fp):
        b = DynamicEulerBernoulliBeam(files[name], force_params=ForceParams(**fp) if fp else None)
        b.create_system_func()
        b.create_input_func()
        return b

    # ---- test_dynamic_beam.py: system creation, the three solve_ivp tests, fluid_coefficients
    for name in ("t4lin", "t4nl"):
        b = dyn(name)
        n = b.beam_model.M.shape[0]
        out[f"{name}/dyn_zero_ones"] = b.get_dynamic_system()(0, np.zeros(2 * n), np.ones(n))
    t0 = time.time()
    for name, fp in (("t4lin", {}), ("t4nl", {}), ("t4lin", dict(fluid_density=1000.0, enable_fluid_effects=True)),
                     ("t4nl", dict(fluid_density=1000.0, enable_fluid_effects=True)),
                     ("t4nl", dict(fluid_density=2000.0, enable_fluid_effects=True))):
        b = dyn(name, **fp)
        n = b.beam_model.M.shape[0]
        f = b.get_dynamic_system()
        sol = solve_ivp(lambda t, x: f(t, x, np.sin(t) * np.ones(n)), [0, 0.1], np.zeros(2 * n))
        key = f"{name}/ivp_rho{int(fp.get('fluid_density', 0))}"
        out[key + "/y_end"] = sol.y[:, -1]
        out[key + "/nfev"] = np.array(sol.nfev)
        print(f"g9 {key}: success {sol.success}, nfev {sol.nfev}, {time.time() - t0:.0f} s", flush=True)
    b = dyn("t4fluid", fluid_density=1000.0, enable_fluid_effects=True)
    fl = [c for c in b.force_registry.get_registered_forces() if hasattr(c, "fluid_coefficients")][0]
    for k in ("w_vel_indices", "w_pos_indices", "drag_factors"):
        out[f"t4fluid/fluid_coefficients/{k}"] = np.asarray(fl.fluid_coefficients[k])
    ns = len(b.state_to_node_param)
    out["t4fluid/dyn_ones_zero"] = b.get_dynamic_system()(0.0, np.ones(ns), np.zeros(ns // 2))

    # ---- test_functional_composition.py (beam fc4)
    n = len(dyn("fc4").state_to_node_param) // 2
    x = rng.random(2 * n) * 0.01
    u = rng.random(n) * 0.1
    out["fc4/x"], out["fc4/u"] = x, u
    out["fc4/sys_fluid"] = dyn("fc4", fluid_density=1000.0, enable_fluid_effects=True).get_system_func()(x)
    out["fc4/sys_plain"] = dyn("fc4").get_system_func()(x)
    out["fc4/sys_plain_big"] = dyn("fc4").get_system_func()(10 * x)
    out["fc4/sys_gravity_zero"] = dyn("fc4", enable_gravity_effects=True).get_system_func()(np.zeros(2 * n))
    b = DynamicEulerBernoulliBeam(files["fc4"])
    spring500 = TipSpringDamper(500.0, 0.0)
    b.create_system_func(lambda xx, t: spring500.compute_forces(xx, t))
    tip = np.zeros(2 * n)
    tip[n - 2] = 0.01
    out["fc4/sys_spring500"], out["fc4/sys_spring500_tip"] = b.get_system_func()(x), b.get_system_func()(tip)
    b.create_system_func(lambda xx, t: np.concatenate(([0.0, 100.0 * np.sin(2 * np.pi * t)], np.zeros(len(xx) // 2 - 2))))
    out["fc4/sys_time_force_zero"] = b.get_system_func()(np.zeros(2 * n))
    b = dyn("fc4", fluid_density=1000.0, enable_fluid_effects=True, enable_gravity_effects=True)
    reg = b.force_registry.create_aggregated_function()
    spring200 = TipSpringDamper(200.0, 0.0)
    out["fc4/registry_forces"] = reg(x, 0.0)
    b.create_system_func(lambda xx, t: reg(xx, t) + spring200.compute_forces(xx, t))
    out["fc4/sys_hybrid"] = b.get_system_func()(x)
    b = DynamicEulerBernoulliBeam(files["fc4"])
    b.create_system_func(lambda xx, t: np.concatenate(([0.0, 200.0], np.zeros(len(xx) // 2 - 2))))
    out["fc4/sys_mock200_zero"] = b.get_system_func()(np.zeros(2 * n))
    b = dyn("fc4")
    out["fc4/input_default"] = b.input_func(x, u)
    out["fc4/input_doubled"] = b.input_func(x, 2.0 * u)
    b.input_registry.register(Gain(0.1))
    b.input_registry.register(Gain(0.2))
    out["fc4/input_aggregated_ones"] = b.input_registry.create_aggregated_function()(x, np.ones(n), 0.0)
    x0 = rng.random(2 * n) * 0.001
    f = dyn("fc4").get_dynamic_system()
    sol = solve_ivp(lambda t, xx: f(t, xx, np.zeros(n)), (0, 0.01), x0, t_eval=np.linspace(0, 0.01, 10), method="RK45", rtol=1e-6)
    out["fc4/ivp_x0"], out["fc4/ivp_y"] = x0, sol.y
    print(f"g9 fc4 ivp: success {sol.success}, nfev {sol.nfev}", flush=True)

    # ---- test_advanced_composition.py (beam cx5)
    n = len(dyn("cx5").state_to_node_param) // 2
    x = rng.random(2 * n) * 0.01
    u = rng.random(n) * 0.1
    out["cx5/x"], out["cx5/u"] = x, u
    b = dyn("cx5", fluid_density=1000.0, enable_fluid_effects=True, enable_gravity_effects=True)
    b.force_registry.register(TipSpringDamper(500.0, 5.0))
    b.create_system_func()
    out["cx5/sys_fluid_gravity_spring"] = b.get_system_func()(x)
    b = DynamicEulerBernoulliBeam(files["cx5"])
    b.force_registry.register(TipSpringDamper(100.0))
    b.force_registry.register(TipSpringDamper(200.0))
    b.create_system_func()
    out["cx5/sys_two_springs"] = b.get_system_func()(x)
    for scale in (1.0, 2.5):
        b = DynamicEulerBernoulliBeam(files["cx5"])
        b.create_system_func(lambda xx, t, s=scale: s * np.concatenate(([0.0, 100.0], np.zeros(len(xx) // 2 - 2))))
        out[f"cx5/sys_const_force_x{scale}"] = b.get_system_func()(np.zeros(2 * n))
    b = DynamicEulerBernoulliBeam(files["cx5"])
    for i in range(50):
        b.force_registry.register(TipSpringDamper(100.0 + i, 1.0 + 0.1 * i))
    b.create_system_func()
    out["cx5/sys_fifty_springs"] = b.get_system_func()(x)
    b = DynamicEulerBernoulliBeam(files["cx5"])
    keep = [TipSpringDamper(100.0 + i) for i in range(20)]
    for fo in keep:
        b.force_registry.register(fo)
    for fo in keep[:10]:
        b.force_registry.unregister(fo)
    b.create_system_func()
    out["cx5/sys_ten_of_twenty_springs"] = b.get_system_func()(x)
    b = DynamicEulerBernoulliBeam(files["cx5"])
    sp = TipSpringDamper(1000.0)
    b.force_registry.register(sp)
    b.create_system_func()
    b.create_input_func()
    out["cx5/sys_spring1000"] = b.get_system_func()(x)
    out["cx5/input_default"] = b.input_func(x, u)
    sp.enabled = False
    out["cx5/sys_spring1000_disabled"] = b.get_system_func()(x)
    small = np.ones(2 * n) * 0.001
    out["cx5/sys_plain_small_ones"] = dyn("cx5").get_system_func()(small)
    b = dyn("cx5", fluid_density=1000.0, enable_fluid_effects=True, enable_gravity_effects=True)
    b.force_registry.register(TipSpringDamper(500.0, 10.0))
    reg = b.force_registry.create_aggregated_function()
    b.create_system_func(lambda xx, t: reg(xx, t) + np.concatenate((np.zeros(len(xx) // 2 - 2), [50.0 * np.sin(10.0 * t), 0.0])))
    xs = 0.1 * x
    out["cx5/xs"] = xs
    out["cx5/sys_full"], out["cx5/dyn_full"] = b.get_system_func()(xs), b.get_dynamic_system()(0.1, xs, np.zeros(n))
    b = dyn("cx5")
    out["cx5/input_ones"] = b.input_func(x, np.ones(n))

    # ---- test_unified_beam_system.py
    seg = NonlinearSegment(Properties(length=1.0, elastic_modulus=200e9, moment_inertia=1e-6, density=7850, cross_area=1e-4,
                                      segment_id=0, element_type="nonlinear"))
    st6 = np.array([0.01, 0.001, 0.1, 0.02, 0.002, 0.2])
    out["seg_nl/state"], out["seg_nl/forces"] = st6, seg.get_stiffness_func()(st6)
    hyb = EulerBernoulliBeam(beams["mix2"][COLS[:6]])
    nd = hyb.get_mass_matrix().shape[0]
    q = rng.random(nd) * 0.01
    out["mix2/q"], out["mix2/beam_stiffness"] = q, hyb.get_stiffness_function()(q)
    b = dyn("mix2")
    nd = b.beam_model.get_mass_matrix().shape[0]
    x = rng.random(2 * nd) * 0.01
    out["mix2/x"], out["mix2/sys"], out["mix2/dyn"] = x, b.get_system_func()(x), b.get_dynamic_system()(0.0, x, np.zeros(nd))
    b = dyn("hyb3")
    nd = b.beam_model.get_mass_matrix().shape[0]
    x = np.zeros(2 * nd)
    x[nd:] = 0.01
    out["hyb3/dyn_initial_velocity"] = b.get_dynamic_system()(0.0, x, np.zeros(nd))
    out["hyb3/is_hybrid"] = np.array(b.beam_model.is_hybrid())

    np.savez_compressed(os.path.join(HERE, "g9_reference_suite.npz"), **out)
    for f in files.values():
        os.unlink(f)
    print(f"g9: {len(out)} arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g34", "g5", "g6", "g7", "g8", "g8long", "g9"]
    if "g9" in which:
        g9_reference_suite()
    if "g8long" in which:
        g8_long()
    if "g8" in which:
        g8_lsoda()
    if "g7" in which:
        g7_rk45()
    if "g6" in which:
        g6_lqr_loop()
    if "g1" in which:
        g1_elements()
    if "g2" in which:
        g2_assembly()
    if "g34" in which:
        g34_forces_rhs()
    if "g5" in which:
        g5_rollouts()
