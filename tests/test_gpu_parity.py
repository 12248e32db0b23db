"""GPU parity tests proper: the HIP path (through the C ABI, via BeamEnsemble) against
  (1) the golden vectors generated from the reference itself (tests/golden/*.npz), and
  (2) the CPU oracle on the same seeded inputs,
plus size-independent properties at BASELINE.json's full sizes.

Tolerances: north_star asks 1e-6 relative fp64; the HIP path and the reference differ only
by rounding (cyclic-reduction solve vs explicit inverse, regrouped polynomial, FMA contraction), so
the fp64 assertions are far tighter and written next to each check.
"""
import os
import sys

import numpy as np
import pytest

from tests.helpers import (assert_blocks, beam_columns, block_errs, force_kwargs, nitinol_columns, oracle_beam, rel_err,
                           rollout_conditioning)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def ensemble(cols, n_beams, kw=None, dtype=None, node_bc=None, corrected_axial=False):
    from continuum_robot.batched import BeamEnsemble
    from continuum_robot.models.force_params import ForceParams

    kw = kw or {}
    fp = ForceParams(fluid_density=kw.get("fluid_density", 0.0), enable_fluid_effects=kw.get("enable_fluid", False),
                     gravity_vector=list(kw.get("gravity", [0.0, -9.81, 0.0])),
                     enable_gravity_effects=kw.get("enable_gravity", False))
    return BeamEnsemble(cols, n_beams, force_params=fp, dtype=dtype or torch.float64, node_bc=node_bc,
                        corrected_axial=corrected_axial)


def test_native_library_is_the_loaded_path():
    from continuum_robot import _native as nat

    assert nat.load().crb_version() == 106
    assert torch.cuda.is_available()


G2_BEAMS = ["test4_lin", "test4_nl", "mixed5", "hetero7"]
G2_SETS = ["none", "fixed0", "pinned0", "fixed0_pinned2", "pinned0_pinnedN"]


@pytest.mark.parametrize("bname", G2_BEAMS)
@pytest.mark.parametrize("sname", G2_SETS)
def test_internal_force_matches_reference(golden, bname, sname):
    z = golden["g2_assembly"]
    key = f"{bname}/{sname}"
    q, k = z[f"{key}/q"], z[f"{key}/k_q"]
    ens = ensemble(beam_columns(z, bname), q.shape[0], node_bc=z[f"{key}/node_bc"].astype(np.uint8))
    got = ens.internal_force(q).cpu().numpy()
    assert_blocks(got, k, ens.free_index, 1e-12)   # per DOF block of k(q) (axial / transverse / moment)


G34_BEAMS = ["test4_lin", "test4_nl", "mixed5", "hetero7", "test4_nl_pinned0", "mixed5_fixed0_pinned2",
             "hetero7_free", "hetero7_pinned0_fixed3"]
FORCE_SETS = ["none", "drag", "grav", "both", "grav_xy", "both_xy"]


@pytest.mark.parametrize("bname", G34_BEAMS)
@pytest.mark.parametrize("fname", FORCE_SETS)
def test_rhs_matches_reference(golden, bname, fname):
    """dynamic_system(t, x, u) on the reference's own outputs: every (state, input) pair is one beam."""
    z = golden["g34_forces_rhs"]
    key = f"{bname}/{fname}"
    X, U, ref = z[f"{key}/x"], z[f"{key}/u"], z[f"{key}/xdot"]
    xs = np.repeat(X, U.shape[0], axis=0)
    us = np.tile(U, (X.shape[0], 1))
    ens = ensemble(beam_columns(z, bname), xs.shape[0], force_kwargs(z, key))
    got = ens.rhs(xs, us).cpu().numpy()
    assert_blocks(got, ref.reshape(got.shape), ens.free_index, 1e-10)   # every block of [v ; a]; bar 1e-6
    # u = None is the zero input (first row of U)
    got0 = ens.rhs(xs).cpu().numpy().reshape(ref.shape)
    assert_blocks(got0[:, 0], ref[:, 0], ens.free_index, 1e-10)


G5 = ["lin10_grav", "lin64_grav", "lin64_grav_x0", "nl64_drag", "nl256_drag", "nl256_drag_a2", "mixed5_both",
      "hetero7_both", "hetero7_p0f3_grav"]


@pytest.mark.parametrize("name", G5)
def test_rk4_rollouts_match_reference(golden, name):
    """Fused multi-step RK4 against RK4 over the reference's own RHS (tests/golden/make_golden.py)."""
    z = golden["g5_rollouts"]
    B = 3  # identical beams: also checks that beams of a workgroup do not interact
    ens = ensemble(beam_columns(z, name), B, force_kwargs(z, name))
    n = ens.n
    ens.set_state(np.tile(z[f"{name}/x0"], (B, 1)))
    dt, amp, dur = float(z[f"{name}/dt"]), float(z[f"{name}/amp"]), float(z[f"{name}/duration"])
    done = 0
    for c in z[f"{name}/checkpoints"]:
        ens.step(int(c) - done, dt, impulse_amp=np.full(B, amp), impulse_duration=dur, impulse_index=-2)
        done = int(c)
        got = ens.unpack_state().cpu().numpy()
        ref = z[f"{name}/x_{c}"]
        # every DOF block within 1e-9 of the REFERENCE's own rollout (bar: 1e-6, north_star); the axial blocks of the
        # long nonlinear chains beyond ~600 steps are ill-conditioned (shipped f1; helpers.assert_blocks) and are
        # held to the oracle's own sensitivity there
        cond = None
        if name.startswith("nl") and done > 600:
            ob = oracle_beam(beam_columns(z, name), **force_kwargs(z, name))
            cond = rollout_conditioning(ob, z[f"{name}/x0"], dt, done, amp, duration=dur)
        errs = assert_blocks(got[0], ref, ens.free_index, 1e-9, what=(name, c), cond=cond, steps=done)
        if name == "nl256_drag" and done == 1000:
            # the metric's horizon against the REFERENCE's own output, numbers at the call site (measured round 2: u 8e-9,
            # du/dt 6e-6 -- the C oracle itself is 3.4e-6 from this golden there; w 5e-15, phi 2e-14, dw/dt 7e-13, dphi/dt 6e-12)
            for k, bound in (("w", 1e-12), ("phi", 1e-12), ("dw_dt", 1e-10), ("dphi_dt", 1e-10), ("u", 1e-6), ("du_dt", 5e-5)):
                assert errs[k] <= bound, (k, errs[k], bound)
        assert abs(got[0, n - 2] - ref[n - 2]) <= 1e-11 * abs(ref[n - 2])  # tip displacement
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])


def test_batched_rollout_matches_oracle_per_beam():
    """64 nonlinear + drag beams with distinct impulse amplitudes, zero initial state."""
    cols = nitinol_columns(64, "nonlinear")
    kw = dict(fluid_density=1000.0, enable_fluid=True)
    B, steps, dt = 64, 200, 2e-5
    amps = 0.1 * (1.0 + np.arange(B) / B)
    ens = ensemble(cols, B, kw)
    ens.step(steps, dt, impulse_amp=amps)
    got = ens.unpack_state().cpu().numpy()
    ob = oracle_beam(cols, **kw)
    ref, _ = ob.rk4_impulse_batch(np.zeros((B, 2 * ob.n)), dt, steps, amps)
    assert_blocks(got, ref, ens.free_index, 1e-9)
    tips = ens.tip_displacement().cpu().numpy()
    assert np.allclose(tips, ref[:, ob.n - 2], rtol=1e-9, atol=0)
    assert np.all(np.diff(tips) > 0)  # larger impulse, larger tip displacement


def test_linear_gravity_ensemble_with_random_initial_states_matches_oracle():
    """BASELINE config 2 in small: per-beam x0 ~ N(0, sigma) on w/phi DOFs."""
    cols = nitinol_columns(64, "linear")
    kw = dict(enable_gravity=True)
    B, steps, dt = 32, 300, 2e-5
    ob = oracle_beam(cols, **kw)
    n = ob.n
    rng = np.random.default_rng(1234)
    x0 = np.concatenate([rng.normal(0, 1e-5, (B, n)), rng.normal(0, 1e-3, (B, n))], axis=1)
    x0[:, 0:n:3] = 0.0
    x0[:, n::3] = 0.0
    amps = 0.1 * (1.0 + np.arange(B) / B)
    ens = ensemble(cols, B, kw)
    ens.set_state(x0)
    ens.step(steps, dt, impulse_amp=amps)
    ref, _ = ob.rk4_impulse_batch(x0, dt, steps, amps)
    assert_blocks(ens.unpack_state().cpu().numpy(), ref, ens.free_index, 1e-9)


def test_chunked_launches_are_bitwise_identical_to_one_launch():
    cols = nitinol_columns(32, "nonlinear")
    kw = dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True)
    amps = np.linspace(0.05, 0.2, 7)
    a = ensemble(cols, 7, kw)
    b = ensemble(cols, 7, kw)
    a.step(600, 2e-5, impulse_amp=amps)
    for chunk in (1, 99, 400, 100):  # crosses the t < 0.01 switch-off at step 500
        b.step(chunk, 2e-5, impulse_amp=amps)
    assert a.time == b.time
    assert torch.equal(a.state, b.state)


def test_held_force_and_pack_roundtrip():
    cols = nitinol_columns(12, "linear")
    ob = oracle_beam(cols)
    n, B = ob.n, 5
    rng = np.random.default_rng(9)
    x0 = rng.normal(0, 1e-4, (B, 2 * n))
    u = rng.normal(0, 0.05, (B, n))
    ens = ensemble(cols, B)
    ens.set_state(x0)
    assert np.array_equal(ens.unpack_state().cpu().numpy(), x0)
    assert float(ens.state[..., 3].abs().max()) == 0.0 and float(ens.state[:, :, 0].abs().max()) == 0.0
    ens.step(150, 2e-5, held_force=u)
    ref = np.array([ob.rk4_held(x0[b], 2e-5, 150, u[b]) for b in range(B)])
    assert_blocks(ens.unpack_state().cpu().numpy(), ref, ens.free_index, 1e-10)


@pytest.mark.parametrize("n_e,kind,kw", [(64, "linear", dict(enable_gravity=True)),
                                          (128, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True)),
                                          (256, "mixed", dict(fluid_density=1000.0, enable_fluid=True))])
def test_held_force_on_lean_size_beams(n_e, kind, kw, monkeypatch):
    """A per-node input force held over the launch (zero-order-hold control) through the lean stepper's HELD
    instantiation, against the oracle and against the general kernel."""
    kinds = ["nonlinear" if i % 3 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    cols = nitinol_columns(n_e, kinds)
    ob = oracle_beam(cols, **kw)
    n, B = ob.n, 3
    rng = np.random.default_rng(n_e)
    x0 = rng.normal(0, 1e-5, (B, 2 * n))
    u = rng.normal(0, 1e-3, (B, n))       # (small: the shipped nonlinear element diverges under large axial loads)
    steps = 60
    ens = ensemble(cols, B, kw)
    ens.set_state(x0)
    ens.step(steps, 2e-5, held_force=u)
    got = ens.unpack_state().cpu().numpy()
    ref = np.array([ob.rk4_held(x0[b], 2e-5, steps, u[b]) for b in range(B)])
    assert np.isfinite(ref).all()
    assert_blocks(got, ref, ens.free_index, 1e-9)
    monkeypatch.setenv("CRB_DISABLE_LEAN", "1")
    gen = ensemble(cols, B, kw)
    gen.set_state(x0)
    gen.step(steps, 2e-5, held_force=u)
    assert_blocks(gen.unpack_state().cpu().numpy(), got, ens.free_index, 1e-10)


@pytest.mark.parametrize("n_e,B", [(1, 1), (2, 70), (10, 13), (63, 5), (65, 3), (130, 2), (300, 2), (600, 2), (1024, 1)])
def test_ragged_sizes_and_partial_groups(n_e, B):
    """Beam sizes around the wavefront/workgroup boundaries, batch sizes that leave a group partly empty."""
    kinds = ["nonlinear" if i % 2 else "linear" for i in range(n_e)]
    cols = nitinol_columns(n_e, kinds)
    kw = dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True, gravity=[1.0, -9.81, 0.0])
    ob = oracle_beam(cols, **kw)
    amps = 0.05 * (1.0 + np.arange(B))
    ens = ensemble(cols, B, kw)
    ens.step(120, 2e-5, impulse_amp=amps)
    ref, _ = ob.rk4_impulse_batch(np.zeros((B, 2 * ob.n)), 2e-5, 120, amps)
    assert_blocks(ens.unpack_state().cpu().numpy(), ref, ens.free_index, 1e-9)


@pytest.mark.parametrize("n_e,B", [(33, 3), (64, 2), (100, 3), (128, 2), (200, 2), (300, 2), (512, 2)])
def test_lean_stepper_sizes(n_e, B):
    """The lean fused stepper (no gravity) for every waves-per-beam count it is built for, incl. beams
    that leave padding threads and the > 64 KiB LDS case."""
    kinds = ["nonlinear" if i % 3 else "linear" for i in range(n_e)]
    cols = nitinol_columns(n_e, kinds)
    kw = dict(fluid_density=1000.0, enable_fluid=True)
    ob = oracle_beam(cols, **kw)
    amps = 0.05 * (1.0 + np.arange(B))
    ens = ensemble(cols, B, kw)
    ens.step(120, 2e-5, impulse_amp=amps)
    ref, _ = ob.rk4_impulse_batch(np.zeros((B, 2 * ob.n)), 2e-5, 120, amps)
    assert_blocks(ens.unpack_state().cpu().numpy(), ref, ens.free_index, 1e-9)


@pytest.mark.parametrize("kind", ["linear", "nonlinear"])
@pytest.mark.parametrize("n_e,root", [(64, "FIXED"), (128, "FIXED"), (256, "FIXED"), (512, "FIXED"),
                                      (100, "PINNED"), (200, "NONE")])
def test_lean_stepper_single_kind_topologies(kind, n_e, root):
    """Beams of ONE element kind take the lean stepper's straight-line force code (element mode
    EM_LINEAR / EM_NONLINEAR, crb_lean.h) for every waves-per-beam count.  With a root that keeps its
    slot (PINNED or unconstrained) the first thread has no element to its left and runs the same formula
    on zero coefficients."""
    cols = nitinol_columns(n_e, kind, [root] + ["NONE"] * (n_e - 1))
    kw = dict(fluid_density=1000.0, enable_fluid=True)
    B = 3
    amps = 0.05 * (1.0 + np.arange(B))
    ob = oracle_beam(cols, **kw)
    ens = ensemble(cols, B, kw)
    steps = 100
    ens.step(steps, 2e-5, impulse_amp=amps)
    ref, _ = ob.rk4_impulse_batch(np.zeros((B, 2 * ob.n)), 2e-5, steps, amps)
    assert_blocks(ens.unpack_state().cpu().numpy(), ref, ens.free_index, 1e-9)


@pytest.mark.parametrize("n_e,kind,kw", [(10, "linear", dict(enable_gravity=True)),
                                          (16, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True)),
                                          (5, "mixed", dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True))])
@pytest.mark.parametrize("poison", ["nan", "inf", "huge"])
def test_packed_beams_are_isolated_from_a_diverged_wave_mate(n_e, kind, kw, poison):
    """Beams of fewer than 64 slots share a wave (PACK form of the lean stepper): one beam seeded non-finite (or so
    large that it overflows within a few steps) must leave every other beam BITWISE equal to a run without it --
    the reference's beams are independent; a lane shift across a beam boundary is cancelled by a select, not by a
    zero factor (0 * NaN = NaN)."""
    kinds = ["nonlinear" if i % 2 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    cols = nitinol_columns(n_e, kinds)
    B = 23                                  # several waves' worth, the last one partly filled
    bad = [0, 7, B - 1]                     # first lane group, a middle one, the last beam
    amps = 0.05 * (1.0 + np.arange(B) / B)
    clean = ensemble(cols, B, kw)
    assert clean.plan.beams_per_group > 1   # really the packed layout
    rng = np.random.default_rng(5)
    x0 = rng.normal(0.0, 1e-6, (B, 2 * clean.n))
    clean.set_state(x0)
    clean.step(60, 2e-5, impulse_amp=amps)
    want = clean.unpack_state()
    assert bool(torch.isfinite(want).all())
    x0p = x0.copy()
    x0p[bad] = {"nan": np.nan, "inf": np.inf, "huge": 1e200}[poison]
    dirty = ensemble(cols, B, kw)
    dirty.set_state(x0p)
    dirty.step(60, 2e-5, impulse_amp=amps)
    got = dirty.unpack_state()
    good = [b for b in range(B) if b not in bad]
    assert torch.equal(got[good], want[good])
    if poison != "huge":     # (a huge but finite linear beam stays finite; what matters is that its wave-mates did not notice)
        assert not bool(torch.isfinite(got[bad]).all())


def test_corrected_axial_option_matches_oracle():
    cols = nitinol_columns(16, "nonlinear")
    ob = oracle_beam(cols, corrected_axial=True)
    ens = ensemble(cols, 2, corrected_axial=True)
    ens.step(200, 2e-5, impulse_amp=np.array([0.1, 0.3]))
    ref, _ = ob.rk4_impulse_batch(np.zeros((2, 2 * ob.n)), 2e-5, 200, np.array([0.1, 0.3]))
    assert_blocks(ens.unpack_state().cpu().numpy(), ref, ens.free_index, 1e-9)


def test_corrected_axial_option_on_a_lean_size_beam():
    """CRB_CORRECTED_AXIAL on a beam long enough for the lean stepper (all-nonlinear topology: the corrected
    force takes the per-lane element path, not the EM_NONLINEAR straight-line one)."""
    cols = nitinol_columns(128, "nonlinear")
    kw = dict(fluid_density=1000.0, enable_fluid=True)
    ob = oracle_beam(cols, corrected_axial=True, **kw)
    ens = ensemble(cols, 2, kw, corrected_axial=True)
    amps = np.array([0.1, 0.3])
    ens.step(200, 2e-5, impulse_amp=amps)
    ref, _ = ob.rk4_impulse_batch(np.zeros((2, 2 * ob.n)), 2e-5, 200, amps)
    assert_blocks(ens.unpack_state().cpu().numpy(), ref, ens.free_index, 1e-9)
    plain = ensemble(cols, 2, kw)
    plain.step(200, 2e-5, impulse_amp=amps)
    assert rel_err(plain.unpack_state().cpu().numpy(), ref) > 1e-6   # the two element variants do differ


def test_fp32_plan_tracks_fp64_within_measured_drift():
    """BASELINE config 4's dtype on a small ensemble, against the ORACLE (fp64), per DOF block.  fp32 is NOT held
    to 1e-6: cond(M) ~ 1e4 costs ~4 digits per solve; tolerances = measured drift x ~3 over the
    200-step horizon (FP32_TOL below; measured values next to them)."""
    cols = nitinol_columns(256, "nonlinear")
    kw = dict(fluid_density=1000.0, enable_fluid=True)
    amps = np.array([0.1, 0.15, 0.2])
    e32 = ensemble(cols, 3, kw, dtype=torch.float32)
    e32.step(200, 2e-5, impulse_amp=amps)
    got = e32.unpack_state().double().cpu().numpy()
    assert np.all(np.isfinite(got))
    ob = oracle_beam(cols, **kw)
    ref, _ = ob.rk4_impulse_batch(np.zeros((3, 2 * ob.n)), 2e-5, 200, amps)
    _assert_fp32(block_errs(got, ref, e32.free_index))


# fp32 plans vs the fp64 oracle, nonlinear + drag, 200 steps, per DOF block: the bound asserted = ~3x the worst
# over the whole 4096 x 256 ensemble measured on MI355X (tests/measure_fp32_blocks.py: u 1.9e-6, w 8.4e-7, phi 1.0e-6,
# du/dt 3.5e-5, dw/dt 1.7e-6, dphi/dt 9.2e-6).  The transverse blocks -- what the examples read -- keep 6 digits;
# the axial rate is a second-order quantity (|du/dt| ~ 1e-7 against |dw/dt| ~ 1e-2) driven by a difference of O(1)
# terms and keeps 4.
FP32_TOL = {"u": 6e-6, "w": 3e-6, "phi": 3e-6, "du_dt": 1e-4, "dw_dt": 5e-6, "dphi_dt": 3e-5}


def _assert_fp32(errs):
    for k, e in errs.items():
        assert e <= FP32_TOL[k], (k, e, errs)


def test_full_size_config3_properties():
    """4096 beams x 256 nonlinear elements + drag (BASELINE config 3) for the metric's 1000 steps, in launches of
    100 as bench.py issues them:
    (a) beams with equal input are bitwise equal wherever they sit in the grid,
    (b) tip displacement is monotone in the impulse amplitude,
    (c) the first / middle / last beam match the oracle in every DOF block at 200 and at 1000 steps."""
    cols = nitinol_columns(256, "nonlinear")
    kw = dict(fluid_density=1000.0, enable_fluid=True)
    B, dt = 4096, 2e-5
    amps = 0.1 * (1.0 + np.arange(B) / B)
    amps[1::512] = amps[0]  # replicas of beam 0 scattered over the grid
    ens = ensemble(cols, B, kw)
    ob = oracle_beam(cols, **kw)
    done = 0
    for horizon in (200, 1000):
        while done < horizon:
            ens.step(100, dt, impulse_amp=amps)
            done += 100
        x = ens.unpack_state()
        assert bool(torch.isfinite(x).all())
        for b in range(1, B, 512):
            assert torch.equal(x[b], x[0])
        tips = ens.tip_displacement().cpu().numpy()
        order = np.argsort(amps, kind="stable")
        assert np.all(np.diff(tips[order]) >= 0)
        for b in (0, B // 2, B - 1):
            ref = ob.rk4_impulse(np.zeros(2 * ob.n), dt, horizon, amps[b])
            # 200 steps: every block <= 1e-10 (bar 1e-6).  1000 steps: w, phi and their rates <= 1e-10; the axial
            # blocks are ill-conditioned there (shipped f1: helpers.assert_blocks) and are held to the oracle's own
            # sensitivity to a 64-ulp change of the impulse amplitude
            cond = rollout_conditioning(ob, np.zeros(2 * ob.n), dt, horizon, amps[b]) if horizon > 600 else None
            assert_blocks(x[b].cpu().numpy(), ref, ens.free_index, 1e-10, what=(horizon, b), cond=cond, steps=horizon)
            assert abs(tips[b] - ref[ob.n - 2]) <= 1e-11 * abs(ref[ob.n - 2])


def test_full_size_config4_fp32_against_oracle():
    """BASELINE config 4 at full size on one GPU (4096 x 256 nonlinear + drag, fp32, 200 steps -- the 8 x 512
    shards of the 8-GPU layout are contiguous slices of this ensemble, stepped by the same kernel): every 16th beam
    plus the two ends against the fp64 oracle, per DOF block, tolerance = measured single-precision drift x ~3."""
    cols = nitinol_columns(256, "nonlinear")
    kw = dict(fluid_density=1000.0, enable_fluid=True)
    B, steps, dt = 4096, 200, 2e-5
    amps = 0.1 * (1.0 + np.arange(B) / B)
    ens = ensemble(cols, B, kw, dtype=torch.float32)
    ens.step(steps, dt, impulse_amp=amps)
    got = ens.unpack_state().double().cpu().numpy()
    assert np.all(np.isfinite(got))
    sel = np.unique(np.concatenate([np.arange(0, B, 16), [B - 1]]))
    ob = oracle_beam(cols, **kw)
    ref, _ = ob.rk4_impulse_batch(np.zeros((sel.size, 2 * ob.n)), dt, steps, amps[sel])
    _assert_fp32(block_errs(got[sel], ref, ens.free_index))
    # the 8-GPU layout: shard r of 8 is beams [512 r, 512 (r + 1)); a 512-beam plan stepping shard 3's amplitudes
    # must reproduce those rows bit for bit (no dependence on the grid position)
    shard = ensemble(cols, 512, kw, dtype=torch.float32)
    shard.step(steps, dt, impulse_amp=amps[3 * 512:4 * 512])
    assert torch.equal(shard.unpack_state(), ens.unpack_state()[3 * 512:4 * 512])


_CARE_GAIN = {}


def care_gain(ens):
    """LQR gain of examples/lqr_control.py:46-84 (Q = diag(100 I, 10 I), R = I) through the scipy-CARE shim,
    solved once per session (768-state CARE: ~40 s)."""
    from continuum_robot.control import LinearQuadraticRegulator

    key = (ens.n_elem, ens.n)
    if key not in _CARE_GAIN:
        K, M = ens.plan.stiffness(), ens.plan.mass()
        n = K.shape[0]
        Q = np.eye(2 * n)
        Q[:n, :n] *= 100
        Q[n:, n:] *= 10
        _CARE_GAIN[key] = LinearQuadraticRegulator(K, M, Q, np.eye(n)).compute_gain_matrix()
    return _CARE_GAIN[key]


def test_full_size_config5_lqr_ensemble_against_oracle():
    """BASELINE config 5's per-GPU shard: 2048 beams x 128 linear elements + gravity, the REAL LQR gain (scipy-CARE
    shim, Q/R of lqr_control.py:61-66) applied in every RK4 stage, per-beam impulse 10 (1 + b/B) N, random x0,
    dt = 5e-6, 400 steps: first / middle / last beam against the oracle's closed loop, per DOF block."""
    cols = nitinol_columns(128, "linear")
    kw = dict(enable_gravity=True)
    B, steps, dt = 2048, 400, 5e-6
    ens = ensemble(cols, B, kw)
    n = ens.n
    gain = care_gain(ens)
    rng = np.random.default_rng(1234)
    x0 = np.concatenate([rng.normal(0, 1e-5, (B, n)), rng.normal(0, 1e-3, (B, n))], axis=1)
    x0[:, 0:n:3] = 0.0
    x0[:, n::3] = 0.0
    amps = 10.0 * (1.0 + np.arange(B) / B)
    ens.set_state(x0)
    ens.step_feedback(steps, dt, gain, impulse_amp=amps)
    got = ens.unpack_state().cpu().numpy()
    assert np.isfinite(got).all()
    ob = oracle_beam(cols, **kw)
    for b in (0, B // 2, B - 1):
        want = ob.rk4_feedback(x0[b], dt, steps, gain, amp=amps[b])
        assert_blocks(got[b], want, ens.free_index, 1e-9, what=b)


def test_full_size_config2_against_oracle_sample():
    """1024 beams x 64 linear elements + gravity (BASELINE config 2), random x0, 200 steps."""
    cols = nitinol_columns(64, "linear")
    kw = dict(enable_gravity=True)
    B, steps, dt = 1024, 200, 2e-5
    ob = oracle_beam(cols, **kw)
    n = ob.n
    rng = np.random.default_rng(1234)
    x0 = np.concatenate([rng.normal(0, 1e-5, (B, n)), rng.normal(0, 1e-3, (B, n))], axis=1)
    x0[:, 0:n:3] = 0.0
    x0[:, n::3] = 0.0
    amps = 0.1 * (1.0 + np.arange(B) / B)
    ens = ensemble(cols, B, kw)
    ens.set_state(x0)
    ens.step(steps, dt, impulse_amp=amps)
    got = ens.unpack_state().cpu().numpy()
    sel = np.arange(0, B, 37)
    ref, _ = ob.rk4_impulse_batch(x0[sel], dt, steps, amps[sel])
    assert_blocks(got[sel], ref, ens.free_index, 1e-9)


def test_error_paths():
    from continuum_robot import _native as nat

    cols = nitinol_columns(4, "linear")
    ens = ensemble(cols, 2)
    with pytest.raises(ValueError):
        ens.set_state(np.zeros((2, 5)))
    with pytest.raises(IndexError):
        ens.step(1, 2e-5, impulse_amp=np.ones(2), impulse_index=ens.n)
    with pytest.raises(nat.NativeError, match="dt must be positive"):
        ens.step(1, 0.0)
    with pytest.raises(ValueError, match="CSV must contain columns"):
        ensemble({k: v for k, v in cols.items() if k != "density"}, 1)


@pytest.mark.parametrize("case", ["test4_lin/fixed0", "mixed5/fixed0_pinned2", "hetero7/pinned0_pinnedN", "hetero7/none",
                                  "nitinol256", "nitinol300_f32"])
def test_device_assembly_matches_host_assembly(golden, case):
    """crb_assemble_kernel (device plans) against the plain-C++ assembly of host-only plans: element
    packs, masks, drag/segment-mass constants, node-block mass and every cyclic-reduction table."""
    from continuum_robot import _native as nat

    kw = dict(fluid_density=870.0, enable_fluid=True, enable_gravity=True, gravity=[3.0, -9.81, 0.0])
    dtype = "f64"
    if case.startswith("nitinol"):
        n = int(case[7:10])
        cols, node_bc = nitinol_columns(n, ["nonlinear" if i % 3 else "linear" for i in range(n)]), None
        dtype = "f32" if case.endswith("f32") else "f64"
    else:
        z = golden["g2_assembly"]
        bname = case.split("/")[0]
        cols, node_bc = beam_columns(z, bname), z[f"{case}/node_bc"].astype(np.uint8)
    dev = nat.Plan(cols, node_bc=node_bc, device=0, dtype=dtype, **kw)
    host = nat.Plan(cols, node_bc=node_bc, device=-1, dtype=dtype, **kw)
    for f in ("n_free", "node_offset", "n_slots", "threads", "beams_per_group", "pcr_levels", "pcr_levels_full"):
        assert getattr(dev, f) == getattr(host, f), f
    assert rel_err(dev.mass(), host.mass()) < 1e-14
    (lv_d, fin_d, nrm_d), (lv_h, fin_h, nrm_h) = dev.pcr_tables(), host.pcr_tables()
    used = host.pcr_levels
    if used:
        assert rel_err(lv_d[:used], lv_h[:used]) < 1e-12
    assert rel_err(fin_d, fin_h) < 1e-12
    assert np.allclose(nrm_d, nrm_h, rtol=1e-10, atol=0)
    td, th = dev.slot_tables(), host.slot_tables()
    tol = 1e-6 if dtype == "f32" else 1e-15
    for k in ("drag", "half_mass"):
        assert np.allclose(td[k], th[k], rtol=tol, atol=0), k
    assert np.array_equal(td["mask"], th["mask"]) and np.array_equal(td["grav"], th["grav"])
    assert np.array_equal(td["elem_kind"], th["elem_kind"])


@pytest.mark.parametrize("name", ["lqr6", "lqr24"])
def test_lqr_feedback_rollout_matches_reference(golden, name):
    """Stage-split RK4 with u = K(r - x) per stage (BASELINE config 5's loop) against the reference
    (tests/golden/make_golden.py:g6_lqr_loop) and, with per-beam amplitudes and references, the oracle."""
    z = golden["g6_lqr_loop"]
    cols, kw = beam_columns(z, name), force_kwargs(z, name)
    gain, dt, steps, amp = z[f"{name}/gain"], float(z[f"{name}/dt"]), int(z[f"{name}/steps"]), float(z[f"{name}/amp"])
    B = 4
    ens = ensemble(cols, B, kw)
    ens.step_feedback(steps, dt, gain, impulse_amp=np.full(B, amp))
    got = ens.unpack_state().cpu().numpy()
    assert_blocks(got[0], z[f"{name}/x_final"], ens.free_index, 1e-8, what=name)
    assert np.array_equal(got[0], got[3])
    # distinct amplitudes + a nonzero reference, shorter horizon, against the oracle
    ob = oracle_beam(cols, **kw)
    n = ob.n
    amps = amp * (1.0 + np.arange(B) / B)
    ref = np.zeros((B, 2 * n))
    ref[:, n - 2] = 1e-3 * (1 + np.arange(B))
    ens = ensemble(cols, B, kw)
    ens.step_feedback(300, dt, gain, reference=ref, impulse_amp=amps)
    got = ens.unpack_state().cpu().numpy()
    for b in range(B):
        want = ob.rk4_feedback(np.zeros(2 * n), dt, 300, gain, reference=ref[b], amp=amps[b])
        assert_blocks(got[b], want, ens.free_index, 1e-9, what=b)


@pytest.mark.parametrize("n_e,kind,kw,hetero", [
    (64, "linear", dict(enable_gravity=True), False),                               # one wave per beam, gravity
    (128, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True), False),        # two waves (config 5's shape)
    (128, "linear", dict(enable_gravity=True), True),                                # per-beam tables: reload per beam
    (200, "mixed", dict(fluid_density=1000.0, enable_fluid=True), False),            # four waves, padding threads
    (300, "linear", dict(), False),                                                  # eight waves
])
def test_lean_stage_kernel_feedback_rollout_matches_oracle(n_e, kind, kw, hetero, monkeypatch):
    """The stage-split stepper (crb_feedback_force + crb_rk4_stage) on beams long enough for the lean stage
    kernel (crb_stage_lean_kernel), with fewer workgroups than beams so that each walks over several beams
    (uneven split), a dense random gain, per-beam references and amplitudes, from a random state."""
    monkeypatch.setenv("CRB_STAGE_GROUPS", "2")
    rng = np.random.default_rng(n_e)
    kinds = ["nonlinear" if i % 3 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    base = nitinol_columns(n_e, kinds)
    B = 5
    per_beam = [_scaled(base, rng) for _ in range(B)] if hetero else [base] * B
    ens = ensemble(per_beam if hetero else base, B, kw)
    n = ens.n
    gain = rng.normal(0.0, 2e-2, (n, 2 * n))   # (small: the rotational DOFs have tiny inertia, RK4 must stay stable)
    ref = rng.normal(0.0, 1e-4, (B, 2 * n))
    x0 = rng.normal(0.0, 1e-4, (B, 2 * n))
    amps = 0.05 * (1.0 + np.arange(B))
    ens.set_state(x0)
    steps, dt = 40, 2e-5
    ens.step_feedback(steps, dt, gain, reference=ref, impulse_amp=amps)
    got = ens.unpack_state().cpu().numpy()
    for b in range(B):
        want = oracle_beam(per_beam[b], **kw).rk4_feedback(x0[b], dt, steps, gain, reference=ref[b], amp=amps[b])
        assert_blocks(got[b], want, ens.free_index, 1e-9, what=b)
    # the generic stage kernel gives the same answer
    monkeypatch.setenv("CRB_DISABLE_LEAN_STAGE", "1")
    ens2 = ensemble(per_beam if hetero else base, B, kw)
    ens2.set_state(x0)
    ens2.step_feedback(steps, dt, gain, reference=ref, impulse_amp=amps)
    assert_blocks(ens2.unpack_state().cpu().numpy(), got, ens.free_index, 1e-10)


@pytest.mark.parametrize("lean", [True, False])
def test_strided_recording_matches_chunked_stepping(lean):
    """f-4: on-device t_eval-style recording of the tip displacement / velocity."""
    # (the shipped nonlinear element under a distributed load diverges within ~250 steps -- SURVEY App. B-1 --
    #  so the gravity variant, which takes the generic kernel, uses linear elements)
    cols = nitinol_columns(64, "nonlinear" if lean else "linear")
    kw = dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=not lean)
    amps = np.array([0.1, 0.17, 0.2])
    a, b = ensemble(cols, 3, kw), ensemble(cols, 3, kw)
    t, rec = a.step(600, 2e-5, impulse_amp=amps, record=(64, "w"), record_every=50)
    assert rec.shape == (3, 12)
    want = []
    for _ in range(12):
        b.step(50, 2e-5, impulse_amp=amps)
        want.append(b.tip_displacement().clone())
    assert torch.equal(rec, torch.stack(want, dim=1)) and torch.equal(a.state, b.state) and t == b.time
    _, recv = a.step(30, 2e-5, impulse_amp=amps, record=(10, "dphi_dt"), record_every=7)
    assert recv.shape == (3, 4)
    assert torch.equal(recv[:, -1] != 0, torch.ones(3, dtype=torch.bool, device=recv.device))


def _scaled(cols, rng):
    """Per-beam E, rho, r scaled by U(0.9, 1.1) (SURVEY §8(d) heterogeneous variant)."""
    sE, sr, srho = rng.uniform(0.9, 1.1, 3)
    out = dict(cols)
    out["elastic_modulus"] = cols["elastic_modulus"] * sE
    out["density"] = cols["density"] * srho
    out["cross_area"] = cols["cross_area"] * sr**2
    out["moment_inertia"] = cols["moment_inertia"] * sr**4
    out["wetted_area"] = cols["wetted_area"] * sr
    return out


@pytest.mark.parametrize("n_e,kind,kw", [
    (64, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True)),           # lean stepper, one wave
    (256, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True)),          # lean stepper, four waves
    (40, "linear", dict(enable_gravity=True, fluid_density=500.0, enable_fluid=True)),  # generic stepper
    (10, "linear", dict(enable_gravity=True)),                                  # several beams per wave
])
def test_heterogeneous_ensemble_matches_per_beam_oracle(n_e, kind, kw):
    """f-3: per-beam coefficients (crb_plan_create_ensemble, batched device assembly)."""
    rng = np.random.default_rng(4321)
    B = 9
    base = nitinol_columns(n_e, kind)
    per_beam = [_scaled(base, rng) for _ in range(B)]
    if kind == "linear":  # per-beam element TYPES too
        per_beam[3]["type"] = np.array(["nonlinear" if i % 2 else "linear" for i in range(n_e)])
    amps = 0.1 * (1.0 + np.arange(B) / B)
    ens = ensemble(per_beam, B, kw)
    steps = 150
    ens.step(steps, 2e-5, impulse_amp=amps)
    got = ens.unpack_state().cpu().numpy()
    for b in range(B):
        ob = oracle_beam(per_beam[b], **kw)
        want = ob.rk4_impulse(np.zeros(2 * ob.n), 2e-5, steps, amps[b])
        assert_blocks(got[b], want, ens.free_index, 1e-9, what=b)
    # rhs through the per-beam tables as well
    x = rng.normal(0, 1e-3, (B, 2 * ens.n))
    xd = ens.rhs(x).cpu().numpy()
    for b in (0, B - 1):
        assert_blocks(xd[b], oracle_beam(per_beam[b], **kw).rhs(x[b]), ens.free_index, 1e-10, what=b)
    longer = ensemble([base, nitinol_columns(n_e + 1, kind)], 2, kw)     # different lengths are one ensemble now (f-3)
    assert longer.mixed_topology and list(longer.n_elem_per_beam) == [n_e, n_e + 1]


@pytest.mark.parametrize("tile", ["32", "48"])
@pytest.mark.parametrize("n_e,B,bcs", [(4, 3, None), (24, 70, None), (128, 130, None), (50, 200, None),
                                        (7, 5, ["PINNED", "NONE", "NONE", "FIXED", "NONE", "NONE", "NONE"])])
def test_fused_mfma_feedback_force_matches_matmul(n_e, B, bcs, tile, monkeypatch):
    """crb_feedback_force (gather + fp64 MFMA GEMM + scatter) against (r - x) @ K^T in torch, with
    asymmetric random data (an MFMA lane-map mistake cannot hide) and ragged tile edges (rows, columns and
    the K tail), for both tile shapes the dispatcher chooses between (32 x 32 and 64 x 48 outputs)."""
    import ctypes as C

    monkeypatch.setenv("CRB_FEEDBACK_TILE", tile)

    from continuum_robot import _native as nat

    cols = nitinol_columns(n_e, "linear", bcs)
    ens = ensemble(cols, B, dict(enable_gravity=True))
    n = ens.n
    g = torch.Generator(device="cpu").manual_seed(n_e)
    K = torch.randn((n, 2 * n), generator=g, dtype=torch.float64).to(ens.device)
    X = torch.randn((B, 2 * n), generator=g, dtype=torch.float64).to(ens.device)
    R = torch.randn((B, 2 * n), generator=g, dtype=torch.float64).to(ens.device)
    xs = ens.pack_state(X)
    for ref in (None, R):
        u = torch.zeros((B, ens.n_node, 4), dtype=torch.float64, device=ens.device)
        nat.check(nat.load().crb_feedback_force(ens.plan.h, C.c_void_p(xs.data_ptr()), C.c_void_p(K.data_ptr()),
                                                C.c_void_p(ref.data_ptr()) if ref is not None else None,
                                                C.c_void_p(u.data_ptr()), None))
        torch.cuda.synchronize()
        want = ((ref if ref is not None else 0) - X) @ K.t()
        got = ens.unpack_vec(u)
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < 1e-13
        assert float(u[..., 3].abs().max()) == 0.0  # pad lane untouched


@pytest.mark.parametrize("name", ["lin40_grav", "nl64_drag"])
def test_adaptive_rk45_matches_scipy_on_the_reference_rhs(golden, name):
    """f-2: crb_solve_rk45 (Dormand-Prince 5(4), per-beam step control, one launch) against
    scipy.integrate.solve_ivp(method="RK45") run over the REFERENCE RHS (tests/golden/make_golden.py:g7_rk45):
    same accepted-step count, same number of RHS evaluations, same terminal state."""
    z = golden["g7_rk45"]
    B = 3
    ens = ensemble(beam_columns(z, name), B, force_kwargs(z, name))
    st = ens.solve_rk45(float(z[f"{name}/t_end"]), rtol=float(z[f"{name}/rtol"]), atol=float(z[f"{name}/atol"]),
                        impulse_amp=np.full(B, float(z[f"{name}/amp"])), impulse_duration=float(z[f"{name}/duration"]))
    assert np.all(st["status"] == 0)
    assert np.all(st["accepted"] == int(z[f"{name}/accepted"])), (st["accepted"], int(z[f"{name}/accepted"]))
    assert np.all(st["nfev"] == int(z[f"{name}/nfev"]))
    got = ens.unpack_state().cpu().numpy()
    assert_blocks(got[0], z[f"{name}/x_final"], ens.free_index, 1e-8, what=name)
    assert np.array_equal(got[0], got[2])
    # t_eval: dense output of the tip displacement / velocity on the grid the golden run was sampled on
    n_eval = z[f"{name}/tip_w_eval"].size
    for param, key in (("w", "tip_w_eval"), ("dw_dt", "tip_dw_eval")):
        ens = ensemble(beam_columns(z, name), B, force_kwargs(z, name))
        st = ens.solve_rk45(float(z[f"{name}/t_end"]), rtol=float(z[f"{name}/rtol"]), atol=float(z[f"{name}/atol"]),
                            impulse_amp=np.full(B, float(z[f"{name}/amp"])), impulse_duration=float(z[f"{name}/duration"]),
                            record=(ens.n_elem, param), t_eval=(0.0, float(z[f"{name}/eval_dt"]), n_eval))
        y = st["y"].cpu().numpy()
        assert y.shape == (B, n_eval) and rel_err(y[1], z[f"{name}/{key}"]) < 1e-8


def test_adaptive_rk45_per_beam_step_control_matches_scipy_over_oracle():
    """Different impulse amplitudes -> different step sequences per beam; each against scipy over the oracle RHS."""
    from scipy.integrate import solve_ivp

    cols = nitinol_columns(48, "nonlinear")
    kw = dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True)
    ob = oracle_beam(cols, **kw)
    n, B = ob.n, 4
    amps = np.array([0.02, 0.1, 0.5, 2.0])
    dur, t_end, rtol, atol = 5e-4, 1.2e-3, 1e-5, 1e-8
    ens = ensemble(cols, B, kw)
    st = ens.solve_rk45(t_end, rtol=rtol, atol=atol, impulse_amp=amps, impulse_duration=dur)
    got = ens.unpack_state().cpu().numpy()
    assert len(set(st["accepted"].tolist())) > 1  # the beams really took different numbers of steps
    for b in range(B):
        def fun(t, x, b=b):
            u = np.zeros(n)
            if t < dur:
                u[-2] = amps[b]
            return ob.rhs(x, u)

        sol = solve_ivp(fun, (0.0, t_end), np.zeros(2 * n), method="RK45", rtol=rtol, atol=atol)
        assert st["accepted"][b] == len(sol.t) - 1 and st["nfev"][b] == sol.nfev, (b, st["accepted"][b], len(sol.t) - 1)
        assert_blocks(got[b], sol.y[:, -1], ens.free_index, 1e-8, what=b)


def test_fp32_plans_run_every_entry_point():
    """fp32 (BASELINE config 4's dtype) through every API: values track fp64 at single precision."""
    cols = nitinol_columns(64, "linear")
    kw = dict(enable_gravity=True, fluid_density=1000.0, enable_fluid=True)
    B = 3
    rng = np.random.default_rng(2)
    e64, e32 = ensemble(cols, B, kw), ensemble(cols, B, kw, dtype=torch.float32)
    n = e64.n
    x = rng.normal(0, 1e-3, (B, 2 * n))
    u = rng.normal(0, 0.1, (B, n))
    for a, b in ((e64.rhs(x, u), e32.rhs(x, u)), (e64.internal_force(x[:, :n]), e32.internal_force(x[:, :n]))):
        assert rel_err(b.double().cpu().numpy(), a.cpu().numpy()) < 2e-3  # cond(M) ~ 1e4 in single precision
    amps = np.array([0.1, 0.2, 0.3])
    for ens in (e64, e32):
        ens.zero_state()
        ens.step(100, 2e-5, impulse_amp=amps, held_force=u * 0.01)
    assert rel_err(e32.tip_displacement().double().cpu().numpy(), e64.tip_displacement().cpu().numpy()) < 5e-3
    gain = rng.normal(0, 1e-2, (n, 2 * n))
    for ens in (e64, e32):
        ens.zero_state()
        ens.step_feedback(40, 5e-6, gain, impulse_amp=amps)  # fp64: fused MFMA kernel; fp32: torch.matmul path
    assert rel_err(e32.tip_displacement().double().cpu().numpy(), e64.tip_displacement().cpu().numpy()) < 5e-3
    st = {}
    for name, ens in (("f64", e64), ("f32", e32)):
        ens.zero_state()
        st[name] = ens.solve_rk45(4e-4, rtol=1e-3, atol=1e-6, impulse_amp=amps, impulse_duration=2e-4)
        assert np.all(st[name]["status"] == 0)
    assert rel_err(e32.tip_displacement().double().cpu().numpy(), e64.tip_displacement().cpu().numpy()) < 2e-2


def test_new_entry_points_reject_bad_arguments():
    from continuum_robot import _native as nat

    ens = ensemble(nitinol_columns(40, "linear"), 2)
    with pytest.raises(nat.NativeError, match="t_end must be greater"):
        ens.solve_rk45(0.0)
    with pytest.raises(nat.NativeError, match="tolerances"):
        ens.solve_rk45(1e-3, rtol=0.0)
    with pytest.raises(ValueError):
        ens.step_feedback(1, 5e-6, np.zeros((3, 3)))
    with pytest.raises(nat.NativeError, match="bad record"):
        ens.step(10, 2e-5, record=(999, "w"))
    with pytest.raises(ValueError, match="n_beams entries"):
        ensemble([nitinol_columns(8, "linear")], 2)
    with pytest.raises(nat.NativeError, match="h must be positive"):
        ens.step_implicit(1, 0.0)
    with pytest.raises(nat.NativeError, match="n_iter"):
        ens.step_implicit(1, 1e-3, n_iter=0)
    with pytest.raises(nat.NativeError, match="needs an fp64 plan"):
        ensemble(nitinol_columns(40, "linear"), 2, dtype=torch.float32).step_implicit(1, 1e-3)
    with pytest.raises(nat.NativeError, match="more than 256"):
        ensemble(nitinol_columns(300, "linear"), 1).step_implicit(1, 1e-3)


def test_native_feedback_rollout_entry_point_contract():
    """crb_step_rk4_feedback: argument checks, the accumulated clock it returns, and equality with the same
    loop issued stage by stage through crb_feedback_force + crb_rk4_stage."""
    import ctypes as C

    from continuum_robot import _native as nat

    lib = nat.load()
    cols = nitinol_columns(70, "linear")
    B, dt, steps = 3, 2e-5, 12
    rng = np.random.default_rng(3)
    ens = ensemble(cols, B, dict(enable_gravity=True))
    n = ens.n
    gain = torch.tensor(rng.normal(0.0, 2e-2, (n, 2 * n)), device=ens.device)
    x0 = rng.normal(0.0, 1e-4, (B, 2 * n))
    ens.set_state(x0)
    work = torch.empty((int(lib.crb_feedback_work_bytes(ens.plan.h)),), dtype=torch.uint8, device=ens.device)
    # (three state-sized buffers, a force-sized one and the device clock for the stage-split loop; plans the persistent stepper
    #  covers -- this one: 70 thread-carried nodes -- ask for its buffers when those are larger)
    assert work.numel() >= 3 * ens.state.numel() * 8 + ens.state.numel() * 4 + 256
    t_end = C.c_double(-1.0)
    vp = lambda t: C.c_void_p(t.data_ptr())
    # bad arguments
    assert lib.crb_step_rk4_feedback(ens.plan.h, vp(ens.state), 0.0, dt, steps, None, None, None, vp(work), C.byref(t_end), None) != 0
    assert b"null pointer" in lib.crb_last_error()
    assert lib.crb_step_rk4_feedback(ens.plan.h, vp(ens.state), 0.0, -1.0, steps, vp(gain), None, None, vp(work), C.byref(t_end), None) != 0
    # the rollout, against the stage-by-stage loop
    nat.check(lib.crb_step_rk4_feedback(ens.plan.h, vp(ens.state), 0.25, dt, steps, vp(gain), None, None, vp(work),
                                        C.byref(t_end), None))
    t = 0.25
    for _ in range(steps):
        t = t + dt
    assert t_end.value == t
    ref = ensemble(cols, B, dict(enable_gravity=True))
    ref.set_state(x0)
    acc, bufs = torch.empty_like(ref.state), (torch.empty_like(ref.state), torch.empty_like(ref.state))
    u = torch.zeros((B, ref.n_node, 4), dtype=torch.float64, device=ref.device)
    t = 0.25
    for _ in range(steps):
        cur = ref.state
        for s, ts in enumerate((t, t + 0.5 * dt, t + 0.5 * dt, t + dt)):
            nat.check(lib.crb_feedback_force(ref.plan.h, vp(cur), vp(gain), None, vp(u), None))
            nat.check(lib.crb_rk4_stage(ref.plan.h, vp(ref.state), vp(cur), vp(acc), vp(bufs[s & 1]), vp(u), s, ts, dt, None, None))
            cur = bufs[s & 1]
        t = t + dt
    torch.cuda.synchronize()
    assert torch.equal(ens.state, ref.state)
    # opt-in hipGraph replay of one captured step (CRB_USE_GRAPH=1; a second rollout reuses the captured step):
    # an impulse that switches off mid-run is timed by the device clock exactly as by the host loop
    amps = torch.tensor([0.1, 0.2, 0.3], dtype=torch.float64, device=ens.device)
    finals = []
    for graph in (True, False):
        e = ensemble(cols, B, dict(enable_gravity=True))
        e.set_state(x0)
        if graph:
            os.environ["CRB_USE_GRAPH"] = "1"
        try:
            e.step_feedback(10, dt, gain, impulse_amp=amps, impulse_duration=5.5 * dt)
            e.step_feedback(10, dt, gain, impulse_amp=amps, impulse_duration=5.5 * dt)
        finally:
            os.environ.pop("CRB_USE_GRAPH", None)
        assert abs(e.time - 20 * dt) < 1e-15
        finals.append(e.unpack_state())
    assert torch.equal(finals[0], finals[1])


@pytest.mark.parametrize("n_e,kind", [(64, "linear"), (100, "mixed"), (200, "nonlinear")])
def test_adaptive_rk45_lean_rhs_equals_general_rhs(n_e, kind, monkeypatch):
    """crb_solve_rk45 evaluates the RHS through the lean exchange structure for plans without gravity (1, 2
    and 4 waves per beam here); the general RHS (CRB_DISABLE_LEAN) must take the same steps and land on
    the same state up to rounding, in fp64 and in fp32."""
    kinds = ["nonlinear" if i % 3 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    cols = nitinol_columns(n_e, kinds)
    kw = dict(fluid_density=1000.0, enable_fluid=True)
    B = 3
    amps = 0.05 * (1.0 + np.arange(B))
    for dtype, tol in ((torch.float64, 1e-9), (torch.float32, 2e-3)):
        out = []
        for disable in (False, True):
            if disable:
                monkeypatch.setenv("CRB_DISABLE_LEAN", "1")
            else:
                monkeypatch.delenv("CRB_DISABLE_LEAN", raising=False)
            ens = ensemble(cols, B, kw, dtype=dtype)
            st = ens.solve_rk45(4e-4, rtol=1e-6, atol=1e-9, impulse_amp=amps, t0=0.0)
            assert (np.asarray(st["status"]) == 0).all()
            out.append((np.asarray(st["accepted"]), np.asarray(st["nfev"]), ens.unpack_state().double().cpu().numpy()))
        if dtype == torch.float64:
            assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
        assert rel_err(out[0][2], out[1][2]) < tol


@pytest.mark.parametrize("tile", ["32", "48"])
@pytest.mark.parametrize("n_e,B", [(24, 70), (128, 130)])
def test_fp32_feedback_force_and_rollout(n_e, B, tile, monkeypatch):
    """fp32 plans: crb_feedback_force on v_mfma_f32_16x16x4_f32 (its C/D row map differs from the fp64
    instruction's) against torch.matmul, and the native closed-loop rollout against the fp64 one."""
    import ctypes as C

    from continuum_robot import _native as nat

    monkeypatch.setenv("CRB_FEEDBACK_TILE", tile)
    cols = nitinol_columns(n_e, "linear")
    e32 = ensemble(cols, B, dict(enable_gravity=True), dtype=torch.float32)
    n = e32.n
    g = torch.Generator(device="cpu").manual_seed(n_e)
    K = torch.randn((n, 2 * n), generator=g, dtype=torch.float32).to(e32.device)
    X = torch.randn((B, 2 * n), generator=g, dtype=torch.float32).to(e32.device)
    R = torch.randn((B, 2 * n), generator=g, dtype=torch.float32).to(e32.device)
    xs = e32.pack_state(X)
    for ref in (None, R):
        u = torch.zeros((B, e32.n_node, 4), dtype=torch.float32, device=e32.device)
        nat.check(nat.load().crb_feedback_force(e32.plan.h, C.c_void_p(xs.data_ptr()), C.c_void_p(K.data_ptr()),
                                                C.c_void_p(ref.data_ptr()) if ref is not None else None,
                                                C.c_void_p(u.data_ptr()), None))
        want = ((ref if ref is not None else 0) - X).double() @ K.double().t()
        got = e32.unpack_vec(u).double()
        assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < 5e-6
        assert float(u[..., 3].abs().max()) == 0.0
    # rollout: fp32 tracks fp64 (small gain, short horizon)
    rng = np.random.default_rng(n_e)
    gain = rng.normal(0.0, 2e-2, (n, 2 * n))
    x0 = rng.normal(0.0, 1e-4, (B, 2 * n))
    outs = []
    for dtype in (torch.float64, torch.float32):
        e = ensemble(cols, B, dict(enable_gravity=True), dtype=dtype)
        e.set_state(x0)
        e.step_feedback(20, 2e-5, gain, impulse_amp=np.full(B, 0.05))
        outs.append(e.unpack_state().double().cpu().numpy())
    assert rel_err(outs[1], outs[0]) < 2e-3


@pytest.mark.parametrize("n_e,kw", [(12, dict(enable_gravity=True)),                       # general kernel, several beams per wave
                                    (128, dict(fluid_density=1000.0, enable_fluid=True))])  # lean stepper
def test_whole_state_snapshots_equal_chunked_stepping(n_e, kw):
    """step(..., record="all"): snapshots of the whole state every k steps -- the reference's sol.y for every
    DOF on a t_eval grid (example_utilities.py:173-205 reads beam shapes from it) -- equal stepping in chunks
    of k and unpacking, bit for bit."""
    cols = nitinol_columns(n_e, "linear" if n_e < 64 else "nonlinear")
    B, k, n_rec = 5, 20, 6
    amps = 0.05 * (1.0 + np.arange(B))
    ens = ensemble(cols, B, kw)
    t, snaps = ens.step(k * n_rec + 7, 2e-5, impulse_amp=amps, record="all", record_every=k)
    assert snaps.shape == (n_rec, B, 2, ens.n_node, 4)
    red = ens.unpack_snapshots(snaps)
    assert float(snaps[..., 3].abs().max()) == 0.0
    ref = ensemble(cols, B, kw)
    for i in range(n_rec):
        ref.step(k, 2e-5, impulse_amp=amps)
        assert torch.equal(red[i], ref.unpack_state()), i
    ref.step(7, 2e-5, impulse_amp=amps)
    assert torch.equal(ens.unpack_state(), ref.unpack_state())


def test_adaptive_rk45_dense_output_of_the_whole_state():
    """solve_rk45(record="all", t_eval=...): every DOF of sol.y on the grid, equal (bit for bit) to the
    single-DOF dense output of the same integration, for positions and velocities of several nodes."""
    cols = nitinol_columns(64, "nonlinear")
    kw = dict(fluid_density=1000.0, enable_fluid=True)
    B = 3
    amps = 0.05 * (1.0 + np.arange(B))
    grid = (1e-5, 2.5e-5, 12)
    ens = ensemble(cols, B, kw)
    st = ens.solve_rk45(4e-4, rtol=1e-6, atol=1e-9, impulse_amp=amps, t0=0.0, record="all", t_eval=grid)
    snaps = st["y"]
    assert snaps.shape == (grid[2], B, 2, ens.n_node, 4)
    red = ens.unpack_snapshots(snaps)
    for node, param in ((64, "w"), (64, "dw_dt"), (20, "phi"), (1, "u")):
        one = ensemble(cols, B, kw)
        s1 = one.solve_rk45(4e-4, rtol=1e-6, atol=1e-9, impulse_amp=amps, t0=0.0, record=(node, param), t_eval=grid)
        vel = param.endswith("_dt")
        idx = one.reduced_index(node, param[1:-3] if vel else param) + (one.n if vel else 0)
        assert torch.equal(red[:, :, idx].t().contiguous(), torch.as_tensor(s1["y"], device=red.device)), (node, param)


@pytest.mark.parametrize("seed", range(int(os.environ.get("CRB_FUZZ_N", "24"))))   # CRB_FUZZ_N=400 for a long hunt
def test_randomised_topologies_against_oracle(seed):
    """Differential test over random beams: size (1..300 elements: several beams per wave, one wave, up to
    eight waves, padding threads), element kinds, boundary-condition sets (incl. constrained interior nodes and
    free roots), force sets, held input or impulse, random initial states, per-beam coefficients or shared."""
    rng = np.random.default_rng(1000 + seed)
    n_e = int(rng.choice([1, 2, 5, 9, 17, 31, 40, 64, 65, 100, 128, 200, 257, 300]))
    mode = int(rng.integers(0, 3))
    kinds = (["linear"] * n_e if mode == 0 else ["nonlinear"] * n_e if mode == 1 else
             [("nonlinear" if rng.random() < 0.5 else "linear") for _ in range(n_e)])
    bcs = []
    for i in range(n_e):
        r = rng.random()
        bcs.append("FIXED" if (i == 0 and r < 0.6) or r > 0.995 else ("PINNED" if r > 0.97 else "NONE"))
    cols = nitinol_columns(n_e, kinds, bcs)
    kw = {}
    if rng.random() < 0.5:
        kw.update(fluid_density=1000.0, enable_fluid=True)
    if rng.random() < 0.5:
        kw.update(enable_gravity=True)
    B = int(rng.integers(1, 7))
    hetero = rng.random() < 0.3
    per_beam = [_scaled(cols, rng) for _ in range(B)] if hetero else [cols] * B
    obs = [oracle_beam(c, **kw) for c in (per_beam if hetero else [cols])]
    n = obs[0].n
    ens = ensemble(per_beam if hetero else cols, B, kw)
    assert ens.n == n
    x0 = rng.normal(0.0, 1e-5, (B, 2 * n))
    ens.set_state(x0)
    steps = int(rng.integers(3, 40))
    branch = rng.random()
    if branch < 0.25:     # closed loop: u = K (r - x) at every stage, dense random gain, impulse on top
        gain = rng.normal(0.0, 1e-2, (n, 2 * n))
        ref = rng.normal(0.0, 1e-5, (B, 2 * n))
        amps = rng.uniform(0.01, 0.1, B)
        idx = int(rng.integers(0, n))
        ens.step_feedback(steps, 2e-5, gain, reference=ref, impulse_amp=amps, impulse_index=idx, t0=0.0)
        want = np.array([obs[b if hetero else 0].rk4_feedback(x0[b], 2e-5, steps, gain, reference=ref[b], amp=amps[b], idx=idx)
                         for b in range(B)])
    elif branch < 0.6:
        u = rng.normal(0.0, 1e-3, (B, n))
        ens.step(steps, 2e-5, held_force=u)
        want = np.array([obs[b if hetero else 0].rk4_held(x0[b], 2e-5, steps, u[b]) for b in range(B)])
    else:
        amps = rng.uniform(0.01, 0.1, B)
        idx = int(rng.integers(0, n))
        dur = float(rng.uniform(0.0, steps * 2e-5))
        ens.step(steps, 2e-5, impulse_amp=amps, impulse_index=idx, impulse_duration=dur, t0=0.0)
        want = np.array([obs[b if hetero else 0].rk4_impulse(x0[b], 2e-5, steps, amps[b], duration=dur, idx=idx)
                         for b in range(B)])
    got = ens.unpack_state().cpu().numpy()
    assert np.isfinite(want).all()
    assert_blocks(got, want, ens.free_index, 1e-8, what=(n_e, mode, bcs[:3], kw, B, hetero))


@pytest.mark.parametrize("n_e,kind,kw", [(4, "linear", dict(enable_gravity=True)),
                                          (10, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True)),
                                          (5, "mixed", dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True))])
def test_adaptive_rk45_on_the_reference_test_sizes(n_e, kind, kw):
    """The reference's own tests hand 4-10 element beams to solve_ivp(method="RK45")
    (tests/test_dynamic_beam.py:218-220): beams of a few slots run one per wave in crb_solve_rk45, each against
    scipy over the oracle RHS -- same accepted steps, same nfev, same state, same dense output."""
    from scipy.integrate import solve_ivp

    kinds = ["nonlinear" if i % 2 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    cols = nitinol_columns(n_e, kinds)
    ob = oracle_beam(cols, **kw)
    n, B = ob.n, 3
    amps = np.array([0.05, 0.2, 1.0])
    dur, t_end, rtol, atol = 4e-4, 1.5e-3, 1e-6, 1e-9
    grid = (1e-4, 1e-4, 14)
    ens = ensemble(cols, B, kw)
    st = ens.solve_rk45(t_end, rtol=rtol, atol=atol, impulse_amp=amps, impulse_duration=dur, record=(n_e, "w"), t_eval=grid)
    got = ens.unpack_state().cpu().numpy()
    tev = grid[0] + grid[1] * np.arange(grid[2])
    for b in range(B):
        def fun(t, x, b=b):
            u = np.zeros(n)
            if t < dur:
                u[-2] = amps[b]
            return ob.rhs(x, u)

        sol = solve_ivp(fun, (0.0, t_end), np.zeros(2 * n), method="RK45", rtol=rtol, atol=atol, t_eval=tev)
        assert st["nfev"][b] == sol.nfev, (b, st["nfev"][b], sol.nfev)
        assert rel_err(torch.as_tensor(st["y"]).cpu().numpy()[b], sol.y[n - 2]) < 1e-8
    # terminal state against scipy without t_eval (same integration)
    for b in range(B):
        def fun(t, x, b=b):
            u = np.zeros(n)
            if t < dur:
                u[-2] = amps[b]
            return ob.rhs(x, u)

        sol = solve_ivp(fun, (0.0, t_end), np.zeros(2 * n), method="RK45", rtol=rtol, atol=atol)
        assert st["accepted"][b] == len(sol.t) - 1
        assert_blocks(got[b], sol.y[:, -1], ens.free_index, 1e-8, what=b)


# ------------------------------------------------------------------ f-3: ensembles the reference runs as separate processes
def _example_files(n_seg=6):
    """The three beam files of examples/example_utilities.py:89-113 (data of :25-34): all-linear, all-nonlinear,
    linear base / nonlinear tip, FIXED at node 0."""
    half = n_seg // 2
    return {"linear": nitinol_columns(n_seg, "linear"), "nonlinear": nitinol_columns(n_seg, "nonlinear"),
            "mixed": nitinol_columns(n_seg, ["linear"] * half + ["nonlinear"] * (n_seg - half))}


def test_reference_parallel_examples_run_as_one_ensemble():
    """The six tasks of examples/beam_comparison_fluid.py:52-76 (linear / nonlinear / mixed, each without and with
    fluid) and the three of beam_comparison_gravity.py:53-66, which the reference maps over a process pool
    (:82-83), each as ONE ensemble with per-beam element types and per-beam ForceParams: every beam against its own
    oracle, per DOF block, tip impulse 0.1 N for t < 0.01 s as example_utilities.py:144-148."""
    from continuum_robot.batched import BeamEnsemble
    from continuum_robot.models.force_params import ForceParams

    files = _example_files()
    none, fluid = ForceParams(), ForceParams(fluid_density=1000.0, enable_fluid_effects=True)
    grav = ForceParams(enable_gravity_effects=True)
    for tasks, steps in (([("linear", none), ("nonlinear", none), ("mixed", none),
                           ("linear", fluid), ("nonlinear", fluid), ("mixed", fluid)], 600),
                         ([("linear", grav), ("nonlinear", grav), ("mixed", grav)], 150)):   # (nonlinear + distributed load
        ens = BeamEnsemble.from_dataframes([files[k] for k, _ in tasks], force_params=[fp for _, fp in tasks])  # diverges later)
        assert not ens.mixed_topology and ens.n == 18
        B = len(tasks)
        ens.step(steps, 2e-5, impulse_amp=np.full(B, 0.1))
        got = ens.unpack_state().cpu().numpy()
        for b, (k, fp) in enumerate(tasks):
            ob = oracle_beam(files[k], fluid_density=fp.fluid_density, enable_fluid=fp.enable_fluid_effects,
                             enable_gravity=fp.enable_gravity_effects)
            want = ob.rk4_impulse(np.zeros(2 * ob.n), 2e-5, steps, 0.1)
            assert np.isfinite(want).all()
            assert_blocks(got[b], want, ens.free_index, 1e-9, what=(k, fp.enable_fluid_effects, fp.enable_gravity_effects))
        # the RHS entry point through the per-beam tables as well
        x = np.random.default_rng(3).normal(0, 1e-3, (B, 2 * ens.n))
        xd = ens.rhs(x).cpu().numpy()
        for b, (k, fp) in enumerate(tasks):
            ob = oracle_beam(files[k], fluid_density=fp.fluid_density, enable_fluid=fp.enable_fluid_effects,
                             enable_gravity=fp.enable_gravity_effects)
            assert_blocks(xd[b], ob.rhs(x[b]), ens.free_index, 1e-10, what=b)


def _lqr_gain_of(cols, node_bc=None):
    """The gain lqr_control.py:46-84 designs for ONE model: CARE on the beam's LINEAR model (every element taken as linear --
    a nonlinear element has no stiffness matrix, euler_bernoulli_beam.py:422-511), Q = diag(100 I, 10 I), R = I."""
    from continuum_robot import _native as nat
    from continuum_robot.control import LinearQuadraticRegulator

    lin = dict(cols)
    lin["type"] = np.array(["linear"] * len(cols["length"]))
    plan = nat.Plan(lin, node_bc=node_bc, device=-1)
    K, M = plan.stiffness(), plan.mass()
    n = K.shape[0]
    Q = np.eye(2 * n)
    Q[:n, :n] *= 100
    Q[n:, n:] *= 10
    return LinearQuadraticRegulator(K, M, Q, np.eye(n)).compute_gain_matrix()


def test_per_beam_gains_close_the_loop_on_heterogeneous_ensembles():
    """crb_step_rk4_feedback_grouped / step_feedback(gain=[K_b ...]): one LQR gain per model, as the reference designs them
    (lqr_control.py:46-84).  (a) The six tasks of examples/beam_comparison_fluid.py:52-76 as one ensemble, each beam under the
    CARE gain of its own linearised model (three distinct gains, each shared by the dry and the wet task: three groups);
    (b) an ensemble whose beams differ in length and boundary conditions (mixed topology: per-beam reduced orderings, padded
    references), one beam without any feedback.  Every beam against RK4 over ITS oracle RHS with ITS gain in every stage."""
    from continuum_robot.batched import BeamEnsemble
    from continuum_robot.models.force_params import ForceParams

    files = _example_files()
    none, fluid = ForceParams(), ForceParams(fluid_density=1000.0, enable_fluid_effects=True)
    tasks = [("linear", none), ("nonlinear", none), ("mixed", none), ("linear", fluid), ("nonlinear", fluid), ("mixed", fluid)]
    ens = BeamEnsemble.from_dataframes([files[k] for k, _ in tasks], force_params=[fp for _, fp in tasks])
    gain_of = {k: _lqr_gain_of(files[k]) for k in files}
    gains = [gain_of[k] for k, _ in tasks]
    B, steps, dt = len(tasks), 300, 5e-6
    amps = 0.1 * (1.0 + np.arange(B))
    ens.step_feedback(steps, dt, gains, impulse_amp=amps)
    got = ens.unpack_state().cpu().numpy()
    for b, (k, fp) in enumerate(tasks):
        ob = oracle_beam(files[k], fluid_density=fp.fluid_density, enable_fluid=fp.enable_fluid_effects)
        want = ob.rk4_feedback(np.zeros(2 * ob.n), dt, steps, gains[b], amp=amps[b])
        assert np.isfinite(want).all() and np.abs(want).max() > 0
        assert_blocks(got[b], want, ens.free_index, 1e-9, what=(k, fp.enable_fluid_effects))
    # (b) lengths, boundary conditions and gravity differ; references; beam 2 uncontrolled
    rng = np.random.default_rng(21)
    ne = [6, 4, 6, 5, 4, 6]
    bcs_of = [["FIXED"] + ["NONE"] * 5, ["PINNED"] + ["NONE"] * 3, ["FIXED"] + ["NONE"] * 5, ["FIXED", "NONE", "PINNED", "NONE", "NONE"],
              ["PINNED"] + ["NONE"] * 3, ["FIXED"] + ["NONE"] * 5]
    beams = [nitinol_columns(n, "linear", bc) for n, bc in zip(ne, bcs_of)]
    fps = [ForceParams(enable_gravity_effects=bool(b % 2)) for b in range(len(ne))]
    ens = BeamEnsemble.from_dataframes(beams, force_params=fps)
    assert ens.mixed_topology
    B = len(ne)
    gains = [_lqr_gain_of(beams[b]) for b in range(B)]
    gains[2] = None
    obs = [oracle_beam(beams[b], enable_gravity=fps[b].enable_gravity_effects) for b in range(B)]
    x0 = [rng.normal(0.0, 1e-5, 2 * ob.n) for ob in obs]
    refs = [rng.normal(0.0, 1e-4, 2 * ob.n) for ob in obs]
    ens.set_state(ens.pad_states(x0))
    steps = 120
    ens.step_feedback(steps, dt, gains, reference=ens.pad_states(refs), impulse_amp=np.full(B, 0.05), impulse_index=-2)
    for b in range(B):
        K = gains[b] if gains[b] is not None else np.zeros((obs[b].n, 2 * obs[b].n))
        want = obs[b].rk4_feedback(x0[b], dt, steps, K, reference=refs[b], amp=0.05)
        assert_blocks(ens.beam_state(b), want, obs[b].red2full(), 1e-9, what=("mixed", b))
    # like models under equal gains share a group: {0, 5}, {1, 4}, {3}; beam 2 has none
    group_of, mats = ens._gain_groups(gains)
    assert group_of[0] == group_of[5] and group_of[1] == group_of[4] and group_of[2] == -1 and len(mats) == 3
    with pytest.raises(ValueError, match="expected shape"):
        ens.step_feedback(1, dt, [gains[0]] * B)


@pytest.mark.parametrize("size", ["small", "lean"])
def test_ensemble_with_per_beam_lengths_boundary_conditions_and_force_params(size):
    """One ensemble of beams that differ in element COUNT (padding nodes), boundary-condition column (FIXED / PINNED /
    free root, a constrained interior node), element types, fluid density, drag on/off, gravity on/off and gravity
    vector: each beam against its own oracle (RK4 rollout with the impulse at each beam's own tip, and the RHS), in the
    reference's per-beam reduced ordering.  `small`: several beams per wave; `lean`: beams of 60..130 elements."""
    from continuum_robot.batched import BeamEnsemble
    from continuum_robot.models.force_params import ForceParams

    rng = np.random.default_rng(11)
    ne = [6, 4, 6, 5, 3, 6, 2] if size == "small" else [130, 96, 130, 60, 128, 77]
    beams, fps = [], []
    for b, n in enumerate(ne):
        kinds = [("nonlinear" if rng.random() < 0.5 else "linear") for _ in range(n)]
        bcs = ["NONE"] * n
        bcs[0] = ["FIXED", "PINNED", "FIXED", "NONE", "FIXED", "FIXED", "PINNED"][b % 7]
        if b % 3 == 2 and n > 3:
            bcs[n // 2] = "PINNED"
        cols = nitinol_columns(n, kinds, bcs)
        cols["elastic_modulus"] = cols["elastic_modulus"] * rng.uniform(0.9, 1.1)
        beams.append(cols)
        fps.append(ForceParams(fluid_density=float(rng.uniform(500, 1500)), enable_fluid_effects=bool(b % 2),
                               enable_gravity_effects=bool(b % 3 != 1),
                               gravity_vector=[float(rng.uniform(-2, 2)), -9.81, 0.0]))
    ens = BeamEnsemble.from_dataframes(beams, force_params=fps)
    B = len(ne)
    assert ens.mixed_topology and ens.n_elem == max(ne) and list(ens.n_elem_per_beam) == ne
    obs = [oracle_beam(beams[b], fluid_density=fps[b].fluid_density, enable_fluid=fps[b].enable_fluid_effects,
                       enable_gravity=fps[b].enable_gravity_effects, gravity=fps[b].get_gravity_vector()) for b in range(B)]
    assert [ob.n for ob in obs] == list(ens.n_per_beam) and ens.n == max(ob.n for ob in obs)
    for b in range(B):
        assert np.array_equal(ens.free_index_per_beam[b], obs[b].red2full())
    x0 = [rng.normal(0.0, 1e-6, 2 * ob.n) for ob in obs]
    ens.set_state(ens.pad_states(x0))
    # pack -> unpack round trip keeps every beam's vector and zeroes the padding
    back = ens.unpack_state().cpu().numpy()
    for b in range(B):
        assert np.array_equal(ens.beam_state(b), x0[b])
        assert np.all(back[b, obs[b].n:ens.n] == 0) and np.all(back[b, ens.n + obs[b].n:] == 0)
    xd = ens.rhs().cpu().numpy()
    for b in range(B):
        assert_blocks(ens.beam_state(b, xd), obs[b].rhs(x0[b]), obs[b].red2full(), 1e-10, what=("rhs", b))
    amps = rng.uniform(0.01, 0.05, B)
    steps = 40
    ens.step(steps, 2e-5, impulse_amp=amps, impulse_index=-2)        # -2 = EACH beam's own tip w
    for b in range(B):
        want = obs[b].rk4_impulse(x0[b], 2e-5, steps, amps[b])
        assert np.isfinite(want).all()
        assert_blocks(ens.beam_state(b), want, obs[b].red2full(), 1e-9, what=("step", b, ne[b]))
    tips = ens.tip_displacement().cpu().numpy()
    for b in range(B):
        assert tips[b] == ens.beam_state(b)[obs[b].n - 2]
    # adaptive RK45 with per-beam error norms (N = each beam's own state size)
    ens.set_state(ens.pad_states(x0))
    st = ens.solve_rk45(3e-4, rtol=1e-6, atol=1e-9, impulse_amp=amps, impulse_duration=1e-4, t0=0.0)
    assert np.all(st["status"] == 0)
    from scipy.integrate import solve_ivp

    for b in (0, 1, B - 1):
        n_b = obs[b].n

        def fun(t, x, b=b, n_b=n_b):
            u = np.zeros(n_b)
            if t < 1e-4:
                u[-2] = amps[b]
            return obs[b].rhs(x, u)

        sol = solve_ivp(fun, (0.0, 3e-4), x0[b], method="RK45", rtol=1e-6, atol=1e-9)
        assert st["nfev"][b] == sol.nfev and st["accepted"][b] == len(sol.t) - 1, (b, st["nfev"][b], sol.nfev)
        assert_blocks(ens.beam_state(b), sol.y[:, -1], obs[b].red2full(), 1e-8, what=("rk45", b))
    # one gain matrix for the whole ensemble needs one free-DOF set
    with pytest.raises(Exception, match="one free-DOF set"):
        ens.step_feedback(1, 5e-6, np.zeros((ens.n, 2 * ens.n)))


# ------------------------------------------------------------------ batched functional composition (SURVEY §7 "hard parts")
class _TipSpringDamper:
    """The reference's StateAwareForce (tests/test_advanced_composition.py:36-65) restated for an ensemble: a
    spring-damper on the last transverse DOF, on torch tensors x[B, 2n] -> [B, n]."""

    def __init__(self, stiffness=1000.0, damping=10.0, enabled=True):
        self.stiffness, self.damping, self.enabled = stiffness, damping, enabled

    def compute_forces(self, x, t):
        assert t == 0.0            # the reference passes 0.0 as the force time (dynamic_beam_model.py:265)
        n = x.shape[1] // 2
        f = torch.zeros((x.shape[0], n), dtype=x.dtype, device=x.device)
        f[:, n - 2] = -self.stiffness * x[:, n - 2] - self.damping * x[:, n + n - 2]
        return f

    def is_enabled(self):
        return self.enabled


def _rk4_over_oracle(rhs, x, t, dt, steps):
    """classical RK4 with the clock convention of crb_step_rk4 (t accumulates by addition) over rhs(t, x)"""
    for _ in range(steps):
        th, t1 = t + 0.5 * dt, t + dt
        k1 = rhs(t, x)
        k2 = rhs(th, x + 0.5 * dt * k1)
        k3 = rhs(th, x + 0.5 * dt * k2)
        k4 = rhs(t1, x + dt * k3)
        x = x + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        t = t1
    return x, t


@pytest.mark.parametrize("n_e,kind", [(8, "mixed"), (70, "nonlinear")])
def test_batched_functional_composition_matches_oracle(n_e, kind):
    """step_composed: registry forces (built-in drag + gravity fused in the kernel, a state-dependent user force on the
    device), a time-varying input u(t) and the tip impulse, against RK4 over the oracle RHS with the same forces;
    `enabled` toggled between calls takes effect (force_registry.py:66-67); a forces_func REPLACES the registry
    (dynamic_beam_model.py:253-254)."""
    kinds = ["nonlinear" if i % 2 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    cols = nitinol_columns(n_e, kinds)
    kw = dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True)
    B, dt = 3, 2e-5
    ens = ensemble(cols, B, kw)
    n = ens.n
    assert len(ens.force_registry) == 2          # the two built-in markers, like the reference's auto-registration
    spring = _TipSpringDamper(stiffness=800.0, damping=5.0)
    off = _TipSpringDamper(enabled=False)
    ens.force_registry.register(spring)
    ens.force_registry.register(off)             # register() ignores a disabled component (force_registry.py:20-21)
    assert len(ens.force_registry) == 3 and off not in ens.force_registry
    forces = ens.force_registry.get_registered_forces()
    forces.clear()
    assert len(ens.force_registry) == 3          # a copy was returned (:49)
    rng = np.random.default_rng(n_e)
    x0 = rng.normal(0.0, 1e-6, (B, 2 * n))
    amps = np.array([0.05, 0.1, 0.2])
    w = 2 * np.pi * 900.0
    dof = n - 5

    def u_of_t(t):                               # callable input u(t) (dynamic_beam_model.py:357-360)
        u = torch.zeros((B, n), dtype=torch.float64, device=ens.device)
        u[:, dof] = 0.02 * float(np.sin(w * t)) * torch.arange(1, B + 1, device=ens.device)
        return u

    obs = {(d, g): oracle_beam(cols, fluid_density=1000.0, enable_fluid=d, enable_gravity=g)
           for d in (True, False) for g in (True, False)}

    def oracle_rhs(b, drag, grav, use_spring, t_imp_end):
        def rhs(t, x):
            u = np.zeros(n)
            u[dof] = 0.02 * np.sin(w * t) * (b + 1)
            if t < t_imp_end:
                u[n - 2] += amps[b]
            if use_spring:
                u[n - 2] += -800.0 * x[n - 2] - 5.0 * x[2 * n - 2]
            return obs[(drag, grav)].rhs(x, u)
        return rhs

    ens.set_state(x0)
    imp_end = 13.5 * dt
    ens.step_composed(20, dt, u=u_of_t, impulse_amp=amps, impulse_duration=imp_end)
    ens._auto_drag.enabled = False               # toggled at run time: the next call runs without the fused drag
    spring.enabled = False
    ens.step_composed(10, dt, u=u_of_t, impulse_amp=amps, impulse_duration=imp_end)
    spring.enabled = True
    ens._auto_gravity.enabled = False
    ens.step_composed(10, dt, u=u_of_t, impulse_amp=amps, impulse_duration=imp_end)
    got = ens.unpack_state().cpu().numpy()
    assert abs(ens.time - 40 * dt) < 1e-15
    for b in range(B):
        x, t = _rk4_over_oracle(oracle_rhs(b, True, True, True, imp_end), x0[b], 0.0, dt, 20)
        x, t = _rk4_over_oracle(oracle_rhs(b, False, True, False, imp_end), x, t, dt, 10)
        x, t = _rk4_over_oracle(oracle_rhs(b, False, False, True, imp_end), x, t, dt, 10)
        assert_blocks(got[b], x, ens.free_index, 1e-9, what=("registry", b))
    # forces_func replaces the registry: only the user force acts, the built-in terms are off
    ens.set_state(x0)
    ens.step_composed(15, dt, forces_func=lambda x, t: spring.compute_forces(x, t), u=0.01 * np.ones((B, n)), t0=0.0)
    got = ens.unpack_state().cpu().numpy()
    for b in range(B):
        def rhs(t, x, b=b):
            u = np.full(n, 0.01)
            u[n - 2] += -800.0 * x[n - 2] - 5.0 * x[2 * n - 2]
            return obs[(False, False)].rhs(x, u)

        x, _ = _rk4_over_oracle(rhs, x0[b], 0.0, dt, 15)
        assert_blocks(got[b], x, ens.free_index, 1e-9, what=("forces_func", b))
    # a user force of the wrong shape is a ValueError (test_advanced_composition.py:341 accepts it), exceptions propagate
    with pytest.raises(ValueError, match="dimension mismatch"):
        ens.step_composed(1, dt, forces_func=lambda x, t: torch.zeros((B, n + 1), device=ens.device))

    def boom(x, t):
        raise RuntimeError("user force failed")

    with pytest.raises(RuntimeError, match="user force failed"):
        ens.step_composed(1, dt, forces_func=boom)
    # with nothing but the built-in forces the composed path equals the fused stepper
    a, b2 = ensemble(cols, B, kw), ensemble(cols, B, kw)
    a.set_state(x0)
    b2.set_state(x0)
    a.step_composed(12, dt, impulse_amp=amps)
    b2.step(12, dt, impulse_amp=amps)
    assert_blocks(a.unpack_state().cpu().numpy(), b2.unpack_state().cpu().numpy(), a.free_index, 1e-11)


# ------------------------------------------------------------------ implicit stepper (stiff integrations, f-2)
@pytest.mark.parametrize("n_e,kind,kw,h", [
    (10, "linear", dict(enable_gravity=True), 1e-3),                                   # BASELINE config 1's beam
    (6, "mixed", dict(fluid_density=1000.0, enable_fluid=True), 1e-4),                 # the parallel examples' mixed beam
    (64, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True), 2e-4),            # one wave per beam, 6 levels
    (130, "mixed", dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True), 5e-4),   # 4 waves, padding threads
    (256, "linear", dict(enable_gravity=True), 1e-3),                                  # 8 levels: the largest supported
    (256, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True), 1e-4),           # small step: A's reduction stops at 5 of 8 levels
    (128, "linear", dict(enable_gravity=True, gravity=[2.0, -9.81, 0.0]), 1e-4),       # the same with two waves per beam
    (200, "mixed", dict(fluid_density=1000.0, enable_fluid=True), 1e-4),               # truncated, padding threads
    (256, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True), 2e-4),           # 6 levels (two waves per SIMD without gravity)
    (256, "linear", dict(enable_gravity=True), 3e-4),                                  # 6 - 7 levels with gravity
    (100, "linear", dict(enable_gravity=True), 2e-4),                                  # two waves per beam, 6 of 7 levels
])
def test_implicit_stepper_matches_oracle(n_e, kind, kw, h):
    """crb_step_implicit (implicit midpoint rule, modified Newton with A = M + h^2/4 K0 by cyclic reduction) against
    the oracle's dense-LU statement of the same scheme, per DOF block, at step sizes 5x .. 50x beyond the explicit
    stability limit; chunked calls equal one call up to the restart of the iteration."""
    half = n_e // 2
    kinds = (["linear"] * half + ["nonlinear"] * (n_e - half)) if kind == "mixed" else kind
    cols = nitinol_columns(n_e, kinds)
    ob = oracle_beam(cols, **kw)
    B, steps = 3, 40
    amps = np.array([0.05, 0.1, 0.2])
    rng = np.random.default_rng(n_e)
    x0 = rng.normal(0.0, 1e-6, (B, 2 * ob.n))
    dur = 10.0 * h          # the impulse switches off on a step boundary, inside the run
    ens = ensemble(cols, B, kw)
    ens.set_state(x0)
    t = ens.step_implicit(steps, h, n_iter=3, impulse_amp=amps, impulse_duration=dur)
    assert abs(t - steps * h) < 1e-12
    got = ens.unpack_state().cpu().numpy()
    for b in range(B):
        want = ob.implicit(x0[b], h, steps, n_iter=3, amp=amps[b], duration=dur)
        assert np.isfinite(want).all()
        # cond(A) ~ 1e6..1e9 at these step sizes (alpha K0 dominates M): cyclic reduction and the oracle's dense LU
        # round differently; the measured agreement is 1e-9..1e-8 per block
        assert_blocks(got[b], want, ens.free_index, 1e-7, what=(n_e, b))
    # held force + recording through the same entry point
    ens.set_state(x0)
    u = rng.normal(0.0, 1e-3, (B, ob.n))
    _, rec = ens.step_implicit(20, h, n_iter=2, held_force=u, record=(n_e, "w"), record_every=5, t0=0.0)
    assert rec.shape == (B, 4)
    want = ob.implicit(x0[1], h, 20, n_iter=2, u_held=u[1])
    assert_blocks(ens.unpack_state().cpu().numpy()[1], want, ens.free_index, 1e-7)
    assert abs(float(rec[1, -1]) - want[ob.n - 2]) <= 1e-7 * abs(want[ob.n - 2])


def test_implicit_stepper_integrates_config1_to_the_lsoda_golden(golden):
    """BASELINE config 1 (10 linear elements + gravity, the reference's CPU example) integrated to t = 0.02 / 0.05 /
    0.1 s in ONE launch each, against scipy LSODA (rtol 1e-10) over the REFERENCE RHS (tests/golden/g8_lsoda.npz,
    anchor tip w(0.1) = -0.0725343890628583).  At h = 1e-4 s the tip displacement is within LSODA's DEFAULT tolerance
    band (atol 1e-6 + rtol 1e-3 |w|, what the example itself asks of its integrator) at every time -- the error is
    ~1e-6 m of unresolved high-frequency content, constant in time, so the relative figure falls as the beam deflects."""
    z = golden["g8_lsoda"]
    name = "lin10_grav"
    ens = ensemble(beam_columns(z, name), 2, force_kwargs(z, name))
    tight = z[f"{name}/x_tight"]
    w_bound = {0.02: 1e-3, 0.05: 3e-4, 0.1: 6e-5}      # measured 3.8e-4, 9.4e-5, 1.9e-5 (w block, relative)
    for k, t_end in enumerate(z[f"{name}/times"]):
        ens.zero_state()
        ens.step_implicit(int(round(t_end / 1e-4)), 1e-4, n_iter=2, impulse_amp=np.full(2, 0.1))
        got = ens.unpack_state().cpu().numpy()
        errs = block_errs(got[0], tight[k], ens.free_index)
        assert errs["w"] < w_bound[round(float(t_end), 2)], (t_end, errs)
        tip, ref = got[0, ens.n - 2], tight[k][ens.n - 2]
        assert abs(tip - ref) < 1e-6 + 1e-3 * abs(ref), (t_end, tip, ref)       # LSODA's default tolerance band
        assert abs(tip - ref) < 2e-6                                            # (measured 1.2e-6, 1.5e-6, 5.9e-7 m)
        assert np.array_equal(got[0], got[1])
    assert abs(got[0, ens.n - 2] / tight[-1][ens.n - 2] - 1.0) < 2e-5          # t = 0.1 s: 8e-6 relative


@pytest.mark.parametrize("n_e,kind,kw,rho,held", [
    (1, "linear", dict(), 0.5, False),
    (10, "linear", dict(enable_gravity=True), 0.0, False),                              # BASELINE config 1, asymptotic annihilation
    (10, "linear", dict(enable_gravity=True), 0.8, True),
    (6, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True), 0.5, False),
    (64, "linear", dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True), 0.6, False),
    (130, "linear", dict(enable_gravity=True), 0.3, True),                              # four waves, padding threads
    (256, "linear", dict(fluid_density=1000.0, enable_fluid=True), 0.9, False),         # 8 levels
])
def test_damped_implicit_stepper_matches_the_oracle(n_e, kind, kw, rho, held):
    """crb_step_implicit_damped (generalised-alpha, spectral radius rho at infinite frequency) against the oracle's statement
    of the same scheme (orc_implicit_alpha) per DOF block: the impulse sampled at t + (1 - alpha_f) h, a held force, the
    acceleration history started from the RHS at t0 -- and rho = 1 reproduces crb_step_implicit bit for bit."""
    kinds = ["nonlinear" if i % 3 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    cols = nitinol_columns(n_e, kinds)
    B, h, steps = 3, 1e-4, 40
    ens = ensemble(cols, B, kw)
    ob = oracle_beam(cols, **kw)
    rng = np.random.default_rng(7 * n_e)
    x0 = 1e-4 * rng.standard_normal((B, 2 * ens.n))
    u = 0.05 * rng.standard_normal((B, ens.n)) if held else None
    amps = np.array([0.1, 0.0, 0.2])
    ens.set_state(x0)
    ens.step_implicit(steps, h, n_iter=2, impulse_amp=amps, impulse_duration=17.3 * h, held_force=u, rho_inf=rho, t0=0.0)
    got = ens.unpack_state().cpu().numpy()
    for b in range(B):
        want = ob.implicit_alpha(x0[b], h, steps, rho, n_iter=2, amp=amps[b], duration=17.3 * h, u_held=None if u is None else u[b])
        assert_blocks(got[b], want, ens.free_index, 1e-8, what=(b, rho))
    one, mid = ensemble(cols, B, kw), ensemble(cols, B, kw)
    one.set_state(x0)
    mid.set_state(x0)
    one.step_implicit(steps, h, impulse_amp=amps, held_force=u, rho_inf=1.0)
    mid.step_implicit(steps, h, impulse_amp=amps, held_force=u)
    assert np.array_equal(one.state.cpu().numpy(), mid.state.cpu().numpy())
    with pytest.raises(Exception, match="rho_inf"):
        ens.step_implicit(1, h, rho_inf=1.5)


def test_damped_implicit_stepper_filters_what_the_step_cannot_resolve(golden):
    """What the damped variant is for: BASELINE config 1 at the examples' h = 1e-4 s.  The midpoint rule keeps the bending modes
    above 1 / h ringing with the wrong phase (golden G8: the dphi/dt block is 0.2 .. 0.3 of its norm away from LSODA-tight);
    LSODA at its default tolerances filters them (its BDF formulas damp what it does not resolve), and so does
    generalised-alpha with rho_inf = 0: after the impulse has ended the velocity blocks follow the DEFAULT-tolerance LSODA
    run of G8 about twice as closely as the midpoint rule does, while the displacements stay where the midpoint rule has them."""
    z = golden["g8_lsoda"]
    cols, kw = beam_columns(z, "lin10_grav"), force_kwargs(z, "lin10_grav")
    times, tight, dflt = z["lin10_grav/times"], z["lin10_grav/x_tight"], z["lin10_grav/x_default_tol"]
    out = {}
    for rho in (1.0, 0.0):
        ens = ensemble(cols, 1, kw)
        rows, t = [], 0.0
        for t1 in times:
            ens.step_implicit(int(round((t1 - t) / 1e-4)), 1e-4, impulse_amp=np.full(1, 0.1), rho_inf=rho)
            t = float(t1)
            rows.append(ens.unpack_state().cpu().numpy()[0])
        out[rho] = np.array(rows)
    n = out[1.0].shape[1] // 2
    for ti in range(len(times)):
        band = 1e-6 + 1e-3 * np.abs(tight[ti])
        for rho in (1.0, 0.0):
            assert np.max(np.abs(out[rho][ti][:n] - tight[ti][:n]) / band[:n]) < 25.0, (rho, ti)     # positions: a few bands at h = 1e-4 (DESIGN)
    print("dphi/dt block distance to default-tolerance LSODA:", [(float(np.linalg.norm(out[r][-1][n + 2::3] - dflt[-1][n + 2::3]) /
          np.linalg.norm(dflt[-1][n + 2::3]))) for r in (1.0, 0.0)])
    for blk in (1, 2):                                    # dw/dt, dphi/dt at t = 0.1 s
        d_mid = np.linalg.norm(out[1.0][-1][n + blk::3] - dflt[-1][n + blk::3])
        d_dmp = np.linalg.norm(out[0.0][-1][n + blk::3] - dflt[-1][n + blk::3])
        assert d_dmp < 0.6 * d_mid, (blk, d_dmp, d_mid)       # measured 0.52 (dw/dt), 0.43 (dphi/dt: 0.058 against 0.135 of the block norm)


def test_beam_shapes_follow_the_reference_indexing():
    """BeamEnsemble.beam_shapes: extract_beam_shapes (examples/example_utilities.py:173-205) on whole-state snapshots --
    by default with the reference's own indexing (sol.y[n_pos + 1::3]: the velocity half), restated
    here on the unpacked snapshots."""
    cols = nitinol_columns(6, "linear")
    ens = ensemble(cols, 2, dict(enable_gravity=True))
    _, snaps = ens.step(40, 2e-5, impulse_amp=np.array([0.1, 0.2]), record="all", record_every=10)
    x, y = ens.beam_shapes(snaps, 0.25)
    assert x.shape == y.shape == (4, 2, 7)
    sol_y = ens.unpack_snapshots(snaps).cpu().numpy()            # [n_t, B, 2n]
    n_pos = ens.n
    for i in range(4):
        for b in range(2):
            pos = sol_y[i, b][n_pos + 1::3]
            want = np.concatenate([[0.0], [pos[j] if j < len(pos) else 0 for j in range(6)]])
            assert np.array_equal(y[i, b], want) and np.allclose(x[i, b], 0.25 * np.arange(7))
    _, yw = ens.beam_shapes(snaps, 0.25, as_reference=False)
    assert np.array_equal(yw[-1, 1, 1:], sol_y[-1, 1][1:n_pos:3]) and not np.array_equal(yw, y)


@pytest.mark.parametrize("n_e,kind,kw,held", [
    (64, "linear", dict(enable_gravity=True), False),                                   # one wave per beam, gravity
    (130, "mixed", dict(fluid_density=1000.0, enable_fluid=True), False),               # four waves, padding threads
    (256, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True), True),            # the headline shape + held input
    (10, "linear", dict(enable_gravity=True), False),                                   # packed: six beams per wave
    (16, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True), True),             # packed + held input
])
def test_workgroups_walking_over_beams_match_one_workgroup_per_beam(n_e, kind, kw, held, monkeypatch):
    """The lean stepper's workgroups walk over several beams with the solve tables kept in registers (the grid is
    capped at what is resident; CRB_LEAN_MAX_GROUPS caps it at 3 here so that 23 beams -- or 4 wave-groups of packed
    beams -- are split unevenly over the workgroups): bitwise equal to one workgroup per beam (CRB_LEAN_NO_WALK), incl.
    strided recording and whole-state snapshots from inside the walk."""
    kinds = ["nonlinear" if i % 3 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    cols = nitinol_columns(n_e, kinds)
    B = 23
    rng = np.random.default_rng(n_e)
    x0 = rng.normal(0.0, 1e-6, (B, 6 * n_e))
    amps = 0.05 * (1.0 + np.arange(B) / B)
    u = rng.normal(0.0, 1e-3, (B, 3 * n_e)) if held else None
    outs = []
    for walk in (True, False):
        if walk:
            monkeypatch.setenv("CRB_LEAN_MAX_GROUPS", "3")
            monkeypatch.delenv("CRB_LEAN_NO_WALK", raising=False)
        else:
            monkeypatch.delenv("CRB_LEAN_MAX_GROUPS", raising=False)
            monkeypatch.setenv("CRB_LEAN_NO_WALK", "1")
        ens = ensemble(cols, B, kw)
        ens.set_state(x0)
        _, tip = ens.step(30, 2e-5, impulse_amp=amps, held_force=u, record=(n_e, "w"), record_every=10)
        _, snaps = ens.step(20, 2e-5, impulse_amp=amps, held_force=u, record="all", record_every=10)
        outs.append((ens.unpack_state(), tip, snaps))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    ob = oracle_beam(cols, **kw)
    if not held:
        want, _ = ob.rk4_impulse_batch(x0, 2e-5, 50, amps)
        assert_blocks(outs[0][0].cpu().numpy(), want, ensemble(cols, 1, kw).free_index, 1e-9)


def test_implicit_stepper_integrates_config1_for_the_examples_full_second(golden):
    """The reference's CPU example end to end on the device: BASELINE config 1, 1 s of simulated time = 10000 implicit
    steps in ONE launch with the tip displacement sampled every 0.1 s on the device (the example's t_eval), against
    LSODA (rtol 1e-8) over the REFERENCE RHS (tests/golden/g8_lsoda.npz: lin10_grav_1s, tip w(1 s) = -0.41624141)."""
    z = golden["g8_lsoda"]
    ens = ensemble(beam_columns(z, "lin10_grav"), 2, force_kwargs(z, "lin10_grav"))
    tight, times = z["lin10_grav_1s/x_tight"], z["lin10_grav_1s/times"]
    t, tip = ens.step_implicit(10000, 1e-4, n_iter=2, impulse_amp=np.full(2, 0.1), record=(ens.n_elem, "w"), record_every=1000)
    assert abs(t - 1.0) < 1e-9 and tip.shape == (2, 10)
    tip = tip.cpu().numpy()
    for k in range(10):
        ref = tight[k][ens.n - 2]
        assert abs(tip[0, k] - ref) < 1e-6 + 1e-3 * abs(ref), (times[k], tip[0, k], ref)    # LSODA's default tolerance band
        assert abs(tip[0, k] - ref) < 1e-5 * max(abs(ref), 0.1)                            # (measured <= 7e-6 relative)
    assert np.array_equal(tip[0], tip[1])
    errs = block_errs(ens.unpack_state().cpu().numpy()[0], tight[-1], ens.free_index)
    assert errs["w"] < 5e-5 and errs["phi"] < 4e-3, errs


@pytest.mark.parametrize("n_e,kind,kw", [(64, "linear", dict(enable_gravity=True)),
                                          (100, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True)),
                                          (256, "mixed", dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True)),
                                          (10, "linear", dict(enable_gravity=True)),                                # six beams per wave
                                          (5, "mixed", dict(fluid_density=1000.0, enable_fluid=True)),              # twelve, 3 levels
                                          (30, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True))])   # two, 5 levels
def test_implicit_lean_kernel_equals_the_general_one(n_e, kind, kw, monkeypatch):
    """crb_implicit_lean_kernel (lean exchange structure, workgroups walking over beams: 7 beams on 3 workgroups here)
    against crb_implicit_kernel (general kernel machinery, CRB_DISABLE_LEAN_IMPLICIT) and against the oracle, with
    per-beam inputs and whole-state snapshots recorded from inside the walk."""
    half = n_e // 2
    kinds = (["linear"] * half + ["nonlinear"] * (n_e - half)) if kind == "mixed" else kind
    cols = nitinol_columns(n_e, kinds)
    B, h, steps = 7, 2e-4, 24
    rng = np.random.default_rng(n_e)
    x0 = rng.normal(0.0, 1e-6, (B, 6 * n_e))
    amps = 0.05 * (1.0 + np.arange(B) / B)
    u = rng.normal(0.0, 1e-3, (B, 3 * n_e))
    outs = []
    for lean in (True, False):
        if lean:
            monkeypatch.setenv("CRB_LEAN_MAX_GROUPS", "3")
            monkeypatch.delenv("CRB_DISABLE_LEAN_IMPLICIT", raising=False)
        else:
            monkeypatch.setenv("CRB_DISABLE_LEAN_IMPLICIT", "1")
        ens = ensemble(cols, B, kw)
        ens.set_state(x0)
        _, snaps = ens.step_implicit(steps, h, n_iter=2, impulse_amp=amps, impulse_duration=8 * h, held_force=u,
                                     record="all", record_every=8)
        outs.append((ens.unpack_state().cpu().numpy(), ens.unpack_snapshots(snaps).cpu().numpy()))
    fi = ensemble(cols, 1, kw).free_index
    assert_blocks(outs[0][0], outs[1][0], fi, 1e-8)
    for k in range(3):
        assert_blocks(outs[0][1][k], outs[1][1][k], fi, 1e-8, what=k)
    assert np.array_equal(outs[0][1][-1], outs[0][0])       # the last snapshot is the terminal state
    ob = oracle_beam(cols, **kw)
    for b in (0, B - 1):
        want = ob.implicit(x0[b], h, steps, n_iter=2, amp=amps[b], duration=8 * h, u_held=u[b])
        assert_blocks(outs[0][0][b], want, fi, 1e-7, what=b)


def test_examples_solve_ivp_call_for_an_ensemble(golden):
    """BeamEnsemble.solve_ivp: the examples' ``solve_ivp(..., method="LSODA", t_eval=np.arange(0, T, DT))``
    (examples/example_utilities.py:153-159) for a whole ensemble.  The stiff methods map to the implicit stepper (tip
    series against LSODA-tight over the REFERENCE RHS: config 1, tests/golden/g8_lsoda.npz), RK45 to the scipy-exact
    adaptive kernel, RK4 to the fused explicit stepper; the three agree on the tip displacement where all are accurate."""
    z = golden["g8_lsoda"]
    cols, kw = beam_columns(z, "lin10_grav"), force_kwargs(z, "lin10_grav")
    tight = z["lin10_grav_1s/x_tight"]
    ens = ensemble(cols, 2, kw)
    n = ens.n
    t_eval = np.arange(0.0, 0.3005, 0.001)
    sol = ens.solve_ivp((0.0, 0.3005), t_eval, method="LSODA", substeps=10, impulse_amp=np.full(2, 0.1))
    assert sol.t.shape == (301,) and tuple(sol.y.shape) == (2, 2 * n, 301) and sol.success
    y = sol.y.cpu().numpy()
    assert np.all(y[:, :, 0] == 0.0)                                   # first column: the initial state
    for k, t in ((100, 0.1), (200, 0.2), (300, 0.3)):
        ref = tight[int(round(t / 0.1)) - 1][n - 2]
        assert abs(y[0, n - 2, k] - ref) < 1e-6 + 1e-3 * abs(ref)      # the example's own tolerance band
    assert abs(ens.time - 0.3) < 1e-9
    # RK45 and RK4 over the first 10 ms (explicit methods: short horizon): they agree with each other closely, and the
    # implicit series (h = 1e-4 s) with them within the examples' absolute tolerance (1e-6 m; the first milliseconds of
    # an impulse response are where the unresolved modes show most: measured 2e-6)
    series = {}
    for method, extra in (("RK45", dict(rtol=1e-8, atol=1e-11)), ("RK4", dict(substeps=50))):
        e2 = ensemble(cols, 2, kw)
        s2 = e2.solve_ivp((0.0, 0.0105), np.arange(0.0, 0.0105, 0.001), method=method, impulse_amp=np.full(2, 0.1), **extra)
        y2 = s2.y.cpu().numpy()
        assert y2.shape == (2, 2 * n, 11)
        series[method] = y2[0, n - 2, :]
        assert np.max(np.abs(y2[0, n - 2, 1:] - y[0, n - 2, 1:11])) < 3e-6, method
    assert np.allclose(series["RK45"], series["RK4"], rtol=1e-5, atol=1e-12)      # (RK4 at dt = 2e-5: 2e-6 of its own)
    with pytest.raises(ValueError, match="uniform"):
        ens.solve_ivp((0.0, 1.0), np.array([0.0, 0.1, 0.3]))
    with pytest.raises(ValueError, match="unknown method"):
        ens.solve_ivp((0.0, 1.0), np.arange(0, 1, 0.1), method="Euler")


@pytest.mark.parametrize("controller", ["device", "host"])
@pytest.mark.parametrize("name", ["lin10_grav", "lin6_fluid", "mixed6_fluid"])
def test_solve_ivp_controls_the_step_by_the_tolerances(golden, name, controller):
    """``solve_ivp(method="LSODA")`` without ``substeps``: the step size follows rtol / atol (step doubling on the implicit
    midpoint rule), as the reference's call means it (examples/example_utilities.py:153-159, default tolerances).  The
    yardstick is the reference's own integrator: golden G8 holds scipy LSODA over the REFERENCE RHS at tight tolerances
    (1e-10 / 1e-12) and at the defaults.  In units of the default tolerance band (1e-6 + 1e-3 |y|) the controlled run has
    to land every position DOF inside the band and be, over the WHOLE state (velocities of the stiff modes included),
    about as close to the tight run as default-tolerance LSODA is (measured: RMS 7 ... 49 against LSODA's 17 ... 55, largest
    entry 34 ... 340 against 70 ... 190) -- which the uncontrolled h = 1e-4 run is not (RMS 350 ... 690).  A tolerance
    1000 times tighter brings the error down by about as much."""
    z = golden["g8_lsoda"]
    cols, kw = beam_columns(z, name), force_kwargs(z, name)
    times, tight, dflt = z[name + "/times"], z[name + "/x_tight"], z[name + "/x_default_tol"]
    amp, T = float(z[name + "/amp"]), float(z[name + "/times"][-1])
    t_eval = np.arange(0.0, T + 0.0005, 0.001)

    def scaled_errors(**tol):
        ens = ensemble(cols, 2, kw)
        sol = ens.solve_ivp((0.0, T + 0.0005), t_eval, method="LSODA", impulse_amp=np.full(2, amp), controller=controller, **tol)
        y = sol.y.cpu().numpy()
        assert np.array_equal(y[0], y[1]) and abs(ens.time - t_eval[-1]) < 1e-9 and len(sol.substeps) == t_eval.size - 1
        assert sol.controller == controller
        out = []
        for ti, t in enumerate(times):
            band = 1e-6 + 1e-3 * np.abs(tight[ti])
            out.append((np.abs(y[0][:, int(round(t / 0.001))] - tight[ti]) / band, np.abs(dflt[ti] - tight[ti]) / band))
        return ens.n, out, sol.substeps

    n, errs, used = scaled_errors()
    assert min(used) >= 40        # (h <= 2.5e-5 s: the controller does resolve the stiff modes, as LSODA does)
    for ours, lsoda in errs:
        assert ours[:n].max() < 1.0, (name, ours[:n].max())
        assert np.sqrt((ours ** 2).mean()) < 2.0 * np.sqrt((lsoda ** 2).mean()), name
        assert ours.max() < 4.0 * lsoda.max(), (name, ours.max(), lsoda.max())
    _, errs_t, used_t = scaled_errors(rtol=1e-6, atol=1e-9)
    assert max(used_t) > 8 * max(used)
    for (ours_t, _), (ours, _) in zip(errs_t, errs):
        assert ours_t.max() < 2e-2 * max(ours.max(), 50.0), (name, ours_t.max(), ours.max())
    ens = ensemble(cols, 2, kw)
    with pytest.raises(ValueError, match="substeps"):
        ens.solve_ivp((0.0, 0.01), t_eval[:11], method="LSODA", substeps="fine")
    with pytest.raises(ValueError, match="control"):
        ens.solve_ivp((0.0, 0.01), t_eval[:11], method="LSODA", control="some")
    with pytest.raises(ValueError, match="integer"):
        ens.solve_ivp((0.0, 0.01), t_eval[:11], method="RK4", substeps="auto")
    with pytest.raises(RuntimeError, match="tolerances ask for more"):
        ens.solve_ivp((0.0, 0.002), t_eval[:3], method="LSODA", rtol=1e-15, atol=1e-18, impulse_amp=np.full(2, amp), controller=controller)
    with pytest.raises(ValueError, match="controller"):
        ens.solve_ivp((0.0, 0.002), t_eval[:3], method="LSODA", controller="gpu")


@pytest.mark.parametrize("controller", ["device", "host"])
@pytest.mark.parametrize("name,T", [("lqr6", 0.03), ("lqr24", 0.012)])
def test_solve_ivp_closed_loop_follows_the_tolerances(golden, name, T, controller):
    """``solve_ivp(..., gain=K)``: the closed loop of examples/lqr_control.py:94-125 (u = K (0 - x) + impulse inside the RHS,
    ``solve_ivp(method="LSODA", rtol=1e-8, atol=1e-10)``) for an ensemble -- RK4 with the feedback in every stage, its step
    chosen by the same step-doubling controller, the input's switch-off a breakpoint of the integration.  Checked against
    scipy's LSODA at the example's tolerances over the ORACLE's closed-loop RHS (gain of golden G6, 6 and 24 elements):
    the whole trajectory agrees to 0.1 of the default tolerance band and to within the global error of the tight LSODA run
    itself (measured: 250 of ITS tolerance units in the velocities, 0.05 in the positions)."""
    from scipy.integrate import solve_ivp

    z = golden["g6_lqr_loop"]
    cols, kw = beam_columns(z, name), force_kwargs(z, name)
    K, amp = z[f"{name}/gain"], float(z[f"{name}/amp"])
    ob = oracle_beam(cols, **kw)
    n = ob.n

    def closed_loop(t, x):
        u = -K @ x
        if t < 0.01:
            u[n - 2] += amp
        return ob.rhs(x, u)

    t_eval = np.arange(0.0, T + 0.0005, 0.001)
    ref = solve_ivp(closed_loop, (0.0, t_eval[-1]), np.zeros(2 * n), method="LSODA", t_eval=t_eval, rtol=1e-8, atol=1e-10)
    assert ref.success
    B = 3
    ens = ensemble(cols, B, kw)
    sol = ens.solve_ivp((0.0, t_eval[-1]), t_eval, method="LSODA", rtol=1e-8, atol=1e-10, impulse_amp=np.full(B, amp), gain=K,
                        controller=controller)
    assert sol.controller == controller      # ("device": the whole closed-loop span is ONE launch, crb_solve_controlled)
    y = sol.y.cpu().numpy()
    assert y.shape == (B, 2 * n, t_eval.size) and np.array_equal(y[0], y[B - 1]) and len(sol.substeps) == t_eval.size - 1
    assert np.max(np.abs(y[0] - ref.y) / (1e-6 + 1e-3 * np.abs(ref.y))) < 0.1
    tight = np.abs(y[0] - ref.y) / (1e-10 + 1e-8 * np.abs(ref.y))
    assert tight[:n].max() < 1.0 and tight[n:].max() < 1e3, (tight[:n].max(), tight[n:].max())
    # integer substeps: no control, the plain stage-split rollout interval by interval
    e2, e3 = ensemble(cols, B, kw), ensemble(cols, B, kw)
    s2 = e2.solve_ivp((0.0, 0.003), t_eval[:4], method="RK4", substeps=200, impulse_amp=np.full(B, amp), gain=K)
    e3.step_feedback(600, 0.001 / 200, K, impulse_amp=np.full(B, amp))
    assert np.allclose(s2.y[:, :, -1].cpu().numpy(), e3.unpack_state().cpu().numpy(), rtol=1e-12, atol=1e-15)
    with pytest.raises(ValueError, match="closed loop"):
        e2.solve_ivp((0.0, 0.003), t_eval[:4], method="RK45", gain=K)


def _replay_controlled(ob, x0, t0, dt_eval, used, amp, t_switch, gain=None, n_iter=1):
    """The oracle's fixed-step schemes driven with the step counts the in-kernel controller ACCEPTED: per t_eval interval
    ``used[k]`` steps of dt_eval / used[k] (the implicit scheme restarts its iterate at every interval, as the kernel does)."""
    x, out = np.array(x0, dtype=np.float64), []
    for k, m in enumerate(used):
        t_a = t0 + k * dt_eval
        if gain is None:
            x = ob.implicit(x, dt_eval / m, int(m), n_iter=n_iter, amp=amp, duration=t_switch, t0=t_a)
        else:
            x = ob.rk4_feedback(x, dt_eval / m, int(m), gain, amp=amp, duration=t_switch, t0=t_a)
        out.append(x.copy())
    return np.array(out)


@pytest.mark.parametrize("n_e,kind,kw,lean", [
    (1, "linear", dict(), True),                                                        # one slot: no reduction level (general RHS)
    (3, "nonlinear", dict(enable_gravity=True), True),                                  # 2 levels, lean + gravity
    (10, "linear", dict(enable_gravity=True), True),                                    # BASELINE config 1: 4 levels
    (10, "linear", dict(enable_gravity=True), False),                                   # ... through the general RHS
    (6, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True), False),             # nonlinear elements, general RHS
    (27, "linear", dict(fluid_density=1000.0, enable_fluid=True), True),                # 5 levels
    (64, "linear", dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True), True),      # 6 levels, a full wave
    (100, "linear", dict(enable_gravity=True), True),                                   # two waves per beam, 7 levels
    (100, "linear", dict(fluid_density=1000.0, enable_fluid=True), False),              # ... general RHS, 7 levels
    (200, "linear", dict(fluid_density=1000.0, enable_fluid=True), True),               # four waves per beam, 8 levels
])
def test_controlled_implicit_kernel_takes_the_oracles_steps(n_e, kind, kw, lean, monkeypatch):
    """crb_solve_controlled, implicit scheme: whatever step counts the in-kernel controller accepts, the states it records are
    the ORACLE's implicit-midpoint states for exactly those step counts (every instance of crb_controlled_kernel: the lean
    iteration with 1 / 2 / 4 waves per beam and 1 .. 8 reduction levels, with and without gravity, and the general RHS) --
    so the controller only ever chooses among the oracle's own trajectories.  Loose tolerances and a chosen first rate make
    the choice short: 8 fine steps in the first interval, then fewer as the estimate allows."""
    if not lean:
        monkeypatch.setenv("CRB_DISABLE_LEAN_IMPLICIT", "1")
    kinds = ["nonlinear" if i % 3 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    cols = nitinol_columns(n_e, kinds)
    B, dt_eval, n_int = 3, 1e-3, 4
    ens = ensemble(cols, B, kw)
    ob = oracle_beam(cols, **kw)
    rng = np.random.default_rng(n_e)
    x0 = 1e-4 * rng.standard_normal((B, 2 * ens.n))
    ens.set_state(x0)
    amps = np.array([0.05, 0.1, 0.0])
    snaps, stats, used = ens.solve_controlled(n_int, dt_eval, rtol=1e-2, atol=1e-5, impulse_amp=amps, impulse_duration=2e-3,
                                              first_rate=4.0 / dt_eval, t0=0.0)
    y = ens.unpack_snapshots(snaps).cpu().numpy()            # [n_int, B, 2n]
    assert np.all(stats[:, 2] == 0) and np.array_equal(stats[:, 0], used.sum(axis=1)) and used.min() >= 2
    for b in range(B):
        for k in range(n_int):       # interval by interval from the kernel's own recorded state (roundoff does not pile up)
            start = x0[b] if k == 0 else y[k - 1, b]
            want = _replay_controlled(ob, start, k * dt_eval, dt_eval, used[b, k:k + 1], amps[b], 2e-3)
            assert_blocks(y[k, b], want[0], ens.free_index, 2e-9, what=(b, k, used[b]))     # (up to 1024 steps per interval)
    assert np.array_equal(ens.unpack_state().cpu().numpy(), y[-1]) and abs(ens.time - n_int * dt_eval) < 1e-15


@pytest.mark.parametrize("lean", [True, False])
@pytest.mark.parametrize("name", ["lqr6", "lqr24"])
def test_controlled_closed_loop_kernel_takes_the_oracles_steps(golden, name, lean, monkeypatch):
    """crb_solve_controlled, closed-loop RK4 (gain in LDS, u = K (r - x) in every stage): the recorded states are the oracle's
    ``rk4_feedback`` states for the step counts the controller accepted.  With loose tolerances the controller keeps halving
    the rate until the COARSE solution crosses RK4's stability limit (h > ~8.6e-6 s for these loops): its estimate is then
    not finite, the piece is repeated with twice the steps, and the accepted fine solution is a stable one again."""
    if not lean:        # (the general right-hand side instead of the one-wave lean one)
        monkeypatch.setenv("CRB_DISABLE_LEAN_FEEDBACK", "1")
    z = golden["g6_lqr_loop"]
    cols, kw = beam_columns(z, name), force_kwargs(z, name)
    K, amp = z[f"{name}/gain"], float(z[f"{name}/amp"])
    ob = oracle_beam(cols, **kw)
    B, dt_eval, n_int = 2, 1e-3, 5
    ens = ensemble(cols, B, kw)
    amps = np.array([amp, 0.5 * amp])
    snaps, stats, used = ens.solve_controlled(n_int, dt_eval, rtol=1e-2, atol=1e-5, gain=K, impulse_amp=amps, impulse_duration=0.0025,
                                              t0=0.0)
    y = ens.unpack_snapshots(snaps).cpu().numpy()
    assert np.all(stats[:, 2] == 0) and np.all(np.isfinite(y))
    assert used.min() >= 128 and stats[:, 1].max() >= 1, (used, stats)      # (64 steps per ms is unstable; a doubling happened)
    for b in range(B):
        x, t = np.zeros(2 * ob.n), 0.0
        for k in range(n_int):
            # (the interval the impulse ends in is cut at 2.5 ms: its two halves take used / 2 steps each only when the
            #  controller chose the same rung for both -- replay whole intervals outside it)
            if k == 2:
                x = y[k, b].copy()
                continue
            m = int(used[b, k])
            x = ob.rk4_feedback(x, dt_eval / m, m, K, amp=amps[b], duration=0.0025, t0=k * dt_eval)
            assert_blocks(y[k, b], x, ens.free_index, 1e-9, what=(b, k, m))
            x = y[k, b].copy()


@pytest.mark.parametrize("n_e,kw", [(3, dict(enable_gravity=True)), (10, dict(enable_gravity=True)),
                                     (10, dict(fluid_density=1000.0, enable_fluid=True)), (20, dict()), (31, dict(enable_gravity=True))])
def test_packed_controlled_kernel_shares_a_step_sequence_per_wave(n_e, kw):
    """crb_solve_controlled with ``per_wave`` (``controller="device-packed"``): G = 64 / slots short beams per wave, ONE step
    sequence per wave -- the worst of its beams decides.  Every beam's recorded states are still its own oracle's
    implicit-midpoint states for the accepted step counts; the beams of a wave report the same counts; a beam never takes
    FEWER steps than it does when it decides alone; and the last, partly filled wave is handled."""
    cols = nitinol_columns(n_e, "linear")
    ens = ensemble(cols, 1, kw)
    G = int(ens.plan.layout.beams_per_group)
    assert G == 64 // int(ens.plan.layout.n_slots) and G >= 2
    B = 2 * G + 1                                        # two full waves and one beam in a third
    ens = ensemble(cols, B, kw)
    ob = oracle_beam(cols, **kw)
    rng = np.random.default_rng(n_e)
    x0 = 1e-5 * rng.standard_normal((B, 2 * ens.n))
    amps = np.where(np.arange(B) % G == 1, 0.2, 1e-3)    # one hard-hit beam per wave
    amps[-1] = 0.0
    dt_eval, n_int = 1e-3, 4
    ens.set_state(x0)
    snaps, stats, used = ens.solve_controlled(n_int, dt_eval, rtol=1e-3, atol=1e-6, impulse_amp=amps, impulse_duration=2e-3, t0=0.0,
                                              per_wave=True)
    y = ens.unpack_snapshots(snaps).cpu().numpy()
    assert np.all(stats[:, 2] == 0)
    for w in range(3):
        rows = used[w * G:min((w + 1) * G, B)]
        assert np.all(rows == rows[0]), (w, rows)
    for b in (0, 1, G - 1, G, G + 1, 2 * G):
        for k in range(n_int):
            start = x0[b] if k == 0 else y[k - 1, b]
            m = int(used[b, k])
            want = ob.implicit(start, dt_eval / m, m, n_iter=1, amp=amps[b], duration=2e-3, t0=k * dt_eval)
            assert_blocks(y[k, b], want, ens.free_index, 2e-9, what=(b, k, m))
    alone = ensemble(cols, B, kw)
    alone.set_state(x0)
    _, _, used_alone = alone.solve_controlled(n_int, dt_eval, rtol=1e-3, atol=1e-6, impulse_amp=amps, impulse_duration=2e-3, t0=0.0)
    assert np.all(used.sum(axis=1) >= used_alone.sum(axis=1)) and used.sum() > used_alone.sum()
    # solve_ivp takes the packed form when asked, and refuses it where it does not exist
    e2 = ensemble(cols, B, kw)
    sol = e2.solve_ivp((0.0, 0.0035), np.arange(0.0, 0.0035, 0.001), method="LSODA", impulse_amp=amps, controller="device-packed")
    assert sol.controller == "device-packed" and sol.substeps_per_beam.shape == (B, 3)
    big = ensemble(nitinol_columns(70, "linear"), 2, kw)
    with pytest.raises(ValueError, match="device-packed"):
        big.solve_ivp((0.0, 0.0035), np.arange(0.0, 0.0035, 0.001), method="LSODA", controller="device-packed")


def test_controlled_solver_records_one_dof_instead_of_snapshots(golden):
    """crb_solve_controlled's series output (``record=(node, param)``): the tip displacement on the t_eval grid -- what the examples
    read from sol.y (lqr_control.py:168, example_utilities.py:173-205) -- without the whole-state snapshots: equal, bit for bit,
    to that DOF of the snapshots of the same run; the open loop and the closed loop, displacement and velocity."""
    z = golden["g6_lqr_loop"]
    cols, kw = beam_columns(z, "lqr6"), force_kwargs(z, "lqr6")
    K = z["lqr6/gain"]
    B, n_int = 5, 6
    amps = 0.1 * (1.0 + np.arange(B))
    for gain, tol in ((None, dict(rtol=1e-3, atol=1e-6)), (K, dict(rtol=1e-6, atol=1e-9))):
        for param in ("w", "dw_dt"):
            a, b = ensemble(cols, B, kw), ensemble(cols, B, kw)
            snaps, st_a, used_a = a.solve_controlled(n_int, 1e-3, gain=gain, impulse_amp=amps, impulse_duration=2.5e-3, t0=0.0, **tol)
            series, st_b, used_b = b.solve_controlled(n_int, 1e-3, gain=gain, impulse_amp=amps, impulse_duration=2.5e-3, t0=0.0,
                                                      record=(a.n_elem, param), **tol)
            assert tuple(series.shape) == (B, n_int) and np.array_equal(used_a, used_b)
            y = a.unpack_snapshots(snaps).cpu().numpy()          # [n_int, B, 2n]
            idx = a.reduced_index(a.n_elem, "w") + (a.n if param == "dw_dt" else 0)
            assert np.array_equal(series.cpu().numpy(), y[:, :, idx].T), (gain is not None, param)
            assert np.array_equal(a.state.cpu().numpy(), b.state.cpu().numpy())
    with pytest.raises(Exception, match="series"):
        ensemble(cols, 2, kw).solve_controlled(2, 1e-3, record=(99, "w"))


def test_controlled_steppers_give_every_beam_its_own_step_sequence(golden):
    """Per-beam control: in ONE launch a beam that is hit hard takes more steps than one that is barely moved or left alone,
    and each beam's trajectory is bit-identical to the run of that beam as an ensemble of one -- the beams are as independent
    as the reference's separate solve_ivp calls (examples/beam_comparison_fluid.py:82-83).  Also an impulse that ends INSIDE
    a t_eval interval (the interval is cut there) and the position-only norm."""
    z = golden["g8_lsoda"]
    cols, kw = beam_columns(z, "lin6_fluid"), force_kwargs(z, "lin6_fluid")      # (no gravity: an unforced beam stays at rest)
    amps = np.array([0.1, 1e-4, 0.0, 0.3])
    t_eval = np.arange(0.0, 0.0305, 0.001)
    ens = ensemble(cols, 4, kw)
    sol = ens.solve_ivp((0.0, t_eval[-1]), t_eval, method="LSODA", impulse_amp=amps, impulse_duration=0.0104, controller="device")
    per_beam = sol.substeps_per_beam
    assert per_beam.shape == (4, t_eval.size - 1)
    total = per_beam.sum(axis=1)
    assert total[0] > total[1] > total[2] and total[2] < 0.1 * total[0], total       # (the beam at rest: 2 steps per piece in the end)
    assert np.all(per_beam[2, 12:] == 2)
    assert sol.substeps == [int(v) for v in per_beam.max(axis=0)]
    y = sol.y.cpu().numpy()
    for b in range(4):
        one = ensemble(cols, 1, kw)
        s1 = one.solve_ivp((0.0, t_eval[-1]), t_eval, method="LSODA", impulse_amp=amps[b:b + 1], impulse_duration=0.0104,
                           controller="device")
        assert np.array_equal(s1.y.cpu().numpy()[0], y[b]) and np.array_equal(s1.substeps_per_beam[0], per_beam[b]), b
    # the cut interval [0.010, 0.011]: the host-loop controller (worst beam decides) agrees within the tolerances
    host = ensemble(cols, 4, kw)
    sh = host.solve_ivp((0.0, t_eval[-1]), t_eval, method="LSODA", impulse_amp=amps, impulse_duration=0.0104, controller="host")
    yh = sh.y.cpu().numpy()
    n = ens.n
    assert np.max(np.abs(y[:, :n] - yh[:, :n]) / (1e-6 + 1e-3 * np.abs(yh[:, :n]))) < 1.0
    # positions-only norm: fewer steps; the local control of the positions alone lets their GLOBAL error grow to ~20 default
    # bands over these 30 ms (measured 18.3), which is why the whole-state norm is the default
    pos = ensemble(cols, 4, kw)
    sp = pos.solve_ivp((0.0, t_eval[-1]), t_eval, method="LSODA", impulse_amp=amps, impulse_duration=0.0104, controller="device",
                       control="positions")
    assert sp.substeps_per_beam.sum() < per_beam.sum()
    assert np.max(np.abs(sp.y.cpu().numpy()[:, :n] - y[:, :n]) / (1e-6 + 1e-3 * np.abs(y[:, :n]))) < 40.0


def test_config1_one_second_at_default_tolerances_in_one_launch(golden):
    """BASELINE config 1 as the example runs it (examples/example_utilities.py:153-159: 1 s, LSODA, scipy's default
    tolerances, t_eval every millisecond) in ONE launch with the controller in the kernel.  Yardstick: golden G8's
    default-tolerance LSODA run over the REFERENCE RHS, both measured against the tight run (1e-10 / 1e-12) in units of the
    default band 1e-6 + 1e-3 |y|, per DOF block and over the ten recorded times: every position block stays inside the band
    (LSODA: 0.00 .. 0.00; ours <= 0.3), the velocity blocks are as close to the tight run as default LSODA's own worst block
    is (LSODA: du up to 224, dphi up to 9; ours: du <= 1, dw / dphi up to 311 / 2161 at t = 0.6 s, where the host-loop
    controller has 500 / 3657) -- velocity content of modes above 1 / h keeps its amplitude, not its phase."""
    z = golden["g8_lsoda"]
    cols, kw = beam_columns(z, "lin10_grav"), force_kwargs(z, "lin10_grav")
    times, tight, dflt = z["lin10_grav_1s/times"], z["lin10_grav_1s/x_tight"], z["lin10_grav_1s/x_default_tol"]
    t_eval = np.arange(0.0, 1.0005, 0.001)
    ens = ensemble(cols, 2, kw)
    sol = ens.solve_ivp((0.0, 1.0005), t_eval, method="LSODA", impulse_amp=np.full(2, 0.1))
    assert sol.controller == "device" and len(sol.substeps) == 1000
    y = sol.y.cpu().numpy()
    n = ens.n
    assert np.array_equal(y[0], y[1])
    ours_rms, lsoda_rms = [], []
    for ti, t in enumerate(times):
        band = 1e-6 + 1e-3 * np.abs(tight[ti])
        ours = np.abs(y[0][:, int(round(t / 0.001))] - tight[ti]) / band
        lsoda = np.abs(dflt[ti] - tight[ti]) / band
        assert ours[:n].max() < 1.0, (t, ours[:n].max())                 # every position DOF inside the default band
        assert ours[n::3].max() < 2.0, (t, ours[n::3].max())             # axial velocities: far inside LSODA's own error (16 .. 224)
        ours_rms.append(np.sqrt((ours ** 2).mean()))
        lsoda_rms.append(np.sqrt((lsoda ** 2).mean()))
    # over the whole state the median recorded time is as close to the tight run as default LSODA is; the worst one
    # (t = 0.6 s: a velocity-phase excursion of the bending modes) within the host-loop controller's own figure
    assert np.median(ours_rms) < 1.5 * np.median(lsoda_rms), (np.median(ours_rms), np.median(lsoda_rms))
    assert max(ours_rms) < 600.0
    tip = y[0, n - 2, -1]
    assert abs(tip / tight[-1][n - 2] - 1.0) < 1e-4                       # tip w(1 s) = -0.41624141


@pytest.mark.parametrize("seed", range(int(os.environ.get("CRB_FUZZ_MIXED_N", "12"))))   # CRB_FUZZ_MIXED_N=200 for a long hunt
def test_randomised_mixed_ensembles_against_oracle(seed):
    """Differential test over random HETEROGENEOUS ensembles (f-3): every beam draws its own element count, element
    kinds, boundary-condition column, material scaling and ForceParams (drag / gravity on or off, fluid density, gravity
    vector); RK4 rollout with the impulse at each beam's own tip or a held input, the RHS, and the implicit stepper,
    each beam against its own oracle in its own reduced ordering."""
    from continuum_robot.batched import BeamEnsemble
    from continuum_robot.models.force_params import ForceParams

    rng = np.random.default_rng(5000 + seed)
    B = int(rng.integers(2, 8))
    big = rng.random() < 0.4
    sizes = rng.integers(40, 200, B) if big else rng.integers(1, 30, B)
    beams, fps = [], []
    for b in range(B):
        n = int(sizes[b])
        kinds = [("nonlinear" if rng.random() < 0.5 else "linear") for _ in range(n)]
        bcs = []
        for i in range(n):
            r = rng.random()
            bcs.append("FIXED" if (i == 0 and r < 0.6) or r > 0.99 else ("PINNED" if r > 0.96 else "NONE"))
        cols = nitinol_columns(n, kinds, bcs)
        cols = _scaled(cols, rng)
        beams.append(cols)
        fps.append(ForceParams(fluid_density=float(rng.uniform(500, 1500)), enable_fluid_effects=bool(rng.random() < 0.5),
                               enable_gravity_effects=bool(rng.random() < 0.5),
                               gravity_vector=[float(rng.uniform(-3, 3)), float(rng.uniform(-12, -6)), 0.0]))
    ens = BeamEnsemble.from_dataframes(beams, force_params=fps)
    obs = [oracle_beam(beams[b], fluid_density=fps[b].fluid_density, enable_fluid=fps[b].enable_fluid_effects,
                       enable_gravity=fps[b].enable_gravity_effects, gravity=fps[b].get_gravity_vector()) for b in range(B)]
    assert [ob.n for ob in obs] == list(ens.n_per_beam)
    x0 = [rng.normal(0.0, 1e-6, 2 * ob.n) for ob in obs]
    ens.set_state(ens.pad_states(x0))
    xd = ens.rhs().cpu().numpy()
    for b in range(B):
        assert_blocks(ens.beam_state(b, xd), obs[b].rhs(x0[b]), obs[b].red2full(), 1e-9, what=("rhs", seed, b))
    steps = int(rng.integers(3, 30))
    if rng.random() < 0.5:
        amps = rng.uniform(0.01, 0.05, B)
        ens.step(steps, 2e-5, impulse_amp=amps, impulse_index=-2, t0=0.0)
        want = [obs[b].rk4_impulse(x0[b], 2e-5, steps, amps[b]) for b in range(B)]
    else:
        u = [rng.normal(0.0, 1e-3, ob.n) for ob in obs]
        upad = np.zeros((B, ens.n))
        for b in range(B):
            upad[b, :obs[b].n] = u[b]
        ens.step(steps, 2e-5, held_force=upad, t0=0.0)
        want = [obs[b].rk4_held(x0[b], 2e-5, steps, u[b]) for b in range(B)]
    for b in range(B):
        assert np.isfinite(want[b]).all()
        assert_blocks(ens.beam_state(b), want[b], obs[b].red2full(), 1e-8, what=("rk4", seed, b, int(sizes[b])))
    # implicit stepper on the same ensemble
    ens.set_state(ens.pad_states(x0))
    amps = rng.uniform(0.01, 0.05, B)
    ens.step_implicit(10, 1e-4, n_iter=3, impulse_amp=amps, impulse_index=-2, impulse_duration=5e-4, t0=0.0)
    for b in range(B):
        w = obs[b].implicit(x0[b], 1e-4, 10, n_iter=3, amp=amps[b], duration=5e-4)
        assert np.isfinite(w).all()
        assert_blocks(ens.beam_state(b), w, obs[b].red2full(), 1e-6, what=("implicit", seed, b, int(sizes[b])))


@pytest.mark.parametrize("seed", range(6))
def test_controlled_implicit_kernel_on_heterogeneous_ensembles(seed):
    """crb_solve_controlled on ensembles whose beams differ in length, boundary conditions, material and ForceParams (per-beam
    table ladders, per-beam error-norm sizes, each beam forced at its own tip): every beam's recorded states are its OWN
    oracle's implicit-midpoint states for the step counts the controller accepted for THAT beam, interval by interval."""
    from continuum_robot.batched import BeamEnsemble
    from continuum_robot.models.force_params import ForceParams

    rng = np.random.default_rng(7000 + seed)
    B = int(rng.integers(2, 7))
    sizes = rng.integers(40, 130, B) if seed % 3 == 2 else rng.integers(2, 30, B)
    beams, fps = [], []
    for b in range(B):
        n = int(sizes[b])
        bcs = ["FIXED" if i == 0 else ("PINNED" if rng.random() > 0.97 else "NONE") for i in range(n)]
        beams.append(_scaled(nitinol_columns(n, "linear", bcs), rng))
        fps.append(ForceParams(fluid_density=float(rng.uniform(500, 1500)), enable_fluid_effects=bool(rng.random() < 0.5),
                               enable_gravity_effects=bool(rng.random() < 0.5),
                               gravity_vector=[float(rng.uniform(-3, 3)), float(rng.uniform(-12, -6)), 0.0]))
    ens = BeamEnsemble.from_dataframes(beams, force_params=fps)
    obs = [oracle_beam(beams[b], fluid_density=fps[b].fluid_density, enable_fluid=fps[b].enable_fluid_effects,
                       enable_gravity=fps[b].enable_gravity_effects, gravity=fps[b].get_gravity_vector()) for b in range(B)]
    x0 = [rng.normal(0.0, 1e-5, 2 * ob.n) for ob in obs]
    ens.set_state(ens.pad_states(x0))
    amps = rng.uniform(0.01, 0.1, B)
    dt_eval, n_int = 1e-3, 3
    snaps, stats, used = ens.solve_controlled(n_int, dt_eval, rtol=1e-2, atol=1e-6, impulse_amp=amps, impulse_duration=1.5e-3,
                                              first_rate=4.0 / dt_eval, t0=0.0)
    assert np.all(stats[:, 2] == 0) and used.min() >= 2
    y = ens.unpack_snapshots(snaps).cpu().numpy()            # [n_int, B, 2 n_max] padded
    for b in range(B):
        start = x0[b]
        for k in range(n_int):
            if k == 1:      # (the interval the impulse ends in is cut in two pieces with their own rungs: continue from the record)
                start = ens.beam_state(b, y[k])
                continue
            m = int(used[b, k])
            want = obs[b].implicit(start, dt_eval / m, m, n_iter=1, amp=amps[b], duration=1.5e-3, t0=k * dt_eval)
            assert_blocks(ens.beam_state(b, y[k]), want, obs[b].red2full(), 5e-9, what=(seed, b, k, m, int(sizes[b])))
            start = ens.beam_state(b, y[k])


@pytest.mark.parametrize("seed", range(int(os.environ.get("CRB_FUZZ_CTRL_N", "16"))))   # CRB_FUZZ_CTRL_N=300 for a long hunt
def test_randomised_controlled_runs_replay_on_the_oracle(seed):
    """Differential test of crb_solve_controlled over random shapes (1 .. 200 elements: every instance family of the kernel),
    element kinds, boundary conditions, forces, tolerances, first rates, impulse timing (on the grid, inside an interval, or
    absent), a held force, the position-only norm and the packed form: whatever step counts the controller accepts, each
    recorded interval is the oracle's implicit-midpoint result for that count from the recorded start of the interval."""
    rng = np.random.default_rng(9000 + seed)
    n_e = int(rng.choice([1, 2, 3, 5, 8, 10, 13, 16, 21, 31, 33, 47, 64, 65, 100, 128, 129, 200]))
    kind = "nonlinear" if (n_e <= 5 and rng.random() < 0.5) else "linear"
    bcs = ["FIXED"] + [("PINNED" if rng.random() > 0.97 else "NONE") for _ in range(n_e - 1)]
    cols = _scaled(nitinol_columns(n_e, kind, bcs), rng)
    kw = dict()
    if rng.random() < 0.5:
        kw.update(enable_gravity=True)
    if rng.random() < 0.5:
        kw.update(fluid_density=float(rng.uniform(500, 1500)), enable_fluid=True)
    B = int(rng.integers(1, 12))
    ens = ensemble(cols, B, kw)
    ob = oracle_beam(cols, **kw)
    x0 = float(rng.choice([0.0, 1e-6, 1e-4])) * rng.standard_normal((B, 2 * ens.n))
    amps = rng.uniform(0.0, 0.2, B)
    dt_eval = float(rng.choice([5e-4, 1e-3, 2e-3]))
    n_int = int(rng.integers(2, 5))
    mode = int(rng.integers(0, 3))                       # impulse: ends on the grid / inside interval 1 / absent
    t_sw = {0: dt_eval, 1: 1.37 * dt_eval, 2: None}[mode]
    held = 0.02 * rng.standard_normal((B, ens.n)) if rng.random() < 0.3 else None
    packable = int(ens.plan.layout.beams_per_group) > 1 and 1 <= int(ens.plan.layout.pcr_levels_full) <= 5
    per_wave = bool(packable and rng.random() < 0.5)
    control = "positions" if rng.random() < 0.3 else "all"
    args = dict(rtol=float(rng.choice([1e-2, 1e-3, 1e-4])), atol=float(rng.choice([1e-5, 1e-7])),
                impulse_amp=None if mode == 2 else amps, impulse_duration=t_sw if t_sw else 0.01, held_force=held,
                first_rate=float(rng.choice([0.0, 2.0, 64.0])) / dt_eval, t0=0.0, control=control, max_rungs=12)
    ens.set_state(x0)
    try:
        try:
            snaps, stats, used = ens.solve_controlled(n_int, dt_eval, per_wave=per_wave, **args)
        except Exception as e:                            # (a pinned interior node makes gravity non-canonical: no packed form)
            if not (per_wave and "per_wave" in str(e)):
                raise
            per_wave = False
            ens.set_state(x0)
            snaps, stats, used = ens.solve_controlled(n_int, dt_eval, **args)
    except RuntimeError as e:                             # (a tolerance the ladder of 12 rungs cannot meet is a legitimate outcome)
        assert "tolerances ask for more" in str(e)
        return
    y = ens.unpack_snapshots(snaps).cpu().numpy()
    assert np.all(stats[:, 2] == 0) and np.all(np.isfinite(y))
    for b in range(B):
        for k in range(n_int):
            if mode == 1 and k == 1:
                continue                                  # (the cut interval: two pieces with their own step counts)
            start = x0[b] if k == 0 else y[k - 1, b]
            m = int(used[b, k])
            want = ob.implicit(start, dt_eval / m, m, n_iter=1, amp=0.0 if mode == 2 else amps[b], duration=t_sw if t_sw else 0.0,
                               t0=k * dt_eval, u_held=None if held is None else held[b])
            assert_blocks(y[k, b], want, ens.free_index, 5e-9, what=(seed, n_e, kind, b, k, m, per_wave, control))


def test_example_scripts_run_and_agree_with_the_oracle():
    """examples/beam_comparison_ensemble.py (the reference's beam_comparison_* task lists as one ensemble) and
    examples/lqr_ensemble.py (its lqr_control.py loop over many impulses): both run, the comparison's linear dry rod
    follows the oracle's implicit trajectory, the LQR loop ends closer to rest than the open loop."""
    import importlib.util

    ex = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples")
    sys.path.insert(0, ex)
    try:
        mods = {}
        for name in ("beam_comparison_ensemble", "lqr_ensemble"):
            spec = importlib.util.spec_from_file_location(f"crb_example_{name}", os.path.join(ex, name + ".py"))
            mods[name] = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mods[name])
        from _common import rod
    finally:
        sys.path.remove(ex)
    y = mods["beam_comparison_ensemble"].main(["--elements", "6", "--t-final", "0.01", "--substeps", "10"])
    assert y.shape == (9, 36, 11) and np.isfinite(y).all()
    ob = oracle_beam({c: rod(6, "linear")[c].to_numpy() for c in rod(6, "linear").columns})
    want = ob.implicit(np.zeros(36), 1e-4, 100, n_iter=2, amp=0.1)
    assert_blocks(y[0, :, -1], want, ob.red2full(), 1e-9, what="linear dry rod of the example")
    # the default: the reference's call with its tolerances, the step size chosen per rod inside the kernel.  The uncontrolled
    # h = 1e-4 s run is what DESIGN says it is next to it: the tip deflection inside the default tolerance band, the rotations
    # of the unresolved modes a few tens of bands off (measured 26)
    yc = mods["beam_comparison_ensemble"].main(["--elements", "6", "--t-final", "0.01"])
    assert yc.shape == y.shape and np.isfinite(yc).all()
    bands = np.abs(yc[:, :18, -1] - y[:, :18, -1]) / (1e-6 + 1e-3 * np.abs(yc[:, :18, -1]))
    assert bands[:, 16].max() < 2.0 and bands.max() < 60.0, (bands[:, 16].max(), bands.max())
    rows = mods["lqr_ensemble"].main(["--elements", "4", "--beams", "8", "--t-final", "0.01"])
    (_, open_tip, _), (_, lqr_tip, _) = rows
    assert np.isfinite(open_tip).all() and np.isfinite(lqr_tip).all()
    assert np.abs(lqr_tip).max() < np.abs(open_tip).max()
    rows = mods["lqr_ensemble"].main(["--elements", "4", "--beams", "8", "--t-final", "0.01", "--lsoda"])
    assert len(rows) == 3 and np.allclose(rows[2][1], rows[1][1], rtol=1e-3, atol=1e-9), np.abs(rows[2][1] - rows[1][1]).max()   # the tolerance-controlled run (the fixed-step run sees the end of the impulse in one stage of one step)


@pytest.mark.parametrize("lean", [True, False])
@pytest.mark.parametrize("n_e,B,kind,kw,bcs", [
    (6, 5, "linear", dict(enable_gravity=True), None),                                  # the reference's LQR example size
    (6, 70, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True), None),          # several workgroups, 10 beams a wave
    (10, 3, "mixed", dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True), "pinned"),
    (10, 40, "linear", dict(enable_gravity=True), None),                                # lqr_control.py's own size: K e on the matrix cores
    (9, 13, "mixed", dict(fluid_density=1000.0, enable_fluid=True), None),              # 27 rows: matrix cores, partly filled last wave
    (27, 4, "linear", dict(enable_gravity=True), None),                                 # gain of 102 KB: LDS above 64 KB
    (1, 9, "linear", dict(), None),
])
def test_fused_feedback_stepper_matches_the_stage_split_one_and_the_oracle(n_e, B, kind, kw, bcs, lean, monkeypatch):
    """Closed-loop rollouts of beams that live in one wave take ONE launch -- the packed lean stepper with the feedback in its
    stages (crb_step_lean_kernel<..., FB>: 3 .. 5 reduction levels, gravity absent or canonical) or the general kernel
    (crb_beam_kernel<..., FB>; CRB_DISABLE_LEAN_FEEDBACK forces it): the gain in LDS, u = K (r - x) formed per stage, on the
    matrix cores for gains of 21 .. 32 rows -- instead of the stage-split path's eight launches per step:
    against RK4 over the oracle RHS with the feedback in every stage (lqr_control.py:95-111), per DOF block, with
    per-beam references and amplitudes from random states, and against the stage-split path (CRB_FUSED_FEEDBACK=0)."""
    monkeypatch.setenv("CRB_FUSED_FEEDBACK", "1")   # (whenever the gain fits LDS: also where the default would not choose it)
    if not lean:
        monkeypatch.setenv("CRB_DISABLE_LEAN_FEEDBACK", "1")
    rng = np.random.default_rng(100 + n_e)
    kinds = ["nonlinear" if i % 3 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    bc = None
    if bcs == "pinned":
        bc = ["PINNED"] + ["NONE"] * (n_e - 1)
        bc[n_e // 2] = "PINNED"
    cols = nitinol_columns(n_e, kinds, bc) if bc else nitinol_columns(n_e, kinds)
    ens = ensemble(cols, B, kw)
    n = ens.n
    gain = rng.normal(0.0, 2e-2, (n, 2 * n))
    ref = rng.normal(0.0, 1e-4, (B, 2 * n))
    x0 = rng.normal(0.0, 1e-4, (B, 2 * n))
    amps = 0.05 * (1.0 + np.arange(B))
    steps, dt = 60, 1e-5
    ens.set_state(x0)
    t = ens.step_feedback(steps, dt, gain, reference=ref, impulse_amp=amps)
    got = ens.unpack_state().cpu().numpy()
    ob = oracle_beam(cols, **kw)
    for b in range(0, B, max(1, B // 6)):
        want = ob.rk4_feedback(x0[b], dt, steps, gain, reference=ref[b], amp=amps[b])
        assert_blocks(got[b], want, ens.free_index, 1e-9, what=b)
    # no reference, no impulse
    ens.set_state(x0)
    ens.step_feedback(steps, dt, gain)
    got0 = ens.unpack_state().cpu().numpy()
    assert_blocks(got0[B - 1], ob.rk4_feedback(x0[B - 1], dt, steps, gain), ens.free_index, 1e-9)
    # the stage-split path (GEMM + stage kernel per stage) agrees to rounding
    monkeypatch.setenv("CRB_FUSED_FEEDBACK", "0")
    ens2 = ensemble(cols, B, kw)
    ens2.set_state(x0)
    t2 = ens2.step_feedback(steps, dt, gain, reference=ref, impulse_amp=amps)
    assert t2 == t
    assert_blocks(ens2.unpack_state().cpu().numpy(), got, ens.free_index, 1e-10)


@pytest.mark.parametrize("n_e", [6, 10])
def test_fused_feedback_stepper_walks_over_groups_of_beams(n_e, monkeypatch):
    """The packed lean stepper with the feedback loads the gain (and its matrix-core fragments) once per workgroup and WALKS over
    several groups of beams when the ensemble is larger than what is resident (CRB_LEAN_MAX_GROUPS = 2 here: two workgroups for
    70 beams): reduced indices, references and amplitudes are per group, the error vectors of a finished group must not leak
    into the next one; the last group is partly filled."""
    monkeypatch.setenv("CRB_FUSED_FEEDBACK", "1")
    monkeypatch.setenv("CRB_LEAN_MAX_GROUPS", "2")
    cols, kw, B = nitinol_columns(n_e, "linear"), dict(enable_gravity=True), 70
    rng = np.random.default_rng(50 + n_e)
    ens = ensemble(cols, B, kw)
    n = ens.n
    gain = rng.normal(0.0, 2e-2, (n, 2 * n))
    ref = rng.normal(0.0, 1e-4, (B, 2 * n))
    x0 = rng.normal(0.0, 1e-4, (B, 2 * n))
    amps = 0.02 * (1.0 + np.arange(B))
    ens.set_state(x0)
    ens.step_feedback(40, 1e-5, gain, reference=ref, impulse_amp=amps)
    got = ens.unpack_state().cpu().numpy()
    ob = oracle_beam(cols, **kw)
    for b in (0, 1, 8, 9, 17, 35, 62, 63, 68, 69):
        want = ob.rk4_feedback(x0[b], 1e-5, 40, gain, reference=ref[b], amp=amps[b])
        assert_blocks(got[b], want, ens.free_index, 1e-9, what=b)


@pytest.mark.parametrize("lean", [True, False])
@pytest.mark.parametrize("n_e", [6, 10])
def test_fused_feedback_stepper_in_single_precision(n_e, lean, monkeypatch):
    """fp32 plans run the same fused closed-loop kernels (the product K e on v_mfma_f32_16x16x4_f32 for 10 elements, from LDS
    for 6): the rollout tracks the fp64 one at single precision."""
    monkeypatch.setenv("CRB_FUSED_FEEDBACK", "1")
    if not lean:
        monkeypatch.setenv("CRB_DISABLE_LEAN_FEEDBACK", "1")
    cols, kw, B = nitinol_columns(n_e, "linear"), dict(enable_gravity=True), 23
    rng = np.random.default_rng(n_e)
    outs = []
    gain = rng.normal(0.0, 2e-2, (3 * n_e, 6 * n_e))
    x0 = rng.normal(0.0, 1e-4, (B, 6 * n_e))
    ref = rng.normal(0.0, 1e-4, (B, 6 * n_e))
    for dtype in (torch.float64, torch.float32):
        e = ensemble(cols, B, kw, dtype=dtype)
        assert e.feedback_path() == "fused"
        e.set_state(x0)
        e.step_feedback(30, 1e-5, gain, reference=ref, impulse_amp=np.full(B, 0.05))
        outs.append(e.unpack_state().double().cpu().numpy())
    assert rel_err(outs[1], outs[0]) < 2e-3


@pytest.mark.parametrize("n_e,B,kind,kw,bcs,with_ref,groups", [
    (128, 70, "linear", dict(enable_gravity=True), None, False, None),      # config 5's shape, two row blocks (the second ragged)
    (128, 200, "linear", dict(enable_gravity=True), None, True, 2),          # four row blocks on two groups: groups walk; reference
    (100, 64, "nonlinear", dict(fluid_density=1000.0, enable_fluid=True), None, False, None),   # padding slots, per-lane element branch
    (70, 65, "mixed", dict(fluid_density=1000.0, enable_fluid=True, enable_gravity=True), None, True, None),
    (64, 130, "linear", dict(enable_gravity=True), None, False, None),      # one wave per beam: groups of four workgroups
    (40, 65, "linear", dict(), None, True, 1),
    (96, 66, "linear", dict(enable_gravity=False), "pinned_tip", True, None),   # constrained DOFs inside the padded slot order
])
def test_persistent_closed_loop_stepper_matches_the_oracle_and_the_stage_split_path(n_e, B, kind, kw, bcs, with_ref, groups, monkeypatch):
    """crb_loop_kernel (csrc/crb_loop.h): the closed-loop rollout of lqr_control.py:95-125 as ONE persistent launch -- groups
    of workgroups own 64 beams, keep their slice of the gain in registers, alternate between the fp64-MFMA product and
    the stage arithmetic and hand tiles to each other through L2.  Against RK4 over the oracle RHS with u = K (r - x) in
    every stage, per DOF block, per-beam amplitudes (and references) from random states -- first / last beam of every row
    block of 64 and the ones next to the block borders -- and against the stage-split path (CRB_LOOP=0), every beam."""
    monkeypatch.setenv("CRB_LOOP", "1")
    if groups:
        monkeypatch.setenv("CRB_LOOP_MAX_GROUPS", str(groups))
    rng = np.random.default_rng(1000 + n_e + B)
    kinds = ["nonlinear" if i % 3 else "linear" for i in range(n_e)] if kind == "mixed" else kind
    bc = None
    if bcs == "pinned_tip":
        bc = ["FIXED"] + ["NONE"] * (n_e - 1)
        bc[n_e // 2] = "PINNED"
    cols = nitinol_columns(n_e, kinds, bc) if bc else nitinol_columns(n_e, kinds)
    ens = ensemble(cols, B, kw)
    n = ens.n
    gain = rng.normal(0.0, 2e-2, (n, 2 * n))
    ref = rng.normal(0.0, 1e-4, (B, 2 * n)) if with_ref else None
    x0 = rng.normal(0.0, 1e-4, (B, 2 * n))
    amps = 0.05 * (1.0 + np.arange(B) / B)
    steps, dt = 14, 2e-5
    ens.set_state(x0)
    t = ens.step_feedback(steps, dt, gain, reference=ref, impulse_amp=amps, impulse_duration=6.5 * dt)
    assert ens.feedback_status() == 0
    got = ens.unpack_state().cpu().numpy()
    assert np.isfinite(got).all()
    ob = oracle_beam(cols, **kw)
    for b in sorted({0, 1, 7, 8, 62, 63, 64 % B, 65 % B, B // 2, B - 2, B - 1}):
        want = ob.rk4_feedback(x0[b], dt, steps, gain, reference=None if ref is None else ref[b], amp=amps[b], duration=6.5 * dt)
        assert_blocks(got[b], want, ens.free_index, 1e-10, what=b)
    # a second call continues from where the first one stopped (fresh buffers, same clock): equal to one rollout of twice the length
    ens.step_feedback(steps, dt, gain, reference=ref, impulse_amp=amps, impulse_duration=6.5 * dt)
    ens_long = ensemble(cols, B, kw)
    ens_long.set_state(x0)
    ens_long.step_feedback(2 * steps, dt, gain, reference=ref, impulse_amp=amps, impulse_duration=6.5 * dt)
    assert torch.equal(ens.state, ens_long.state)
    monkeypatch.setenv("CRB_LOOP", "0")
    ens2 = ensemble(cols, B, kw)
    ens2.set_state(x0)
    t2 = ens2.step_feedback(steps, dt, gain, reference=ref, impulse_amp=amps, impulse_duration=6.5 * dt)
    assert t2 == t
    assert_blocks(ens2.unpack_state().cpu().numpy(), got, ens.free_index, 1e-10)


def test_persistent_closed_loop_stepper_with_release_acquire_fences(monkeypatch):
    """The same launch with agent-scope release / acquire fences around every hand-off (CRB_LOOP_FENCES=1: the memory model's
    own form, 50 us per step dearer) gives bitwise the result of the write-through / L1-bypassing form it ships with."""
    monkeypatch.setenv("CRB_LOOP", "1")
    cols = nitinol_columns(128, "linear")
    kw = dict(enable_gravity=True)
    rng = np.random.default_rng(5)
    B = 130
    outs = []
    for fences in ("0", "1"):
        monkeypatch.setenv("CRB_LOOP_FENCES", fences)
        ens = ensemble(cols, B, kw)
        n = ens.n
        if not outs:
            gain = rng.normal(0.0, 2e-2, (n, 2 * n))
            x0 = rng.normal(0.0, 1e-4, (B, 2 * n))
        ens.set_state(x0)
        ens.step_feedback(10, 2e-5, gain, impulse_amp=np.full(B, 0.1))
        assert ens.feedback_status() == 0
        outs.append(ens.state.clone())
    assert torch.equal(outs[0], outs[1])


def test_full_size_config5_total_on_one_gpu():
    """BASELINE config 5's TOTAL -- 16384 beams x 128 linear elements + gravity, LQR feedback per stage -- on one GPU: 256 row
    blocks on the 32 resident groups of the persistent stepper (each group rolls eight row blocks out, one after the
    other).  A dense random gain of the real one's size, 40 steps; a beam of every eighth row block against the oracle."""
    cols = nitinol_columns(128, "linear")
    kw = dict(enable_gravity=True)
    B, steps, dt = 16384, 40, 5e-6
    ens = ensemble(cols, B, kw)
    n = ens.n
    rng = np.random.default_rng(77)
    gain = rng.normal(0.0, 2e-2, (n, 2 * n))
    x0 = np.concatenate([rng.normal(0, 1e-5, (B, n)), rng.normal(0, 1e-3, (B, n))], axis=1)
    amps = 10.0 * (1.0 + np.arange(B) / B)
    ens.set_state(x0)
    ens.step_feedback(steps, dt, gain, impulse_amp=amps)
    assert ens.feedback_status() == 0
    got = ens.unpack_state().cpu().numpy()
    assert np.isfinite(got).all()
    ob = oracle_beam(cols, **kw)
    for b in list(range(5, B, 8 * 64 + 9)) + [B - 1]:
        want = ob.rk4_feedback(x0[b], dt, steps, gain, amp=amps[b])
        assert_blocks(got[b], want, ens.free_index, 1e-10, what=b)


def test_per_beam_status_reports_the_launch_in_which_a_beam_went_non_finite():
    """crb_plan_set_status / BeamEnsemble.status: 0 while a beam is finite, else the ensemble's step count at the end of the
    launch that first left a non-finite value in it -- through the lean stepper, the general stepper, the implicit stepper
    and both closed-loop paths.  (a) poisoned beams are reported by the first launch, their neighbours never;
    (b) the shipped nonlinear element (segments.py:178-208: no -EA/L u2 coupling) diverges on a 256-element chain past
    ~1500 steps of dt = 2e-5: every beam is reported in the launch in which torch.isfinite first fails for it."""
    # (a) lean stepper, packed beams (several per wave) and one beam per workgroup
    for n_e, B in ((10, 40), (128, 9)):
        ens = ensemble(nitinol_columns(n_e, "nonlinear"), B, dict(fluid_density=1000.0, enable_fluid=True))
        st = ens.status
        x0 = np.zeros((B, 2 * ens.n))
        x0[3, 5] = np.nan
        x0[B - 1, ens.n + 1] = np.inf
        ens.set_state(x0)
        ens.step(7, 2e-5, impulse_amp=np.full(B, 0.1))
        ens.step(5, 2e-5, impulse_amp=np.full(B, 0.1))
        want = np.zeros(B, dtype=np.int32)
        want[[3, B - 1]] = 7
        assert np.array_equal(st.cpu().numpy(), want), (n_e, st)
        fin = torch.isfinite(ens.unpack_state()).all(dim=1).cpu().numpy()
        assert np.array_equal(fin, want == 0)
    # general stepper (gravity on a pinned beam), implicit stepper, closed loop (persistent and stage-split)
    cols = nitinol_columns(70, "linear")
    for mode in ("general", "implicit", "loop", "split"):
        kw = dict(enable_gravity=True)
        c = nitinol_columns(70, "linear", ["PINNED"] + ["NONE"] * 69) if mode == "general" else cols
        os.environ["CRB_LOOP"] = "1" if mode == "loop" else "0"
        try:
            ens = ensemble(c, 66, kw)
            st = ens.status
            x0 = np.zeros((66, 2 * ens.n))
            x0[65, 2] = np.nan
            ens.set_state(x0)
            if mode == "general":
                ens.step(4, 2e-5)
            elif mode == "implicit":
                ens.step_implicit(4, 1e-4)
            else:
                ens.step_feedback(4, 5e-6, np.zeros((ens.n, 2 * ens.n)))
            want = np.zeros(66, dtype=np.int32)
            want[65] = 1 if mode == "split" else 4   # (the stage-split loop is one launch per stage: it reports per step)
            assert np.array_equal(st.cpu().numpy(), want), (mode, st)
        finally:
            os.environ.pop("CRB_LOOP", None)
    # (b)
    B = 16
    ens = ensemble(nitinol_columns(256, "nonlinear"), B, dict(fluid_density=1000.0, enable_fluid=True))
    st = ens.status
    amps = 0.1 * (1.0 + np.arange(B) / B)
    first_bad = np.zeros(B, dtype=np.int64)
    for launch in range(1, 9):
        ens.step(500, 2e-5, impulse_amp=amps)
        fin = torch.isfinite(ens.unpack_state()).all(dim=1).cpu().numpy()
        first_bad = np.where((first_bad == 0) & ~fin, 500 * launch, first_bad)
    assert (first_bad > 1000).all() and (first_bad > 0).all(), first_bad   # stable for the metric's 1000 steps, gone by 4000
    assert np.array_equal(st.cpu().numpy(), first_bad.astype(np.int32))


def test_bench_two_ranks_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` end to end on this ONE GPU (CRB_BENCH_REHEARSAL=1: both ranks on GPU 0, the exchange
    over gloo): the launcher, the shards, the chunked rollout with its asynchronous all-gather, the max-reduced clock
    and the JSON line of rank 0.  The number it prints measures nothing; that the two shards of the gathered ensemble
    are finite, complete and parity-checked against the oracle does."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CRB_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", "config4", "--steps", "20",
                          "--warmup", "0", "--no-cpu-baseline", "--repeats", "1", "--gather-chunks", "2"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2 and line["config"]["gather_chunks"] == 2
    assert line["config"]["beams_per_gpu"] == [2048, 2048] and line["scaling"] == "strong"
    assert line["check"]["finite"] and line["check"]["gathered_beams"] == 4096
    assert "rehearsal" in line["config"]
    errs = line["check"]["block_err_vs_oracle_last_beam"]
    assert max(errs.values()) < 1e-4, errs          # fp32 against the fp64 oracle at 20 steps


def test_bench_takes_the_rccl_path_with_one_rank():
    """`CRB_BENCH_FORCE_DIST=1 python bench.py --gpus 1`: launcher, torch.distributed over the REAL "nccl" (= RCCL) backend,
    barrier, the all-gather of the terminal states and the max-reduced clock, with one rank -- the N > 1 code path alive on a
    one-GPU box (the 2 / 4 / 8-GPU runs are the round-end driver's)."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CRB_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CRB_BENCH_REHEARSAL"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--no-cpu-baseline", "--repeats", "2"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["ranks_seen"] == 1 and line["config"]["launched_by"] == "bench.py launcher"
    assert line["config"]["collective"].startswith("all_gather_into_tensor") and "rehearsal" not in line["config"]
    assert line["config"]["exchange_alone_ms"] > 0
    assert line["check"]["finite"] and line["check"]["gathered_beams"] == 4096
    assert max(line["check"]["block_err_vs_oracle_last_beam"].values()) < 1e-9


@pytest.mark.parametrize("n_e,B", [(10, 7), (64, 3), (130, 2)])
@pytest.mark.parametrize("span", [0.05, 0.78, 3.0, 40.0])
def test_gravity_rotation_kernels_over_small_and_large_angles(n_e, B, span):
    """Gravity rotates each segment's weight by cos / sin of its mean rotation (gravity_forces.py:117-124): RHS and a
    few steps from states whose rotations span +-span rad (small, at pi/4, beyond it, many periods: the device
    library's range reduction) against the oracle (libm), per DOF block, for the lean and the general kernels."""
    rng = np.random.default_rng(int(span * 100) + n_e)
    cols = nitinol_columns(n_e, "linear")
    kw = dict(enable_gravity=True, gravity=[1.5, -9.81, 0.0])
    ob = oracle_beam(cols, **kw)
    n = ob.n
    x = rng.normal(0.0, 1e-3, (B, 2 * n))
    x[:, 2:n:3] = rng.uniform(-span, span, (B, n // 3))          # the rotations
    x[0, 2:n:3] = np.linspace(-span, span, n // 3)               # one beam sweeping the interval end to end
    for env in (None, "CRB_DISABLE_LEAN"):
        if env:
            os.environ[env] = "1"
        try:
            ens = ensemble(cols, B, kw)
            xd = ens.rhs(x).cpu().numpy()
            ens.set_state(x)
            ens.step(3, 1e-6)
            got = ens.unpack_state().cpu().numpy()
        finally:
            if env:
                del os.environ[env]
        for b in range(B):
            assert_blocks(xd[b], ob.rhs(x[b]), ens.free_index, 1e-12, what=("rhs", env, b))
            assert_blocks(got[b], ob.rk4_impulse(x[b], 1e-6, 3, 0.0), ens.free_index, 1e-11, what=("rk4", env, b))


def test_cantilever_rings_at_the_euler_bernoulli_natural_frequencies():
    """Physics check of the whole path (assembly, M^-1, stepper): the free vibration of the examples' 10-element Nitinol
    cantilever after a tip impulse, integrated for 8 s with the implicit stepper, has its spectral peaks at the analytic
    Euler-Bernoulli natural frequencies f_n = (beta_n L)^2 sqrt(EI / (rho A L^4)) / (2 pi), beta_n L = 1.8751, 4.6941
    (the formula the reference's examples print next to their plots, examples/example_utilities.py:208-240)."""
    n_e = 10
    cols = nitinol_columns(n_e, "linear")
    ens = ensemble(cols, 2, {})
    h, steps, every = 1e-3, 8000, 5
    _, rec = ens.step_implicit(steps, h, n_iter=3, impulse_amp=np.array([0.1, 0.2]), impulse_duration=0.01,
                               record=(n_e, "w"), record_every=every)
    w = rec.cpu().numpy()
    assert w.shape == (2, steps // every) and np.isfinite(w).all()
    assert np.allclose(w[1], 2.0 * w[0], rtol=1e-9, atol=1e-15)          # linear: the response scales with the impulse
    L = float(np.sum(cols["length"]))
    EI = float(cols["elastic_modulus"][0] * cols["moment_inertia"][0])
    rhoA = float(cols["density"][0] * cols["cross_area"][0])
    analytic = [(bl ** 2) * np.sqrt(EI / (rhoA * L ** 4)) / (2 * np.pi) for bl in (1.875104, 4.694091)]
    sig = (w[0] - w[0].mean()) * np.hanning(w.shape[1])
    spec = np.abs(np.fft.rfft(sig, 8 * sig.size))                       # zero-padded: 1/64 Hz bins
    freq = np.fft.rfftfreq(8 * sig.size, every * h)
    for f_n in analytic:
        band = (freq > 0.6 * f_n) & (freq < 1.4 * f_n)
        peak = freq[band][np.argmax(spec[band])]
        assert abs(peak - f_n) < 0.02 * f_n + 0.02, (peak, f_n)


def test_implicit_reduction_levels_follow_the_step_size(monkeypatch):
    """The cyclic reduction of A = M + h^2/4 K0 stops where its multipliers fall below the unit roundoff (5 of 8 levels at the
    examples' h = 1e-4 for a 256-node rod, all 8 at h = 1e-3): the truncated run equals the run with every level
    (CRB_STIFF_ALL_LEVELS=1) to rounding, for the lean and the general kernels."""
    cols = nitinol_columns(256, "nonlinear")
    kw = dict(fluid_density=1000.0, enable_fluid=True)
    amps = np.array([0.05, 0.2])
    outs = {}
    for general in (False, True):
        for all_levels in (False, True):
            if general:
                monkeypatch.setenv("CRB_DISABLE_LEAN_IMPLICIT", "1")
            else:
                monkeypatch.delenv("CRB_DISABLE_LEAN_IMPLICIT", raising=False)
            if all_levels:
                monkeypatch.setenv("CRB_STIFF_ALL_LEVELS", "1")
            else:
                monkeypatch.delenv("CRB_STIFF_ALL_LEVELS", raising=False)
            ens = ensemble(cols, 2, kw)
            ens.step_implicit(50, 1e-4, n_iter=2, impulse_amp=amps)
            outs[(general, all_levels)] = ens.unpack_state().cpu().numpy()
    ref = outs[(True, True)]
    for key, got in outs.items():
        assert np.isfinite(got).all()
        assert_blocks(got, ref, ens.free_index, 1e-10, what=key)


@pytest.mark.parametrize("poison", ["nan", "inf"])
def test_packed_implicit_beams_are_isolated_from_a_diverged_wave_mate(poison):
    """Several short beams share a wave in the packed lean implicit kernel; a beam whose state is non-finite must not
    reach its wave-mates (selects at the beam boundaries, not zero weights): every other beam is bitwise equal to the run
    without the poisoned one."""
    cols = nitinol_columns(10, "linear")
    kw = dict(enable_gravity=True)
    B = 13
    rng = np.random.default_rng(77)
    x0 = rng.normal(0.0, 1e-6, (B, 60))
    amps = 0.05 * (1.0 + np.arange(B) / B)
    a = ensemble(cols, B, kw)
    a.set_state(x0)
    a.step_implicit(30, 1e-4, n_iter=2, impulse_amp=amps)
    want = a.unpack_state()
    bad = [3, 8]
    x1 = x0.copy()
    x1[bad, 5] = np.nan if poison == "nan" else np.inf
    b = ensemble(cols, B, kw)
    b.set_state(x1)
    b.step_implicit(30, 1e-4, n_iter=2, impulse_amp=amps)
    got = b.unpack_state()
    good = [k for k in range(B) if k not in bad]
    assert torch.equal(got[good], want[good])
    assert not bool(torch.isfinite(got[bad]).all())
