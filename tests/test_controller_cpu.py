"""CPU tests of the step-size controller behind BeamEnsemble.solve_ivp (host logic only: the controller is run over a
stand-in ensemble whose "stepper" is a textbook scheme on a linear oscillator, no GPU and no kernel involved).

Reference call sites it serves: solve_ivp(..., method="LSODA", t_eval=..., [rtol, atol]) of
examples/example_utilities.py:153-159 and examples/lqr_control.py:113-125."""
import numpy as np
import pytest
import torch

from continuum_robot.batched import BeamEnsemble


class _Stub:
    """What _solve_controlled touches of an ensemble: state [B, 2, n_node, 4], unpack_state(), n_per_beam, device, time."""

    def __init__(self, B=2, omega=300.0):
        self.device = torch.device("cpu")
        self.n_per_beam = np.full(B, 1)
        self.state = torch.zeros((B, 2, 1, 4), dtype=torch.float64)
        self.state[:, 0, 0, 0] = torch.arange(1, B + 1, dtype=torch.float64) * 1e-2      # q
        self.time = 0.0
        self.omega = omega
        self.calls = []

    def unpack_state(self):
        return self.state[:, :, 0, 0].clone()

    def midpoint(self, m, h, t, on=None, force=0.0):
        """implicit midpoint rule (order 2) on q'' = -omega^2 q + f(t), f = force while the input is on"""
        self.calls.append((m, h, t, on))
        w2 = self.omega ** 2
        f = force if (on or (on is None and False)) else 0.0
        q, v = self.state[:, 0, 0, 0].clone(), self.state[:, 1, 0, 0].clone()
        for _ in range(m):
            # [q1; v1] = [q0; v0] + h * A * ([q0; v0] + [q1; v1]) / 2 + h * [0; f]
            a = 0.25 * h * h * w2
            q1 = ((1 - a) * q + h * v + 0.5 * h * h * f) / (1 + a)
            v1 = v - 0.5 * h * w2 * (q + q1) + h * f
            q, v = q1, v1
        self.state[:, 0, 0, 0], self.state[:, 1, 0, 0] = q, v


def _exact(q0, omega, t):
    return q0 * np.cos(omega * t), -q0 * omega * np.sin(omega * t)


@pytest.mark.parametrize("rtol,atol", [(1e-3, 1e-6), (1e-6, 1e-9)])
def test_step_doubling_meets_the_tolerance_and_adapts_the_step(rtol, atol):
    ens = _Stub()
    first = ens.unpack_state().unsqueeze(0)
    n_t, dt = 21, 1e-3
    ys, used = BeamEnsemble._solve_controlled(ens, lambda m, h, t, on=None: ens.midpoint(m, h, t, on), 2, 1, 0.0, dt, n_t, first,
                                              rtol, atol, "all")
    assert tuple(ys.shape) == (n_t, 2, 2) and len(used) == n_t - 1 and abs(ens.time - (n_t - 1) * dt) < 1e-15
    y = ys.numpy()
    for b, q0 in enumerate((1e-2, 2e-2)):
        q, v = _exact(q0, ens.omega, np.arange(n_t) * dt)
        # local control: the global error after 20 intervals stays within a few tens of tolerance units
        assert np.max(np.abs(y[:, b, 0] - q) / (atol + rtol * np.abs(q).max())) < 40
        assert np.max(np.abs(y[:, b, 1] - v) / (atol + rtol * np.abs(v).max())) < 40
    # second order: a tolerance 1000 times tighter needs about sqrt(1000) times the steps
    assert min(used) >= 2 and (max(used) > 30 if rtol < 1e-4 else max(used) <= 64)
    # every accepted interval was integrated twice from the same state (m and 2m steps), in that order
    assert ens.calls[0][0] * 2 == ens.calls[1][0] and ens.calls[0][2] == ens.calls[1][2] == 0.0


def test_the_integration_is_cut_at_the_end_of_the_impulse():
    ens = _Stub(B=1)
    ens.state.zero_()
    first = ens.unpack_state().unsqueeze(0)
    t_switch = 0.00237                       # inside the third interval
    adv = lambda m, h, t, on=None: ens.midpoint(m, h, t, on, force=5.0)
    ys, used = BeamEnsemble._solve_controlled(ens, adv, 2, 1, 0.0, 1e-3, 6, first, 1e-6, 1e-9, "all", t_switch)
    starts = sorted({round(t, 12) for _, _, t, _ in ens.calls})
    assert round(t_switch, 12) in starts                         # a piece starts exactly at the switch
    for m, h, t, on in ens.calls:                                # no piece straddles it, and the flag follows the side
        assert (t + m * h <= t_switch + 1e-12) == bool(on), (t, m * h, on)
    # the exact response: forced oscillator up to t_switch, free afterwards
    w = ens.omega
    qs, vs = 5.0 / w**2 * (1 - np.cos(w * t_switch)), 5.0 / w * np.sin(w * t_switch)
    t_end = 5e-3 - t_switch
    q_end = qs * np.cos(w * t_end) + vs / w * np.sin(w * t_end)
    assert abs(ys[-1, 0, 0].item() - q_end) < 1e-4 * abs(q_end) + 1e-9


def test_estimates_that_are_not_finite_double_the_steps_and_the_budget_is_enforced():
    ens = _Stub(B=1)
    first = ens.unpack_state().unsqueeze(0)

    def unstable_above(m, h, t, on=None):                        # an explicit scheme beyond its stability limit: NaN
        if h > 2.6e-4:
            ens.state = ens.state * float("nan")
        else:
            ens.midpoint(m, h, t, on)

    ys, used = BeamEnsemble._solve_controlled(ens, unstable_above, 2, 1, 0.0, 1e-3, 3, first, 1e-2, 1e-4, "all")
    assert torch.isfinite(ys).all() and min(used) >= 8           # 1e-3 / 8 = 1.25e-4 is the first fine step below the limit
    ens2 = _Stub(B=1)
    with pytest.raises(RuntimeError, match="tolerances ask for more"):
        BeamEnsemble._solve_controlled(ens2, lambda m, h, t, on=None: ens2.midpoint(m, h, t, on), 2, 1, 0.0, 1e-3, 3,
                                       ens2.unpack_state().unsqueeze(0), 1e-14, 1e-17, "all", None, 64)
    with pytest.raises(ValueError, match="control"):
        BeamEnsemble._solve_controlled(ens2, None, 2, 1, 0.0, 1e-3, 3, first, 1e-3, 1e-6, "velocities")


def test_position_control_ignores_the_velocity_half():
    ens = _Stub(B=1, omega=3000.0)                               # omega h >> 1 at the first steps: velocity phase is lost
    first = ens.unpack_state().unsqueeze(0)
    adv = lambda m, h, t, on=None: ens.midpoint(m, h, t, on)
    _, used_all = BeamEnsemble._solve_controlled(ens, adv, 2, 1, 0.0, 1e-3, 4, first, 1e-3, 1e-6, "all")
    ens = _Stub(B=1, omega=3000.0)
    _, used_pos = BeamEnsemble._solve_controlled(ens, lambda m, h, t, on=None: ens.midpoint(m, h, t, on), 2, 1, 0.0, 1e-3, 4,
                                                 first, 1e-3, 1e-6, "positions")
    assert max(used_pos) <= max(used_all)
