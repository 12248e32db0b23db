"""BeamEnsemble -- B independent beams of one topology stepped on an MI355X.

This is the batched entry point of the drop-in: it owns the state tensor
``[B, 2, n_node, 4]`` on the GPU (PyTorch-ROCm memory) and calls the HIP kernels of libcrbeam.so
through the C ABI (include/crbeam.h).  The planning / control layers hold one ensemble per robot
model and roll it out; nothing here runs on the CPU -- without the extension or a GPU every
operation raises.

Reference interfaces mirrored (file:line under /root/reference/src/continuum_robot/models/):
  constructor arguments   dynamic_beam_model.py:25-29 (CSV schema :78-90, ForceParams)
  rhs()                   get_dynamic_system()(t, x, u), dynamic_beam_model.py:343-362
  internal_force()        get_stiffness_function(), euler_bernoulli_beam.py:364-368
  step()                  the scipy.solve_ivp call sites (examples/example_utilities.py:153-159)
"""
import ctypes as C
import pathlib
from typing import Optional, Sequence, Union

import numpy as np
import pandas as pd
import torch

from . import _native as nat
from .models.force_params import ForceParams

_CSV_REQUIRED = ["length", "elastic_modulus", "moment_inertia", "density", "cross_area", "type", "boundary_condition"]
_PARAM = {"u": 0, "w": 1, "phi": 2}


def _columns(parameters, need_fluid):
    if isinstance(parameters, (str, pathlib.Path)):
        parameters = pd.read_csv(parameters)  # FileNotFoundError propagates, as in the reference (:46)
    if isinstance(parameters, pd.DataFrame):
        cols = {c: parameters[c].to_numpy() for c in parameters.columns}
    elif isinstance(parameters, dict):
        cols = dict(parameters)
    else:
        raise TypeError("Parameters must be filepath, pandas DataFrame or a dict of columns")
    required = _CSV_REQUIRED + (["wetted_area", "drag_coef"] if need_fluid else [])
    if not all(c in cols for c in required):
        raise ValueError(f"CSV must contain columns: {', '.join(required)}")
    bad = set(str(b) for b in cols["boundary_condition"]) - {"FIXED", "PINNED", "NONE"}
    if bad:
        raise ValueError(f"Invalid boundary conditions: {bad}")
    return cols


class _NoContext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CONTEXT = _NoContext()


class _FusedForce:
    """Marker for a built-in force that the RHS kernel evaluates itself (drag: fluid_forces.py:103-142, gravity:
    gravity_forces.py:66-148).  It sits in the ensemble's registry like the reference's auto-registered force
    objects (dynamic_beam_model.py:220-241); switching ``enabled`` off takes effect at the next call, because the
    registry is re-read on every call (force_registry.py:66-67)."""

    def __init__(self, kind: str):
        self.kind = kind
        self.enabled = True

    def is_enabled(self) -> bool:
        return self.enabled

    def compute_forces(self, x, t):
        raise RuntimeError(f"the built-in {self.kind} force is evaluated inside the RHS kernel, not on the host")


class BatchedForceRegistry:
    """ForceRegistry of an ensemble (reference: models/force_registry.py:6-88, same methods and semantics): force
    components whose ``compute_forces(x, t)`` maps the reduced states ``x[B, 2n]`` (a torch tensor on the ensemble's
    device) to generalised forces ``[B, n]``.  ``register`` ignores a component that is disabled at that moment
    (:20-21); the aggregate re-checks ``is_enabled()`` on every call (:66-67); ``get_registered_forces`` returns a copy."""

    def __init__(self):
        self._forces = []

    def register(self, force_instance) -> None:
        if force_instance.is_enabled():
            self._forces.append(force_instance)

    def unregister(self, force_instance) -> bool:
        if force_instance in self._forces:
            self._forces.remove(force_instance)
            return True
        return False

    def clear(self) -> None:
        self._forces.clear()

    def get_registered_forces(self):
        return self._forces.copy()

    def __len__(self) -> int:
        return len(self._forces)

    def __contains__(self, force_instance) -> bool:
        return force_instance in self._forces


class BeamEnsemble:
    """B independent beams stepped together on one GPU.

    ``parameters``: one parameter set (CSV path / DataFrame / dict of columns) shared by all ``n_beams`` beams, or a
    list of ``n_beams`` of them -- a heterogeneous ensemble (SURVEY f-3): every beam has its own element columns and
    types and may have its own element COUNT and boundary-condition column.  ``force_params`` is then one
    ``ForceParams`` for all beams or a list of ``n_beams`` (what examples/beam_comparison_fluid.py:49-83 runs as six
    processes -- three beams without, three with fluid -- is ONE ensemble here).

    Reduced vectors ([q_red ; v_red], the reference's state ordering) are ``[B, 2 n]``.  When the beams' free-DOF
    sets differ (``mixed_topology``), ``n`` is the largest beam's and beam b uses the first ``n_per_beam[b]`` entries
    of each half (the rest is ignored on input and zero on output); ``beam_state(b)`` / ``pad_states`` convert."""

    def __init__(self, parameters, n_beams: int, force_params=None, dtype=torch.float64,
                 device: Union[str, torch.device, int] = "cuda", corrected_axial: bool = False, node_bc=None):
        per_beam_fp = isinstance(force_params, (list, tuple))
        if per_beam_fp:
            if not isinstance(parameters, (list, tuple)):
                parameters = [parameters] * n_beams      # one beam description, per-beam ForceParams
            if len(force_params) != n_beams:
                raise ValueError("a list of ForceParams must have n_beams entries")
            self.force_params = [fp or ForceParams() for fp in force_params]
        else:
            self.force_params = force_params or ForceParams()
        fp_of = (lambda b: self.force_params[b]) if per_beam_fp else (lambda b: self.force_params)
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("dtype must be torch.float64 or torch.float32")
        nat.load()  # raises if the extension was not built
        if not torch.cuda.is_available():
            raise RuntimeError("BeamEnsemble needs a HIP device: the beam stepper has no CPU path")
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.type != "cuda":
            raise RuntimeError("BeamEnsemble needs a HIP device: the beam stepper has no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._device_index = int(self.device.index)
        self.dtype = dtype
        if isinstance(parameters, (list, tuple)):
            # heterogeneous ensemble: one parameter set per beam
            if len(parameters) != n_beams:
                raise ValueError("a list of parameter sets must have n_beams entries")
            self.columns = [_columns(p, fp_of(b).enable_fluid_effects) for b, p in enumerate(parameters)]
        else:
            self.columns = _columns(parameters, fp_of(0).enable_fluid_effects)
        fps = [fp_of(b) for b in range(n_beams)] if per_beam_fp else None
        with torch.cuda.device(self.device):    # (plan creation selects the plan's device: keep the caller's current one)
            self.plan = nat.Plan(
                self.columns, n_beams=n_beams, node_bc=node_bc,
                fluid_density=[f.fluid_density for f in fps] if fps else fp_of(0).fluid_density,
                enable_fluid=[f.enable_fluid_effects for f in fps] if fps else fp_of(0).enable_fluid_effects,
                gravity=[f.get_gravity_vector() for f in fps] if fps else fp_of(0).get_gravity_vector(),
                enable_gravity=[f.enable_gravity_effects for f in fps] if fps else fp_of(0).enable_gravity_effects,
                corrected_axial=corrected_axial, dtype="f64" if dtype == torch.float64 else "f32", device=self.device.index)
        p = self.plan
        self.n_beams, self.n_elem, self.n_node, self.n = n_beams, p.n_elem, p.n_node, p.n_free
        self.mixed_topology = p.mixed_topology
        self.free_index = p.free_index.copy()                 # beam 0's (every beam's unless mixed_topology)
        if self.mixed_topology:
            self.free_index_per_beam = [p.beam_free_index(b) for b in range(n_beams)]
            self.n_per_beam = np.array([fi.size for fi in self.free_index_per_beam])
        else:
            self.free_index_per_beam = [self.free_index] * n_beams
            self.n_per_beam = np.full(n_beams, self.n)
        # (element counts can differ with ONE free-DOF set: a longer beam whose extra nodes are all FIXED)
        self.n_elem_per_beam = (np.array([p.beam_info(b)[0] for b in range(n_beams)]) if p.per_beam
                                else np.full(n_beams, self.n_elem))
        self.state = torch.zeros((n_beams, 2, self.n_node, 4), dtype=dtype, device=self.device)
        self.time = 0.0
        self._lib = nat.load()
        # functional composition (step_composed): the built-in forces appear in the registry as markers, user forces
        # are torch callables; plans with a built-in force switched off are created on demand
        self._plan_args = dict(n_beams=n_beams, node_bc=node_bc, corrected_axial=corrected_axial,
                               dtype="f64" if dtype == torch.float64 else "f32", device=self.device.index)
        self._fps = fps if fps else [fp_of(0)]
        self._plans = {}
        self.force_registry = BatchedForceRegistry()
        self._auto_drag = self._auto_gravity = None
        if any(f.enable_fluid_effects for f in self._fps):
            self._auto_drag = _FusedForce("drag")
            self.force_registry.register(self._auto_drag)
        if any(f.enable_gravity_effects for f in self._fps):
            self._auto_gravity = _FusedForce("gravity")
            self.force_registry.register(self._auto_gravity)
        self._plans[(self._auto_drag is not None, self._auto_gravity is not None)] = self.plan

    @classmethod
    def from_dataframes(cls, frames: Sequence, force_params=None, **kwargs):
        """One beam per entry of ``frames`` (DataFrames, CSV paths or column dicts -- the reference's beam files,
        euler_bernoulli_beam.py:26-109; boundary conditions from each file's own column,
        dynamic_beam_model.py:205-218), ``force_params`` one ForceParams or one per beam."""
        frames = list(frames)
        return cls(frames, len(frames), force_params=force_params, **kwargs)

    # ------------------------------------------------------------------ ragged reduced vectors (mixed_topology)
    def pad_states(self, states: Sequence) -> np.ndarray:
        """per-beam reference state vectors [2 n_b] -> the ensemble's [B, 2 n] layout ([q_b, 0.. ; v_b, 0..])"""
        if len(states) != self.n_beams:
            raise ValueError("expected one state vector per beam")
        out = np.zeros((self.n_beams, 2 * self.n))
        for b, x in enumerate(states):
            nb = int(self.n_per_beam[b])
            x = np.asarray(x, dtype=np.float64)
            if x.shape != (2 * nb,):
                raise ValueError(f"beam {b}: expected a state of {2 * nb} entries, got {x.shape}")
            out[b, :nb] = x[:nb]
            out[b, self.n:self.n + nb] = x[nb:]
        return out

    def beam_state(self, b: int, x_red=None) -> np.ndarray:
        """the reference's state vector [q_red ; v_red] of beam ``b`` (host array of 2 n_b entries)"""
        x = self.unpack_state() if x_red is None else x_red
        row = x[b].detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x[b])
        nb = int(self.n_per_beam[b])
        return np.concatenate([row[:nb], row[self.n:self.n + nb]])

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _on_device(self):
        """Context in which the library calls run: the plan's device current.  The library selects it itself
        (hipSetDevice); when it already is the current one there is nothing to restore afterwards, and the call path of
        a short launch saves the context manager's two device switches."""
        if torch.cuda.current_device() == self._device_index:
            return _NO_CONTEXT
        return torch.cuda.device(self.device)

    def _dev(self, a, shape=None):
        t = torch.as_tensor(a, dtype=self.dtype, device=self.device).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def reduced_index(self, node: int, param: str, beam: int = 0) -> int:
        """Reduced position index of (node, 'u'|'w'|'phi') in beam ``beam``; KeyError when constrained or absent
        (reference: EulerBernoulliBeam.get_dof_index, euler_bernoulli_beam.py:404-420)."""
        full = 3 * node + _PARAM[param]
        hit = np.nonzero(self.free_index_per_beam[beam] == full)[0]
        if node < 0 or node >= self.n_node or hit.size == 0:
            raise KeyError(f"Invalid node/parameter combination: ({node}, {param})")
        return int(hit[0])

    # ------------------------------------------------------------------ layout conversion
    def pack_state(self, x_red) -> torch.Tensor:
        """reduced [B, 2n] (reference ordering [q_red ; v_red]) -> device layout [B, 2, n_node, 4]"""
        x_red = self._dev(x_red, (self.n_beams, 2 * self.n))
        out = torch.empty_like(self.state)
        with self._on_device():
            nat.check(self._lib.crb_pack_state(self.plan.h, self._ptr(x_red), self._ptr(out), self._stream()))
        return out

    def unpack_state(self, x=None, out=None) -> torch.Tensor:
        """Device layout -> the reference's reduced ordering [B, 2n]; ``out``: a contiguous [B, 2n] tensor to write into (e.g.
        this rank's slot of an all-gather's result)."""
        x = self.state if x is None else x
        if out is None:
            out = torch.empty((self.n_beams, 2 * self.n), dtype=self.dtype, device=self.device)
        elif tuple(out.shape) != (self.n_beams, 2 * self.n) or out.dtype != self.dtype or not out.is_contiguous():
            raise ValueError("unpack_state: out must be a contiguous [n_beams, 2n] tensor of the ensemble's dtype")
        with self._on_device():
            nat.check(self._lib.crb_unpack_state(self.plan.h, self._ptr(x), self._ptr(out), self._stream()))
        return out

    def pack_vec(self, v_red) -> torch.Tensor:
        v_red = self._dev(v_red, (self.n_beams, self.n))
        out = torch.empty((self.n_beams, self.n_node, 4), dtype=self.dtype, device=self.device)
        with self._on_device():
            nat.check(self._lib.crb_pack_vec(self.plan.h, self._ptr(v_red), self._ptr(out), self._stream()))
        return out

    def unpack_vec(self, v) -> torch.Tensor:
        out = torch.empty((self.n_beams, self.n), dtype=self.dtype, device=self.device)
        with self._on_device():
            nat.check(self._lib.crb_unpack_vec(self.plan.h, self._ptr(v), self._ptr(out), self._stream()))
        return out

    def unpack_snapshots(self, snaps: torch.Tensor) -> torch.Tensor:
        """[n_rec, B, 2, n_node, 4] snapshots of step(..., record="all") -> reduced [n_rec, B, 2n] (sol.y ordering)."""
        return torch.stack([self.unpack_state(snaps[k]) for k in range(snaps.shape[0])])

    def beam_shapes(self, snaps: torch.Tensor, dx: float, as_reference: bool = True):
        """Beam x, y coordinates over time from whole-state snapshots (``step(..., record="all")`` /
        ``solve_rk45(record="all")`` / ``step_implicit(record="all")``), the reference's
        ``extract_beam_shapes(sol, n_segments, dx)`` (examples/example_utilities.py:173-205) for every beam:
        x[t, b, j] = j * dx, y[t, b, 0] = 0 (fixed base), y[t, b, j + 1] = pos[j].

        ``as_reference=True`` reproduces the reference's indexing ``pos = sol.y[n_pos + 1::3]`` -- entries of the
        VELOCITY half of the state (the transverse RATES dw/dt of the free nodes, not the
        displacements); ``False`` reads the transverse displacements ``sol.y[1:n_pos:3]`` the docstring there means.
        Returns numpy arrays of shape [n_t, B, n_elem + 1]."""
        red = self.unpack_snapshots(snaps).detach().cpu().numpy()        # [n_t, B, 2n]: sol.y per beam, transposed
        n_t, n_pts = red.shape[0], self.n_elem + 1
        x = np.broadcast_to(np.arange(n_pts) * float(dx), (n_t, self.n_beams, n_pts)).copy()
        y = np.zeros((n_t, self.n_beams, n_pts))
        pos = red[:, :, self.n + 1::3] if as_reference else red[:, :, 1:self.n:3]
        m = min(pos.shape[2], self.n_elem)
        y[:, :, 1:1 + m] = pos[:, :, :m]
        return x, y

    def set_state(self, x_red, time: float = 0.0) -> None:
        self.state = self.pack_state(x_red)
        self.time = float(time)
        self.reset_status()

    def zero_state(self) -> None:
        self.state.zero_()
        self.time = 0.0
        self.reset_status()

    # ------------------------------------------------------------------ per-beam status
    @property
    def status(self) -> torch.Tensor:
        """int32 [B]: 0 while a beam's state is finite, else the number of steps the ensemble had taken at the END of the launch
        in which the beam went non-finite (fixed-step steppers: step, step_implicit, step_feedback, step_composed's stages).
        The first access switches the reporting on; `set_state` / `zero_state` / `reset_status` start it afresh.  The
        reference's solve_ivp just returns NaN rows when its shipped nonlinear element (segments.py:178-208) diverges."""
        if getattr(self, "_status", None) is None:
            self._status = torch.zeros((self.n_beams,), dtype=torch.int32, device=self.device)
            with self._on_device():
                nat.check(self._lib.crb_plan_set_status(self.plan.h, self._ptr(self._status), 0))
        return self._status

    def reset_status(self) -> None:
        if getattr(self, "_status", None) is not None:
            self._status.zero_()
            with self._on_device():
                nat.check(self._lib.crb_plan_set_status(self.plan.h, self._ptr(self._status), 0))

    def _impulse(self, desc, keep, impulse_amp, impulse_duration, impulse_index):
        """Fill the impulse part of a crb_input_desc: amplitudes [B] on reduced position index ``impulse_index`` of
        EVERY beam (-2 = each beam's own tip w, example_utilities.py:147) while t < impulse_duration."""
        amp = self._dev(impulse_amp, (self.n_beams,))
        nodes, dofs = [], []
        for b in (range(self.n_beams) if self.mixed_topology else (0,)):
            fi = self.free_index_per_beam[b]
            idx = impulse_index if impulse_index >= 0 else fi.size + impulse_index
            if not 0 <= idx < fi.size:
                raise IndexError("impulse_index out of range")
            nodes.append(int(fi[idx]) // 3)
            dofs.append(int(fi[idx]) % 3)
        if len(set(dofs)) != 1:
            raise ValueError("impulse_index addresses different DOF kinds (u / w / phi) in different beams")
        desc.kind, desc.node, desc.dof = nat.CRB_INPUT_IMPULSE, nodes[0], dofs[0]
        desc.duration = float(impulse_duration)
        desc.amp = amp.data_ptr()
        keep.append(amp)
        if len(set(nodes)) > 1:       # beams of different length / boundary conditions: each forced at its own node
            node_b = torch.as_tensor(nodes, dtype=torch.int32, device=self.device)
            desc.node_b = node_b.data_ptr()
            keep.append(node_b)

    # ------------------------------------------------------------------ the hot path
    def internal_force(self, q_red) -> torch.Tensor:
        """k(q) for every beam, reduced [B, n]."""
        q = self._dev(q_red, (self.n_beams, self.n))
        x = self.pack_state(torch.cat([q, torch.zeros_like(q)], dim=1))
        k = torch.empty((self.n_beams, self.n_node, 4), dtype=self.dtype, device=self.device)
        with self._on_device():
            nat.check(self._lib.crb_internal_force(self.plan.h, self._ptr(x), self._ptr(k), self._stream()))
        return self.unpack_vec(k)

    def rhs_device(self, x: torch.Tensor, u: Optional[torch.Tensor] = None) -> torch.Tensor:
        """xdot in device layout for a state (and optional force) in device layout."""
        out = torch.empty_like(x)
        with self._on_device():
            nat.check(self._lib.crb_rhs(self.plan.h, self._ptr(x), self._ptr(u), self._ptr(out), self._stream()))
        return out

    def rhs(self, x_red=None, u_red=None) -> torch.Tensor:
        """dynamic_system(t, x, u) for every beam: reduced [B, 2n] in, reduced [B, 2n] out."""
        x = self.state if x_red is None else self.pack_state(x_red)
        u = None if u_red is None else self.pack_vec(u_red)
        return self.unpack_state(self.rhs_device(x, u))

    def step(self, n_steps: int, dt: float, impulse_amp=None, impulse_duration: float = 0.01,
             impulse_index: int = -2, held_force=None, t0: Optional[float] = None, record=None, record_every: int = 1):
        """Advance the resident state by ``n_steps`` RK4 steps in one kernel launch.

        impulse_amp    per-beam amplitudes [B] of the examples' forcing: that value on reduced
                       position index ``impulse_index`` (-2 = tip w, example_utilities.py:147)
                       while t < impulse_duration
        held_force     reduced [B, n] generalised force held constant over the call
        record         (node, 'u'|'w'|'phi'|'du_dt'|'dw_dt'|'dphi_dt'): sample that DOF on the device after
                       every ``record_every``-th step (the t_eval output of the reference's solve_ivp
                       calls); the call then returns (clock, samples[B, n_steps // record_every]).
                       "all": whole-state snapshots instead, (clock, snaps[n_rec, B, 2, n_node, 4])
        Returns the clock after the call (accumulated by addition, as the oracle does).
        """
        if t0 is not None:
            self.time = float(t0)
        desc = nat.InputDesc()
        desc.kind = nat.CRB_INPUT_NONE
        keep = []
        if impulse_amp is not None:
            self._impulse(desc, keep, impulse_amp, impulse_duration, impulse_index)
        if held_force is not None:
            held = self.pack_vec(held_force)
            desc.f_held = held.data_ptr()
            keep.append(held)
        t_end = C.c_double(0.0)
        rec, samples = None, None
        if isinstance(record, str) and record == "all":
            # whole-state snapshots (every DOF of the reference's sol.y on the t_eval grid): device layout
            # [n_rec, B, 2, n_node, 4]; reduced ordering through unpack_snapshots()
            samples = torch.zeros((int(n_steps) // int(record_every),) + tuple(self.state.shape), dtype=self.dtype,
                                  device=self.device)
            rec = nat.RecordDesc(0, -1, 0, int(record_every), samples.data_ptr())
            keep.append(samples)
        elif record is not None:
            node, param = record
            vel = param.startswith("d") and param.endswith("_dt")
            samples = torch.zeros((self.n_beams, int(n_steps) // int(record_every)), dtype=self.dtype, device=self.device)
            rec = nat.RecordDesc(int(vel), int(node), _PARAM[param[1:-3] if vel else param], int(record_every),
                                 samples.data_ptr())
            keep.append(samples)
        with self._on_device():
            nat.check(self._lib.crb_step_rk4_rec(self.plan.h, self._ptr(self.state), self.time, float(dt), int(n_steps),
                                                 C.byref(desc), C.byref(rec) if rec is not None else None,
                                                 C.byref(t_end), self._stream()))
        self._keep = keep  # device buffers must outlive the asynchronous launch
        self.time = t_end.value
        return (self.time, samples) if record is not None else self.time

    def step_implicit(self, n_steps: int, h: float, n_iter: int = 2, impulse_amp=None, impulse_duration: float = 0.01,
                      impulse_index: int = -2, held_force=None, t0: Optional[float] = None, record=None,
                      record_every: int = 1, rho_inf: float = 1.0):
        """Advance the resident state by ``n_steps`` steps of size ``h`` of the implicit midpoint rule in one launch
        (crb_step_implicit): the stiff end of the reference's call sites -- the examples integrate 1 s with
        ``solve_ivp(method="LSODA")`` (example_utilities.py:153-159) because explicit steppers are limited to
        dt <= ~7e-5 s; here h = 1e-3 ... 1e-4 s is stable.  ``n_iter`` modified-Newton iterations per step (2
        reproduces the converged step); inputs are sampled at the step midpoint.  ``record`` as in ``step``.
        Displacements converge at second order in h; velocity components of modes with |lambda| h >> 1 are not
        resolved (amplitude kept, phase not).  ``rho_inf`` < 1: the numerically damped member of the family
        (generalised-alpha, crb_step_implicit_damped) -- those modes lose the factor ``rho_inf`` per step instead, as
        they do under LSODA's BDF formulas; 0 removes them within a step or two, 1 is the midpoint rule."""
        if t0 is not None:
            self.time = float(t0)
        desc = nat.InputDesc()
        desc.kind = nat.CRB_INPUT_NONE
        keep = []
        if impulse_amp is not None:
            self._impulse(desc, keep, impulse_amp, impulse_duration, impulse_index)
        if held_force is not None:
            held = self.pack_vec(held_force)
            desc.f_held = held.data_ptr()
            keep.append(held)
        t_end = C.c_double(0.0)
        rec, samples = None, None
        if isinstance(record, str) and record == "all":
            samples = torch.zeros((int(n_steps) // int(record_every),) + tuple(self.state.shape), dtype=self.dtype,
                                  device=self.device)
            rec = nat.RecordDesc(0, -1, 0, int(record_every), samples.data_ptr())
            keep.append(samples)
        elif record is not None:
            node, param = record
            vel = param.startswith("d") and param.endswith("_dt")
            samples = torch.zeros((self.n_beams, int(n_steps) // int(record_every)), dtype=self.dtype, device=self.device)
            rec = nat.RecordDesc(int(vel), int(node), _PARAM[param[1:-3] if vel else param], int(record_every),
                                 samples.data_ptr())
            keep.append(samples)
        with self._on_device():
            nat.check(self._lib.crb_step_implicit_damped(self.plan.h, self._ptr(self.state), self.time, float(h), int(n_steps),
                                                         int(n_iter), float(rho_inf), C.byref(desc),
                                                         C.byref(rec) if rec is not None else None, C.byref(t_end), self._stream()))
        self._keep = keep
        self.time = t_end.value
        return (self.time, samples) if record is not None else self.time

    def solve_ivp(self, t_span, t_eval, method: str = "LSODA", impulse_amp=None, impulse_duration: float = 0.01,
                  impulse_index: int = -2, held_force=None, substeps: Union[int, str, None] = None,
                  rtol: float = 1e-3, atol: float = 1e-6, control: str = "all", gain=None, reference=None,
                  controller: str = "auto", rho_inf: float = 1.0):
        """The examples' integration call for the whole ensemble (examples/example_utilities.py:153-159:
        ``solve_ivp(f, t_span, x0, method="LSODA", t_eval=np.arange(t0, t1, DT))``) from the RESIDENT state, with the
        examples' forcing.  ``t_eval`` must be a uniform grid starting at ``t_span[0]`` (what ``np.arange`` gives).

        method   "LSODA" / "BDF" / "Radau" (the stiff solvers the examples use): the A-stable implicit stepper
                 (``step_implicit``).  ``substeps="auto"`` (the default, as the reference's call means it): the step size
                 is CONTROLLED by ``rtol`` / ``atol`` (``_solve_implicit_controlled``: every interval is integrated with
                 h and with h / 2, the difference is the error estimate, all beams take the step the worst one needs;
                 at the examples' default tolerances the whole state, velocities included, lands as close to a
                 tight-tolerance LSODA run as LSODA itself does at these tolerances -- h = 3 ... 6 us for the 10-element
                 example).  An integer: that many steps per ``t_eval`` interval, no control, one launch for the whole
                 span -- DT = 1e-3 with 10 substeps is h = 1e-4 s (displacements within the examples' tolerance band,
                 velocity content of the unresolved modes not);
                 "RK45": ``solve_rk45`` (scipy's algorithm, per-beam step control, rtol / atol) with
                 its dense output on the grid;   "RK4": the fused explicit stepper with ``substeps`` steps per interval
                 (default 10; stable for dt <= ~7e-5 s).
        gain     [n, 2n]: the CLOSED loop of examples/lqr_control.py:94-125 (``u = K (reference - x)`` inside the RHS, default
                 reference 0, plus the impulse) -- RK4 with the feedback evaluated in every stage (``step_feedback``) for
                 every method except "RK45"; the stiff methods choose its step by ``rtol`` / ``atol`` with the same
                 controller ((fine - coarse) / 15 for the fourth-order scheme; a step beyond RK4's stability limit shows
                 as a failed estimate and is halved), "RK4" takes ``substeps`` as given.
        rho_inf  (integer ``substeps`` of the stiff methods) < 1: the damped implicit scheme, see ``step_implicit``.
        controller  where the step-size control of ``substeps="auto"`` runs.  "device": inside the kernel, every beam with its
                 own step sequence, the whole span in ONE launch (``solve_controlled`` / crb_solve_controlled; the closed loop
                 only for gains that fit the LDS, beams of up to ~30 elements);  "host": the same controller as a host loop
                 over fixed-step launches, the worst beam deciding for the ensemble (``_solve_controlled``; any gain, through
                 ``step_feedback``);  "device-packed" (implicit scheme, beams of 2 .. 32 thread-carried nodes): G = 64 / slots
                 beams share a wave and ONE step sequence, the worst of them deciding -- thousands of short beams fill the
                 chip with a fifth of the waves;  "auto": the device whenever it can, packed once one workgroup per beam
                 would need more than two rounds of resident waves (2048).
        Returns an object with ``t`` [n_t] and ``y`` [B, 2n, n_t] (``y[b]`` is the reference's ``sol.y`` of beam b,
        first column = the state at ``t_span[0]``), ``success``, ``method``; the resident state ends at the last
        ``t_eval`` point reached by whole intervals (RK45: at ``t_span[1]``)."""
        t_eval = np.asarray(t_eval, dtype=np.float64)
        if t_eval.ndim != 1 or t_eval.size < 1 or abs(t_eval[0] - t_span[0]) > 1e-12 * max(1.0, abs(t_span[0])):
            raise ValueError("t_eval must be a 1-D grid starting at t_span[0]")
        n_t = t_eval.size
        dt_eval = float(t_eval[1] - t_eval[0]) if n_t > 1 else float(t_span[1] - t_span[0])
        if n_t > 2 and np.max(np.abs(np.diff(t_eval) - dt_eval)) > 1e-9 * dt_eval:
            raise ValueError("t_eval must be uniform (np.arange / np.linspace)")
        self.time = float(t_span[0])
        first = self.unpack_state().unsqueeze(0)                       # the state at t_span[0]
        kind = method.upper()
        if substeps is None:
            substeps = "auto" if kind in ("LSODA", "BDF", "RADAU", "IMPLICIT") else 10
        elif not isinstance(substeps, str) and int(substeps) < 1:
            raise ValueError('substeps must be a positive integer or "auto"')
        kw = dict(impulse_amp=impulse_amp, impulse_duration=impulse_duration, impulse_index=impulse_index,
                  held_force=held_force)
        stiff = kind in ("LSODA", "BDF", "RADAU", "IMPLICIT")
        if isinstance(substeps, str) and substeps != "auto":
            raise ValueError('substeps must be a positive integer or "auto"')
        if gain is not None:
            if held_force is not None or kind == "RK45" or not (stiff or kind == "RK4"):
                raise ValueError("the closed loop (gain=...) runs with the stiff methods or RK4, without a held force")
            fkw = dict(reference=reference, impulse_amp=impulse_amp, impulse_duration=impulse_duration, impulse_index=impulse_index)

            def advance(m, h, t_start, forced=None):
                self.step_feedback(m, h, gain, t0=t_start, **(fkw if forced is None else forced_input(fkw, forced)))
        else:
            def advance(m, h, t_start, forced=None):
                self.step_implicit(m, h, t0=t_start, **(kw if forced is None else forced_input(kw, forced)))

        def forced_input(args, on):
            # the controller integrates between breakpoints of the input: inside a piece the impulse is simply on or off
            # (a stage time that lands ON the switch would otherwise put a first-order error into a fourth-order step)
            args = dict(args)
            if on:
                args["impulse_duration"] = float("inf")
            else:
                args["impulse_amp"] = None
            return args

        t_switch = None if impulse_amp is None else float(impulse_duration)
        ctrl_stats = None
        if controller not in ("auto", "device", "device-packed", "host"):
            raise ValueError('controller must be "auto", "device", "device-packed" or "host"')
        if n_t == 1:
            ys = first
        elif gain is not None and not isinstance(substeps, str):
            out = [first]
            for k in range(n_t - 1):
                advance(int(substeps), dt_eval / int(substeps), float(t_span[0]) + k * dt_eval)
                out.append(self.unpack_state().unsqueeze(0))
            self.time = float(t_span[0]) + (n_t - 1) * dt_eval
            ys = torch.cat(out, dim=0)
        elif (stiff or gain is not None) and isinstance(substeps, str) and self._device_controller(controller, gain):
            snaps, stats, per_beam = self.solve_controlled(n_t - 1, dt_eval, rtol=rtol, atol=atol, control=control, gain=gain,
                                                           reference=reference, t0=float(t_span[0]),
                                                           per_wave=self._packs_per_wave(controller, gain), **kw)
            ys = torch.cat([first, self.unpack_snapshots(snaps)], dim=0)
            used = [int(v) for v in per_beam.max(axis=0)]
            ctrl_stats = (stats, per_beam)
        elif gain is not None:
            # (RK4 is only conditionally stable: start near the closed loop's limit for the Nitinol examples, 8.6e-6 s)
            ys, used = self._solve_controlled(advance, 4, max(1, int(np.ceil(dt_eval / 5e-6))), float(t_span[0]), dt_eval, n_t,
                                              first, rtol, atol, control, t_switch)
        elif stiff and isinstance(substeps, str):
            ys, used = self._solve_controlled(advance, 2, max(1, int(np.ceil(dt_eval / 1e-4))), float(t_span[0]), dt_eval, n_t,
                                              first, rtol, atol, control, t_switch)
        elif kind in ("LSODA", "BDF", "RADAU", "IMPLICIT"):
            _, snaps = self.step_implicit((n_t - 1) * int(substeps), dt_eval / int(substeps), record="all",
                                          record_every=int(substeps), rho_inf=rho_inf, **kw)
            ys = torch.cat([first, self.unpack_snapshots(snaps)], dim=0)
        elif kind == "RK4":
            if isinstance(substeps, str):
                raise ValueError("RK4 takes an integer number of substeps")
            _, snaps = self.step((n_t - 1) * int(substeps), dt_eval / int(substeps), record="all",
                                 record_every=int(substeps), **kw)
            ys = torch.cat([first, self.unpack_snapshots(snaps)], dim=0)
        elif kind == "RK45":
            st = self.solve_rk45(float(t_span[1]), rtol=rtol, atol=atol, record="all",
                                 t_eval=(float(t_eval[0]), dt_eval, n_t), **kw)
            if np.any(st["status"] != 0):
                raise RuntimeError("solve_ivp(RK45): a beam stopped before t_span[1]")
            ys = self.unpack_snapshots(st["y"])
        else:
            raise ValueError(f"unknown method {method!r}: LSODA / BDF / Radau (implicit), RK45, RK4")

        class OdeResult:   # the fields of scipy's OdeResult that the examples read
            pass

        sol = OdeResult()
        sol.t, sol.y, sol.success, sol.method = t_eval.copy(), ys.permute(1, 2, 0).contiguous(), True, kind
        if isinstance(substeps, str) and stiff and n_t > 1:
            sol.substeps = used      # steps per t_eval interval that the controller accepted (device: the most any beam took)
            if ctrl_stats is not None:
                sol.controller = "device-packed" if self._packs_per_wave(controller, gain) else "device"
                sol.substeps_per_beam = ctrl_stats[1]        # [B, n_t - 1]
                sol.doublings = ctrl_stats[0][:, 1].copy()   # repeated pieces per beam
            else:
                sol.controller = "host"
        return sol

    def _device_controller(self, controller, gain) -> bool:
        """Whether ``solve_ivp(substeps="auto")`` runs its controller inside the kernel (crb_solve_controlled's conditions)."""
        if controller == "host":
            return False
        if controller == "device-packed":
            controller = "device"
        ok = self.dtype == torch.float64 and int(self.plan.layout.threads) <= 256
        if ok and gain is not None:
            n2p = (2 * self.n + 7) // 8 * 8
            lds = 14 * int(self.plan.layout.threads) * 8 + (n2p + n2p * self.n + 32) * 8 + 512
            ok = not isinstance(gain, (list, tuple)) and not self.mixed_topology and lds <= 160 * 1024
        if controller == "device" and not ok:
            raise ValueError("controller=\"device\": fp64 plans with beams of up to 256 thread-carried nodes; the closed loop "
                             "needs one gain that fits the LDS (beams of up to ~30 elements)")
        return ok

    def _packs_per_wave(self, controller, gain) -> bool:
        """Whether the device controller packs short beams G to a wave (``per_wave``): asked for, or -- "auto" -- when one
        workgroup per beam would need more than two rounds of resident waves (one wave per SIMD: 1024 waves are resident;
        4096 x 10 elements for 1 s: 0.75 s with one beam per wave, the host loop 0.28 s)."""
        lay = self.plan.layout
        can = gain is None and int(lay.beams_per_group) > 1 and int(lay.pcr_levels_full) <= 5 and int(lay.pcr_levels_full) >= 1
        if controller == "device-packed":
            if not can:
                raise ValueError("controller=\"device-packed\": implicit scheme, beams of 2 .. 32 thread-carried nodes")
            return True
        return controller == "auto" and can and self.n_beams * (int(lay.threads) // 64) > 2048

    def solve_controlled(self, n_intervals: int, dt_eval: float, rtol: float = 1e-3, atol: float = 1e-6, control: str = "all",
                         gain=None, reference=None, impulse_amp=None, impulse_duration: float = 0.01, impulse_index: int = -2,
                         held_force=None, t0: Optional[float] = None, n_iter: int = 1, first_rate: float = 0.0,
                         max_rungs: int = 0, record=True, per_wave: bool = False):
        """``n_intervals`` intervals of length ``dt_eval`` from the resident state with the step size chosen per beam by
        ``rtol`` / ``atol`` INSIDE the kernel, one launch (crb_solve_controlled, csrc/crb_ctrl.h): the implicit midpoint rule,
        or -- with ``gain`` -- RK4 with the feedback in every stage.  ``n_iter``: modified-Newton iterations per implicit step (1: at
        the controller's step sizes the previous step's iterate is converged after one, profiles/exp_niter.py).  Returns (snapshots [n_intervals, B, 2, n_node, 4] or
        None -- or, with ``record=(node, param)``, that DOF's series [B, n_intervals] --, stats [B, 4] (fine steps accepted,
        doublings, status, last rung), steps per beam and interval [B, n_intervals]); raises when a beam could not meet the
        tolerances."""
        if control not in ("all", "positions"):
            raise ValueError('control must be "all" or "positions"')
        if t0 is not None:
            self.time = float(t0)
        desc = nat.InputDesc()
        desc.kind = nat.CRB_INPUT_NONE
        keep = []
        if impulse_amp is not None:
            self._impulse(desc, keep, impulse_amp, impulse_duration, impulse_index)
        if held_force is not None:
            held = self.pack_vec(held_force)
            desc.f_held = held.data_ptr()
            keep.append(held)
        K = None if gain is None else self._dev(gain, (self.n, 2 * self.n))
        ref = None if reference is None else self._dev(reference, (self.n_beams, 2 * self.n))
        n_intervals = int(n_intervals)
        # record: True = whole-state snapshots, False = none, (node, param) = that DOF's series [B, n_intervals] instead (what the
        # examples read; param "w" / "dw_dt" ...): the first return value is then the series
        series, sp, sn, sd = None, 0, 0, 0
        if isinstance(record, tuple):
            node, param = record
            vel = param.startswith("d") and param.endswith("_dt")
            sp, sn, sd = int(vel), int(node), _PARAM[param[1:-3] if vel else param]
            series = torch.zeros((self.n_beams, max(n_intervals, 1)), dtype=self.dtype, device=self.device)
        ctl = nat.ControlDesc(float(rtol), float(atol), float(first_rate), int(control == "positions"), int(n_iter), int(max_rungs),
                              int(bool(per_wave)), sp, sn, sd, 0, series.data_ptr() if series is not None else None)
        snaps = torch.zeros((n_intervals,) + tuple(self.state.shape), dtype=self.dtype, device=self.device) if record is True else None
        stats = torch.zeros((self.n_beams, 4), dtype=torch.int32, device=self.device)
        used = torch.zeros((self.n_beams, max(n_intervals, 1)), dtype=torch.int32, device=self.device)
        with self._on_device():
            nat.check(self._lib.crb_solve_controlled(self.plan.h, self._ptr(self.state), self.time, float(dt_eval), n_intervals,
                                                     C.byref(ctl), C.byref(desc), self._ptr(K), self._ptr(ref), self._ptr(snaps),
                                                     self._ptr(stats), self._ptr(used), self._stream()))
        self._keep = keep + [K, ref, snaps, stats, used, series]
        st = stats.cpu().numpy()
        if np.any(st[:, 2] != 0):
            bad = int(np.flatnonzero(st[:, 2] != 0)[0])
            raise RuntimeError(f"solve_controlled: the tolerances ask for more steps per interval than the ladder holds "
                               f"(beam {bad}; {int(np.count_nonzero(st[:, 2]))} beams in all)")
        self.time = self.time + n_intervals * float(dt_eval)
        return (series[:, :n_intervals] if series is not None else snaps), st, used.cpu().numpy()[:, :n_intervals]

    def _solve_controlled(self, advance, order, m_first, t0, dt_eval, n_t, first, rtol, atol, control, t_switch=None,
                          max_substeps=1 << 14):
        """Step-size control by step doubling for a fixed-step scheme of the given order (``advance(m, h, t, on)`` takes m
        steps of size h from the resident state at time t with the impulse on or off: the implicit midpoint rule, order
        2, or the closed-loop RK4, order 4): every ``t_eval`` interval -- cut in two at ``t_switch``, the end of the
        impulse, when that falls inside it -- is integrated from the same state with m steps and with 2m steps;
        (fine - coarse) / (2^order - 1)
        estimates the error of the fine solution, which is measured like scipy measures its own (RMS over a beam's state
        of err / (atol + rtol |y|), tolerances of the reference's call, example_utilities.py:153-159 /
        lqr_control.py:117-125) -- the worst beam decides for the ensemble.  Above 1 the interval is repeated with
        twice the steps (the fine solution becomes the coarse one; so is an estimate that is not finite -- an explicit
        scheme beyond its stability limit), the accepted solution is the fine one; an estimate far below 1 halves m for
        the next interval.  ``control="positions"`` measures the position half of the state
        only (velocity components of modes far above 1 / h keep their amplitude but not their phase, which the full
        norm answers with the small steps LSODA itself takes).  Returns (states [n_t, B, 2n], accepted m per interval)."""
        if control not in ("all", "positions"):
            raise ValueError('control must be "all" or "positions"')
        n_dof = torch.as_tensor(np.asarray(self.n_per_beam, dtype=np.float64) * (2.0 if control == "all" else 1.0),
                                dtype=torch.float64, device=self.device)
        rows = slice(None) if control == "all" else slice(0, 1)

        def estimate(fine, coarse, start):
            scale = atol + rtol * torch.maximum(fine[:, rows].abs(), start[:, rows].abs())
            e = ((fine[:, rows] - coarse[:, rows]) / (float(2 ** order - 1) * scale)).double()
            return float(torch.sqrt((e * e).sum(dim=(1, 2, 3)) / n_dof).max().item())

        def run(start, t_a, length, on, m):
            self.state = start.clone()
            advance(m, length / m, t_a, on)
            return self.state

        rate = float(m_first) / dt_eval       # steps per second of the coarse solution; doubled / halved by the controller
        shrink = 0.8 / float(2 ** order)      # half the steps multiply the estimate by 2^order: still below 1 with margin
        out, used = [first], []
        for k in range(n_t - 1):
            t_k, t_n = t0 + k * dt_eval, t0 + (k + 1) * dt_eval
            pieces = [(t_k, t_n)]
            if t_switch is not None and t_k + 1e-9 * dt_eval < t_switch < t_n - 1e-9 * dt_eval:
                pieces = [(t_k, t_switch), (t_switch, t_n)]
            taken = 0
            for t_a, t_b in pieces:
                on = None if t_switch is None else (0.5 * (t_a + t_b) < t_switch)
                m = max(1, int(np.ceil(rate * (t_b - t_a) - 1e-9)))
                start = self.state
                coarse = run(start, t_a, t_b - t_a, on, m)
                while True:
                    fine = run(start, t_a, t_b - t_a, on, 2 * m)
                    err = estimate(fine, coarse, start)
                    if err <= 1.0:
                        break
                    m, coarse, rate = 2 * m, fine, 2.0 * rate
                    if 2 * m > max_substeps:
                        raise RuntimeError(f"solve_ivp: the tolerances ask for more than {max_substeps} steps per t_eval "
                                           f"interval at t = {t_a:.6g} s (error estimate {err:.3g})")
                taken += 2 * m
                self.state = fine
                if err < shrink * 0.25 and m > 1:
                    rate *= 0.5
            used.append(taken)
            self.time = t_n
            out.append(self.unpack_state().unsqueeze(0))
        return torch.cat(out, dim=0), used

    def solve_rk45(self, t_end: float, rtol: float = 1e-3, atol: float = 1e-6, impulse_amp=None,
                   impulse_duration: float = 0.01, impulse_index: int = -2, held_force=None,
                   first_step=None, t0: Optional[float] = None, max_steps: int = 0, record=None, t_eval=None):
        """Integrate every beam from the current clock to ``t_end`` with adaptive Dormand-Prince 5(4),
        per-beam step control, in ONE kernel launch -- the algorithm (and defaults) of
        ``scipy.integrate.solve_ivp(method="RK45")`` that the reference's tests call
        (tests/test_dynamic_beam.py:218-220).  Returns a dict of per-beam statistics:
        accepted / rejected steps, nfev, status (0 = reached t_end), next_step.

        record=(node, param) with t_eval=(start, step, count): also returns "y" [B, count], that DOF on the
        uniform grid start + k*step by scipy's dense output; record="all": "y" is the whole state on the grid,
        [count, B, 2, n_node, 4] (unpack_snapshots() gives the reduced ordering) (what ``sol.y[i]`` holds when solve_ivp is
        given ``t_eval=np.arange(...)``, example_utilities.py:158)."""
        if t0 is not None:
            self.time = float(t0)
        desc = nat.InputDesc()
        desc.kind = nat.CRB_INPUT_NONE
        keep = []
        if impulse_amp is not None:
            self._impulse(desc, keep, impulse_amp, impulse_duration, impulse_index)
        if held_force is not None:
            held = self.pack_vec(held_force)
            desc.f_held = held.data_ptr()
            keep.append(held)
        h = torch.zeros((self.n_beams,), dtype=torch.float64, device=self.device)
        if first_step is not None:
            h += torch.as_tensor(first_step, dtype=torch.float64, device=self.device)
        stats = torch.zeros((self.n_beams, 4), dtype=torch.int32, device=self.device)
        rec, ys, grid = None, None, (0.0, 0.0, 0)
        if isinstance(record, str) and record == "all":   # every DOF of sol.y: snapshots [count, B, 2, n_node, 4]
            grid = (float(t_eval[0]), float(t_eval[1]), int(t_eval[2]))
            ys = torch.zeros((grid[2],) + tuple(self.state.shape), dtype=self.dtype, device=self.device)
            rec = nat.RecordDesc(0, -1, 0, 1, ys.data_ptr())
        elif record is not None:
            node, param = record
            vel = param.startswith("d") and param.endswith("_dt")
            grid = (float(t_eval[0]), float(t_eval[1]), int(t_eval[2]))
            ys = torch.zeros((self.n_beams, grid[2]), dtype=self.dtype, device=self.device)
            rec = nat.RecordDesc(int(vel), int(node), _PARAM[param[1:-3] if vel else param], 1, ys.data_ptr())
        with self._on_device():
            nat.check(self._lib.crb_solve_rk45_eval(self.plan.h, self._ptr(self.state), self.time, float(t_end), float(rtol),
                                                    float(atol), C.byref(desc), self._ptr(h), self._ptr(stats),
                                                    int(max_steps), C.byref(rec) if rec is not None else None, grid[0],
                                                    grid[1], grid[2], self._stream()))
        self._keep = keep + [h, stats, ys]
        st = stats.cpu().numpy()
        out = {"accepted": st[:, 0], "rejected": st[:, 1], "nfev": st[:, 2], "status": st[:, 3],
               "next_step": h.cpu().numpy()}
        if np.any(st[:, 3] != 0):
            # a beam that gave up (step too small / max_steps) stopped BEFORE t_end: the ensemble no longer has one
            # clock, so the clock is not advanced and the caller is told (scipy reports status -1 the same way)
            import warnings

            bad = np.nonzero(st[:, 3] != 0)[0]
            warnings.warn(f"solve_rk45: {bad.size} of {self.n_beams} beams stopped before t_end (first: beam {int(bad[0])}); "
                          "ensemble clock left unchanged", RuntimeWarning, stacklevel=2)
        else:
            self.time = float(t_end)
        if ys is not None:
            out["y"] = ys
        return out

    def step_feedback(self, n_steps: int, dt: float, gain, reference=None, impulse_amp=None,
                      impulse_duration: float = 0.01, impulse_index: int = -2, t0: Optional[float] = None) -> float:
        """Closed-loop rollout: RK4 with the state feedback u = K (r - x) evaluated inside the RHS at
        EVERY stage, as examples/lqr_control.py:95-111 does through FullStateLinear.compute_input
        (control/full_state_linear.py:81), plus the optional tip impulse.

        gain       [n, 2n] LQR gain (reduced ordering, e.g. LinearQuadraticRegulator.compute_gain_matrix()) for the whole
                   ensemble, or a list of B matrices (None = no feedback for that beam): one gain per beam of a heterogeneous
                   ensemble, each in its own beam's reduced ordering; equal gains on like beams run as one group
        reference  [B, 2n] or None (= regulation to 0)
        Stage-split path: per stage one GEMM over the whole ensemble, [B, 2n] x [2n, n] -- the fused
        MFMA kernel crb_feedback_force (gather + GEMM + scatter, in the plan's dtype) -- then one launch
        of the stage kernel (crb_rk4_stage); the loop itself is crb_step_rk4_feedback.
        Note: the LQR loop is stiff (|lambda|max ~ 3e5 1/s for the Nitinol example): RK4 needs
        dt <= ~8e-6 s, not the 2e-5 s of the open-loop configs.
        """
        if t0 is not None:
            self.time = float(t0)
        if isinstance(gain, (list, tuple)):
            return self._step_feedback_grouped(n_steps, dt, gain, reference, impulse_amp, impulse_duration, impulse_index)
        K = self._dev(gain, (self.n, 2 * self.n))
        ref = None if reference is None else self._dev(reference, (self.n_beams, 2 * self.n))
        desc = nat.InputDesc()
        desc.kind = nat.CRB_INPUT_NONE
        keep = []
        if impulse_amp is not None:
            self._impulse(desc, keep, impulse_amp, impulse_duration, impulse_index)
        # the whole loop is one native call (crb_step_rk4_feedback issues every launch)
        work = torch.empty((int(self._lib.crb_feedback_work_bytes(self.plan.h)),), dtype=torch.uint8, device=self.device)
        t_end = C.c_double(0.0)
        with self._on_device():
            nat.check(self._lib.crb_step_rk4_feedback(self.plan.h, self._ptr(self.state), self.time, float(dt), int(n_steps),
                                                      self._ptr(K), self._ptr(ref), C.byref(desc), self._ptr(work),
                                                      C.byref(t_end), self._stream()))
        self._keep = keep + [K, ref, work]
        self._feedback_work = work
        self.time = float(t_end.value)
        return self.time

    def _gain_groups(self, gains):
        """Per-beam gains -> (group of every beam, one device gain per group): beams with the same free-DOF set and an equal gain
        matrix form a group (the reference designs one gain per model, lqr_control.py:46-84); None = no feedback for that beam."""
        if len(gains) != self.n_beams:
            raise ValueError(f"per-beam gains: expected {self.n_beams} matrices, got {len(gains)}")
        index, group_of, mats = {}, np.full(self.n_beams, -1, dtype=np.int32), []
        for b, K in enumerate(gains):
            if K is None:
                continue
            K = np.ascontiguousarray(K, dtype=np.float64)
            nb = int(self.n_per_beam[b])
            if K.shape != (nb, 2 * nb):
                raise ValueError(f"gain of beam {b}: expected shape {(nb, 2 * nb)} (its own reduced state), got {K.shape}")
            key = (self.free_index_per_beam[b].tobytes(), K.tobytes())
            if key not in index:
                index[key] = len(mats)
                mats.append(torch.as_tensor(K, dtype=self.dtype, device=self.device).contiguous())
            group_of[b] = index[key]
        if not mats:
            raise ValueError("per-beam gains: every entry is None")
        return group_of, mats

    def _step_feedback_grouped(self, n_steps, dt, gains, reference, impulse_amp, impulse_duration, impulse_index):
        """step_feedback with one gain per beam (heterogeneous ensembles, `from_dataframes`): grouped by free-DOF set and gain,
        one MFMA launch per group and stage (crb_step_rk4_feedback_grouped).  `reference`: [B, 2 n_max] padded reduced states."""
        group_of, mats = self._gain_groups(gains)
        ref = None if reference is None else self._dev(reference, (self.n_beams, 2 * self.n))
        desc = nat.InputDesc()
        desc.kind = nat.CRB_INPUT_NONE
        keep = []
        if impulse_amp is not None:
            self._impulse(desc, keep, impulse_amp, impulse_duration, impulse_index)
        ptrs = (C.c_void_p * len(mats))(*[m.data_ptr() for m in mats])
        work = torch.empty((int(self._lib.crb_feedback_work_bytes(self.plan.h)),), dtype=torch.uint8, device=self.device)
        t_end = C.c_double(0.0)
        with self._on_device():
            nat.check(self._lib.crb_step_rk4_feedback_grouped(
                self.plan.h, self._ptr(self.state), self.time, float(dt), int(n_steps), len(mats),
                group_of.ctypes.data_as(C.POINTER(C.c_int32)), ptrs, self._ptr(ref), C.byref(desc), self._ptr(work), C.byref(t_end),
                self._stream()))
        self._keep = keep + mats + [ref, work]
        self._feedback_work = work
        self.time = float(t_end.value)
        return self.time

    def feedback_path(self) -> str:
        """Which form `step_feedback` takes for this ensemble: "persistent" (one launch per rollout, csrc/crb_loop.h), "fused"
        (small beams, gain in LDS) or "stage-split" (one GEMM + one stage launch per RK4 stage)."""
        return {0: "stage-split", 1: "fused", 2: "persistent"}[int(self._lib.crb_feedback_path(self.plan.h))]

    def feedback_status(self) -> int:
        """0, or which hand-off of the last `step_feedback` gave up (the persistent stepper's workgroups wait for each
        other with a time limit instead of hanging the device; the state is unusable then).  Synchronises the stream."""
        work = getattr(self, "_feedback_work", None)
        if work is None:
            return 0
        status = C.c_int32(0)
        with self._on_device():
            nat.check(self._lib.crb_feedback_status(self.plan.h, self._ptr(work), C.byref(status), self._stream()))
        return int(status.value)

    # ------------------------------------------------------------------ functional composition
    def _plan_with(self, drag_on: bool, gravity_on: bool):
        """The ensemble's plan with the built-in drag / gravity terms of the RHS kernel switched on or off."""
        key = (bool(drag_on), bool(gravity_on))
        if key not in self._plans:
            fps, per = self._fps, len(self._fps) > 1
            pick = (lambda f: [f(p) for p in fps]) if per else (lambda f: f(fps[0]))
            with self._on_device():
                self._plans[key] = nat.Plan(
                    self.columns, fluid_density=pick(lambda p: p.fluid_density),
                    enable_fluid=pick(lambda p: bool(p.enable_fluid_effects and drag_on)),
                    gravity=pick(lambda p: p.get_gravity_vector()),
                    enable_gravity=pick(lambda p: bool(p.enable_gravity_effects and gravity_on)), **self._plan_args)
        return self._plans[key]

    def step_composed(self, n_steps: int, dt: float, forces_func=None, u=None, impulse_amp=None,
                      impulse_duration: float = 0.01, impulse_index: int = -2, t0: Optional[float] = None) -> float:
        """RK4 with the reference's functional force composition, batched (dynamic_beam_model.py:243-274, 343-362):
        xdot = [v ; Minv(-k(q) + forces(x, 0.0) + u(t))] with

        forces_func  ``callable(x[B, 2n], t) -> [B, n]`` (torch, on the ensemble's device) used INSTEAD of the registry,
                     as ``create_system_func(forces_func)`` does (:253-254), called with t = 0.0 (:265, quirk B-3);
                     None: the aggregate of ``self.force_registry`` -- the enabled built-in drag / gravity markers run
                     fused inside the RHS kernel, every other enabled component (``compute_forces(x, t)``, e.g. the
                     reference's StateAwareForce, tests/test_advanced_composition.py:36-65) is summed on the device;
                     ``enabled`` is re-read at EVERY evaluation (force_registry.py:66-67)
        u            ``[B, n]`` held input, or ``callable(t) -> [B, n]`` evaluated at every stage time (:357-360)
        impulse_*    the examples' tip impulse on top (as in ``step``).

        Stage-split path: per stage the state is unpacked to the reduced ordering, the callables run as torch ops,
        and one crb_rk4_stage launch takes their sum as its input force.  Returns the clock (accumulated by addition).
        """
        if t0 is not None:
            self.time = float(t0)
        desc = nat.InputDesc()
        desc.kind = nat.CRB_INPUT_NONE
        keep = []
        if impulse_amp is not None:
            self._impulse(desc, keep, impulse_amp, impulse_duration, impulse_index)
        acc, bufs = torch.empty_like(self.state), (torch.empty_like(self.state), torch.empty_like(self.state))
        zeros = torch.zeros((self.n_beams, self.n_node, 4), dtype=self.dtype, device=self.device)
        u_held = None if (u is None or callable(u)) else self._dev(u, (self.n_beams, self.n))
        t = self.time
        dt = float(dt)
        with self._on_device():
            for _ in range(int(n_steps)):
                cur = self.state
                for s, ts in enumerate((t, t + 0.5 * dt, t + 0.5 * dt, t + dt)):   # crb_step_rk4's clock convention
                    total = None
                    if forces_func is not None:
                        plan = self._plan_with(False, False)
                        total = self._force_tensor(forces_func(self.unpack_state(cur), 0.0))
                    else:
                        drag_on = grav_on = False
                        x_red = None
                        for force in self.force_registry.get_registered_forces():
                            if not force.is_enabled():
                                continue
                            if force is self._auto_drag:
                                drag_on = True
                            elif force is self._auto_gravity:
                                grav_on = True
                            else:
                                if x_red is None:
                                    x_red = self.unpack_state(cur)
                                part = self._force_tensor(force.compute_forces(x_red, 0.0))
                                total = part.clone() if total is None else total + part
                        plan = self._plan_with(drag_on, grav_on)
                    if u is not None:
                        ut = self._force_tensor(u(ts)) if callable(u) else u_held
                        total = ut if total is None else total + ut
                    u_stage = zeros if total is None else self.pack_vec(total)
                    nxt = bufs[s & 1]
                    nat.check(self._lib.crb_rk4_stage(plan.h, self._ptr(self.state), self._ptr(cur), self._ptr(acc),
                                                      self._ptr(nxt), self._ptr(u_stage), s, float(ts), dt, C.byref(desc),
                                                      self._stream()))
                    cur = nxt
                t = t + dt
        self._keep = keep + [acc, bufs, zeros]
        self.time = t
        return self.time

    def _force_tensor(self, f) -> torch.Tensor:
        """A user callable's result as a [B, n] tensor of the ensemble's dtype and device; the wrong shape is a
        ValueError (the reference's M_inv.dot raises a dimension mismatch, dynamic_beam_model.py:270)."""
        f = torch.as_tensor(f, dtype=self.dtype, device=self.device)
        if tuple(f.shape) != (self.n_beams, self.n):
            raise ValueError(f"dimension mismatch: force of shape {tuple(f.shape)} for {(self.n_beams, self.n)} position DOFs")
        return f.contiguous()

    def gather(self, node: int, param: str, velocity: bool = False) -> torch.Tensor:
        out = torch.empty((self.n_beams,), dtype=self.dtype, device=self.device)
        with self._on_device():
            nat.check(self._lib.crb_gather_dof(self.plan.h, self._ptr(self.state), int(velocity), int(node),
                                               _PARAM[param], self._ptr(out), self._stream()))
        return out

    def tip_displacement(self) -> torch.Tensor:
        """w of the last node of every beam (the examples' 'tip displacement', lqr_control.py:168)."""
        if len(set(self.n_elem_per_beam.tolist())) == 1:
            return self.gather(self.n_elem, "w")
        rows = torch.arange(self.n_beams, device=self.device)      # beams of different length: each beam's own last node
        nodes = torch.as_tensor(self.n_elem_per_beam, device=self.device)
        return self.state[rows, 0, nodes, 1].clone()
