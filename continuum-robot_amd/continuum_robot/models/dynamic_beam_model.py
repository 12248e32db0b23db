"""DynamicEulerBernoulliBeam: the state-space model x' = [v ; Minv(-k(q) + f(x) + u)].

Reference: src/continuum_robot/models/dynamic_beam_model.py:16-364 -- same constructor
(CSV + ForceParams), attributes and closure API (create_system_func / create_input_func /
get_system_func / get_dynamic_system), same validation messages.

Where the work runs.  ``system(x)`` is one launch of the RHS kernel (crb_rhs): k(q), the mass solve
and -- for the registry-based default -- the auto-registered drag and gravity terms all run fused on
the GPU (the registry is re-read on every call, so toggling ``force.enabled`` or changing the gravity
vector still takes effect immediately).  Only what cannot be lowered is evaluated on the host, as the
functional-composition API demands: a user-supplied ``forces_func`` and user-registered force objects
are arbitrary Python callables; their sum enters the kernel as its input vector.  Rollouts of many
beams should use ``to_ensemble()`` / ``continuum_robot.batched.BeamEnsemble``.
"""
import pathlib
from typing import Callable, Dict, Union

import numpy as np
import pandas as pd
from scipy import sparse

from .abstractions import BoundaryConditionType, ElementType
from .euler_bernoulli_beam import EulerBernoulliBeam
from .fluid_forces import FluidDragForce
from .force_params import ForceParams
from .force_registry import ForceRegistry, InputRegistry
from .gravity_forces import GravityForce

_BASE_COLS = ["length", "elastic_modulus", "moment_inertia", "density", "cross_area", "type", "boundary_condition"]


class DynamicEulerBernoulliBeam:
    def __init__(self, filename: Union[str, pathlib.Path], force_params: ForceParams = None):
        self.force_params = force_params or ForceParams()
        self.params = pd.read_csv(filename)
        self._validate_parameters()
        self.boundary_conditions = self._process_boundary_conditions()

        self.beam_model = EulerBernoulliBeam(self.params)
        self.beam_model.apply_boundary_conditions(self.boundary_conditions)
        self.constrained_dofs = self.beam_model.get_constrained_dofs()

        self._M_inv = None
        self.system_func = None
        self.input_func = None
        self.force_registry = ForceRegistry()
        self.input_registry = InputRegistry()
        self._initialize_state_mapping()
        self._auto_register_forces()

    # ------------------------------------------------------------------ construction helpers
    def _validate_parameters(self) -> None:
        required = list(_BASE_COLS)
        if self.force_params.enable_fluid_effects:
            required += ["wetted_area", "drag_coef"]
        if not all(col in self.params.columns for col in required):
            raise ValueError(f"CSV must contain columns: {', '.join(required)}")
        invalid_types = set(self.params["type"].str.lower()) - {t.value for t in ElementType}
        if invalid_types:
            raise ValueError(f"Invalid element types: {invalid_types}")
        invalid_bcs = set(self.params["boundary_condition"]) - {"FIXED", "PINNED", "NONE"}
        if invalid_bcs:
            raise ValueError(f"Invalid boundary conditions: {invalid_bcs}")
        if self.force_params.enable_fluid_effects:
            if self.force_params.fluid_density <= 0:
                raise ValueError("Fluid density must be positive")
            if (self.params["drag_coef"] < 0).any():
                raise ValueError("Drag coefficients cannot be negative")
            if (self.params["wetted_area"] < 0).any():
                raise ValueError("Wetted areas cannot be negative")

    def _process_boundary_conditions(self) -> Dict[int, BoundaryConditionType]:
        """CSV row i constrains NODE i; the last node cannot be constrained through the CSV."""
        named = {"FIXED": BoundaryConditionType.FIXED, "PINNED": BoundaryConditionType.PINNED}
        conditions = {i: named[bc] for i, bc in enumerate(self.params["boundary_condition"]) if bc in named}
        if len(conditions) == len(self.params) + 1:
            raise ValueError("Cannot constrain all nodes with boundary conditions")
        return conditions

    def _initialize_state_mapping(self):
        """state = [positions ; velocities]; velocity names are d<param>_dt."""
        pos = self.beam_model.dof_to_node_param
        n = len(pos)
        self.state_to_node_param = dict(pos)
        self.state_to_node_param.update({i + n: (f"d{p}_dt", node) for i, (p, node) in pos.items()})
        self.node_param_to_state = {v: k for k, v in self.state_to_node_param.items()}
        self._original_state_to_node_param = self.state_to_node_param.copy()
        self._original_node_param_to_state = self.node_param_to_state.copy()

    def _auto_register_forces(self) -> None:
        fp = self.force_params
        self._auto_drag = self._auto_gravity = None
        self._fused = {}
        if fp.enable_fluid_effects:
            self._auto_drag = FluidDragForce(
                fluid_data=self.params[["wetted_area", "drag_coef"]], state_mapping=self.state_to_node_param,
                fluid_density=fp.fluid_density, enabled=True)
            self.force_registry.register(self._auto_drag)
        if fp.enable_gravity_effects:
            self._auto_gravity = GravityForce(
                beam_params=self.params[["density", "cross_area", "length"]],
                gravity_vector=fp.get_gravity_vector(), enabled=True)
            self.force_registry.register(self._auto_gravity)

    # ------------------------------------------------------------------ attributes
    @property
    def M_inv(self):
        """Explicit inverse of the reduced mass matrix (scipy sparse), built on first access.
        The reference forms it eagerly (dynamic_beam_model.py:60) and multiplies by it three times
        per RHS; the kernels solve with the cyclic-reduction factors instead, so this is only kept
        for code that reads the attribute."""
        if self._M_inv is None:
            from scipy.sparse.linalg import inv

            self._M_inv = inv(sparse.csc_matrix(self.beam_model.M))
        return self._M_inv

    def get_state_to_node_param(self, state_idx):
        if state_idx not in self.state_to_node_param:
            raise KeyError(f"Invalid state index: {state_idx}")
        return self.state_to_node_param[state_idx]

    def get_state_index(self, node_idx, param):
        if (param, node_idx) not in self.node_param_to_state:
            raise KeyError(f"Invalid node/parameter combination: ({node_idx}, {param})")
        return self.node_param_to_state[(param, node_idx)]

    def get_state_mapping(self):
        return self.state_to_node_param.copy()

    def get_node_param_mapping(self):
        return self.node_param_to_state.copy()

    # ------------------------------------------------------------------ device path
    def _fused_ensemble(self, drag_on: bool, gravity):
        """One-beam GPU ensemble with the given built-in force terms fused into the RHS kernel."""
        key = (bool(drag_on), None if gravity is None else tuple(float(g) for g in gravity))
        if key not in self._fused:
            from ..batched import BeamEnsemble

            fp = ForceParams(fluid_density=self.force_params.fluid_density if drag_on else 0.0,
                             enable_fluid_effects=bool(drag_on),
                             gravity_vector=list(key[1]) if key[1] is not None else [0.0, -9.81, 0.0],
                             enable_gravity_effects=key[1] is not None)
            self._fused[key] = BeamEnsemble(self.params, 1, force_params=fp)
        return self._fused[key]

    def _registry_rhs(self, x: np.ndarray, u: np.ndarray = None) -> np.ndarray:
        """system(x) of the registry-based default: enabled auto-registered drag / gravity are lowered to
        the kernel's fused force terms, every other registered force is a Python callable summed on the
        host (force time argument 0.0, reference :265)."""
        drag_on, gravity, extra = False, None, None
        for force in self.force_registry.get_registered_forces():
            if not force.is_enabled():
                continue
            if force is self._auto_drag and force.fluid_coefficients is not None:
                drag_on = True
            elif force is self._auto_gravity:
                gravity = force.get_gravity_vector()
            else:
                part = np.asarray(force.compute_forces(x, 0.0), dtype=np.float64)
                extra = part.copy() if extra is None else extra + part
        if u is not None:
            extra = np.asarray(u, dtype=np.float64) if extra is None else extra + u
        ens = self._fused_ensemble(drag_on, gravity)
        n = ens.n
        x = np.asarray(x, dtype=np.float64)
        if x.ndim != 1 or x.shape[0] != 2 * n:
            raise ValueError(f"State vector length {x.shape} must be {2 * n}")
        if extra is not None and extra.shape != (n,):
            raise ValueError(f"dimension mismatch: force vector of length {extra.shape} for {n} position DOFs")
        # one launch on host vectors (crb_rhs_host): the closures are called ~6e5 times per simulated second by LSODA
        return ens.plan.rhs_host(x, extra)

    def _structural_rhs(self, x: np.ndarray, generalized_force: np.ndarray) -> np.ndarray:
        """[v ; Minv(-k(q) + force)] by one launch of the RHS kernel (one beam)."""
        ens = self.beam_model._device_ensemble()
        n = ens.n
        x = np.asarray(x, dtype=np.float64)
        force = np.asarray(generalized_force, dtype=np.float64)
        if x.ndim != 1 or x.shape[0] != 2 * n:
            raise ValueError(f"State vector length {x.shape} must be {2 * n}")
        if force.ndim != 1 or force.shape[0] != n:
            raise ValueError(f"dimension mismatch: force vector of length {force.shape} for {n} position DOFs")
        return ens.plan.rhs_host(x, force)

    def to_ensemble(self, n_beams: int, **kwargs):
        """The fused batched stepper for this model (drag / gravity inside the kernel)."""
        from ..batched import BeamEnsemble

        return BeamEnsemble(self.params, n_beams, force_params=self.force_params, **kwargs)

    # ------------------------------------------------------------------ closure API
    def create_system_func(self, forces_func: Callable = None) -> None:
        """system(x) = [v ; Minv(-k(q) + forces_func(x, 0.0))]; forces default to the registry.
        (The force time argument is always 0.0, as in the reference, :265.)"""
        if forces_func is None:
            self.system_func = self._registry_rhs  # built-in forces fused on the GPU
            return

        def system(x):
            additional = forces_func(x, 0.0)  # arbitrary user callable: host by construction
            return self._structural_rhs(x, np.asarray(additional, dtype=np.float64))

        self.system_func = system

    def create_input_func(self) -> None:
        """input_function(x, u, t) = [0 ; Minv u] with the reference's argument checks (:309-321)."""

        def input_function(x: np.ndarray, u: np.ndarray, t: float = 0.0) -> np.ndarray:
            if not isinstance(x, np.ndarray) or not isinstance(u, np.ndarray):
                raise ValueError("State and input must be numpy arrays")
            if x.ndim != 1 or u.ndim != 1:
                raise ValueError("State and input must be 1D arrays")
            n = len(x) // 2
            if len(u) != n:
                raise ValueError(
                    f"Input vector length {len(u)} must match position DOFs {n}. Expected {n}, got {len(u)}")
            # Minv u alone: the RHS kernel at the zero state (k(0) = 0, no drag, no gravity in this plan)
            out = self._structural_rhs(np.zeros(2 * n), u)
            out[:n] = 0.0
            return out

        self.input_func = input_function
        self._default_input_func = input_function

    def get_system_func(self) -> Callable:
        if self.system_func is None:
            raise RuntimeError("System function not yet created")
        return self.system_func

    def get_dynamic_system(self) -> Callable:
        if self.system_func is None or self.input_func is None:
            raise RuntimeError("System and input functions must be created first")

        def dynamic_system(t: float, x: np.ndarray, u: Union[np.ndarray, Callable]) -> np.ndarray:
            force = u(t) if callable(u) else u
            # The default closures (registry forces, plain input): ONE right-hand side launch with the input added
            # before the single mass solve -- Minv(-k + f + u) instead of Minv(-k + f) + Minv u, a rounding-level
            # difference for half the launches of every solve_ivp RHS call.  Looked up per call, like
            # the reference's closure: replacing system_func / input_func afterwards still takes effect.
            if self.system_func == self._registry_rhs and self.input_func is getattr(self, "_default_input_func", None):
                return self._fused_call(x, force)
            return self.system_func(x) + self.input_func(x, force, t)

        return dynamic_system

    def get_composed_dynamic_system(self) -> Callable:
        """The literal two-call composition ``system(x) + input(x, u, t)`` of the reference
        (dynamic_beam_model.py:343-362): two launches, two mass solves."""
        if self.system_func is None or self.input_func is None:
            raise RuntimeError("System and input functions must be created first")

        def dynamic_system(t: float, x: np.ndarray, u: Union[np.ndarray, Callable]) -> np.ndarray:
            force = u(t) if callable(u) else u
            return self.system_func(x) + self.input_func(x, force, t)

        return dynamic_system

    def _fused_call(self, x, force):
        if not isinstance(x, np.ndarray) or not isinstance(force, np.ndarray):
            raise ValueError("State and input must be numpy arrays")
        if x.ndim != 1 or force.ndim != 1:
            raise ValueError("State and input must be 1D arrays")
        n = len(x) // 2
        if len(force) != n:
            raise ValueError(
                f"Input vector length {len(force)} must match position DOFs {n}. Expected {n}, got {len(force)}")
        return self._registry_rhs(x, force)

    def get_fused_dynamic_system(self) -> Callable:
        """``dynamic_system(t, x, u)`` with the registry forces AND the input in ONE right-hand side launch:
        [v ; Minv(-k(q) + f(x, 0) + u)] -- the sum ``system(x) + input(x, u, t)`` of get_dynamic_system() with a
        single mass solve (rounding-level difference, one launch instead of two per call).  Same argument checks."""
        if self.system_func is None or self.input_func is None:
            raise RuntimeError("System and input functions must be created first")
        if self.system_func != self._registry_rhs:
            return self.get_dynamic_system()      # a user forces_func: keep the two-call composition

        def dynamic_system(t: float, x: np.ndarray, u: Union[np.ndarray, Callable]) -> np.ndarray:
            force = u(t) if callable(u) else u
            if not isinstance(x, np.ndarray) or not isinstance(force, np.ndarray):
                raise ValueError("State and input must be numpy arrays")
            if x.ndim != 1 or force.ndim != 1:
                raise ValueError("State and input must be 1D arrays")
            n = len(x) // 2
            if len(force) != n:
                raise ValueError(
                    f"Input vector length {len(force)} must match position DOFs {n}. Expected {n}, got {len(force)}")
            return self._registry_rhs(x, force)

        return dynamic_system
