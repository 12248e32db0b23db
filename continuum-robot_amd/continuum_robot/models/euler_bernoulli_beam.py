"""EulerBernoulliBeam: the global finite-element beam (the reference's "unified beam system").

Reference: src/continuum_robot/models/euler_bernoulli_beam.py:16-511 -- same constructor,
attributes (``parameters``, ``segments``, ``M``, ``stiffness_func``, DOF maps) and methods.
What differs is where the arithmetic runs:
  * mass / stiffness matrices: assembled by libcrbeam's plan builder (crb_plan_get_mass /
    crb_plan_get_stiffness, the code that also builds the kernels' tables); scipy CSR / numpy on
    the way out, as the reference returns them;
  * k(q) (get_stiffness_function): the element + assembly kernel on the GPU
    (crb_internal_force), fed and read in the reference's reduced ordering.
Constrained DOFs are masked inside the plan rather than deleted from arrays; the reduced
ordering seen by callers is the reference's (ascending free DOF index).
"""
import pathlib
from typing import Callable, Dict, List, Set, Union

import numpy as np
import pandas as pd
from scipy import sparse

from .. import _native as nat
from .abstractions import BoundaryConditionType, ElementType, IBeam, create_properties_from_dataframe
from .segments import SegmentFactory

_REQUIRED = ["length", "elastic_modulus", "moment_inertia", "density", "cross_area", "type"]
_PARAMS = ("u", "w", "phi")


class EulerBernoulliBeam(IBeam):
    def __init__(self, parameters: Union[str, pathlib.Path, pd.DataFrame]):
        if isinstance(parameters, (str, pathlib.Path)):
            try:
                self.parameters = pd.read_csv(parameters)
            except FileNotFoundError:
                raise FileNotFoundError(f"Parameter file {parameters} not found")
        elif isinstance(parameters, pd.DataFrame):
            self.parameters = parameters.copy()
        else:
            raise TypeError("Parameters must be filepath or pandas DataFrame")
        self._validate_parameters()

        factory = SegmentFactory()
        self.segments = [factory.create_segment(create_properties_from_dataframe(self.parameters, i))
                         for i in range(len(self.parameters))]
        super().__init__(self.segments)

        self.M = None
        self.stiffness_func = None
        self._initialize_dof_mapping()
        self._boundary_conditions: Dict[int, BoundaryConditionType] = {}
        self._boundary_conditions_applied = False
        self._constrained_dofs: Set[int] = set()
        self._host_plan = None
        self._ensemble = None

        self.assemble_mass_matrix()
        self.stiffness_func = self.create_stiffness_function()

    # ------------------------------------------------------------------ validation / maps
    def _validate_parameters(self) -> None:
        if not all(col in self.parameters.columns for col in _REQUIRED):
            raise ValueError(f"DataFrame must contain columns: {', '.join(_REQUIRED)}")
        if (self.parameters[_REQUIRED[:-1]] <= 0).any().any():
            raise ValueError("All numeric parameters must be positive")
        invalid = set(self.parameters["type"].str.lower()) - {t.value for t in ElementType}
        if invalid:
            raise ValueError(f"Invalid element types: {invalid}")

    def _initialize_dof_mapping(self):
        n_nodes = len(self.parameters) + 1
        self.dof_to_node_param = {3 * node + k: (p, node) for node in range(n_nodes) for k, p in enumerate(_PARAMS)}
        self.node_param_to_dof = {v: k for k, v in self.dof_to_node_param.items()}
        self._original_dof_to_node_param = self.dof_to_node_param.copy()
        self._original_node_param_to_dof = self.node_param_to_dof.copy()

    def _update_dof_mapping(self):
        if not self._boundary_conditions_applied:
            return
        free = [d for d in sorted(self._original_dof_to_node_param) if d not in self._constrained_dofs]
        self.dof_to_node_param = {new: self._original_dof_to_node_param[old] for new, old in enumerate(free)}
        self.node_param_to_dof = {v: k for k, v in self.dof_to_node_param.items()}

    # ------------------------------------------------------------------ native plans
    def _columns(self):
        p = self.parameters
        cols = {c: p[c].to_numpy() for c in _REQUIRED}
        cols["boundary_condition"] = ["NONE"] * len(p)
        return cols

    def _node_bc(self) -> np.ndarray:
        codes = np.zeros(len(self.parameters) + 1, dtype=np.uint8)
        for node, bc in self._boundary_conditions.items():
            codes[node] = nat.CRB_BC_FIXED if bc == BoundaryConditionType.FIXED else nat.CRB_BC_PINNED
        return codes

    def _plan(self):
        """Host-side plan of the current boundary-condition set (matrices, index maps)."""
        if self._host_plan is None:
            self._host_plan = nat.Plan(self._columns(), node_bc=self._node_bc(), device=-1)
        return self._host_plan

    def _device_ensemble(self):
        """One-beam GPU ensemble of the current boundary-condition set (k(q) evaluation)."""
        if self._ensemble is None:
            from ..batched import BeamEnsemble

            self._ensemble = BeamEnsemble(self._columns(), 1, node_bc=self._node_bc())
        return self._ensemble

    def _invalidate(self):
        self._host_plan = None
        self._ensemble = None

    # ------------------------------------------------------------------ assembly
    def assemble_mass_matrix(self) -> np.ndarray:
        """Global consistent mass matrix (scipy CSR), reduced when boundary conditions are applied."""
        self.M = sparse.csr_matrix(self._plan().mass())
        return self.M

    def create_stiffness_function(self) -> Callable:
        """k(q): global internal force of the (possibly mixed linear / nonlinear) beam."""

        def global_stiffness_function(x: np.ndarray) -> np.ndarray:
            ens = self._device_ensemble()
            q = np.asarray(x, dtype=np.float64)
            if q.shape != (ens.n,):
                raise ValueError(f"expected shape {(ens.n,)}, got {q.shape}")
            return ens.plan.internal_force_host(q)    # one launch on host vectors (crb_internal_force_host)

        return global_stiffness_function

    def apply_boundary_conditions(self, conditions: Dict[int, BoundaryConditionType]) -> None:
        if self.M is None or self.stiffness_func is None:
            raise RuntimeError("Matrices must be created before applying boundary conditions")
        n_nodes = len(self.parameters) + 1
        for node_idx in conditions:
            if node_idx < 0 or node_idx >= n_nodes:
                raise ValueError(f"Node index {node_idx} out of range [0, {n_nodes-1}]")
        for bc_type in conditions.values():
            if bc_type not in (BoundaryConditionType.FIXED, BoundaryConditionType.PINNED):
                raise ValueError(f"Unsupported boundary condition type: {bc_type}")
        merged = dict(self._boundary_conditions)
        merged.update(conditions)
        constrained = set()
        for node_idx, bc_type in merged.items():
            base = 3 * node_idx
            constrained.update((base, base + 1) if bc_type == BoundaryConditionType.PINNED
                               else (base, base + 1, base + 2))
        if len(constrained) == 3 * n_nodes:
            raise ValueError("Cannot constrain all degrees of freedom")
        self._boundary_conditions = merged
        self._constrained_dofs = constrained
        self._unconstrained_dofs = sorted(set(range(3 * n_nodes)) - constrained)
        self._boundary_conditions_applied = True
        self._invalidate()
        self.assemble_mass_matrix()
        self.stiffness_func = self.create_stiffness_function()
        self._update_dof_mapping()

    def clear_boundary_conditions(self) -> None:
        if self.M is None or self.stiffness_func is None:
            raise RuntimeError("Matrices must be created before clearing boundary conditions")
        self._boundary_conditions.clear()
        self._constrained_dofs.clear()
        self._boundary_conditions_applied = False
        self._invalidate()
        self.assemble_mass_matrix()
        self.stiffness_func = self.create_stiffness_function()
        self.dof_to_node_param = self._original_dof_to_node_param.copy()
        self.node_param_to_dof = self._original_node_param_to_dof.copy()

    # ------------------------------------------------------------------ accessors
    def get_constrained_dofs(self) -> List[int]:
        return list(self._constrained_dofs)

    def get_boundary_conditions(self) -> Dict[int, BoundaryConditionType]:
        return self._boundary_conditions.copy()

    def has_boundary_conditions(self) -> bool:
        return self._boundary_conditions_applied

    def get_mass_matrix(self) -> np.ndarray:
        if self.M is None:
            raise RuntimeError("Mass matrix not yet created")
        return self.M.toarray()

    def get_stiffness_function(self) -> Callable:
        if self.stiffness_func is None:
            raise RuntimeError("Stiffness function not yet created")
        return self.stiffness_func

    def get_length(self) -> float:
        return self.parameters["length"].sum()

    def get_segment_count(self) -> int:
        return len(self.segments)

    def get_segment_types(self) -> List[ElementType]:
        return [s.get_element_type() for s in self.segments]

    def is_hybrid(self) -> bool:
        return len(set(self.get_segment_types())) > 1

    def get_dof_to_node_param(self, dof_idx: int):
        if dof_idx not in self.dof_to_node_param:
            raise KeyError(f"Invalid DOF index: {dof_idx}")
        return self.dof_to_node_param[dof_idx]

    def get_dof_index(self, node_idx: int, param: str):
        if (param, node_idx) not in self.node_param_to_dof:
            raise KeyError(f"Invalid node/parameter combination: ({node_idx}, {param})")
        return self.node_param_to_dof[(param, node_idx)]

    def get_stiffness_matrix(self) -> np.ndarray:
        """Dense K of an all-linear beam (reduced when boundary conditions are applied)."""
        if self.M is None:
            raise RuntimeError("Mass matrix must be assembled before extracting stiffness matrix")
        for segment in self.segments:
            if segment.get_element_type() != ElementType.LINEAR:
                raise ValueError(
                    f"Cannot extract stiffness matrix from beam with nonlinear segments. "
                    f"Segment {segment.segment_id} is {segment.get_element_type().value}. "
                    "Stiffness matrix is only valid for purely linear beams.")
        return self._plan().stiffness()
