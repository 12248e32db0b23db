"""ctypes binding of libcrbeam.so (include/crbeam.h).

The library is built in-tree by ``make -C continuum-robot_amd`` (or ``__graft_entry__.build()``)
into ``continuum_robot/_lib/libcrbeam.so``.  There is no CPU implementation of the stepper:
if the library is missing, or a launch is attempted without a HIP device, this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CRB_LIB_PATH: point at another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("CRB_LIB_PATH") or os.path.join(_HERE, "_lib", "libcrbeam.so")

CRB_OK, CRB_EINVAL, CRB_EHIP, CRB_ENODEV, CRB_EUNSUPPORTED = 0, -1, -2, -3, -4
CRB_F64, CRB_F32 = 0, 1
CRB_BC_NONE, CRB_BC_FIXED, CRB_BC_PINNED = 0, 1, 2
CRB_FORCE_DRAG, CRB_FORCE_GRAVITY, CRB_CORRECTED_AXIAL = 1, 2, 4
CRB_INPUT_NONE, CRB_INPUT_IMPULSE = 0, 1

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


class BeamDesc(C.Structure):
    _fields_ = [
        ("n_elem", C.c_int32),
        ("length", _dp),
        ("elastic_modulus", _dp),
        ("moment_inertia", _dp),
        ("density", _dp),
        ("cross_area", _dp),
        ("nonlinear", _u8p),
        ("node_bc", _u8p),
        ("wetted_area", _dp),
        ("drag_coef", _dp),
        ("fluid_density", C.c_double),
        ("gravity", C.c_double * 3),
        ("flags", C.c_uint32),
    ]


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "n_beams", "n_elem", "n_node", "n_free", "node_offset", "n_slots", "beams_per_group", "threads",
        "pcr_levels", "pcr_levels_full", "mixed_topology")]


class InputDesc(C.Structure):
    _fields_ = [
        ("kind", C.c_int32),
        ("node", C.c_int32),
        ("dof", C.c_int32),
        ("reserved", C.c_int32),
        ("duration", C.c_double),
        ("amp", C.c_void_p),
        ("f_held", C.c_void_p),
        ("node_b", C.c_void_p),
    ]


class RecordDesc(C.Structure):
    _fields_ = [("plane", C.c_int32), ("node", C.c_int32), ("dof", C.c_int32), ("every", C.c_int32), ("out", C.c_void_p)]


class ControlDesc(C.Structure):
    _fields_ = [("rtol", C.c_double), ("atol", C.c_double), ("first_rate", C.c_double), ("positions_only", C.c_int32),
                ("n_iter", C.c_int32), ("max_rungs", C.c_int32), ("per_wave", C.c_int32), ("series_plane", C.c_int32),
                ("series_node", C.c_int32), ("series_dof", C.c_int32), ("series_pad", C.c_int32), ("series_out", C.c_void_p)]


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libcrbeam error {code}: {message}")
        self.code = code


_lib = None


def load():
    """Load libcrbeam.so; raises if it has not been built (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first (make -C continuum-robot_amd, or "
            "__graft_entry__.build()).  The beam stepper has no CPU fallback.")
    # libcrbeam.so needs libamdhip64.so.7.  PyTorch-ROCm bundles its own copy under the same
    # SONAME, and the process must hold exactly ONE HIP runtime (device pointers and streams come
    # from torch): import torch first so that its runtime is the one both sides resolve to.
    import torch  # noqa: F401

    torch_hip = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(torch_hip):
        C.CDLL(torch_hip, mode=C.RTLD_GLOBAL)
    L = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    L.crb_version.restype = i32
    L.crb_last_error.restype = C.c_char_p
    L.crb_plan_create.argtypes = [C.POINTER(vp), i32, i32, i32, C.POINTER(BeamDesc)]
    L.crb_plan_create_ensemble.argtypes = [C.POINTER(vp), i32, i32, i32, C.POINTER(BeamDesc)]
    L.crb_plan_destroy.argtypes = [vp]
    L.crb_plan_destroy.restype = None
    L.crb_plan_get_layout.argtypes = [vp, C.POINTER(Layout)]
    L.crb_plan_get_free_index.argtypes = [vp, C.POINTER(C.c_int32)]
    L.crb_plan_get_beam_info.argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.crb_plan_get_beam_free_index.argtypes = [vp, i32, C.POINTER(C.c_int32)]
    L.crb_plan_get_pcr_tables.argtypes = [vp, _dp, _dp, _dp]
    L.crb_plan_get_slot_tables.argtypes = [vp, _dp, _dp, _dp, C.POINTER(C.c_int16), C.POINTER(C.c_int32)]
    L.crb_plan_get_mass.argtypes = [vp, _dp]
    L.crb_plan_get_stiffness.argtypes = [vp, _dp]
    for name in ("crb_pack_state", "crb_unpack_state", "crb_pack_vec", "crb_unpack_vec", "crb_internal_force"):
        getattr(L, name).argtypes = [vp, vp, vp, vp]
    L.crb_rhs.argtypes = [vp, vp, vp, vp, vp]
    L.crb_rhs_host.argtypes = [vp, vp, vp, vp]             # (addresses as integers: this is the per-RHS-call path)
    L.crb_internal_force_host.argtypes = [vp, vp, vp]
    L.crb_step_rk4.argtypes = [vp, vp, C.c_double, C.c_double, i32, C.POINTER(InputDesc), _dp, vp]
    L.crb_step_rk4_rec.argtypes = [vp, vp, C.c_double, C.c_double, i32, C.POINTER(InputDesc), C.POINTER(RecordDesc), _dp,
                                   vp]
    L.crb_gather_dof.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.crb_solve_rk45.argtypes = [vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(InputDesc), vp, vp, i32,
                                 vp]
    L.crb_solve_rk45_eval.argtypes = [vp, vp, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(InputDesc), vp, vp,
                                      i32, C.POINTER(RecordDesc), C.c_double, C.c_double, i32, vp]
    L.crb_feedback_force.argtypes = [vp, vp, vp, vp, vp, vp]
    L.crb_step_implicit.argtypes = [vp, vp, C.c_double, C.c_double, i32, i32, C.POINTER(InputDesc),
                                    C.POINTER(RecordDesc), C.POINTER(C.c_double), vp]
    L.crb_step_implicit_damped.argtypes = [vp, vp, C.c_double, C.c_double, i32, i32, C.c_double, C.POINTER(InputDesc),
                                           C.POINTER(RecordDesc), C.POINTER(C.c_double), vp]
    L.crb_solve_controlled.argtypes = [vp, vp, C.c_double, C.c_double, i32, C.POINTER(ControlDesc), C.POINTER(InputDesc), vp, vp,
                                       vp, vp, vp, vp]
    L.crb_rk4_stage.argtypes = [vp, vp, vp, vp, vp, vp, i32, C.c_double, C.c_double, C.POINTER(InputDesc), vp]
    L.crb_feedback_work_bytes.argtypes = [vp]
    L.crb_feedback_status.argtypes = [vp, vp, C.POINTER(C.c_int32), vp]
    L.crb_feedback_path.argtypes = [vp]
    L.crb_feedback_force_grouped.argtypes = [vp, vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), vp, vp, vp]
    L.crb_step_rk4_feedback_grouped.argtypes = [vp, vp, C.c_double, C.c_double, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p), vp,
                                                C.POINTER(InputDesc), vp, C.POINTER(C.c_double), vp]
    L.crb_plan_set_status.argtypes = [vp, vp, C.c_longlong]
    L.crb_feedback_work_bytes.restype = C.c_size_t
    L.crb_step_rk4_feedback.argtypes = [vp, vp, C.c_double, C.c_double, i32, vp, vp, C.POINTER(InputDesc), vp,
                                        C.POINTER(C.c_double), vp]
    _lib = L
    return L


def check(rc):
    if rc != CRB_OK:
        raise NativeError(rc, load().crb_last_error().decode("utf-8", "replace"))


_BC_CODES = {"NONE": CRB_BC_NONE, "FIXED": CRB_BC_FIXED, "PINNED": CRB_BC_PINNED}


def node_bc_from_column(boundary_condition):
    """CSV column -> per-node codes: row i applies to node i, the last node stays free
    (reference: dynamic_beam_model.py:205-218)."""
    codes = [_BC_CODES[str(b).upper()] for b in boundary_condition]
    return np.asarray(codes + [CRB_BC_NONE], dtype=np.uint8)


def _beam_desc(columns, node_bc, fluid_density, enable_fluid, gravity, enable_gravity, corrected_axial):
    """crb_beam_desc for one beam + the numpy arrays it points into (keep them alive)."""
    f8 = lambda v: np.ascontiguousarray(v, dtype=np.float64)  # noqa: E731
    cols = {k: f8(columns[k]) for k in ("length", "elastic_modulus", "moment_inertia", "density", "cross_area")}
    n = cols["length"].shape[0]
    bad = [str(t) for t in columns["type"] if str(t).lower() not in ("linear", "nonlinear")]
    if bad:
        raise ValueError(f"Invalid element types: {set(bad)}")
    nl = np.ascontiguousarray([1 if str(t).lower() == "nonlinear" else 0 for t in columns["type"]], dtype=np.uint8)
    if node_bc is None:
        node_bc = node_bc_from_column(columns["boundary_condition"])
    bc = np.ascontiguousarray(node_bc, dtype=np.uint8)
    if bc.shape != (n + 1,):
        raise ValueError("node_bc must have n_elem + 1 entries")
    has_fluid_cols = "wetted_area" in columns and "drag_coef" in columns and columns["wetted_area"] is not None
    wet = f8(columns["wetted_area"]) if has_fluid_cols else None
    cd = f8(columns["drag_coef"]) if has_fluid_cols else None
    d = BeamDesc()
    d.n_elem = n
    for k, a in cols.items():
        setattr(d, k, a.ctypes.data_as(_dp))
    d.nonlinear = nl.ctypes.data_as(_u8p)
    d.node_bc = bc.ctypes.data_as(_u8p)
    d.wetted_area = wet.ctypes.data_as(_dp) if wet is not None else None
    d.drag_coef = cd.ctypes.data_as(_dp) if cd is not None else None
    d.fluid_density = float(fluid_density)
    g = np.asarray(gravity, dtype=np.float64)
    d.gravity[0], d.gravity[1], d.gravity[2] = float(g[0]), float(g[1]), float(g[2])
    d.flags = ((CRB_FORCE_DRAG if enable_fluid else 0) | (CRB_FORCE_GRAVITY if enable_gravity else 0)
               | (CRB_CORRECTED_AXIAL if corrected_axial else 0))
    return d, (cols, nl, bc, wet, cd)


class Plan:
    """Owner of one ``crb_plan`` (n_beams beams, one dtype, one device).

    ``columns``: the CSV columns of the beam (dict) -- coefficients shared by all beams -- or a
    list of n_beams such dicts for a heterogeneous ensemble (crb_plan_create_ensemble): per-beam element columns,
    element count and boundary conditions.  With a list, ``node_bc``, ``fluid_density``, ``enable_fluid``,
    ``gravity`` and ``enable_gravity`` may each be a list of n_beams values too (per-beam ForceParams)."""

    def __init__(self, columns, n_beams=1, node_bc=None, fluid_density=0.0, enable_fluid=False,
                 gravity=(0.0, -9.81, 0.0), enable_gravity=False, corrected_axial=False, dtype="f64", device=0):
        L = load()
        self.dtype = {"f64": CRB_F64, "f32": CRB_F32, CRB_F64: CRB_F64, CRB_F32: CRB_F32}[dtype]
        self.device = int(device)
        h = C.c_void_p()
        if isinstance(columns, (list, tuple)):
            if len(columns) != n_beams:
                raise ValueError("per-beam coefficients need exactly n_beams column sets")

            def scalars(v, name):
                """one value for all beams, or a sequence of n_beams values"""
                if np.ndim(v) == 0:
                    return lambda b: v
                if len(v) != n_beams:
                    raise ValueError(f"{name}: expected {n_beams} per-beam values, got {len(v)}")
                return lambda b: v[b]

            fd_of, ef_of, eg_of = (scalars(fluid_density, "fluid_density"), scalars(enable_fluid, "enable_fluid"),
                                   scalars(enable_gravity, "enable_gravity"))
            g = np.asarray(gravity, dtype=np.float64)
            if g.shape not in ((3,), (n_beams, 3)):
                raise ValueError(f"gravity: expected a 3-vector or {n_beams} of them")
            g_of = (lambda b: g) if g.ndim == 1 else (lambda b: g[b])
            if node_bc is None:
                bc_of = lambda b: None  # noqa: E731
            elif len(node_bc) == n_beams and all(np.ndim(v) == 1 for v in node_bc):
                bc_of = lambda b: node_bc[b]  # noqa: E731   (one array per beam)
            else:
                bc_of = lambda b: node_bc  # noqa: E731      (one array for all beams)
            descs = (BeamDesc * n_beams)()
            self._keep = []
            for b, cols in enumerate(columns):
                descs[b], keep = _beam_desc(cols, bc_of(b), fd_of(b), bool(ef_of(b)), g_of(b), bool(eg_of(b)), corrected_axial)
                self._keep.append(keep)
            self.per_beam = True
            check(L.crb_plan_create_ensemble(C.byref(h), self.device, self.dtype, int(n_beams), descs))
        else:
            d, self._keep = _beam_desc(columns, node_bc, fluid_density, enable_fluid, gravity, enable_gravity, corrected_axial)
            self.per_beam = False
            check(L.crb_plan_create(C.byref(h), self.device, self.dtype, int(n_beams), C.byref(d)))
        self.h = h
        self._rhs_host = load().crb_rhs_host   # (bound once: the per-RHS-call path of the single-beam closures)
        lay = Layout()
        check(L.crb_plan_get_layout(self.h, C.byref(lay)))
        self.layout = lay
        for name, _ in Layout._fields_:
            setattr(self, name, getattr(lay, name))
        self.mixed_topology = bool(lay.mixed_topology)
        self.n_free0 = self.beam_info(0)[1]       # beam 0's reduced size (== n_free unless the topology is mixed)
        fi = np.empty(self.n_free0, dtype=np.int32)
        check(L.crb_plan_get_free_index(self.h, fi.ctypes.data_as(C.POINTER(C.c_int32))))
        self.free_index = fi

    def beam_info(self, beam):
        """(n_elem, n_free) of one beam of the plan"""
        ne, nf = C.c_int32(0), C.c_int32(0)
        check(load().crb_plan_get_beam_info(self.h, int(beam), C.byref(ne), C.byref(nf)))
        return int(ne.value), int(nf.value)

    def beam_free_index(self, beam):
        """reduced -> full index (3 * node + dof) of one beam"""
        fi = np.empty(self.beam_info(beam)[1], dtype=np.int32)
        check(load().crb_plan_get_beam_free_index(self.h, int(beam), fi.ctypes.data_as(C.POINTER(C.c_int32))))
        return fi

    def __del__(self):
        try:
            if getattr(self, "h", None):
                load().crb_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- host-vector calls (single-beam closures): one launch + one synchronisation each
    def rhs_host(self, x_red, u_red=None):
        """xdot = [v ; Minv(-k(q) + f_drag + f_gravity + u)] for host vectors [n_beams, 2n] / [n_beams, n] (or 1-D for
        one beam), in the reference's reduced ordering."""
        x = np.ascontiguousarray(x_red, dtype=np.float64)
        out = np.empty_like(x)
        u = None if u_red is None else np.ascontiguousarray(u_red, dtype=np.float64)
        # (the C entry point takes bare pointers and copies n_beams * 2n / n_beams * n doubles: sizes are checked here)
        if x.size != self.n_beams * 2 * self.n_free0 or (u is not None and u.size != self.n_beams * self.n_free0):
            raise ValueError(f"rhs_host: expected {self.n_beams} x {2 * self.n_free0} state and {self.n_beams} x {self.n_free0} "
                             f"input values, got {x.size}" + ("" if u is None else f" and {u.size}"))
        rc = self._rhs_host(self.h, x.ctypes.data, u.ctypes.data if u is not None else None, out.ctypes.data)
        if rc:
            check(rc)
        return out

    def internal_force_host(self, q_red):
        q = np.ascontiguousarray(q_red, dtype=np.float64)
        if q.size != self.n_beams * self.n_free0:
            raise ValueError(f"internal_force_host: expected {self.n_beams} x {self.n_free0} positions, got {q.size}")
        out = np.empty_like(q)
        check(load().crb_internal_force_host(self.h, q.ctypes.data, out.ctypes.data))
        return out

    # ---- inspection (host)
    def mass(self):
        M = np.empty((self.n_free0, self.n_free0))
        check(load().crb_plan_get_mass(self.h, M.ctypes.data_as(_dp)))
        return M

    def stiffness(self):
        K = np.empty((self.n_free0, self.n_free0))
        check(load().crb_plan_get_stiffness(self.h, K.ctypes.data_as(_dp)))
        return K

    def pcr_tables(self):
        S, lf = self.n_slots, self.pcr_levels_full
        levels = np.zeros((max(lf, 1), S, 10))
        final = np.zeros((S, 6))
        norms = np.zeros(max(lf, 1))
        check(load().crb_plan_get_pcr_tables(self.h, levels.ctypes.data_as(_dp), final.ctypes.data_as(_dp),
                                             norms.ctypes.data_as(_dp)))
        return levels[:lf], final, norms[:lf]

    def slot_tables(self):
        S = self.n_slots
        drag, hm, mask = np.zeros(S), np.zeros(S), np.zeros((S, 3))
        grav = np.zeros((S, 12), dtype=np.int16)
        kind = np.zeros(S, dtype=np.int32)
        check(load().crb_plan_get_slot_tables(self.h, drag.ctypes.data_as(_dp), hm.ctypes.data_as(_dp),
                                              mask.ctypes.data_as(_dp), grav.ctypes.data_as(C.POINTER(C.c_int16)),
                                              kind.ctypes.data_as(C.POINTER(C.c_int32))))
        return dict(drag=drag, half_mass=hm, mask=mask, grav=grav, elem_kind=kind)
