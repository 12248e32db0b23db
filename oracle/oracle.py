"""ctypes front end of oracle/crb_oracle.c (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcrb_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    """Compile the oracle with gcc (make -C oracle)."""
    src = os.path.join(_HERE, "crb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp, C.c_int, C.c_double,
                                 C.c_int, _dp, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        for name in ("orc_n_red", "orc_n_full"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        L.orc_red2full.argtypes = [C.c_void_p, _ip]
        L.orc_drag_table.argtypes = [C.c_void_p, _ip, _dp]
        L.orc_drag_table.restype = C.c_int
        L.orc_mass_dense.argtypes = [C.c_void_p, _dp]
        L.orc_stiff_dense.argtypes = [C.c_void_p, _dp]
        L.orc_stiff_dense.restype = C.c_int
        L.orc_solve.argtypes = [C.c_void_p, _dp]
        for name in ("orc_internal_force", "orc_drag", "orc_gravity", "orc_forces"):
            getattr(L, name).argtypes = [C.c_void_p, _dp, _dp]
        L.orc_rhs.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.orc_rk4_impulse.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                      C.c_int]
        L.orc_rk4_impulse.restype = C.c_double
        L.orc_rk4_held.argtypes = [C.c_void_p, _dp, C.c_double, C.c_int, _dp]
        L.orc_rk4_feedback.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, C.c_int, _dp, _dp, C.c_double, C.c_double,
                                       C.c_int]
        L.orc_rk4_feedback.restype = C.c_double
        L.orc_implicit.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double,
                                   C.c_double, C.c_int, _dp]
        L.orc_implicit.restype = C.c_double
        L.orc_implicit_alpha.argtypes = [C.c_void_p, _dp, C.c_double, C.c_double, C.c_int, C.c_int, C.c_double, C.c_double,
                                         C.c_double, C.c_int, _dp]
        L.orc_implicit_alpha.restype = C.c_double
        L.orc_rk4_impulse_batch.argtypes = [C.c_void_p, _dp, C.c_int, _dp, C.c_double, C.c_double, C.c_int,
                                            C.c_double, C.c_int, C.c_int]
        L.orc_rk4_impulse_batch.restype = C.c_int
        L.orc_elem_mass.argtypes = [C.c_double, C.c_double, C.c_double, _dp]
        L.orc_elem_stiff_linear.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, _dp]
        L.orc_elem_force_nonlinear.argtypes = [C.c_double, C.c_double, C.c_double, _dp, C.c_int, _dp]
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def elem_mass(L, rho, A):
    out = np.empty(36)
    lib().orc_elem_mass(L, rho, A, _p(out))
    return out.reshape(6, 6)


def elem_stiff_linear(L, E, I, A):
    out = np.empty(36)
    lib().orc_elem_stiff_linear(L, E, I, A, _p(out))
    return out.reshape(6, 6)


def elem_force_nonlinear(L, EA, EI, x, corrected_axial=False):
    x = _d(x)
    out = np.empty(6)
    lib().orc_elem_force_nonlinear(L, EA, EI, _p(x), int(corrected_axial), _p(out))
    return out


_BC = {"NONE": 0, "FIXED": 1, "PINNED": 2, "none": 0, "fixed": 1, "pinned": 2, 0: 0, 1: 1, 2: 2}


class OracleBeam:
    """One beam of the reference's CSV schema (dynamic_beam_model.py:78-90) + ForceParams.

    ``node_bc`` has n_seg+1 entries (node-indexed); ``boundary_condition`` (CSV column,
    row i -> node i, last node free; dynamic_beam_model.py:205-218) is accepted instead.
    """

    def __init__(self, length, elastic_modulus, moment_inertia, density, cross_area, type,
                 boundary_condition=None, node_bc=None, wetted_area=None, drag_coef=None, fluid_density=0.0,
                 enable_fluid=False, gravity=(0.0, -9.81, 0.0), enable_gravity=False, corrected_axial=False):
        n = len(length)
        self.n_seg = n
        if node_bc is None:
            node_bc = [_BC[str(b)] for b in boundary_condition] + [0]
        node_bc = np.ascontiguousarray([_BC[b] if not isinstance(b, (int, np.integer)) else int(b) for b in node_bc],
                                       dtype=np.int32)
        assert node_bc.shape == (n + 1,)
        nl = np.ascontiguousarray([1 if str(t).lower() == "nonlinear" else 0 for t in type], dtype=np.int32)
        a = [_d(v) for v in (length, elastic_modulus, moment_inertia, density, cross_area)]
        wet = _d(wetted_area) if wetted_area is not None else np.zeros(n)
        cd = _d(drag_coef) if drag_coef is not None else np.zeros(n)
        g = _d(gravity)
        self._keep = (a, nl, node_bc, wet, cd, g)
        self.h = C.c_void_p(lib().orc_create(n, *[_p(v) for v in a], nl.ctypes.data_as(_ip),
                                             node_bc.ctypes.data_as(_ip), _p(wet), _p(cd), int(bool(enable_fluid)),
                                             float(fluid_density), int(bool(enable_gravity)), _p(g),
                                             int(bool(corrected_axial))))
        self.n = lib().orc_n_red(self.h)
        self.n_full = lib().orc_n_full(self.h)
        self.node_bc = node_bc

    def __del__(self):
        try:
            if self.h:
                lib().orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- structure
    def red2full(self):
        out = np.empty(self.n, dtype=np.int32)
        lib().orc_red2full(self.h, out.ctypes.data_as(_ip))
        return out

    def drag_table(self):
        pos = np.empty(self.n_seg + 1, dtype=np.int32)
        fac = np.empty(self.n_seg + 1)
        k = lib().orc_drag_table(self.h, pos.ctypes.data_as(_ip), _p(fac))
        return pos[:k], fac[:k]

    def mass(self):
        M = np.empty((self.n, self.n))
        lib().orc_mass_dense(self.h, _p(M))
        return M

    def stiffness(self):
        K = np.empty((self.n, self.n))
        if lib().orc_stiff_dense(self.h, _p(K)) != 0:
            raise ValueError("nonlinear segments: no constant stiffness matrix")
        return K

    def solve(self, b):
        b = _d(b).copy()
        lib().orc_solve(self.h, _p(b))
        return b

    # -- forces / rhs
    def _xf(self, fn, x):
        x = _d(x)
        out = np.empty(self.n)
        fn(self.h, _p(x), _p(out))
        return out

    def internal_force(self, q):
        return self._xf(lib().orc_internal_force, q)

    def drag(self, x):
        return self._xf(lib().orc_drag, x)

    def gravity(self, x):
        return self._xf(lib().orc_gravity, x)

    def forces(self, x):
        return self._xf(lib().orc_forces, x)

    def rhs(self, x, u=None):
        x = _d(x)
        out = np.empty(2 * self.n)
        uu = _d(u) if u is not None else None
        lib().orc_rhs(self.h, _p(x), _p(uu) if uu is not None else None, _p(out))
        return out

    # -- stepping
    def rk4_impulse(self, x0, dt, n_steps, amp, duration=0.01, idx=-2, t0=0.0):
        x = _d(x0).copy()
        lib().orc_rk4_impulse(self.h, _p(x), t0, dt, n_steps, amp, duration, idx)
        return x

    def rk4_held(self, x0, dt, n_steps, u=None):
        x = _d(x0).copy()
        uu = _d(u) if u is not None else None
        lib().orc_rk4_held(self.h, _p(x), dt, n_steps, _p(uu) if uu is not None else None)
        return x

    def rk4_feedback(self, x0, dt, n_steps, gain, reference=None, amp=0.0, duration=0.01, idx=-2, t0=0.0):
        x = _d(x0).copy()
        K = _d(gain)
        assert K.shape == (self.n, 2 * self.n)
        r = _d(reference) if reference is not None else None
        lib().orc_rk4_feedback(self.h, _p(x), t0, dt, n_steps, _p(K), _p(r) if r is not None else None, amp, duration,
                               idx)
        return x

    def implicit(self, x0, h, n_steps, n_iter=3, amp=0.0, duration=0.01, idx=-2, t0=0.0, u_held=None):
        """Implicit midpoint rule + modified Newton (orc_implicit), the CPU statement of crb_step_implicit."""
        x = _d(x0).copy()
        uu = _d(u_held) if u_held is not None else None
        lib().orc_implicit(self.h, _p(x), t0, h, n_steps, n_iter, amp, duration, idx, _p(uu) if uu is not None else None)
        return x

    def implicit_alpha(self, x0, h, n_steps, rho, n_iter=3, amp=0.0, duration=0.01, idx=-2, t0=0.0, u_held=None):
        """Generalised-alpha with spectral radius ``rho`` at infinite frequency (orc_implicit_alpha): the numerically damped
        member of ``implicit``'s family, the CPU statement of crb_step_implicit_damped; rho = 1 is ``implicit`` up to rounding."""
        x = _d(x0).copy()
        uu = _d(u_held) if u_held is not None else None
        lib().orc_implicit_alpha(self.h, _p(x), t0, h, n_steps, n_iter, rho, amp, duration, idx, _p(uu) if uu is not None else None)
        return x

    def rk4_impulse_batch(self, X0, dt, n_steps, amps, duration=0.01, idx=-2, t0=0.0, n_threads=0):
        X = _d(X0).copy()
        amps = _d(amps)
        used = lib().orc_rk4_impulse_batch(self.h, _p(X), X.shape[0], _p(amps), t0, dt, n_steps, duration, idx,
                                           n_threads)
        return X, used
