"""LinearQuadraticRegulator: LQR gain for the linear beam  M q'' + K q = u.

Reference: src/continuum_robot/control/linear_quadratic_regulator.py:5-200 (same constructor,
validation messages, A/B construction, caching and closed-loop stability check).  The reference
solves the Riccati equation with python-control's ``ct.lqr`` (:180); that package is optional
here: when it is not importable the same continuous-time algebraic Riccati equation is solved
with ``scipy.linalg.solve_continuous_are`` and K = R^-1 B^T S (verified substitute).
The control layer stays in Python (north_star); it consumes K and M from
EulerBernoulliBeam.get_stiffness_matrix() / get_mass_matrix().
"""
import numpy as np

try:  # pragma: no cover - depends on the environment
    import control as ct
except Exception:  # python-control (and slycot) are not installed in the build image
    ct = None


def _solve_lqr(A, B, Q, R):
    if ct is not None:
        return ct.lqr(A, B, Q, R)
    from scipy.linalg import solve_continuous_are

    S = solve_continuous_are(A, B, Q, R)
    K = np.linalg.solve(R, B.T @ S)
    E = np.linalg.eigvals(A - B @ K)
    return K, S, E


class LinearQuadraticRegulator:
    def __init__(self, K_beam: np.ndarray, M_beam: np.ndarray, Q: np.ndarray, R: np.ndarray):
        self._validate_beam_matrices(K_beam, M_beam)
        self._validate_weighting_matrices(Q, R)
        self.K_beam, self.M_beam, self.Q, self.R = K_beam, M_beam, Q, R
        self._A = self._B = self._K = self._S = self._E = None

    @staticmethod
    def _validate_beam_matrices(K_beam: np.ndarray, M_beam: np.ndarray) -> None:
        if K_beam.ndim != 2 or K_beam.shape[0] != K_beam.shape[1]:
            raise ValueError("Stiffness matrix must be square")
        if M_beam.ndim != 2 or M_beam.shape[0] != M_beam.shape[1]:
            raise ValueError("Mass matrix must be square")
        if K_beam.shape != M_beam.shape:
            raise ValueError("Stiffness and mass matrices must have the same dimensions")

    @staticmethod
    def _validate_weighting_matrices(Q: np.ndarray, R: np.ndarray) -> None:
        if Q.ndim != 2 or Q.shape[0] != Q.shape[1]:
            raise ValueError("Q matrix must be square")
        if R.ndim != 2 or R.shape[0] != R.shape[1]:
            raise ValueError("R matrix must be square")
        try:
            if np.any(np.linalg.eigvals(Q) < -1e-10):
                raise ValueError("Q matrix must be positive semidefinite")
        except np.linalg.LinAlgError:
            raise ValueError("Q matrix must be positive semidefinite")
        try:
            if np.any(np.linalg.eigvals(R) <= 1e-10):
                raise ValueError("R matrix must be positive definite")
        except np.linalg.LinAlgError:
            raise ValueError("R matrix must be positive definite")

    def _mass_inverse(self) -> np.ndarray:
        try:
            return np.linalg.inv(self.M_beam)
        except np.linalg.LinAlgError:
            raise ValueError("Mass matrix is singular and cannot be inverted")

    def get_A(self) -> np.ndarray:
        """A = [[0, I], [-Minv K, 0]] for x = [q ; q']."""
        if self._A is None:
            n = self.M_beam.shape[0]
            A = np.zeros((2 * n, 2 * n))
            A[:n, n:] = np.eye(n)
            A[n:, :n] = -self._mass_inverse() @ self.K_beam
            self._A = A
        return self._A

    def get_B(self) -> np.ndarray:
        """B = [[0], [Minv]] (full actuation)."""
        if self._B is None:
            n = self.M_beam.shape[0]
            B = np.zeros((2 * n, n))
            B[n:, :] = self._mass_inverse()
            self._B = B
        return self._B

    def compute_gain_matrix(self) -> np.ndarray:
        if self._K is not None:
            return self._K
        A, B = self.get_A(), self.get_B()
        if self.Q.shape[0] != A.shape[0]:
            raise ValueError(f"Q matrix dimension {self.Q.shape[0]} must match state dimension {A.shape[0]}")
        if self.R.shape[0] != B.shape[1]:
            raise ValueError(f"R matrix dimension {self.R.shape[0]} must match input dimension {B.shape[1]}")
        try:
            self._K, self._S, self._E = _solve_lqr(A, B, self.Q, self.R)
        except Exception as e:
            raise ValueError(f"Failed to solve LQR problem: {e}")
        if np.any(np.real(np.linalg.eigvals(A - B @ self._K)) >= 0):
            raise ValueError("LQR solution results in unstable closed-loop system")
        return self._K

    def get_K(self) -> np.ndarray:
        return self.compute_gain_matrix()
