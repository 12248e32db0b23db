/*
 * crb_oracle.c -- CPU restatement of the reference's beam hot path (plain C, fp64).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under continuum-robot_amd/ may include, link,
 * load or call this file; it is the checker for tests/, __graft_entry__.smoke() and
 * the cpu_baseline leg of bench.py, never the thing shipped or measured as product.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function below
 * against tests/golden/ (npz files), which tests/golden/make_golden.py produced by importing
 * the reference itself (cram9030/continuum-robot @ /root/reference, read-only).
 *
 * What is restated (file:line under /root/reference/src/continuum_robot/models/):
 *   orc_elem_mass            segments.py:64-78, 105-119    consistent 6x6 element mass
 *   orc_elem_stiff_linear    segments.py:32-62             constant 6x6 element stiffness
 *   orc_elem_force_nonlinear segments.py:159-472, order :146-155  (shipped f1, bug-compatible)
 *   orc_create               euler_bernoulli_beam.py:139-161 (mass assembly), :221-298 (BC
 *                            reduction), dynamic_beam_model.py:205-218 (row i BC -> node i),
 *                            fluid_forces.py:50-101 (drag factors), gravity_forces.py:54-64
 *   orc_internal_force       euler_bernoulli_beam.py:163-219, 270-289
 *   orc_drag                 fluid_forces.py:103-142
 *   orc_gravity              gravity_forces.py:66-148  (indexes the REDUCED vector with
 *                            FULL-layout indices; reproduced as shipped)
 *   orc_rhs                  dynamic_beam_model.py:256-272, 294-328, 343-362
 *                            xdot = [v ; Minv(-k(q) + f(x, t=0) + u)]
 *   orc_rk4_feedback         the closed loop of examples/lqr_control.py:95-111 (u = K(r-x) per stage)
 *   orc_implicit_alpha       OURS: the damped member of orc_implicit's family (generalised-alpha, rho at infinity)
 *   orc_implicit             OURS (like the RK4 loop: the reference has no integrator): implicit midpoint rule with
 *                            a modified-Newton iteration, the CPU statement of crb_step_implicit -- the stiff end of
 *                            the solve_ivp(LSODA) call sites (examples/example_utilities.py:153-159); pinned by
 *                            tests/golden/g8_lsoda.npz (scipy LSODA at tight tolerances over the REFERENCE RHS)
 *   orc_rk4_*                fixed-step classical RK4 over orc_rhs.  The reference has no
 *                            integrator (callers use scipy.solve_ivp); the loop restated here
 *                            is the one in tests/golden/make_golden.py:rk4.
 *
 * One deliberate numerical difference, rounding-level only: the reference forms the
 * explicit inverse scipy.sparse.linalg.inv(M) (dynamic_beam_model.py:60) and applies it
 * three times per RHS; this file factors the reduced M once (banded Cholesky, half
 * bandwidth 5) and solves once per RHS.  The golden tests bound the difference.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_KD 5 /* half bandwidth of M in the interleaved [u,w,phi] ordering */

typedef struct orc_model {
    int n_seg, n_node, n_full, n_red;
    double *L, *E, *I, *rho, *A, *wet, *cd, *seg_mass;
    int *nonlinear, *bc;
    int fluid_on, grav_on, corrected_axial;
    double fluid_density, g[3];
    int *red2full, *full2red;
    double *chol; /* lower band Cholesky factor, row i holds cols i-KD..i */
    int n_drag;
    int *drag_pos;
    double *drag_fac;
} orc_model;

/* ------------------------------------------------------------------ element kernels */

/* segments.py:64-78 (LinearSegment) == :105-119 (NonlinearSegment) */
void orc_elem_mass(double L, double rho, double A, double M[36]) {
    const double c = rho * A * L / 420.0;
    const double L2 = L * L;
    const double m[36] = {140,     0,       0,  70,       0,       0,
                          0,     156, -22 * L,   0,      54,  13 * L,
                          0, -22 * L,  4 * L2,   0, -13 * L, -3 * L2,
                          70,      0,       0, 140,       0,       0,
                          0,      54, -13 * L,   0,     156,  22 * L,
                          0,  13 * L, -3 * L2,   0,  22 * L,  4 * L2};
    for (int i = 0; i < 36; ++i) M[i] = m[i] * c;
}

/* segments.py:32-62 ; DOF order [u1, w1, phi1, u2, w2, phi2] */
void orc_elem_stiff_linear(double L, double E, double I, double A, double K[36]) {
    const double EI = E * I, EA = E * A;
    const double a = EA / L, b = 12 * EI / (L * L * L), c = 6 * EI / (L * L), d4 = 4 * EI / L, d2 = 2 * EI / L;
    const double k[36] = {a,  0,  0,  -a, 0,  0,
                          0,  b,  -c, 0,  -b, -c,
                          0,  -c, d4, 0,  c,  d2,
                          -a, 0,  0,  a,  0,  0,
                          0,  -b, c,  0,  b,  c,
                          0,  -c, d2, 0,  c,  d4};
    memcpy(K, k, sizeof k);
}

/* segments.py:159-472.  Literals are the reference's, digit for digit; powers are
 * written as repeated products.  out[] is in the reference's RETURN order
 * [f1, f3, f4, f2, f5, f6] (segments.py:146-155) = forces on [u1, w1, th1, u2, w2, th2].
 * corrected_axial != 0 replaces the mis-transcribed f1 by -f2; default 0. */
void orc_elem_force_nonlinear(double L, double A, double D, const double x[6], int corrected_axial, double out[6]) {
    const double u1 = x[0], w1 = x[1], t1 = x[2], u2 = x[3], w2 = x[4], t2 = x[5];
    const double L2 = L * L, L3 = L * L * L;
    const double t1_2 = t1 * t1, t1_3 = t1 * t1 * t1, t2_2 = t2 * t2, t2_3 = t2 * t2 * t2;
    const double w1_2 = w1 * w1, w1_3 = w1 * w1 * w1, w2_2 = w2 * w2, w2_3 = w2 * w2 * w2;

    /* :178-208 */
    double f1 = A *
                (L * (-t1 * (0.0666666666666665 * t1 * L - 0.0166666666666667 * t2 * L - 0.05 * w1 + 0.05 * w2) +
                      t2 * (0.0166666666666667 * t1 * L - 0.0666666666666667 * t2 * L + 0.05 * w1 - 0.05 * w2) + u1) +
                 (-u2 - w1 + w2) * (-0.05 * t1 * L - 0.05 * t2 * L + 0.6 * w1 - 0.6 * w2)) /
                L2;
    /* :227-258 */
    const double f2 = A *
                      (L * (t1 * (0.0666666666666665 * t1 * L - 0.0166666666666667 * t2 * L - 0.05 * w1 + 0.05 * w2) -
                            t2 * (0.0166666666666667 * t1 * L - 0.0666666666666667 * t2 * L + 0.05 * w1 - 0.05 * w2) -
                            u1 + u2) +
                       (w1 - w2) * (-0.05 * t1 * L - 0.05 * t2 * L + 0.6 * w1 - 0.6 * w2)) /
                      L2;
    /* :279-314 */
    const double s3 = 0.0357142857143344 * A * t1_3 * L3
                    - 0.107142857143003 * A * t1_2 * t2 * L3
                    + 1.28571428571433 * A * t1_2 * L2 * w1
                    - 1.28571428571433 * A * t1_2 * L2 * w2
                    - 0.107142857143003 * A * t1 * t2_2 * L3
                    + 1.0 * A * t1 * L2 * u1
                    - 1.0 * A * t1 * L2 * u2
                    - 3.8571428571413 * A * t1 * L * w1_2
                    + 7.7142857142826 * A * t1 * L * w1 * w2
                    - 3.8571428571413 * A * t1 * L * w2_2
                    + 0.0357142857143344 * A * t2_3 * L3
                    + 1.28571428571433 * A * t2_2 * L2 * w1
                    - 1.28571428571433 * A * t2_2 * L2 * w2
                    + 1.0 * A * t2 * L2 * u1
                    - 1.0 * A * t2 * L2 * u2
                    - 3.857142857143 * A * t2 * L * w1_2
                    + 7.71428571428601 * A * t2 * L * w1 * w2
                    - 3.857142857143 * A * t2 * L * w2_2
                    - 12.0 * A * L * u1 * w1
                    + 12.0 * A * L * u1 * w2
                    + 12.0 * A * L * u2 * w1
                    - 12.0 * A * L * u2 * w2
                    + 10.2857142857147 * A * w1_3
                    - 30.857142857144 * A * w1_2 * w2
                    + 30.857142857144 * A * w1 * w2_2
                    - 10.2857142857147 * A * w2_3
                    - 60.0 * D * t1 * L
                    - 60.0 * D * t2 * L
                    + 120.0 * D * w1
                    - 120.0 * D * w2;
    const double f3 = 0.1 * s3 / L3;
    /* :335-365 */
    const double f4 = 0.0285714285714391 * A * t1_3 * L
                    - 0.0107142857142861 * A * t1_2 * t2 * L
                    + 0.0107142857142719 * A * t1_2 * w1
                    - 0.0107142857142719 * A * t1_2 * w2
                    + 0.00714285714286444 * A * t1 * t2_2 * L
                    - 0.0214285714286007 * A * t1 * t2 * w1
                    + 0.0214285714286007 * A * t1 * t2 * w2
                    - 0.133333333333333 * A * t1 * u1
                    + 0.133333333333333 * A * t1 * u2
                    + 0.128571428571433 * A * t1 * w1_2 / L
                    - 0.257142857142867 * A * t1 * w1 * w2 / L
                    + 0.128571428571433 * A * t1 * w2_2 / L
                    - 0.00357142857143344 * A * t2_3 * L
                    - 0.0107142857142719 * A * t2_2 * w1
                    + 0.0107142857142719 * A * t2_2 * w2
                    + 0.0333333333333333 * A * t2 * u1
                    - 0.0333333333333333 * A * t2 * u2
                    + 0.1 * A * u1 * w1 / L
                    - 0.1 * A * u1 * w2 / L
                    - 0.1 * A * u2 * w1 / L
                    + 0.1 * A * u2 * w2 / L
                    - 0.128571428571377 * A * w1_3 / L2
                    + 0.38571428571413 * A * w1_2 * w2 / L2
                    - 0.38571428571413 * A * w1 * w2_2 / L2
                    + 0.128571428571377 * A * w2_3 / L2
                    + 4.0 * D * t1 / L
                    + 2.0 * D * t2 / L
                    - 6.0 * D * w1 / L2
                    + 6.0 * D * w2 / L2;
    /* :386-421 : every term of f5 is the negated term of f3 */
    const double f5 = 0.1 * (-s3) / L3;
    /* :442-472 */
    const double f6 = -0.00357142857143344 * A * t1_3 * L
                    + 0.00714285714286356 * A * t1_2 * t2 * L
                    - 0.0107142857143003 * A * t1_2 * w1
                    + 0.0107142857143003 * A * t1_2 * w2
                    - 0.0107142857142932 * A * t1 * t2_2 * L
                    - 0.021428571428558 * A * t1 * t2 * w1
                    + 0.021428571428558 * A * t1 * t2 * w2
                    + 0.0333333333333333 * A * t1 * u1
                    - 0.0333333333333333 * A * t1 * u2
                    + 0.0285714285714271 * A * t2_3 * L
                    + 0.0107142857142932 * A * t2_2 * w1
                    - 0.0107142857142932 * A * t2_2 * w2
                    - 0.133333333333333 * A * t2 * u1
                    + 0.133333333333333 * A * t2 * u2
                    + 0.128571428571428 * A * t2 * w1_2 / L
                    - 0.257142857142856 * A * t2 * w1 * w2 / L
                    + 0.128571428571428 * A * t2 * w2_2 / L
                    + 0.1 * A * u1 * w1 / L
                    - 0.1 * A * u1 * w2 / L
                    - 0.1 * A * u2 * w1 / L
                    + 0.1 * A * u2 * w2 / L
                    - 0.128571428571433 * A * w1_3 / L2
                    + 0.3857142857143 * A * w1_2 * w2 / L2
                    - 0.3857142857143 * A * w1 * w2_2 / L2
                    + 0.128571428571433 * A * w2_3 / L2
                    + 2.0 * D * t1 / L
                    + 4.0 * D * t2 / L
                    - 6.0 * D * w1 / L2
                    + 6.0 * D * w2 / L2;
    if (corrected_axial) f1 = -f2;
    out[0] = f1; out[1] = f3; out[2] = f4; out[3] = f2; out[4] = f5; out[5] = f6;
}

/* ------------------------------------------------------------------ model */

void orc_destroy(orc_model* m) {
    if (!m) return;
    free(m->L); free(m->E); free(m->I); free(m->rho); free(m->A); free(m->wet); free(m->cd);
    free(m->seg_mass); free(m->nonlinear); free(m->bc); free(m->red2full); free(m->full2red);
    free(m->chol); free(m->drag_pos); free(m->drag_fac);
    free(m);
}

static double* dup_d(const double* s, int n) {
    double* d = (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    if (s) memcpy(d, s, (size_t)n * sizeof(double));
    return d;
}

/* full (unreduced) dense mass, euler_bernoulli_beam.py:139-161 */
static void full_mass(const orc_model* m, double* Mf) {
    const int nf = m->n_full;
    memset(Mf, 0, (size_t)nf * nf * sizeof(double));
    for (int e = 0; e < m->n_seg; ++e) {
        double Me[36];
        orc_elem_mass(m->L[e], m->rho[e], m->A[e], Me);
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) Mf[(size_t)(3 * e + a) * nf + 3 * e + b] += Me[a * 6 + b];
    }
}

void orc_mass_dense(const orc_model* m, double* M /* n_red x n_red */) {
    const int nf = m->n_full, n = m->n_red;
    double* Mf = (double*)malloc((size_t)nf * nf * sizeof(double));
    full_mass(m, Mf);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) M[(size_t)i * n + j] = Mf[(size_t)m->red2full[i] * nf + m->red2full[j]];
    free(Mf);
}

/* euler_bernoulli_beam.py:422-511 ; returns -1 if any segment is nonlinear */
int orc_stiff_dense(const orc_model* m, double* K /* n_red x n_red */) {
    const int nf = m->n_full, n = m->n_red;
    for (int e = 0; e < m->n_seg; ++e)
        if (m->nonlinear[e]) return -1;
    double* Kf = (double*)calloc((size_t)nf * nf, sizeof(double));
    for (int e = 0; e < m->n_seg; ++e) {
        double Ke[36];
        orc_elem_stiff_linear(m->L[e], m->E[e], m->I[e], m->A[e], Ke);
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) Kf[(size_t)(3 * e + a) * nf + 3 * e + b] += Ke[a * 6 + b];
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) K[(size_t)i * n + j] = Kf[(size_t)m->red2full[i] * nf + m->red2full[j]];
    free(Kf);
    return 0;
}

/* node_bc[i]: 0 none, 1 FIXED (u,w,phi dropped), 2 PINNED (u,w dropped);
 * euler_bernoulli_beam.py:240-259.  wet/cd may be NULL when fluid_on == 0. */
orc_model* orc_create(int n_seg, const double* L, const double* E, const double* I, const double* rho,
                      const double* A, const int* nonlinear, const int* node_bc, const double* wet,
                      const double* cd, int fluid_on, double fluid_density, int grav_on, const double* g,
                      int corrected_axial) {
    orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
    m->n_seg = n_seg;
    m->n_node = n_seg + 1;
    m->n_full = 3 * m->n_node;
    m->L = dup_d(L, n_seg); m->E = dup_d(E, n_seg); m->I = dup_d(I, n_seg);
    m->rho = dup_d(rho, n_seg); m->A = dup_d(A, n_seg);
    m->wet = dup_d(wet, n_seg); m->cd = dup_d(cd, n_seg);
    m->nonlinear = (int*)calloc((size_t)n_seg, sizeof(int));
    memcpy(m->nonlinear, nonlinear, (size_t)n_seg * sizeof(int));
    m->bc = (int*)calloc((size_t)m->n_node, sizeof(int));
    memcpy(m->bc, node_bc, (size_t)m->n_node * sizeof(int));
    m->fluid_on = fluid_on; m->fluid_density = fluid_density; m->grav_on = grav_on;
    m->corrected_axial = corrected_axial;
    m->g[0] = g ? g[0] : 0.0; m->g[1] = g ? g[1] : -9.81; m->g[2] = g ? g[2] : 0.0;

    m->full2red = (int*)malloc((size_t)m->n_full * sizeof(int));
    m->red2full = (int*)malloc((size_t)m->n_full * sizeof(int));
    int n = 0;
    for (int i = 0; i < m->n_node; ++i)
        for (int d = 0; d < 3; ++d) {
            const int fixed = (m->bc[i] == 1) || (m->bc[i] == 2 && d < 2);
            m->full2red[3 * i + d] = fixed ? -1 : n;
            if (!fixed) m->red2full[n++] = 3 * i + d;
        }
    m->n_red = n;

    /* gravity_forces.py:54-64 */
    m->seg_mass = (double*)malloc((size_t)n_seg * sizeof(double));
    for (int e = 0; e < n_seg; ++e) m->seg_mass[e] = m->rho[e] * m->A[e] * m->L[e];

    /* fluid_forces.py:50-101 : per node with a free w, factor from the node's own segment
     * row (last row repeated for the tip node) */
    m->drag_pos = (int*)malloc((size_t)m->n_node * sizeof(int));
    m->drag_fac = (double*)malloc((size_t)m->n_node * sizeof(double));
    m->n_drag = 0;
    if (fluid_on)
        for (int k = 0; k < m->n_node; ++k) {
            const int r = m->full2red[3 * k + 1];
            if (r < 0) continue;
            const int row = k < n_seg ? k : n_seg - 1;
            m->drag_pos[m->n_drag] = r;
            m->drag_fac[m->n_drag] = 0.5 * fluid_density * m->cd[row] * m->wet[row];
            ++m->n_drag;
        }

    /* banded Cholesky of the reduced mass (replaces dynamic_beam_model.py:60) */
    double* M = (double*)malloc((size_t)n * n * sizeof(double));
    orc_mass_dense(m, M);
    const int kd = ORC_KD, w = kd + 1;
    m->chol = (double*)calloc((size_t)n * w, sizeof(double));
#define LB(i, j) m->chol[(size_t)(i) * w + (kd + (j) - (i))]
    for (int i = 0; i < n; ++i) {
        const int j0 = i - kd > 0 ? i - kd : 0;
        for (int j = j0; j <= i; ++j) {
            double s = M[(size_t)i * n + j];
            const int k0 = (j - kd > j0) ? j - kd : j0;
            for (int k = k0; k < j; ++k) s -= LB(i, k) * LB(j, k);
            if (i == j) LB(i, i) = sqrt(s);
            else LB(i, j) = s / LB(j, j);
        }
    }
    free(M);
    return m;
}

int orc_n_red(const orc_model* m) { return m->n_red; }
int orc_n_full(const orc_model* m) { return m->n_full; }
void orc_red2full(const orc_model* m, int* out) { memcpy(out, m->red2full, (size_t)m->n_red * sizeof(int)); }
int orc_drag_table(const orc_model* m, int* pos, double* fac) {
    memcpy(pos, m->drag_pos, (size_t)m->n_drag * sizeof(int));
    memcpy(fac, m->drag_fac, (size_t)m->n_drag * sizeof(double));
    return m->n_drag;
}

/* b <- Minv b  (reduced) */
void orc_solve(const orc_model* m, double* b) {
    const int n = m->n_red, kd = ORC_KD, w = kd + 1;
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        const int j0 = i - kd > 0 ? i - kd : 0;
        for (int j = j0; j < i; ++j) s -= LB(i, j) * b[j];
        b[i] = s / LB(i, i);
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        const int j1 = i + kd < n - 1 ? i + kd : n - 1;
        for (int j = i + 1; j <= j1; ++j) s -= LB(j, i) * b[j];
        b[i] = s / LB(i, i);
    }
}

/* euler_bernoulli_beam.py:270-289 wrapping :167-197 */
void orc_internal_force(const orc_model* m, const double* q_red, double* k_red) {
    const int nf = m->n_full;
    double* xf = (double*)calloc((size_t)2 * nf, sizeof(double));
    double* ff = xf + nf;
    for (int i = 0; i < m->n_red; ++i) xf[m->red2full[i]] = q_red[i];
    for (int e = 0; e < m->n_seg; ++e) {
        const double* xe = xf + 3 * e;
        double fe[6];
        if (m->nonlinear[e]) {
            orc_elem_force_nonlinear(m->L[e], m->E[e] * m->A[e], m->E[e] * m->I[e], xe, m->corrected_axial, fe);
        } else {
            double Ke[36];
            orc_elem_stiff_linear(m->L[e], m->E[e], m->I[e], m->A[e], Ke);
            for (int a = 0; a < 6; ++a) {
                double s = 0.0;
                for (int b = 0; b < 6; ++b) s += Ke[a * 6 + b] * xe[b];
                fe[a] = s;
            }
        }
        for (int a = 0; a < 6; ++a) ff[3 * e + a] += fe[a];
    }
    for (int i = 0; i < m->n_red; ++i) k_red[i] = ff[m->red2full[i]];
    free(xf);
}

/* fluid_forces.py:103-142 : F = -c v|v| on the free transverse DOFs */
void orc_drag(const orc_model* m, const double* x, double* f) {
    const int n = m->n_red;
    memset(f, 0, (size_t)n * sizeof(double));
    for (int i = 0; i < m->n_drag; ++i) {
        const double v = x[n + m->drag_pos[i]];
        f[m->drag_pos[i]] = -m->drag_fac[i] * v * fabs(v);
    }
}

/* gravity_forces.py:66-148 -- as shipped: indices 3i.. are applied to the REDUCED vector */
void orc_gravity(const orc_model* m, const double* x, double* f) {
    const int n = m->n_red;
    memset(f, 0, (size_t)n * sizeof(double));
    const double gx = m->g[0], gy = m->g[1];
    for (int i = 0; i < m->n_seg; ++i) {
        const int sp = 3 * i + 2, ep = 3 * (i + 1) + 2;
        double phi;
        if (sp < n && ep < n) phi = 0.5 * (x[sp] + x[ep]);
        else if (sp < n) phi = x[sp];
        else if (ep < n) phi = x[ep];
        else phi = 0.0;
        const double c = cos(phi), s = sin(phi);
        const double fa = (c * gx + s * gy) * m->seg_mass[i] * 0.5;
        const double ft = (-s * gx + c * gy) * m->seg_mass[i] * 0.5;
        if (3 * i < n) f[3 * i] += fa;
        if (3 * i + 1 < n) f[3 * i + 1] += ft;
        if (3 * (i + 1) < n) f[3 * (i + 1)] += fa;
        if (3 * (i + 1) + 1 < n) f[3 * (i + 1) + 1] += ft;
    }
}

/* force_registry.py:59-79 with the auto-registered forces of dynamic_beam_model.py:220-241 */
void orc_forces(const orc_model* m, const double* x, double* f) {
    const int n = m->n_red;
    memset(f, 0, (size_t)n * sizeof(double));
    double* tmp = (double*)malloc((size_t)n * sizeof(double));
    if (m->fluid_on) { orc_drag(m, x, tmp); for (int i = 0; i < n; ++i) f[i] += tmp[i]; }
    if (m->grav_on) { orc_gravity(m, x, tmp); for (int i = 0; i < n; ++i) f[i] += tmp[i]; }
    free(tmp);
}

/* dynamic_beam_model.py:343-362 : xdot = system(x) + input(x,u,t) ; u may be NULL (zero) */
void orc_rhs(const orc_model* m, const double* x, const double* u, double* xdot) {
    const int n = m->n_red;
    double* k = (double*)malloc((size_t)2 * n * sizeof(double));
    double* f = k + n;
    orc_internal_force(m, x, k);
    orc_forces(m, x, f);
    for (int i = 0; i < n; ++i) {
        xdot[i] = x[n + i];
        xdot[n + i] = -k[i] + f[i] + (u ? u[i] : 0.0);
    }
    orc_solve(m, xdot + n);
    free(k);
}

/* One RK4 trajectory with the examples' forcing (example_utilities.py:144-148):
 * u[idx] = amp while t < duration, else 0; idx is a REDUCED position index
 * (negative counts from the end, -2 = tip w of a cantilever).  The clock accumulates
 * by addition (t <- t + dt), stage times t, t + 0.5*dt, t + dt
 * (tests/golden/make_golden.py:rk4).  Returns the final clock value. */
double orc_rk4_impulse(const orc_model* m, double* x, double t0, double dt, int n_steps, double amp,
                       double duration, int idx) {
    const int n = m->n_red, N = 2 * n;
    if (idx < 0) idx += n;
    double* w = (double*)malloc((size_t)(6 * N + n) * sizeof(double));
    double *k1 = w, *k2 = w + N, *k3 = w + 2 * N, *k4 = w + 3 * N, *xs = w + 4 * N, *u = w + 5 * N;
    memset(u, 0, (size_t)n * sizeof(double));
    double t = t0;
    for (int s = 0; s < n_steps; ++s) {
        const double th = t + 0.5 * dt, t1 = t + dt;
        u[idx] = t < duration ? amp : 0.0;
        orc_rhs(m, x, u, k1);
        for (int i = 0; i < N; ++i) xs[i] = x[i] + (0.5 * dt) * k1[i];
        u[idx] = th < duration ? amp : 0.0;
        orc_rhs(m, xs, u, k2);
        for (int i = 0; i < N; ++i) xs[i] = x[i] + (0.5 * dt) * k2[i];
        orc_rhs(m, xs, u, k3);
        for (int i = 0; i < N; ++i) xs[i] = x[i] + dt * k3[i];
        u[idx] = t1 < duration ? amp : 0.0;
        orc_rhs(m, xs, u, k4);
        for (int i = 0; i < N; ++i) x[i] = x[i] + (dt / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        t = t + dt;
    }
    free(w);
    return t;
}

/* RK4 with a held (constant over the call) generalized force vector u[n] (may be NULL) */
void orc_rk4_held(const orc_model* m, double* x, double dt, int n_steps, const double* u) {
    const int n = m->n_red, N = 2 * n;
    double* w = (double*)malloc((size_t)(5 * N) * sizeof(double));
    double *k1 = w, *k2 = w + N, *k3 = w + 2 * N, *k4 = w + 3 * N, *xs = w + 4 * N;
    for (int s = 0; s < n_steps; ++s) {
        orc_rhs(m, x, u, k1);
        for (int i = 0; i < N; ++i) xs[i] = x[i] + (0.5 * dt) * k1[i];
        orc_rhs(m, xs, u, k2);
        for (int i = 0; i < N; ++i) xs[i] = x[i] + (0.5 * dt) * k2[i];
        orc_rhs(m, xs, u, k3);
        for (int i = 0; i < N; ++i) xs[i] = x[i] + dt * k3[i];
        orc_rhs(m, xs, u, k4);
        for (int i = 0; i < N; ++i) x[i] = x[i] + (dt / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    free(w);
}

/* Closed-loop RK4 of examples/lqr_control.py:95-111: at every stage the total input is
 * u = K (r - x_stage) [control/full_state_linear.py:81] plus the tip impulse; K is [n][2n] row-major,
 * r [2n] or NULL (= 0).  Clock as in orc_rk4_impulse. */
double orc_rk4_feedback(const orc_model* m, double* x, double t0, double dt, int n_steps, const double* K,
                        const double* r, double amp, double duration, int idx) {
    const int n = m->n_red, N = 2 * n;
    if (idx < 0) idx += n;
    double* w = (double*)malloc((size_t)(6 * N + n) * sizeof(double));
    double *k1 = w, *k2 = w + N, *k3 = w + 2 * N, *k4 = w + 3 * N, *xs = w + 4 * N, *u = w + 5 * N;
    double t = t0;
#define ORC_FEEDBACK(state, tt)                                               \
    for (int i = 0; i < n; ++i) {                                             \
        double s_ = 0.0;                                                      \
        for (int j = 0; j < N; ++j) s_ += K[(size_t)i * N + j] * ((r ? r[j] : 0.0) - (state)[j]); \
        u[i] = s_;                                                            \
    }                                                                         \
    if ((tt) < duration) u[idx] += amp;
    for (int s = 0; s < n_steps; ++s) {
        const double th = t + 0.5 * dt, t1 = t + dt;
        ORC_FEEDBACK(x, t)
        orc_rhs(m, x, u, k1);
        for (int i = 0; i < N; ++i) xs[i] = x[i] + (0.5 * dt) * k1[i];
        ORC_FEEDBACK(xs, th)
        orc_rhs(m, xs, u, k2);
        for (int i = 0; i < N; ++i) xs[i] = x[i] + (0.5 * dt) * k2[i];
        ORC_FEEDBACK(xs, th)
        orc_rhs(m, xs, u, k3);
        for (int i = 0; i < N; ++i) xs[i] = x[i] + dt * k3[i];
        ORC_FEEDBACK(xs, t1)
        orc_rhs(m, xs, u, k4);
        for (int i = 0; i < N; ++i) x[i] = x[i] + (dt / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
        t = t + dt;
    }
#undef ORC_FEEDBACK
    free(w);
    return t;
}

/* Implicit midpoint rule on M a = F(q, v, t) = -k(q) + forces(x) + u(t):
 *     a_m solves  M a_m = F(q_m, v_m, t_m),  q_m = q0 + h/2 v0 + alpha a_m,  v_m = v0 + h/2 a_m,  t_m = t0 + h/2,
 *     q1 = q0 + h v0 + 2 alpha a_m,  v1 = v0 + h a_m,                        alpha = h^2/4
 * (for linear systems the trapezoidal rule / Newmark average acceleration: A-stable, second order, no numerical
 * damping), each step solved by n_iter iterations  a_m <- Ainv (F(q_m(a_m), v_m(a_m), t_m) + alpha K0 a_m),
 * A = M + alpha K0, K0 = sum of the element tangent stiffnesses AT q = 0: the linear element stiffness
 * (segments.py:32-62), and for a nonlinear element the same matrix except that the SHIPPED f1 (segments.py:178-208,
 * SURVEY App. B-1) has no -EA/L u2 coupling: its row is [EA/L, 0, 0, 0, 0, 0] unless corrected_axial -- K0, and
 * with it A, is then not symmetric.  Inputs are sampled at the step MIDPOINT (u[idx] = amp while t_m < duration), so a
 * piecewise-constant input whose switch times fall on step boundaries is integrated exactly.
 * Dense LU of A without pivoting (A is diagonally dominated by M / alpha-scaled K0 blocks; test sizes).
 * a_guess [n] (may be NULL): in/out starting iterate (the previous step's a_m).  Returns the final clock. */
double orc_implicit(const orc_model* m, double* x, double t0, double h, int n_steps, int n_iter, double amp,
                    double duration, int idx, const double* u_held) {
    const int n = m->n_red, nf = m->n_full, N = 2 * n;
    if (idx < 0) idx += n;
    const double alpha = 0.25 * h * h, hh = 0.5 * h;
    /* K0 (reduced, dense) and A = M + alpha K0 */
    double* Kf = (double*)calloc((size_t)nf * nf, sizeof(double));
    for (int e = 0; e < m->n_seg; ++e) {
        double Ke[36];
        orc_elem_stiff_linear(m->L[e], m->E[e], m->I[e], m->A[e], Ke);
        if (m->nonlinear[e] && !m->corrected_axial) Ke[0 * 6 + 3] = 0.0;   /* d f1 / d u2 of the shipped f1 at q = 0 */
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) Kf[(size_t)(3 * e + a) * nf + 3 * e + b] += Ke[a * 6 + b];
    }
    double* K0 = (double*)malloc((size_t)n * n * sizeof(double));
    double* Am = (double*)malloc((size_t)n * n * sizeof(double));
    orc_mass_dense(m, Am);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            K0[(size_t)i * n + j] = Kf[(size_t)m->red2full[i] * nf + m->red2full[j]];
            Am[(size_t)i * n + j] += alpha * K0[(size_t)i * n + j];
        }
    free(Kf);
    /* dense LU, A = L U in place (unit lower) */
    for (int k = 0; k < n; ++k)
        for (int i = k + 1; i < n; ++i) {
            const double f = Am[(size_t)i * n + k] / Am[(size_t)k * n + k];
            Am[(size_t)i * n + k] = f;
            if (f != 0.0)
                for (int j = k + 1; j < n; ++j) Am[(size_t)i * n + j] -= f * Am[(size_t)k * n + j];
        }
    double* w = (double*)malloc((size_t)(2 * N + 6 * n) * sizeof(double));
    double *xm = w, *tmp = w + N, *u = w + 2 * N, *qp = u + n, *am = qp + n, *g = am + n, *kf = g + n;
    double t = t0;
    memset(am, 0, (size_t)n * sizeof(double));
    for (int s = 0; s < n_steps; ++s) {
        const double tm = t + hh;
        for (int i = 0; i < n; ++i) u[i] = u_held ? u_held[i] : 0.0;
        if (tm < duration) u[idx] += amp;
        for (int i = 0; i < n; ++i) qp[i] = x[i] + hh * x[n + i];
        /* starting iterate: the previous step's a_m (0 for the first step) */
        for (int it = 0; it < n_iter; ++it) {
            for (int i = 0; i < n; ++i) { xm[i] = qp[i] + alpha * am[i]; xm[n + i] = x[n + i] + hh * am[i]; }
            orc_internal_force(m, xm, kf);
            orc_forces(m, xm, tmp);
            for (int i = 0; i < n; ++i) {
                double k0a = 0.0;
                for (int j = 0; j < n; ++j) k0a += K0[(size_t)i * n + j] * am[j];
                g[i] = -kf[i] + tmp[i] + u[i] + alpha * k0a;
            }
            for (int i = 0; i < n; ++i) {          /* L y = g */
                double v = g[i];
                for (int k = 0; k < i; ++k) v -= Am[(size_t)i * n + k] * g[k];
                g[i] = v;
            }
            for (int i = n - 1; i >= 0; --i) {     /* U a = y */
                double v = g[i];
                for (int k = i + 1; k < n; ++k) v -= Am[(size_t)i * n + k] * g[k];
                g[i] = v / Am[(size_t)i * n + i];
            }
            memcpy(am, g, (size_t)n * sizeof(double));
        }
        for (int i = 0; i < n; ++i) {
            x[i] = x[i] + h * x[n + i] + 2.0 * alpha * am[i];
            x[n + i] = x[n + i] + h * am[i];
        }
        t = t + h;
    }
    free(w); free(K0); free(Am);
    return t;
}

/* The numerically DAMPED member of the same family: generalised-alpha (Chung & Hulbert 1993) with spectral radius rho
 * at infinite frequency (rho = 1: the scheme above, no damping; rho = 0: asymptotic annihilation), second order, written
 * so that the iteration keeps the form of orc_implicit (unknown z = a_{n+1-alpha_m}, constant matrix M + kappa K0):
 *     alpha_m = (2 rho - 1)/(rho + 1),  alpha_f = rho/(rho + 1),  gamma = 1/2 - alpha_m + alpha_f,  beta = (1 - alpha_m + alpha_f)^2 / 4
 *     M z = F(q_f, v_f, t_f),   q_f = (1 - alpha_f) q_{n+1} + alpha_f q_n,  v_f likewise,  t_f = t_n + (1 - alpha_f) h,
 *     z = (1 - alpha_m) a_{n+1} + alpha_m a_n,
 *     q_{n+1} = q_n + h v_n + h^2 ((1/2 - beta) a_n + beta a_{n+1}),   v_{n+1} = v_n + h ((1 - gamma) a_n + gamma a_{n+1})
 * i.e. q_f = qp + kappa z, v_f = vp + cv z with kappa = (1 - alpha_f) beta h^2 / (1 - alpha_m), cv = (1 - alpha_f) gamma h / (1 - alpha_m)
 * and predictors qp, vp that hold the a_n terms.  a_0 = Minv F(q_0, v_0, t_0) (the RHS at the start of the call); the
 * starting iterate of a step is the previous step's z (a_0 for the first).  OURS: the reference has no integrator; LSODA,
 * which the examples use (example_utilities.py:153-159), damps unresolved modes through its BDF formulas. */
double orc_implicit_alpha(const orc_model* m, double* x, double t0, double h, int n_steps, int n_iter, double rho, double amp,
                          double duration, int idx, const double* u_held) {
    const int n = m->n_red, nf = m->n_full, N = 2 * n;
    if (idx < 0) idx += n;
    const double am_ = (2.0 * rho - 1.0) / (rho + 1.0), af_ = rho / (rho + 1.0);
    const double gam = 0.5 - am_ + af_, bet = 0.25 * (1.0 - am_ + af_) * (1.0 - am_ + af_);
    const double kappa = (1.0 - af_) * bet * h * h / (1.0 - am_), cv = (1.0 - af_) * gam * h / (1.0 - am_);
    const double c_qv = (1.0 - af_) * h, c_qa = (1.0 - af_) * h * h * (0.5 - bet) - kappa * am_;
    const double c_va = (1.0 - af_) * h * (1.0 - gam) - cv * am_, inv1m = 1.0 / (1.0 - am_);
    double* Kf = (double*)calloc((size_t)nf * nf, sizeof(double));
    for (int e = 0; e < m->n_seg; ++e) {
        double Ke[36];
        orc_elem_stiff_linear(m->L[e], m->E[e], m->I[e], m->A[e], Ke);
        if (m->nonlinear[e] && !m->corrected_axial) Ke[0 * 6 + 3] = 0.0;
        for (int a = 0; a < 6; ++a)
            for (int b = 0; b < 6; ++b) Kf[(size_t)(3 * e + a) * nf + 3 * e + b] += Ke[a * 6 + b];
    }
    double* K0 = (double*)malloc((size_t)n * n * sizeof(double));
    double* Am = (double*)malloc((size_t)n * n * sizeof(double));
    orc_mass_dense(m, Am);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            K0[(size_t)i * n + j] = Kf[(size_t)m->red2full[i] * nf + m->red2full[j]];
            Am[(size_t)i * n + j] += kappa * K0[(size_t)i * n + j];
        }
    free(Kf);
    for (int k = 0; k < n; ++k)
        for (int i = k + 1; i < n; ++i) {
            const double f = Am[(size_t)i * n + k] / Am[(size_t)k * n + k];
            Am[(size_t)i * n + k] = f;
            if (f != 0.0)
                for (int j = k + 1; j < n; ++j) Am[(size_t)i * n + j] -= f * Am[(size_t)k * n + j];
        }
    double* w = (double*)malloc((size_t)(2 * N + 8 * n) * sizeof(double));
    double *xm = w, *tmp = w + N, *u = w + 2 * N, *qp = u + n, *vp = qp + n, *z = vp + n, *an = z + n, *g = an + n, *kf = g + n;
    double t = t0;
    /* a_0 from the RHS at t0 */
    for (int i = 0; i < n; ++i) u[i] = u_held ? u_held[i] : 0.0;
    if (t0 < duration) u[idx] += amp;
    orc_rhs(m, x, u, tmp);
    for (int i = 0; i < n; ++i) { an[i] = tmp[n + i]; z[i] = an[i]; }
    for (int s = 0; s < n_steps; ++s) {
        const double tf = t + (1.0 - af_) * h;
        for (int i = 0; i < n; ++i) u[i] = u_held ? u_held[i] : 0.0;
        if (tf < duration) u[idx] += amp;
        for (int i = 0; i < n; ++i) {
            qp[i] = x[i] + c_qv * x[n + i] + c_qa * an[i];
            vp[i] = x[n + i] + c_va * an[i];
        }
        for (int it = 0; it < n_iter; ++it) {
            for (int i = 0; i < n; ++i) { xm[i] = qp[i] + kappa * z[i]; xm[n + i] = vp[i] + cv * z[i]; }
            orc_internal_force(m, xm, kf);
            orc_forces(m, xm, tmp);
            for (int i = 0; i < n; ++i) {
                double k0a = 0.0;
                for (int j = 0; j < n; ++j) k0a += K0[(size_t)i * n + j] * z[j];
                g[i] = -kf[i] + tmp[i] + u[i] + kappa * k0a;
            }
            for (int i = 0; i < n; ++i) {
                double v = g[i];
                for (int k = 0; k < i; ++k) v -= Am[(size_t)i * n + k] * g[k];
                g[i] = v;
            }
            for (int i = n - 1; i >= 0; --i) {
                double v = g[i];
                for (int k = i + 1; k < n; ++k) v -= Am[(size_t)i * n + k] * g[k];
                g[i] = v / Am[(size_t)i * n + i];
            }
            memcpy(z, g, (size_t)n * sizeof(double));
        }
        for (int i = 0; i < n; ++i) {
            const double a1 = (z[i] - am_ * an[i]) * inv1m;
            x[i] = x[i] + h * x[n + i] + h * h * ((0.5 - bet) * an[i] + bet * a1);
            x[n + i] = x[n + i] + h * ((1.0 - gam) * an[i] + gam * a1);
            an[i] = a1;
        }
        t = t + h;
    }
    free(w); free(K0); free(Am);
    return t;
}

/* B independent trajectories of one model (the reference runs such ensembles through
 * multiprocessing.Pool, beam_comparison_fluid.py:82-83); X is [B][2n], amps [B].
 * Returns the number of threads used. */
int orc_rk4_impulse_batch(const orc_model* m, double* X, int B, const double* amps, double t0, double dt,
                          int n_steps, double duration, int idx, int n_threads) {
    const int N = 2 * m->n_red;
    int used = 1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
    used = n_threads > 0 ? n_threads : omp_get_max_threads();
    if (used > B) used = B;
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; ++b) orc_rk4_impulse(m, X + (size_t)b * N, t0, dt, n_steps, amps[b], duration, idx);
    return used;
}
