// crb_math.h -- per-node / per-element arithmetic of the beam hot path.
//
// Everything here is a small inline function of VALUES (no memory, no thread ids), so the
// same source is compiled into the gfx950 kernels (crb_kernels.h) and, for CPU-only
// debugging of the kernel arithmetic, into tests/native/crb_emul.cpp (test harness; it is
// not a product path).  Reference citations are file:line under
// /root/reference/src/continuum_robot/models/.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define CRB_HD __host__ __device__ __forceinline__
#else
#define CRB_HD inline
#endif

namespace crb {

enum : int { KIND_NONE = 0, KIND_LINEAR = 1, KIND_NONLINEAR = 2 };

// Per-element coefficient pack (a1/a3 of SURVEY §8).  One element = the element to the
// LEFT of a slot's node.
//   linear    : c = { EA/L, 12EI/L^3, 6EI/L^2, 4EI/L, 2EI/L, 0 }        (segments.py:32-62)
//   nonlinear : c = { L, EA/L^2, 0.1 EA/L^3, 2 EI/L^3, EA/(2L^2), EI/L^2 } (segments.py:128-130; the
//               products of EA, EI and the 1/L^2, 0.1/L^3 prefactors the six forces use)
template <typename T>
struct ElemCoef {
    T c[6];
    int32_t kind;
    int32_t pad;
};

template <typename T>
CRB_HD void elem_coef_build(ElemCoef<T>& e, int kind, double L, double E, double I, double A) {
    const double EA = E * A, EI = E * I;
    e.kind = kind;
    e.pad = 0;
    if (kind == KIND_LINEAR) {
        e.c[0] = T(EA / L);
        e.c[1] = T(12 * EI / (L * L * L));
        e.c[2] = T(6 * EI / (L * L));
        e.c[3] = T(4 * EI / L);
        e.c[4] = T(2 * EI / L);
        e.c[5] = T(0);
    } else if (kind == KIND_NONLINEAR) {
        const double iL2 = 1.0 / (L * L), tenth_iL3 = 0.1 / (L * L * L);
        e.c[0] = T(L);
        e.c[1] = T(EA * iL2);
        e.c[2] = T(EA * tenth_iL3);
        e.c[3] = T(20.0 * EI * tenth_iL3);
        e.c[4] = T(0.5 * EA * iL2);
        e.c[5] = T(EI * iL2);
    } else {
        for (int i = 0; i < 6; ++i) e.c[i] = T(0);
    }
}

// Linear element internal force K_e x_e (segments.py:39-62), split into the halves that go
// to the left node (fl) and the right node (fr).  ql/qr = [u, w, phi] of the two nodes.
template <typename T>
CRB_HD void elem_force_linear(const T* c, const T ql[3], const T qr[3], T fl[3], T fr[3]) {
    const T du = ql[0] - qr[0];
    const T dw = ql[1] - qr[1];
    const T fa = c[0] * du;
    const T fw = c[1] * dw - c[2] * (ql[2] + qr[2]);
    const T m0 = -c[2] * dw;
    fl[0] = fa;
    fl[1] = fw;
    fl[2] = m0 + c[3] * ql[2] + c[4] * qr[2];
    fr[0] = -fa;
    fr[1] = -fw;
    fr[2] = m0 + c[4] * ql[2] + c[3] * qr[2];
}

// Nonlinear (von Karman) element internal force, segments.py:159-472, returned in the
// reference's order [f1,f3,f4 | f2,f5,f6] = forces on [u1,w1,th1 | u2,w2,th2]
// (segments.py:146-155).  The 30-term sums are regrouped in a = th1*L, b = th2*L,
// du = u1-u2, dw = w1-w2; every literal is the reference's.  Where the reference's expanded
// literals are not exactly 2x / 3x each other (sympy float artefacts such as
// 7.71428571428601 vs 2*3.857142857143) the remainder is kept as an explicit w1*w2 term, so
// the regrouped polynomial equals the shipped one identically.  f5 = -f3 term by term in the
// reference.  corrected == false keeps the shipped f1 (it lacks the
// -EA/L*u2 coupling); corrected == true uses f1 = -f2.
// base + c * ww, dropped at compile time when the literal remainder c is exactly 0
template <typename T>
CRB_HD T plus_remainder(T base, double c, T ww) {
    return c == 0.0 ? base : base + T(c) * ww;
}

// (literal form: c = { L, EA, EI, 1/L^2, 0.1/L^3 }, not the ElemCoef pack)
template <typename T>
CRB_HD void elem_force_nonlinear_literal(const T* c, const T ql[3], const T qr[3], bool corrected, T fl[3], T fr[3]) {
    const T L = c[0], A = c[1], D = c[2], iL2 = c[3], tenth_iL3 = c[4];
    const T u1 = ql[0], w1 = ql[1], u2 = qr[0], w2 = qr[1];
    const T a = ql[2] * L, b = qr[2] * L;
    const T du = u1 - u2, dw = w1 - w2;
    const T ww = w1 * w2, dw2 = dw * dw;
    const T a2 = a * a, b2 = b * b, ab = a * b;
    const T Ldu = L * du;

    // ---- f1, f2 (segments.py:178-208, 227-258)
    const T T0 = T(0.6) * dw - T(0.05) * (a + b);
    const T P = a * (T(0.0666666666666665) * a - T(0.0166666666666667) * b - T(0.05) * dw) -
                b * (T(0.0166666666666667) * a - T(0.0666666666666667) * b + T(0.05) * dw);
    const T f2 = A * iL2 * (P - Ldu + dw * T0);
    const T f1 = corrected ? -f2 : A * iL2 * (L * u1 - P - (u2 + dw) * T0);

    // ---- f3 = -f5 (segments.py:279-314, 386-421)
    const T Qa = plus_remainder<T>(T(3.8571428571413) * dw2, 2.0 * 3.8571428571413 - 7.7142857142826, ww);
    const T Qb = plus_remainder<T>(T(3.857142857143) * dw2, 2.0 * 3.857142857143 - 7.71428571428601, ww);
    const T Cw = dw * plus_remainder<T>(T(10.2857142857147) * dw2, 3.0 * 10.2857142857147 - 30.857142857144, ww);
    const T P3 = T(0.0357142857143344) * (a2 * a + b2 * b) - T(0.107142857143003) * ab * (a + b) +
                 T(1.28571428571433) * (a2 + b2) * dw + Ldu * (a + b) - a * Qa - b * Qb - T(12.0) * Ldu * dw + Cw;
    const T f3 = tenth_iL3 * (A * P3 + D * (T(120.0) * dw - T(60.0) * (a + b)));

    // ---- f4 (segments.py:335-365)
    const T Q4 = plus_remainder<T>(T(0.128571428571433) * dw2, 2.0 * 0.128571428571433 - 0.257142857142867, ww);
    const T C4 = dw * plus_remainder<T>(T(0.128571428571377) * dw2, 3.0 * 0.128571428571377 - 0.38571428571413, ww);
    const T P4 = T(0.0285714285714391) * a2 * a - T(0.0107142857142861) * a2 * b + T(0.0107142857142719) * a2 * dw +
                 T(0.00714285714286444) * a * b2 - T(0.0214285714286007) * ab * dw - T(0.133333333333333) * a * Ldu +
                 a * Q4 - T(0.00357142857143344) * b2 * b - T(0.0107142857142719) * b2 * dw +
                 T(0.0333333333333333) * b * Ldu + T(0.1) * Ldu * dw - C4;
    const T f4 = iL2 * (A * P4 + D * (T(4.0) * a + T(2.0) * b - T(6.0) * dw));

    // ---- f6 (segments.py:442-472)
    const T Q6 = plus_remainder<T>(T(0.128571428571428) * dw2, 2.0 * 0.128571428571428 - 0.257142857142856, ww);
    const T C6 = dw * plus_remainder<T>(T(0.128571428571433) * dw2, 3.0 * 0.128571428571433 - 0.3857142857143, ww);
    const T P6 = -T(0.00357142857143344) * a2 * a + T(0.00714285714286356) * a2 * b - T(0.0107142857143003) * a2 * dw -
                 T(0.0107142857142932) * a * b2 - T(0.021428571428558) * ab * dw + T(0.0333333333333333) * a * Ldu +
                 T(0.0285714285714271) * b2 * b + T(0.0107142857142932) * b2 * dw - T(0.133333333333333) * b * Ldu +
                 b * Q6 + T(0.1) * Ldu * dw - C6;
    const T f6 = iL2 * (A * P6 + D * (T(2.0) * a + T(4.0) * b - T(6.0) * dw));

    fl[0] = f1;
    fl[1] = f3;
    fl[2] = f4;
    fr[0] = f2;
    fr[1] = -f3;
    fr[2] = f6;
}

// The same forces in symmetric variables: s = a+b, d = a-b, p = a*b.  Up to the sympy float
// artefacts above, the shipped polynomials are those of the von Karman element with rational
// coefficients (1/15, 1/60, 27/7, 72/7, 1/28, 3/28, 9/7, 1/35, 3/280, 1/140, 3/140, 2/15,
// 1/30, 9/70), f6 is f4 with the two nodes exchanged, and P3 is symmetric in (a, b):
//   P   = s*(s/15 - dw/20) - p/6
//   P3  = s*((s^2 - 6p)/28 + L*du - 27/7 dw^2) + dw*(9/7 (s^2 - 2p) - 12 L*du + 72/7 dw^2)
//   P4+P6 = s*(s^2/40 - 11/140 p - L*du/10 + 9/70 dw^2) + dw*(-3/70 p + L*du/5 - 9/35 dw^2)
//   P4-P6 = d*(9/280 s^2 - p/20 + 3/140 s*dw - L*du/6 + 9/70 dw^2)
// which is 57 flops (with the prefactor products of the ElemCoef pack) against 135 for the literal form.  Every literal of the reference differs
// from its rational by <= 1.5e-12 relative (largest: 0.0214285714286007 vs 3/140), i.e. six
// orders below the 1e-6 parity tolerance; tests/native bounds the difference between the two
// forms.  -DCRB_LITERAL_POLY=1 builds the kernels with the literal form instead.
template <typename T>
CRB_HD void elem_force_nonlinear_sym(const T* c, const T ql[3], const T qr[3], bool corrected, T fl[3], T fr[3]) {
    const T L = c[0], cA1 = c[1], cA3 = c[2], cD3 = c[3], cA4 = c[4], cD4 = c[5];
    const T u1 = ql[0], u2 = qr[0];
    const T a = ql[2] * L, b = qr[2] * L;
    const T du = u1 - u2, dw = ql[1] - qr[1];
    const T s = a + b, d = a - b, p = a * b;
    const T s2 = s * s, dw2 = dw * dw, Ldu = L * du;

    // ---- f1, f2 (segments.py:178-208, 227-258): EA/L^2 * (...)
    const T T0 = T(0.6) * dw - T(0.05) * s;
    const T P = s * (T(1.0 / 15.0) * s - T(0.05) * dw) - T(1.0 / 6.0) * p;
    const T f2 = cA1 * (P - Ldu + dw * T0);
    const T f1 = corrected ? -f2 : cA1 * (L * u1 - P - (u2 + dw) * T0);

    // ---- f3 = -f5 (segments.py:279-314, 386-421): 0.1/L^3 * (EA*P3 + EI*(120 dw - 60 s))
    const T P3 = s * (T(1.0 / 28.0) * (s2 - T(6.0) * p) + Ldu - T(27.0 / 7.0) * dw2) +
                 dw * (T(9.0 / 7.0) * (s2 - T(2.0) * p) - T(12.0) * Ldu + T(72.0 / 7.0) * dw2);
    const T g = T(3.0) * s - T(6.0) * dw;   // 4a+2b-6dw = g + d,  2a+4b-6dw = g - d,  120dw-60s = -20 g
    const T f3 = cA3 * P3 - cD3 * g;

    // ---- f4, f6 (segments.py:335-365, 442-472): 1/L^2 * (EA*P4|6 + EI*(g +- d))
    const T S = s * (T(1.0 / 40.0) * s2 - T(11.0 / 140.0) * p - T(0.1) * Ldu + T(9.0 / 70.0) * dw2) +
                dw * (T(0.2) * Ldu - T(3.0 / 70.0) * p - T(9.0 / 35.0) * dw2);
    const T R = T(9.0 / 280.0) * s2 - T(0.05) * p + T(3.0 / 140.0) * (s * dw) - T(1.0 / 6.0) * Ldu +
                T(9.0 / 70.0) * dw2;
    const T X = cA4 * S + cD4 * g;
    const T Y = d * (cA4 * R + cD4);

    fl[0] = f1;
    fl[1] = f3;
    fl[2] = X + Y;
    fr[0] = f2;
    fr[1] = -f3;
    fr[2] = X - Y;
}

#ifndef CRB_LITERAL_POLY
#define CRB_LITERAL_POLY 0
#endif
template <typename T>
CRB_HD void elem_force_nonlinear(const T* c, const T ql[3], const T qr[3], bool corrected, T fl[3], T fr[3]) {
#if CRB_LITERAL_POLY
    const T iL2 = T(1) / (c[0] * c[0]);
    const T lit[5] = {c[0], c[1] / iL2, c[5] / iL2, iL2, T(0.1) * iL2 / c[0]};
    elem_force_nonlinear_literal<T>(lit, ql, qr, corrected, fl, fr);
#else
    elem_force_nonlinear_sym<T>(c, ql, qr, corrected, fl, fr);
#endif
}

template <typename T>
CRB_HD void elem_force(const ElemCoef<T>& e, const T ql[3], const T qr[3], bool corrected, T fl[3], T fr[3]) {
    if (e.kind == KIND_NONLINEAR) {
        elem_force_nonlinear<T>(e.c, ql, qr, corrected, fl, fr);
    } else if (e.kind == KIND_LINEAR) {
        elem_force_linear<T>(e.c, ql, qr, fl, fr);
    } else {
        fl[0] = fl[1] = fl[2] = fr[0] = fr[1] = fr[2] = T(0);
    }
}

// Gravity of one segment (gravity_forces.py:117-128): half the segment weight, rotated by the
// segment's mean rotation into [axial, transverse].  half_mass = 0.5*rho*A*L.
CRB_HD void crb_sincos(double x, double* s, double* c) {
#if defined(__HIP_DEVICE_COMPILE__)
    ::sincos(x, s, c);
#else
    *s = sin(x);
    *c = cos(x);
#endif
}
CRB_HD void crb_sincos(float x, float* s, float* c) {
#if defined(__HIP_DEVICE_COMPILE__)
    ::sincosf(x, s, c);
#else
    *s = sinf(x);
    *c = cosf(x);
#endif
}

template <typename T>
CRB_HD void gravity_segment(T phi, T gx, T gy, T half_mass, T out[2]) {
    T s, c;
    crb_sincos(phi, &s, &c);
    out[0] = (c * gx + s * gy) * half_mass;
    out[1] = (c * gy - s * gx) * half_mass;
}

// Fluid drag on a transverse DOF (fluid_forces.py:138): -c * v * |v|
CRB_HD double crb_abs(double v) { return __builtin_fabs(v); }
CRB_HD float crb_abs(float v) { return __builtin_fabsf(v); }
template <typename T>
CRB_HD T drag_force(T coef, T v) {
    return -coef * v * crb_abs(v);
}

// ------------------------------------------------------------------ mass blocks / PCR
// Node-block form of the consistent mass matrix (segments.py:64-78 assembled as in
// euler_bernoulli_beam.py:139-161).  The axial DOF (u) decouples from bending (w, phi), so a
// node row carries one scalar triple (a_ax, b_ax, c_ax) and one 2x2 triple (A, B, C), with
//   A = coupling to the node on the left, B = diagonal block, C = coupling to the right.
// 2x2 blocks are stored row-major [ww, wp, pw, pp].
struct NodeBlocks {
    double a_ax, b_ax, c_ax;
    double A[4], B[4], C[4];
};

// contributions of one element (length L, rhoA = rho*A) to its LEFT node's C/B and its RIGHT
// node's A/B
CRB_HD void mass_add_as_right_elem(NodeBlocks& n, double L, double rhoA) {
    // this node is the element's first node: block [0:3,0:3] on the diagonal, [0:3,3:6] to the right
    const double m = rhoA * L / 420.0;
    n.b_ax += 140.0 * m;
    n.c_ax += 70.0 * m;
    n.B[0] += 156.0 * m;
    n.B[1] += -22.0 * L * m;
    n.B[2] += -22.0 * L * m;
    n.B[3] += 4.0 * L * L * m;
    n.C[0] += 54.0 * m;
    n.C[1] += 13.0 * L * m;
    n.C[2] += -13.0 * L * m;
    n.C[3] += -3.0 * L * L * m;
}
CRB_HD void mass_add_as_left_elem(NodeBlocks& n, double L, double rhoA, bool has_left_node) {
    // this node is the element's second node: block [3:6,3:6] on the diagonal, [3:6,0:3] to the left
    const double m = rhoA * L / 420.0;
    n.b_ax += 140.0 * m;
    n.B[0] += 156.0 * m;
    n.B[1] += 22.0 * L * m;
    n.B[2] += 22.0 * L * m;
    n.B[3] += 4.0 * L * L * m;
    if (has_left_node) {
        n.a_ax += 70.0 * m;
        n.A[0] += 54.0 * m;
        n.A[1] += -13.0 * L * m;
        n.A[2] += 13.0 * L * m;
        n.A[3] += -3.0 * L * L * m;
    }
}

// alpha * K0 of one element added to the node blocks, K0 = the element's tangent stiffness AT q = 0 -- the
// iteration matrix A = M + alpha K0 of the implicit stepper (crb_stiff.h) has M's block structure.  For a linear
// element K0 is its stiffness (segments.py:39-62); from elem_force_linear, on [w, phi] the four 2x2 blocks are
//   K11 = [[c1,-c2],[-c2,c3]]  K12 = [[-c1,-c2],[c2,c4]]  K21 = [[-c1,c2],[-c2,c4]]  K22 = [[c1,c2],[c2,c3]]
// with c1 = 12EI/L^3, c2 = 6EI/L^2, c3 = 4EI/L, c4 = 2EI/L, and [[c0,-c0],[-c0,c0]], c0 = EA/L, on the axial DOFs.
// A nonlinear element has the same tangent at q = 0, EXCEPT that the shipped f1 (segments.py:178-208, SURVEY
// App. B-1) lacks the -EA/L u2 coupling: d f1 / d u2 = 0, i.e. the axial block is [[c0, 0],[-c0, c0]] (`axial_12`
// false) and A is not symmetric; the cyclic reduction carries the sub- and super-diagonal separately anyway.
CRB_HD void stiff_add_as_right_elem(NodeBlocks& n, double L, double EA, double EI, double alpha, bool axial_12) {
    // this node is the element's first node: K11 on the diagonal, K12 to the right
    const double c0 = alpha * EA / L, c1 = alpha * 12 * EI / (L * L * L), c2 = alpha * 6 * EI / (L * L), c3 = alpha * 4 * EI / L,
                 c4 = alpha * 2 * EI / L;
    n.b_ax += c0;
    if (axial_12) n.c_ax += -c0;
    n.B[0] += c1; n.B[1] += -c2; n.B[2] += -c2; n.B[3] += c3;
    n.C[0] += -c1; n.C[1] += -c2; n.C[2] += c2; n.C[3] += c4;
}
CRB_HD void stiff_add_as_left_elem(NodeBlocks& n, double L, double EA, double EI, double alpha, bool has_left_node) {
    // this node is the element's second node: K22 on the diagonal, K21 to the left
    const double c0 = alpha * EA / L, c1 = alpha * 12 * EI / (L * L * L), c2 = alpha * 6 * EI / (L * L), c3 = alpha * 4 * EI / L,
                 c4 = alpha * 2 * EI / L;
    n.b_ax += c0;
    n.B[0] += c1; n.B[1] += c2; n.B[2] += c2; n.B[3] += c3;
    if (has_left_node) {
        n.a_ax += -c0;
        n.A[0] += -c1; n.A[1] += c2; n.A[2] += -c2; n.A[3] += c4;
    }
}

// linear coefficient pack {EA/L, 12EI/L^3, 6EI/L^2, 4EI/L, 2EI/L} of an element from EITHER pack of ElemCoef
// (the nonlinear pack holds {L, EA/L^2, 0.1EA/L^3, 2EI/L^3, EA/(2L^2), EI/L^2})
template <typename T>
CRB_HD void elem_linear_coefs(const T* c, int kind, T out[5]) {
    if (kind == KIND_NONLINEAR) {
        const T L = c[0];
        out[0] = c[1] * L;
        out[1] = T(6) * c[3];
        out[2] = T(6) * c[5];
        out[3] = T(4) * c[5] * L;
        out[4] = T(2) * c[5] * L;
    } else {   // KIND_LINEAR (KIND_NONE: all zero already)
        for (int i = 0; i < 5; ++i) out[i] = c[i];
    }
}

// Boundary conditions (euler_bernoulli_beam.py:240-265) in place of physically removing
// rows/columns: a constrained DOF keeps its slot, its row/column become the identity.
// free_* are this node's masks, l_* / r_* the neighbours' (a missing neighbour = not free).
CRB_HD void mass_apply_masks(NodeBlocks& n, bool fu, bool fw, bool fp, bool lu, bool lw, bool lp, bool ru, bool rw,
                             bool rp) {
    if (!fu) { n.a_ax = 0; n.c_ax = 0; n.b_ax = 1; }
    if (!lu) n.a_ax = 0;
    if (!ru) n.c_ax = 0;
    const bool fr[2] = {fw, fp}, lf[2] = {lw, lp}, rf[2] = {rw, rp};
    for (int r = 0; r < 2; ++r)
        for (int cc = 0; cc < 2; ++cc) {
            if (!fr[r] || !lf[cc]) n.A[2 * r + cc] = 0;
            if (!fr[r] || !rf[cc]) n.C[2 * r + cc] = 0;
            if (!fr[r] || !fr[cc]) n.B[2 * r + cc] = (r == cc && !fr[r]) ? 1.0 : 0.0;
        }
}

CRB_HD void inv2(const double M[4], double R[4]) {
    const double det = M[0] * M[3] - M[1] * M[2];
    const double id = 1.0 / det;
    R[0] = M[3] * id;
    R[1] = -M[1] * id;
    R[2] = -M[2] * id;
    R[3] = M[0] * id;
}
CRB_HD void mul2(const double X[4], const double Y[4], double R[4]) {
    R[0] = X[0] * Y[0] + X[1] * Y[2];
    R[1] = X[0] * Y[1] + X[1] * Y[3];
    R[2] = X[2] * Y[0] + X[3] * Y[2];
    R[3] = X[2] * Y[1] + X[3] * Y[3];
}

// One level of parallel cyclic reduction on the (constant) mass matrix, factor phase.
// me = this row, lo = row i-s, hi = row i+s (has_lo / has_hi false outside the beam).
// Outputs the level's elimination multipliers
//     al = -A_i * inv(B_lo),  ga = -C_i * inv(B_hi)     (scalar for axial, 2x2 for bending)
// and the row after the level.  The solve phase only ever needs al/ga:
//     r_i <- r_i + al * r_{i-s} + ga * r_{i+s}.
struct PcrLevel {
    double al_ax, ga_ax;
    double al[4], ga[4];
};
CRB_HD void pcr_factor_level(const NodeBlocks& me, const NodeBlocks& lo, bool has_lo, const NodeBlocks& hi, bool has_hi,
                             PcrLevel& lv, NodeBlocks& out) {
    out = me;
    lv.al_ax = lv.ga_ax = 0.0;
    for (int k = 0; k < 4; ++k) lv.al[k] = lv.ga[k] = 0.0;
    out.a_ax = out.c_ax = 0.0;
    for (int k = 0; k < 4; ++k) out.A[k] = out.C[k] = 0.0;
    if (has_lo) {
        lv.al_ax = -me.a_ax / lo.b_ax;
        out.b_ax += lv.al_ax * lo.c_ax;
        out.a_ax = lv.al_ax * lo.a_ax;
        double Bi[4], t[4];
        inv2(lo.B, Bi);
        mul2(me.A, Bi, t);
        for (int k = 0; k < 4; ++k) lv.al[k] = -t[k];
        mul2(lv.al, lo.C, t);
        for (int k = 0; k < 4; ++k) out.B[k] += t[k];
        mul2(lv.al, lo.A, out.A);
    }
    if (has_hi) {
        lv.ga_ax = -me.c_ax / hi.b_ax;
        out.b_ax += lv.ga_ax * hi.a_ax;
        out.c_ax = lv.ga_ax * hi.c_ax;
        double Bi[4], t[4];
        inv2(hi.B, Bi);
        mul2(me.C, Bi, t);
        for (int k = 0; k < 4; ++k) lv.ga[k] = -t[k];
        mul2(lv.ga, hi.A, t);
        for (int k = 0; k < 4; ++k) out.B[k] += t[k];
        mul2(lv.ga, hi.C, out.C);
    }
}

// Size of a level's multipliers relative to 1, with the rotation DOF scaled by a
// characteristic length Lc so that force/moment units compare (used to decide how many
// levels the solve needs: below the unit roundoff 2^-53 a level no longer changes an fp64 result).
CRB_HD double pcr_level_norm(const PcrLevel& lv, double Lc) {
    double m = fabs(lv.al_ax);
    const double v[9] = {fabs(lv.ga_ax),       fabs(lv.al[0]),      fabs(lv.al[1]) * Lc, fabs(lv.al[2]) / Lc, fabs(lv.al[3]),
                         fabs(lv.ga[0]),       fabs(lv.ga[1]) * Lc, fabs(lv.ga[2]) / Lc, fabs(lv.ga[3])};
    for (int k = 0; k < 9; ++k) m = v[k] > m ? v[k] : m;
    return m;
}

// Solve-phase tables as the kernels read them (T = plan dtype).
//   level record : [al_ax, ga_ax, al[4], ga[4]]  -> 10 values per slot per level
//   final record : [1/b_ax, inv(B)[4]]           -> 5 values per slot (padded to 6)
constexpr int PCR_LEVEL_VALS = 10;
constexpr int PCR_FINAL_VALS = 6;

template <typename T>
CRB_HD void pcr_apply_level(const T* cf, const T rlo[3], const T rhi[3], T r[3]) {
    const T r0 = r[0] + cf[0] * rlo[0] + cf[1] * rhi[0];
    const T r1 = r[1] + cf[2] * rlo[1] + cf[3] * rlo[2] + cf[6] * rhi[1] + cf[7] * rhi[2];
    const T r2 = r[2] + cf[4] * rlo[1] + cf[5] * rlo[2] + cf[8] * rhi[1] + cf[9] * rhi[2];
    r[0] = r0;
    r[1] = r1;
    r[2] = r2;
}
template <typename T>
CRB_HD void pcr_apply_final(const T* cf, const T r[3], T x[3]) {
    x[0] = cf[0] * r[0];
    x[1] = cf[1] * r[1] + cf[2] * r[2];
    x[2] = cf[3] * r[1] + cf[4] * r[2];
}

}  // namespace crb
