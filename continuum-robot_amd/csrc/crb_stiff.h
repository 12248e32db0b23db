// crb_stiff.h -- implicit fixed-step stepper for the stiff end of the reference's call sites (SURVEY f-2): the
// examples integrate 1 s with solve_ivp(method="LSODA") (examples/example_utilities.py:153-159,
// examples/lqr_control.py:117-125) because the beam ODE is stiff (|lambda|max ~ 1e5..1e6 1/s): explicit RK4 / RK45 are
// stability-limited to dt <= ~7e-5 s.  This is the implicit midpoint rule (for linear systems the trapezoidal rule /
// Newmark average acceleration: second order, A-stable, no numerical damping) on
// M a = F(q, v, t) = -k(q) + f_drag(v) + f_gravity(q) + u(t):
//     M a_m = F(q_m, v_m, t_m),   q_m = q0 + h/2 v0 + alpha a_m,   v_m = v0 + h/2 a_m,   t_m = t0 + h/2,   alpha = h^2/4
//     q1 = q0 + h v0 + 2 alpha a_m,   v1 = v0 + h a_m
// solved for a_m by the modified-Newton / fixed-point iteration with the CONSTANT matrix A = M + alpha K0
//     a_m <- Ainv ( F(q_m(a_m), v_m(a_m), t_m) + alpha K0 a_m ),     K0 = element tangent stiffness at q = 0 (crb_math.h)
// (no product with M is needed in this form).  For linear elements -k(q_m) + alpha K0 a_m does not depend on a_m, so the
// first iteration is exact up to the drag / gravity dependence on the state, which is not stiff; for nonlinear elements
// K0 carries their stiff part (incl. the unsymmetric axial tangent of the shipped f1) and the geometric terms are
// O(q).  Two iterations reproduce the converged step in every case measured (oracle, tests/golden/g8).  Inputs are
// sampled at the step midpoint: a piecewise-constant u(t) that switches on step boundaries is integrated exactly.
// A has M's block structure (axial scalar + 2x2 bending blocks per node, block tridiagonal), so Ainv is the same
// parallel cyclic reduction as Minv with a second set of precomputed multipliers (crb_assemble_kernel with
// AsmParams.alpha) -- all ceil(log2 S) levels, A is not as diagonally dominant as M.  One thread per node, whole
// rollout in one launch, state in registers.
#pragma once
#include <hip/hip_runtime.h>

#include "crb_generic.h"

namespace crb {

template <typename T>
struct StiffParams {
    const T* a_levels;        // [levels][S][PCR_LEVEL_VALS] of A (+ beam * alv_stride)
    const T* a_final;         // [S][PCR_FINAL_VALS] of A      (+ beam * afin_stride)
    size_t alv_stride, afin_stride;
    double h;
    int n_iter;
};

template <typename T, int LV, int MAXT, int MINW>
__global__ void __launch_bounds__(MAXT, MINW) crb_implicit_kernel(const KParams<T> p, const StiffParams<T> q) {
    const int NT = blockDim.x;
    const Lds<T> lds = carve_lds<T>(NT);
    Topo tp;
    tp.t = threadIdx.x;
    tp.lane = tp.t & 63;
    tp.S = p.S;
    tp.lognw = p.lognw;
    tp.nwm1 = (1 << p.lognw) - 1;
    int g;
    if (p.G > 1 || p.lognw == 0) {  // whole beams inside a wave
        g = tp.t / p.S;
        tp.j = tp.t - g * p.S;
        tp.base = g * p.S;
    } else {  // one beam per workgroup, slots interleaved over the waves
        g = 0;
        tp.j = (tp.lane << p.lognw) + (tp.t >> 6);
        tp.base = 0;
    }
    const int beam = blockIdx.x * p.G + g;
    tp.valid = (g < p.G) && (tp.j < p.S) && (beam < p.B);
    tp.beam = tp.valid ? beam : 0;
    if (!tp.valid) { tp.j = 0; tp.S = 1; tp.base = tp.t; tp.nwm1 = 0; }   // padding thread: an isolated dummy node
    const bool valid = tp.valid;

    SlotConst<T> sc;
    SolveCoef<T, LV> cf;   // tables of A
    if (valid) {
        sc = p.slot[size_t(beam) * p.slot_stride + tp.j];
#pragma unroll
        for (int l = 0; l < LV; ++l) {
            const T* src = q.a_levels + size_t(beam) * q.alv_stride + (size_t(l) * size_t(p.S) + size_t(tp.j)) * PCR_LEVEL_VALS;
#pragma unroll
            for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = src[k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) cf.fin[k] = q.a_final[size_t(beam) * q.afin_stride + size_t(tp.j) * PCR_FINAL_VALS + k];
    } else {
        sc.elem.kind = KIND_NONE;
#pragma unroll
        for (int k = 0; k < 6; ++k) sc.elem.c[k] = T(0);
        sc.drag = sc.half_mass = T(0);
        sc.mask[0] = sc.mask[1] = sc.mask[2] = T(0);
        sc.grav.phiA = sc.grav.phiB = -1;
#pragma unroll
        for (int c = 0; c < 3; ++c) { sc.grav.segA[c] = sc.grav.segB[c] = -1; sc.grav.comp[c] = 0; }
#pragma unroll
        for (int l = 0; l < LV; ++l)
#pragma unroll
            for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = T(0);
#pragma unroll
        for (int k = 0; k < 5; ++k) cf.fin[k] = T(0);
    }

    const size_t node = size_t(tp.j + p.off);
    const size_t plane = size_t(p.n_node) * 4;
    const size_t xoff = valid ? (size_t(beam) * 2 * plane + node * 4) : 0;
    const size_t aoff = valid ? (size_t(beam) * plane + node * 4) : 0;
    T q0[3] = {T(0), T(0), T(0)}, v0[3] = {T(0), T(0), T(0)}, am[3] = {T(0), T(0), T(0)}, uh[3] = {T(0), T(0), T(0)};
    T amp = T(0);
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            q0[c] = p.x[xoff + c] * sc.mask[c];
            v0[c] = p.x[xoff + plane + c] * sc.mask[c];
            if (p.u_held) uh[c] = p.u_held[aoff + c];
        }
        if (p.amp && tp.j == (p.imp_node_b ? p.imp_node_b[beam] - p.off : p.imp_slot)) amp = p.amp[beam];
    }

    const T h = T(q.h), hh = T(0.5 * q.h), alpha = T(0.25 * q.h * q.h), alpha2 = T(0.5 * q.h * q.h);
    double tc = p.t0;
    for (int step = 0; step < p.n_steps; ++step) {
        const double tm = __dadd_rn(tc, 0.5 * q.h), t1 = __dadd_rn(tc, q.h);
        const T av = (tm < p.duration) ? amp : T(0);
        T uadd[3], qp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uadd[c] = uh[c] + ((c == p.imp_dof) ? av : T(0));
            qp[c] = q0[c] + hh * v0[c];
            am[c] = (step == 0) ? T(0) : am[c];   // starting iterate: the previous step's a_m (0 at the start of a call)
        }
#pragma unroll 1
        for (int it = 0; it < q.n_iter; ++it) {
            T qm[3], vm[3], an[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { qm[c] = qp[c] + alpha * am[c]; vm[c] = v0[c] + hh * am[c]; }
            stage_accel<T, LV, false, false, true>(p, lds, sc, cf, tp, qm, vm, uadd, an, am, alpha);
#pragma unroll
            for (int c = 0; c < 3; ++c) am[c] = an[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            q0[c] = q0[c] + h * v0[c] + alpha2 * am[c];
            v0[c] = v0[c] + h * am[c];
        }
        tc = t1;
        if (p.rec_out && valid && (step + 1) % p.rec_every == 0) {   // strided recording, as crb_step_rk4_rec
            const size_t k = size_t((step + 1) / p.rec_every - 1);
            if (p.rec_slot == REC_ALL_SLOTS) {
                T* snap = p.rec_out + k * size_t(p.B) * 2 * plane + xoff;
#pragma unroll
                for (int c = 0; c < 3; ++c) { snap[c] = q0[c]; snap[plane + c] = v0[c]; }
                snap[3] = T(0);
                snap[plane + 3] = T(0);
            } else if (tp.j == p.rec_slot) {
                T val = q0[0];
#pragma unroll
                for (int c = 1; c < 6; ++c) val = (c == p.rec_comp) ? (c < 3 ? q0[c] : v0[c - 3]) : val;
                p.rec_out[size_t(beam) * p.rec_n + k] = val;
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p.x[xoff + c] = q0[c];
            p.x[xoff + plane + c] = v0[c];
        }
    }
}

}  // namespace crb
