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
#include "crb_lean.h"

namespace crb {

template <typename T>
struct StiffParams {
    const T* a_levels;        // [levels][S][PCR_LEVEL_VALS] of A (+ beam * alv_stride)
    const T* a_final;         // [S][PCR_FINAL_VALS] of A      (+ beam * afin_stride)
    size_t alv_stride, afin_stride;
    double h;
    int n_iter;
    // DAMPED (generalised-alpha with spectral radius rho at infinite frequency; rho = 1 is the midpoint rule above): the
    // unknown z = (1 - alpha_m) a_{n+1} + alpha_m a_n solves M z = F(q_f, v_f, t_f) with q_f = qp + kappa z, v_f = vp + cv z,
    //     qp = q_n + c_qv v_n + c_qa a_n,   vp = v_n + c_va a_n,   t_f = t_n + tf_frac h,
    // by the same iteration with A = M + kappa K0 (the tables above are built for alpha = kappa); then
    //     a_{n+1} = (z - alpha_m a_n) inv1m,  q_{n+1} = q_n + h v_n + c_q0 a_n + c_q1 a_{n+1},  v_{n+1} = v_n + c_v0 a_n + c_v1 a_{n+1}
    const T* a0;              // [B][2][n_node][4]: the RHS at t0 (its acceleration plane is a_0)
    double kappa, cv, c_qv, c_qa, c_va, tf_frac, alpha_m, inv1m, c_q0, c_q1, c_v0, c_v1;
};

template <typename T, int LV, int MAXT, int MINW, bool DAMPED = false>
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
    T acc_n[3] = {T(0), T(0), T(0)};   // DAMPED: a_n
    if (DAMPED && valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { acc_n[c] = q.a0[xoff + plane + c] * sc.mask[c]; am[c] = acc_n[c]; }
    }

    const T h = T(q.h), hh = DAMPED ? T(q.cv) : T(0.5 * q.h), alpha = DAMPED ? T(q.kappa) : T(0.25 * q.h * q.h), alpha2 = T(0.5 * q.h * q.h);
    double tc = p.t0;
    for (int step = 0; step < p.n_steps; ++step) {
        const double tm = __dadd_rn(tc, (DAMPED ? q.tf_frac : 0.5) * q.h), t1 = __dadd_rn(tc, q.h);
        const T av = (tm < p.duration) ? amp : T(0);
        T uadd[3], qp[3], vp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uadd[c] = uh[c] + ((c == p.imp_dof) ? av : T(0));
            if (DAMPED) {
                qp[c] = q0[c] + T(q.c_qv) * v0[c] + T(q.c_qa) * acc_n[c];
                vp[c] = v0[c] + T(q.c_va) * acc_n[c];
            } else {
                qp[c] = q0[c] + hh * v0[c];
                vp[c] = v0[c];
                am[c] = (step == 0) ? T(0) : am[c];   // starting iterate: the previous step's a_m (0 at the start of a call)
            }
        }
#pragma unroll 1
        for (int it = 0; it < q.n_iter; ++it) {
            T qm[3], vm[3], an[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { qm[c] = qp[c] + alpha * am[c]; vm[c] = vp[c] + hh * am[c]; }
            stage_accel<T, LV, false, false, true>(p, lds, sc, cf, tp, qm, vm, uadd, an, am, alpha);
#pragma unroll
            for (int c = 0; c < 3; ++c) am[c] = an[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (DAMPED) {
                const T a1 = (am[c] - T(q.alpha_m) * acc_n[c]) * T(q.inv1m);
                q0[c] = q0[c] + h * v0[c] + (T(q.c_q0) * acc_n[c] + T(q.c_q1) * a1);
                v0[c] = v0[c] + (T(q.c_v0) * acc_n[c] + T(q.c_v1) * a1);
                acc_n[c] = a1;
            } else {
                q0[c] = q0[c] + h * v0[c] + alpha2 * am[c];
                v0[c] = v0[c] + h * am[c];
            }
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
        mark_nonfinite<T>(p, beam, q0, v0);
    }
}

// ------------------------------------------------------------------ lean form
// crb_implicit_lean_kernel: the same scheme through the lean stepper's exchange structure (crb_lean.h) for the plans it
// covers -- one beam per workgroup of 1, 2 or 4 waves (64 ... 256 slots), gravity absent or of the plain cantilever's
// nearest-neighbour form: compile-time topology, one exchange of {q_m, a_m} for the element force and the K0 a term, the
// merged {p, f_left} + level-0 round, levels inside the wave by DPP / ds_bpermute; the workgroup walks over beams with
// its rows of A's tables in registers (all LV = ceil(log2 S) levels: 85 values at 256 slots, one wave per SIMD).
// waves per SIMD the lean implicit kernels are built for: up to 5 levels the tables are no larger than the explicit
// stepper's (two waves), 6 levels fit as well without the gravity terms (with them: >100 spilled registers), the full
// 7 / 8 levels of long beams take the register file whole
__host__ __device__ constexpr int implicit_lean_minw(int lv, bool grav, int lognw = 1) {
    return ((lv <= 5 || (lv == 6 && !grav)) && lean_minw_f64(lv, lognw, grav) == 2) ? 2 : 1;
}
template <typename T>
__host__ __device__ constexpr size_t implicit_lean_lds_bytes(int NT, int lognw) {
    return sizeof(T) * size_t(NT + 1) * size_t(12 + 3 * (lognw > 1 ? lognw - 1 : 0));
}
// One modified-Newton iteration of the implicit midpoint step in the lean form: a_m <- Ainv (F(q_m, v_m) + alpha K0 a_m) with
// q_m = qp + alpha a_m, v_m = v0 + h/2 a_m (qp = q0 + h/2 v0) -- the iteration of crb_implicit_lean_kernel below as a function, for
// the controlled kernel (crb_ctrl.h).  (The fixed-step kernel keeps its own inlined copy: routing it through this function
// changes its register allocation -- measured +32 spilled VGPRs in the <6, 0, no gravity, mixed elements> instance.)
// ldsQ [6][NT+1], ldsA [6][NT+1], ldsB [LOGNW-1][3][NT+1] (used with several waves per beam only).
template <typename T, int LV, int LOGNW, bool GRAV, int EM, bool PACK>
__device__ __forceinline__ void lean_implicit_iterate(const ElemCoef<T>& ec, const T (&lin)[5], bool shipped_nl, bool corrected, T dragc,
                                                      T hm_own, T hm_left, T gx, T gy, const SolveCoef<T, LV>& cf, T* ldsQ, T* ldsA,
                                                      T* ldsB, int t, int lane, int j, int S, bool valid, bool has_left, bool has_right,
                                                      int t_l1, int t_r1, int t_r2, const T (&qp)[3], const T (&v0)[3],
                                                      const T (&uadd)[3], T hh, T alpha, T (&am)[3]) {
    constexpr int NT = 64 << LOGNW;
    T qm[3], vm[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { qm[c] = qp[c] + alpha * am[c]; vm[c] = v0[c] + hh * am[c]; }
    // -- one exchange of {q_m, a_m}: the left neighbour's values (and the right neighbour's phi for gravity)
    T qL[3], zL[3], phiR = T(0);
    if (LOGNW == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            qL[c] = lane_lower<T, 1>(qm[c], lane); zL[c] = lane_lower<T, 1>(am[c], lane);
            if (PACK) { qL[c] = has_left ? qL[c] : T(0); zL[c] = has_left ? zL[c] : T(0); }
        }
        if (GRAV) phiR = lane_higher<T, 1>(qm[2], lane);   // (only used under has_right)
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) { ldsQ[size_t(c) * (NT + 1) + t] = qm[c]; ldsQ[size_t(3 + c) * (NT + 1) + t] = am[c]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) { qL[c] = ldsQ[size_t(c) * (NT + 1) + t_l1]; zL[c] = ldsQ[size_t(3 + c) * (NT + 1) + t_l1]; }
        if (GRAV) phiR = ldsQ[size_t(2) * (NT + 1) + t_r1];
    }
    T fl[3], fr[3], kl[3], kr[3];
    if (EM == EM_NONLINEAR) elem_force_nonlinear<T>(ec.c, qL, qm, false, fl, fr);
    else if (EM == EM_LINEAR) elem_force_linear<T>(ec.c, qL, qm, fl, fr);
    else elem_force<T>(ec, qL, qm, corrected, fl, fr);
    elem_force_linear<T>(lin, zL, am, kl, kr);                   // K0 a_m (tangent at q = 0) ...
    if (EM != EM_LINEAR && shipped_nl) kl[0] = lin[0] * zL[0];    // ... whose shipped-f1 row has no u2 term
    T pp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { fl[c] -= alpha * kl[c]; fr[c] -= alpha * kr[c]; pp[c] = uadd[c] - fr[c]; }
    pp[1] += drag_force<T>(dragc, vm[1]);
    if (GRAV) {
        T g_own[2], g_left[2];
        gravity_segment<T>(has_right ? T(0.5) * (qm[2] + phiR) : qm[2], gx, gy, hm_own, g_own);
        if (LOGNW == 0) {
            g_left[0] = lane_lower<T, 1>(g_own[0], lane);
            g_left[1] = lane_lower<T, 1>(g_own[1], lane);
            if (PACK) { g_left[0] = has_left ? g_left[0] : T(0); g_left[1] = has_left ? g_left[1] : T(0); }
        } else {
            gravity_segment<T>(T(0.5) * (qL[2] + qm[2]), gx, gy, hm_left, g_left);
        }
        pp[0] += g_own[0] + g_left[0];
        pp[1] += g_own[1] + g_left[1];
    }
    // -- merged round {p, fl} + level 0, then the remaining levels and the final block inverse (of A)
    T r[3], rlo[3], rhi[3], an[3];
    if (LOGNW == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const T fl_r1 = lane_higher<T, 1>(fl[c], lane);
            r[c] = pp[c] - ((!PACK || has_right) ? fl_r1 : T(0));   // (one wave: r first, then ITS neighbours -- see the stepper)
            rlo[c] = lane_lower<T, 1>(r[c], lane);
            rhi[c] = lane_higher<T, 1>(r[c], lane);
            if (PACK) { rlo[c] = has_left ? rlo[c] : T(0); rhi[c] = has_right ? rhi[c] : T(0); }
        }
    } else {
        auto col = [&](int k, int th) -> T& { return ldsA[size_t(k) * (NT + 1) + th]; };
#pragma unroll
        for (int c = 0; c < 3; ++c) { col(c, t) = pp[c]; col(3 + c, t) = fl[c]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rlo[c] = col(c, t_l1) - fl[c];
            r[c] = pp[c] - col(3 + c, t_r1);
            rhi[c] = col(c, t_r1) - col(3 + c, t_r2);
        }
    }
    pcr_apply_level<T>(cf.lv[0], rlo, rhi, r);
    lean_reduce_tail<T, LV, LOGNW, PACK>(cf, ldsB, t, lane, j, S, valid, r, an);
#pragma unroll
    for (int c = 0; c < 3; ++c) am[c] = an[c];
    // (LOGNW == 1: round A's columns are rewritten only after the next iteration's q exchange barrier;
    //  LOGNW >= 2: the level-1 barrier in lean_reduce_tail orders this iteration's reads before the next writes)
}

// PACK (one-wave form only): beams of fewer than 64 slots, G = 64 / S of them per wave (lane = g S + j), as in the packed
// explicit stepper: every exchange stays a lane shift, and what a shift drags across a beam boundary is replaced by 0
// with a select (a diverged wave-mate's Inf / NaN must not reach its neighbours through a 0 * NaN).
template <typename T, int LV, int LOGNW, bool GRAV, int EM, bool PACK = false>
__global__ void __launch_bounds__(64 << LOGNW, implicit_lean_minw(LV, GRAV, LOGNW)) crb_implicit_lean_kernel(const KParams<T> p, const StiffParams<T> q) {
    static_assert(LV >= 1, "needs at least one reduction level");
    static_assert(!PACK || LOGNW == 0, "packed beams live inside one wave");
    constexpr int NW = 1 << LOGNW, NT = 64 << LOGNW, NULLT = NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    T* const ldsQ = reinterpret_cast<T*>(crb_smem);            // [6][NT+1]  q_m, a_m
    T* const ldsA = ldsQ + 6 * size_t(NT + 1);                 // [6][NT+1]  p0..2, fl0..2
    T* const ldsB = ldsA + 6 * size_t(NT + 1);                 // [LOGNW-1][3][NT+1]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int S = p.S;
    const int pg = PACK ? lane / S : 0;                         // beam of this lane inside the wave
    const int j = PACK ? lane - pg * S : ((lane << LOGNW) | wave);
    const bool has_slot = PACK ? (pg < p.G) : (j < S);          // this thread carries a slot (of some beam)
    auto thread_of = [](int jj) { return ((jj & (NW - 1)) << 6) | (jj >> LOGNW); };
    const int t_l1 = (has_slot && j >= 1) ? thread_of(j - 1) : NULLT;
    const int t_r1 = (has_slot && j + 1 < S) ? thread_of(j + 1) : NULLT;
    const int t_r2 = (has_slot && j + 2 < S) ? thread_of(j + 2) : NULLT;
    if (LOGNW > 0) {
        if (t < 12 + 3 * (LOGNW > 1 ? LOGNW - 1 : 0)) ldsQ[size_t(t) * (NT + 1) + NULLT] = T(0);   // the "no neighbour" entries
        __syncthreads();
    }
    const bool shared_tables = p.slot_stride == 0 && q.alv_stride == 0 && q.afin_stride == 0;
    const bool corrected = (p.flags & 4u) != 0;
    ElemCoef<T> ec;
    T dragc = T(0), hm_own = T(0), hm_left = T(0);
    T mask[3] = {T(0), T(0), T(0)};
    T lin[5] = {T(0), T(0), T(0), T(0), T(0)};
    bool shipped_nl = false;   // the element left of this node is a nonlinear one with the shipped f1 (no u2 term in its tangent)
    SolveCoef<T, LV> cf;
    auto load_tables = [&](int beam) {
        if (has_slot) {
            const SlotConst<T>* st = p.slot + size_t(beam) * p.slot_stride;
            const SlotConst<T>& sc = st[j];
            ec = sc.elem;
            dragc = (p.flags & 1u) ? sc.drag : T(0);
#pragma unroll
            for (int c = 0; c < 3; ++c) mask[c] = sc.mask[c];
            if (GRAV) { hm_own = sc.half_mass; hm_left = j >= 1 ? st[j - 1].half_mass : T(0); }
            elem_linear_coefs<T>(ec.c, ec.kind, lin);
            shipped_nl = ec.kind == KIND_NONLINEAR && !corrected;
#pragma unroll
            for (int l = 0; l < LV; ++l) {
                const T* src = q.a_levels + size_t(beam) * q.alv_stride + (size_t(l) * size_t(S) + size_t(j)) * PCR_LEVEL_VALS;
#pragma unroll
                for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = src[k];
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) cf.fin[k] = q.a_final[size_t(beam) * q.afin_stride + size_t(j) * PCR_FINAL_VALS + k];
        } else {
            ec.kind = KIND_NONE;
#pragma unroll
            for (int k = 0; k < 6; ++k) ec.c[k] = T(0);
#pragma unroll
            for (int l = 0; l < LV; ++l)
#pragma unroll
                for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = T(0);
#pragma unroll
            for (int k = 0; k < 5; ++k) cf.fin[k] = T(0);
        }
    };
    if (shared_tables) load_tables(0);
    const size_t plane = size_t(p.n_node) * 4;
    const T h = T(q.h), hh = T(0.5 * q.h), alpha = T(0.25 * q.h * q.h), alpha2 = T(0.5 * q.h * q.h);
    const int n_groups = PACK ? (p.B + p.G - 1) / p.G : p.B;

    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int beam = PACK ? grp * p.G + pg : grp;
        const bool valid = has_slot && (!PACK || beam < p.B);
        const bool has_left = valid && j >= 1, has_right = valid && j + 1 < S;
        if (!shared_tables) load_tables(valid ? beam : 0);
        const size_t node = size_t(valid ? j + p.off : 0);
        const size_t xoff = size_t(valid ? beam : 0) * 2 * plane + node * 4;
        const size_t aoff = size_t(valid ? beam : 0) * plane + node * 4;
        T q0[3] = {T(0), T(0), T(0)}, v0[3] = {T(0), T(0), T(0)}, am[3] = {T(0), T(0), T(0)}, uh[3] = {T(0), T(0), T(0)};
        T amp = T(0), gx = p.gx, gy = p.gy;
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                q0[c] = p.x[xoff + c] * mask[c];
                v0[c] = p.x[xoff + plane + c] * mask[c];
                if (p.u_held) uh[c] = p.u_held[aoff + c];
            }
            if (p.amp && j == (p.imp_node_b ? p.imp_node_b[beam] - p.off : p.imp_slot)) amp = p.amp[beam];
            if (GRAV && p.gvec) { gx = p.gvec[2 * size_t(beam)]; gy = p.gvec[2 * size_t(beam) + 1]; }
        }
        double tc = p.t0;
        for (int step = 0; step < p.n_steps; ++step) {
            const double tm = __dadd_rn(tc, 0.5 * q.h), t1 = __dadd_rn(tc, q.h);
            const T av = (tm < p.duration) ? amp : T(0);
            T uadd[3], qp[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                uadd[c] = uh[c] + ((c == p.imp_dof) ? av : T(0));
                qp[c] = q0[c] + hh * v0[c];
                am[c] = (step == 0) ? T(0) : am[c];
            }
#pragma unroll 1
            for (int it = 0; it < q.n_iter; ++it) {
                T qm[3], vm[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { qm[c] = qp[c] + alpha * am[c]; vm[c] = v0[c] + hh * am[c]; }
                // -- one exchange of {q_m, a_m}: the left neighbour's values (and the right neighbour's phi for gravity)
                T qL[3], zL[3], phiR = T(0);
                if (LOGNW == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        qL[c] = lane_lower<T, 1>(qm[c], lane); zL[c] = lane_lower<T, 1>(am[c], lane);
                        if (PACK) { qL[c] = has_left ? qL[c] : T(0); zL[c] = has_left ? zL[c] : T(0); }
                    }
                    if (GRAV) phiR = lane_higher<T, 1>(qm[2], lane);   // (only used under has_right)
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) { ldsQ[size_t(c) * (NT + 1) + t] = qm[c]; ldsQ[size_t(3 + c) * (NT + 1) + t] = am[c]; }
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < 3; ++c) { qL[c] = ldsQ[size_t(c) * (NT + 1) + t_l1]; zL[c] = ldsQ[size_t(3 + c) * (NT + 1) + t_l1]; }
                    if (GRAV) phiR = ldsQ[size_t(2) * (NT + 1) + t_r1];
                }
                T fl[3], fr[3], kl[3], kr[3];
                if (EM == EM_NONLINEAR) elem_force_nonlinear<T>(ec.c, qL, qm, false, fl, fr);
                else if (EM == EM_LINEAR) elem_force_linear<T>(ec.c, qL, qm, fl, fr);
                else elem_force<T>(ec, qL, qm, corrected, fl, fr);
                elem_force_linear<T>(lin, zL, am, kl, kr);                   // K0 a_m (tangent at q = 0) ...
                if (EM != EM_LINEAR && shipped_nl) kl[0] = lin[0] * zL[0];    // ... whose shipped-f1 row has no u2 term
                T pp[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { fl[c] -= alpha * kl[c]; fr[c] -= alpha * kr[c]; pp[c] = uadd[c] - fr[c]; }
                pp[1] += drag_force<T>(dragc, vm[1]);
                if (GRAV) {
                    T g_own[2], g_left[2];
                    gravity_segment<T>(has_right ? T(0.5) * (qm[2] + phiR) : qm[2], gx, gy, hm_own, g_own);
                    if (LOGNW == 0) {
                        g_left[0] = lane_lower<T, 1>(g_own[0], lane);
                        g_left[1] = lane_lower<T, 1>(g_own[1], lane);
                        if (PACK) { g_left[0] = has_left ? g_left[0] : T(0); g_left[1] = has_left ? g_left[1] : T(0); }
                    } else {
                        gravity_segment<T>(T(0.5) * (qL[2] + qm[2]), gx, gy, hm_left, g_left);
                    }
                    pp[0] += g_own[0] + g_left[0];
                    pp[1] += g_own[1] + g_left[1];
                }
                // -- merged round {p, fl} + level 0, then the remaining levels and the final block inverse (of A)
                T r[3], rlo[3], rhi[3], an[3];
                if (LOGNW == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const T fl_r1 = lane_higher<T, 1>(fl[c], lane);
                        r[c] = pp[c] - ((!PACK || has_right) ? fl_r1 : T(0));   // (one wave: r first, then ITS neighbours -- see the stepper)
                        rlo[c] = lane_lower<T, 1>(r[c], lane);
                        rhi[c] = lane_higher<T, 1>(r[c], lane);
                        if (PACK) { rlo[c] = has_left ? rlo[c] : T(0); rhi[c] = has_right ? rhi[c] : T(0); }
                    }
                } else {
                    auto col = [&](int k, int th) -> T& { return ldsA[size_t(k) * (NT + 1) + th]; };
#pragma unroll
                    for (int c = 0; c < 3; ++c) { col(c, t) = pp[c]; col(3 + c, t) = fl[c]; }
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        rlo[c] = col(c, t_l1) - fl[c];
                        r[c] = pp[c] - col(3 + c, t_r1);
                        rhi[c] = col(c, t_r1) - col(3 + c, t_r2);
                    }
                }
                pcr_apply_level<T>(cf.lv[0], rlo, rhi, r);
                lean_reduce_tail<T, LV, LOGNW, PACK>(cf, ldsB, t, lane, j, S, valid, r, an);
#pragma unroll
                for (int c = 0; c < 3; ++c) am[c] = an[c];
                // (LOGNW == 1: round A's columns are rewritten only after the next iteration's q exchange barrier;
                //  LOGNW >= 2: the level-1 barrier in lean_reduce_tail orders this iteration's reads before the next writes)
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                q0[c] = q0[c] + h * v0[c] + alpha2 * am[c];
                v0[c] = v0[c] + h * am[c];
            }
            tc = t1;
            if (p.rec_out && valid && (step + 1) % p.rec_every == 0) {
                const size_t k = size_t((step + 1) / p.rec_every - 1);
                if (p.rec_slot == REC_ALL_SLOTS) {
                    T* snap = p.rec_out + k * size_t(p.B) * 2 * plane + xoff;
#pragma unroll
                    for (int c = 0; c < 3; ++c) { snap[c] = q0[c]; snap[plane + c] = v0[c]; }
                    snap[3] = T(0);
                    snap[plane + 3] = T(0);
                } else if (j == p.rec_slot) {
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
            mark_nonfinite<T>(p, beam, q0, v0);
        }
        if (LOGNW > 0) __syncthreads();   // the next beam's first q exchange must not overtake this beam's last LDS reads
    }
}

}  // namespace crb
