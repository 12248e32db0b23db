// Step-size control inside the kernel for the two fixed-step schemes the examples' integration calls land on
// (examples/example_utilities.py:153-159: solve_ivp(f, t_span, x0, method="LSODA", t_eval=np.arange(...)) at scipy's
// default rtol 1e-3 / atol 1e-6;  examples/lqr_control.py:117-125: the same call on the closed loop at
// rtol 1e-8 / atol 1e-10): the whole span in ONE launch, every beam with its own step sequence.
//
//   scheme  FB = false: the implicit midpoint rule of crb_stiff.h (order 2), iteration matrix A = M + h^2/4 K0;
//           FB = true:  classical RK4 with the state feedback u = K (r - x) inside every stage (order 4, the fused
//                       small-beam form of crb_generic.h: gain in LDS).
//   control step doubling per PIECE (a t_eval interval, cut at the end of the impulse when that falls inside it -- the
//           host lists the pieces): from the piece's start state m = 2^r steps give the coarse solution, 2m steps the
//           fine one, (fine - coarse) / (2^order - 1) estimates the fine solution's error, measured like scipy measures
//           its own (RMS over the beam's reduced state of err / (atol + rtol max(|fine|, |start|))).  Above 1 -- or not
//           finite: an explicit scheme beyond its stability limit -- the fine solution becomes the coarse one and the
//           piece is repeated with twice the steps; the accepted solution is the fine one; an estimate far below 1
//           halves the rate (steps per second) the next piece starts from.  The rate is carried from piece to piece.
//   tables  the implicit scheme needs A's cyclic-reduction tables for every step size it may take: a piece of length
//           L only ever takes h = L / 2^r, so the host pre-factorises the LADDER r = 0 .. n_rungs-1 for each distinct
//           piece length (at most three: whole intervals and the two parts of the interval the impulse ends in) with
//           crb_assemble_kernel before the launch; the kernel re-reads its rows (L2-resident) when r changes.
// One workgroup per beam (every beam has its own step sequence; the PACK instances below put several short beams into one
// wave, which then share a sequence), one thread per node, state and tables in registers, one wave per SIMD.  fp64.
// LNW < 0: the general RHS (stage_accel: any gravity table, run-time topology).  LNW = 0..2 (implicit scheme): the lean
// iteration of crb_stiff.h with 2^LNW waves per beam -- lane shifts instead of LDS round trips, gravity absent or of the
// plain cantilever's form (GRAV): 2.2 instead of 3.8 us per step for the 10-element example.
#pragma once
#include <hip/hip_runtime.h>

#include "crb_generic.h"
#include "crb_stiff.h"

namespace crb {

struct CtrlPiece {
    double t_a, len;
    int32_t ladder;   // which table ladder matches `len` (implicit scheme)
    int32_t on;       // impulse on (1) / off (0) throughout the piece
    int32_t rec;      // snapshot index written after the piece, or -1
    int32_t interval; // t_eval interval the piece belongs to (index into `used`)
};
constexpr int CTRL_MAX_LADDERS = 3;
constexpr int CTRL_STATS = 4;   // fine steps accepted, doublings, status (0 ok, 2 tolerance not reachable), last rung

template <typename T>
struct CtrlParams {
    const T* a_levels[CTRL_MAX_LADDERS];   // [n_rungs][nd][LV][S][PCR_LEVEL_VALS]
    const T* a_final[CTRL_MAX_LADDERS];    // [n_rungs][nd][S][PCR_FINAL_VALS]
    size_t alv_stride, afin_stride;        // per beam (0: one table set for the ensemble)
    size_t lv_rung, fin_rung;              // per rung
    const CtrlPiece* pieces;
    int n_pieces, n_rungs, n_iter, n_intervals;
    double rtol, atol, rate0;
    int n_state;                // 2 n_free: size of the reference's state vector (the RMS runs over it, or over its position half)
    const int32_t* n_state_b;   // [B] per-beam 2 n_free (mixed ensembles) or nullptr
    int positions_only;
    int32_t* stats;             // [B][CTRL_STATS]
    int32_t* used;              // [B][n_intervals] fine steps accepted per t_eval interval, or nullptr
    T* y_out;                   // [n_rec][B][2][n_node][4] or nullptr
    T* series_out;              // [B][n_intervals] one DOF per t_eval point, or nullptr
    int series_slot, series_comp;   // its slot (node - off) and component (3 plane + dof)
};

template <typename T>
__host__ __device__ constexpr size_t ctrl_lds_bytes(int NT, bool fb, int n, int lean_lognw = -1) {
    return (lean_lognw >= 0 ? implicit_lean_lds_bytes<T>(NT, lean_lognw)
                            : lds_bytes<T>(NT) + (fb ? (size_t(fb_padded(2 * n)) + size_t(fb_padded(2 * n)) * n + FBM_UPAD) * sizeof(T) : 0)) +
           64 * sizeof(double);
}

// The explicit right-hand side a = Minv (u - k(q) + drag + gravity) of ONE beam that lives in one wave, in the lean form (the
// stage of crb_step_lean_kernel for LOGNW = 0 as a function: the left element's force from a lane shift of q, r = p - f_left of
// the right neighbour first and ITS neighbours shifted in, the remaining levels by lean_reduce_tail) -- for the closed-loop
// RK4 of the controlled kernel, whose general right-hand side (stage_accel) costs twice as much at these sizes.
template <typename T, int LV, bool GRAV>
__device__ __forceinline__ void lean_wave_rhs(const ElemCoef<T>& ec, bool corrected, T dragc, T hm_own, T gx, T gy, const SolveCoef<T, LV>& cf,
                                              int t, int lane, int j, int S, bool valid, bool has_right, const T (&sq)[3], const T (&sv)[3],
                                              const T (&uin)[3], T (&a)[3]) {
    T qL[3], fl[3], fr[3], pp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) qL[c] = lane_lower<T, 1>(sq[c], lane);    // (lane 0 reads 0: the clamped / absent root)
    elem_force<T>(ec, qL, sq, corrected, fl, fr);
#pragma unroll
    for (int c = 0; c < 3; ++c) pp[c] = uin[c] - fr[c];
    pp[1] += drag_force<T>(dragc, sv[1]);
    if (GRAV) {
        const T phiR = lane_higher<T, 1>(sq[2], lane);
        T g_own[2];
        gravity_segment<T>(has_right ? T(0.5) * (sq[2] + phiR) : sq[2], gx, gy, hm_own, g_own);
        pp[0] += g_own[0] + lane_lower<T, 1>(g_own[0], lane);             // (segment j-1 is the left lane's own segment)
        pp[1] += g_own[1] + lane_lower<T, 1>(g_own[1], lane);
    }
    T r[3], rlo[3], rhi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        r[c] = pp[c] - lane_higher<T, 1>(fl[c], lane);
        rlo[c] = lane_lower<T, 1>(r[c], lane);
        rhi[c] = lane_higher<T, 1>(r[c], lane);
    }
    pcr_apply_level<T>(cf.lv[0], rlo, rhi, r);
    lean_reduce_tail<T, LV, 0>(cf, nullptr, t, lane, j, S, valid, r, a);
}

// PACK (lean form, one wave): beams of fewer than 33 slots, G = 64 / S of them per wave (lane = g S + j) as in the packed
// fixed-step kernels; the beams of a wave share ONE step sequence -- the worst of them decides -- so an ensemble of thousands
// of short beams fills the chip with a fifth of the waves (control per group of G neighbours instead of per beam).
template <typename T, int LV, bool FB, int LNW = -1, bool GRAV = false, bool PACK = false>
__global__ void __launch_bounds__(LNW >= 0 ? (64 << LNW) : 256, 1)
crb_controlled_kernel(const KParams<T> p, const CtrlParams<T> q) {
    static_assert(sizeof(T) == 8, "the controlled steppers are fp64");
    static_assert(!(FB && LNW > 0) && !(FB && PACK), "the closed loop runs one beam per wave: the lean RHS of one wave, or the general one");
    static_assert(LNW < 0 || LV >= 1, "the lean form needs at least one reduction level");
    static_assert(!PACK || LNW == 0, "packed beams live inside one wave");
    constexpr bool SLIM = LNW >= 0;
    constexpr int LOGNW = SLIM ? LNW : 0, NTL = 64 << LOGNW, NULLT = NTL;
    const int NT = blockDim.x;
    const int pg = PACK ? int(threadIdx.x & 63) / p.S : 0;                 // beam of this lane inside the wave
    const int beam_raw = PACK ? int(blockIdx.x) * p.G + pg : int(blockIdx.x);
    const bool beam_ok = !PACK || (pg < p.G && beam_raw < p.B);
    const int beam = beam_ok ? beam_raw : 0;
    const Lds<T> lds = carve_lds<T>(NT);
    // ---- topology: general (Topo) or lean (slot j of the beam in lane j >> LOGNW of wave j & (NW - 1))
    Topo tp;
    tp.t = threadIdx.x;
    tp.lane = tp.t & 63;
    tp.S = p.S;
    tp.lognw = p.lognw;
    tp.nwm1 = (1 << p.lognw) - 1;
    if (PACK) { tp.j = tp.lane - pg * p.S; tp.base = 0; }
    else if (SLIM) { tp.j = (tp.lane << LOGNW) | (tp.t >> 6); tp.base = 0; }
    else if (p.lognw == 0) { tp.j = tp.t; tp.base = 0; }
    else { tp.j = (tp.lane << p.lognw) + (tp.t >> 6); tp.base = 0; }
    tp.beam = beam;
    tp.valid = tp.j < p.S && beam_ok;
    const int jl = tp.j;                        // (lean form: the slot index also of a thread without a slot)
    if (!tp.valid) { tp.j = 0; tp.S = 1; tp.base = tp.t; tp.nwm1 = 0; }   // padding thread: an isolated dummy node
    const bool valid = tp.valid;
    T* const smem0 = lds.q;
    T* const ldsQ = smem0;                                   // lean: [6][NTL+1]  q_m, a_m
    T* const ldsA = ldsQ + 6 * size_t(NTL + 1);              //       [6][NTL+1]  p0..2, fl0..2
    T* const ldsB = ldsA + 6 * size_t(NTL + 1);              //       [LOGNW-1][3][NTL+1]
    auto thread_of = [](int jj) { return ((jj & ((1 << LOGNW) - 1)) << 6) | (jj >> LOGNW); };
    const bool has_left = valid && jl >= 1, has_right = valid && jl + 1 < p.S;
    const int t_l1 = has_left ? thread_of(jl - 1) : NULLT;
    const int t_r1 = has_right ? thread_of(jl + 1) : NULLT;
    const int t_r2 = (valid && jl + 2 < p.S) ? thread_of(jl + 2) : NULLT;
    if (SLIM && LOGNW > 0) {
        if (tp.t < 12 + 3 * (LOGNW > 1 ? LOGNW - 1 : 0)) ldsQ[size_t(tp.t) * (NTL + 1) + NULLT] = T(0);   // the "no neighbour" entries
        __syncthreads();
    }

    SlotConst<T> sc;
    SolveCoef<T, LV> cf;
    if (valid) {
        sc = p.slot[size_t(beam) * p.slot_stride + tp.j];
    } else {
        sc.elem.kind = KIND_NONE;
#pragma unroll
        for (int k = 0; k < 6; ++k) sc.elem.c[k] = T(0);
        sc.drag = sc.half_mass = T(0);
        sc.mask[0] = sc.mask[1] = sc.mask[2] = T(0);
        sc.grav.phiA = sc.grav.phiB = -1;
#pragma unroll
        for (int c = 0; c < 3; ++c) { sc.grav.segA[c] = sc.grav.segB[c] = -1; sc.grav.comp[c] = 0; }
    }
    // lean form: what the iteration reads of the slot
    const bool corrected = (p.flags & 4u) != 0;
    const T dragc = (p.flags & 1u) ? sc.drag : T(0);
    T lin[5] = {T(0), T(0), T(0), T(0), T(0)};
    T hm_left = T(0), gx = p.gx, gy = p.gy;
    bool shipped_nl = false;
    if (SLIM && valid) {
        elem_linear_coefs<T>(sc.elem.c, sc.elem.kind, lin);
        shipped_nl = sc.elem.kind == KIND_NONLINEAR && !corrected;
        if (GRAV) {
            hm_left = jl >= 1 ? p.slot[size_t(beam) * p.slot_stride + jl - 1].half_mass : T(0);
            if (p.gvec) { gx = p.gvec[2 * size_t(beam)]; gy = p.gvec[2 * size_t(beam) + 1]; }
        }
    }
#pragma unroll
    for (int l = 0; l < LV; ++l)
#pragma unroll
        for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = T(0);
#pragma unroll
    for (int k = 0; k < 5; ++k) cf.fin[k] = T(0);
    if (FB && valid) {   // the mass matrix's tables: one set for every step size
#pragma unroll
        for (int l = 0; l < LV; ++l) {
            const T* src = p.pcr_levels + size_t(beam) * p.lv_stride + (size_t(l) * size_t(p.S) + size_t(tp.j)) * PCR_LEVEL_VALS;
#pragma unroll
            for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = src[k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) cf.fin[k] = p.pcr_final[size_t(beam) * p.fin_stride + size_t(tp.j) * PCR_FINAL_VALS + k];
    }

    const size_t node = size_t(tp.j + p.off);
    const size_t plane = size_t(p.n_node) * 4;
    const size_t xoff = valid ? (size_t(beam) * 2 * plane + node * 4) : 0;
    const size_t aoff = valid ? (size_t(beam) * plane + node * 4) : 0;
    T ys[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, uh[3] = {T(0), T(0), T(0)};
    T amp = T(0);
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ys[c] = p.x[xoff + c] * sc.mask[c];
            ys[3 + c] = p.x[xoff + plane + c] * sc.mask[c];
            if (p.u_held) uh[c] = p.u_held[aoff + c];
        }
        if (p.amp && tp.j == (p.imp_node_b ? p.imp_node_b[beam] - p.off : p.imp_slot)) amp = p.amp[beam];
    }

    // FB: reduced indices and reference of this node, the gain into LDS (crb_beam_kernel's layout with one beam per group)
    const int fb_n = p.n_red, fb_n2 = 2 * p.n_red, fb_n2p = fb_padded(fb_n2);
    T* const fbx = lds.r1 + 3 * NT;            // [2n padded]     r - x of the stage
    T* const fbK = fbx + fb_n2p;               // [2n padded][n]  gain, transposed
    T* const fbu = fbK + size_t(fb_n2p) * fb_n;   // [FBM_UPAD]      K e of the stage (matrix-core form, crb_generic.h)
    const bool fb_mfma = FB && fb_on_matrix_cores(1, fb_n);
    T fb_af[FBM_MT][FBM_KS];
    double* const red = (SLIM && !FB) ? reinterpret_cast<double*>(smem0 + size_t(NTL + 1) * size_t(12 + 3 * (LOGNW > 1 ? LOGNW - 1 : 0)))
                             : reinterpret_cast<double*>(FB ? fbu + FBM_UPAD : fbx);   // [NT / 64]; PACK: [64]
    int ridx[3] = {-1, -1, -1};
    T rq[3] = {T(0), T(0), T(0)}, rv[3] = {T(0), T(0), T(0)};
    if (FB) {
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ridx[c] = p.red_map[3 * node + c];
                if (ridx[c] >= 0 && p.fb_ref) {
                    rq[c] = p.fb_ref[size_t(beam) * fb_n2 + ridx[c]];
                    rv[c] = p.fb_ref[size_t(beam) * fb_n2 + fb_n + ridx[c]];
                }
            }
        }
        for (int idx = tp.t; idx < fb_n * fb_n2; idx += NT) {
            const int i = idx / fb_n2, k = idx - i * fb_n2;
            fbK[size_t(k) * fb_n + i] = p.fb_gain[idx];
        }
        for (int idx = tp.t; idx < (fb_n2p - fb_n2) * fb_n; idx += NT) fbK[size_t(fb_n2) * fb_n + idx] = T(0);
        for (int idx = tp.t; idx < fb_n2p - fb_n2; idx += NT) fbx[fb_n2 + idx] = T(0);
        __syncthreads();
        if (fb_mfma) fbm_gain_fragments<T>(fb_af, fbK, fb_n, fb_n2p, tp.lane);
    }

    constexpr double ORDER_DIV = FB ? 15.0 : 3.0;          // 2^order - 1
    constexpr double SHRINK = (FB ? 0.8 / 16.0 : 0.8 / 4.0) * 0.25;   // half the steps multiply the estimate by 2^order
    const int n_state = q.n_state_b ? q.n_state_b[beam] : q.n_state;
    const double n_norm = double(q.positions_only ? n_state / 2 : n_state);
    double rate = q.rate0;
    int status = 0, doublings = 0, r = 0, in_interval = 0;
    long long total = 0;
    T y[6], yc[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) y[c] = yc[c] = ys[c];

#pragma unroll 1
    for (int pc = 0; pc < q.n_pieces; ++pc) {
        const CtrlPiece P = q.pieces[pc];
        r = 0;
        while (r < q.n_rungs - 2 && double(1 << r) < rate * P.len - 1e-9) ++r;
        const T av = P.on ? amp : T(0);
        T uadd[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) uadd[c] = uh[c] + ((c == p.imp_dof) ? av : T(0));
        int phase = 0;      // 0: the coarse solution, 1: a fine one
        double err = 0.0;
#pragma unroll 1
        while (true) {
            const int rr = r + phase, m = 1 << rr;
            const double hd = P.len / double(m);
            const T h = T(hd), hh = T(0.5 * hd), h6 = T(hd / 6.0), alpha = T(0.25 * hd * hd), alpha2 = T(0.5 * hd * hd);
            if (!FB && valid) {   // A's tables for this rung
                const T* lv = q.a_levels[P.ladder] + size_t(rr) * q.lv_rung + size_t(beam) * q.alv_stride;
                const T* fin = q.a_final[P.ladder] + size_t(rr) * q.fin_rung + size_t(beam) * q.afin_stride;
#pragma unroll
                for (int l = 0; l < LV; ++l) {
                    const T* src = lv + (size_t(l) * size_t(p.S) + size_t(tp.j)) * PCR_LEVEL_VALS;
#pragma unroll
                    for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = src[k];
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) cf.fin[k] = fin[size_t(tp.j) * PCR_FINAL_VALS + k];
            }
            T am[3] = {T(0), T(0), T(0)};   // (the starting iterate of a run: 0, like the start of a crb_step_implicit call)
#pragma unroll
            for (int c = 0; c < 6; ++c) y[c] = ys[c];
#pragma unroll 1
            for (int step = 0; step < m; ++step) {
                if (FB) {
                    T acc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, xs[6];
#pragma unroll
                    for (int c = 0; c < 6; ++c) xs[c] = y[c];
#pragma unroll 1
                    for (int s = 0; s < 4; ++s) {
                        T ua[3] = {uadd[0], uadd[1], uadd[2]}, a[3];
                        if (valid) {
#pragma unroll
                            for (int c = 0; c < 3; ++c)
                                if (ridx[c] >= 0) { fbx[ridx[c]] = rq[c] - xs[c]; fbx[fb_n + ridx[c]] = rv[c] - xs[3 + c]; }
                        }
                        T ufb[3];
                        fb_feedback<T>(fb_mfma, fb_af, fbx, fbK, fbu, 1, 0, fb_n, fb_n2p, tp.lane, valid, ridx, ufb);
#pragma unroll
                        for (int c = 0; c < 3; ++c) ua[c] += ufb[c];
                        __syncthreads();   // (several waves per beam: every wave has read the error vector before the next stage overwrites it)
                        if (SLIM) {
                            const T sq[3] = {xs[0], xs[1], xs[2]}, sv[3] = {xs[3], xs[4], xs[5]};
                            lean_wave_rhs<T, LV, GRAV>(sc.elem, corrected, dragc, sc.half_mass, gx, gy, cf, tp.t, tp.lane, jl, p.S, valid, has_right,
                                                       sq, sv, ua, a);
                        } else {
                            stage_accel<T, LV, false, false>(p, lds, sc, cf, tp, xs, xs + 3, ua, a);
                        }
                        const T w = (s == 0 || s == 3) ? T(1) : T(2);
                        const T cs = (s == 2) ? h : hh;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const T kq = xs[3 + c], kv = a[c];
                            acc[c] += w * kq;
                            acc[3 + c] += w * kv;
                            xs[c] = y[c] + cs * kq;
                            xs[3 + c] = y[3 + c] + cs * kv;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 6; ++c) y[c] += h6 * acc[c];
                } else if (SLIM) {
                    T qp[3], v0[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { qp[c] = y[c] + hh * y[3 + c]; v0[c] = y[3 + c]; }
#pragma unroll 1
                    for (int it = 0; it < q.n_iter; ++it)
                        lean_implicit_iterate<T, LV, LOGNW, GRAV, EM_MIXED, PACK>(sc.elem, lin, shipped_nl, corrected, dragc, sc.half_mass,
                                                                                 hm_left, gx, gy, cf, ldsQ, ldsA, ldsB, tp.t, tp.lane, jl,
                                                                                 p.S, valid, has_left, has_right, t_l1, t_r1, t_r2, qp, v0,
                                                                                 uadd, hh, alpha, am);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        y[c] = y[c] + h * y[3 + c] + alpha2 * am[c];
                        y[3 + c] = y[3 + c] + h * am[c];
                    }
                } else {
                    T qp[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) qp[c] = y[c] + hh * y[3 + c];
#pragma unroll 1
                    for (int it = 0; it < q.n_iter; ++it) {
                        T qm[3], vm[3], an[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) { qm[c] = qp[c] + alpha * am[c]; vm[c] = y[3 + c] + hh * am[c]; }
                        stage_accel<T, LV, false, false, true>(p, lds, sc, cf, tp, qm, vm, uadd, an, am, alpha);
#pragma unroll
                        for (int c = 0; c < 3; ++c) am[c] = an[c];
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        y[c] = y[c] + h * y[3 + c] + alpha2 * am[c];
                        y[3 + c] = y[3 + c] + h * am[c];
                    }
                }
            }
            if (phase == 0) {
#pragma unroll
                for (int c = 0; c < 6; ++c) yc[c] = y[c];
                phase = 1;
                continue;
            }
            // the fine solution's error estimate over this beam
            double e2 = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double scale = q.atol + q.rtol * fmax(fabs(double(y[c])), fabs(double(ys[c])));
                const double e = (double(y[c]) - double(yc[c])) / (ORDER_DIV * scale);
                if (c < 3 || !q.positions_only) e2 += e * e;
            }
            if (!valid) e2 = 0.0;
            if (PACK) {
                // per beam: the sum over its S lanes (through LDS); per wave: the worst beam's estimate, a non-finite one as +inf
                red[tp.lane] = e2;
                __syncthreads();
                double sum = 0.0;
                for (int jj = 0; jj < p.S; ++jj) sum += red[(beam_ok ? pg : 0) * p.S + jj];
                __syncthreads();
                double eb = beam_ok ? sqrt(sum / n_norm) : 0.0;
                eb = (eb == eb) ? eb : __builtin_huge_val();
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) eb = fmax(eb, __shfl_xor(eb, o, 64));
                err = eb;
            } else {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o, 64);
                if (NT > 64) {
                    __syncthreads();
                    if (tp.lane == 0) red[tp.t >> 6] = e2;
                    __syncthreads();
                    e2 = 0.0;
                    for (int w = 0; w < NT / 64; ++w) e2 += red[w];
                }
                err = sqrt(e2 / n_norm);
            }
            if (err <= 1.0) break;
            ++r;
            ++doublings;
#pragma unroll
            for (int c = 0; c < 6; ++c) yc[c] = y[c];
            if (r + 1 >= q.n_rungs) { status = 2; break; }
        }
        if (status != 0) break;
#pragma unroll
        for (int c = 0; c < 6; ++c) ys[c] = y[c];
        total += 2ll << r;
        in_interval += 2 << r;
        rate = double(1 << r) / P.len;
        if (err < SHRINK && r > 0) rate *= 0.5;
        if (q.used && valid && tp.j == 0 && (pc + 1 == q.n_pieces || q.pieces[pc + 1].interval != P.interval)) {
            q.used[size_t(beam) * q.n_intervals + P.interval] = in_interval;
        }
        if (pc + 1 == q.n_pieces || q.pieces[pc + 1].interval != P.interval) in_interval = 0;
        if (q.series_out && valid && jl == q.series_slot && (pc + 1 == q.n_pieces || q.pieces[pc + 1].interval != P.interval)) {
            T val = ys[0];
#pragma unroll
            for (int c = 1; c < 6; ++c) val = (c == q.series_comp) ? ys[c] : val;
            q.series_out[size_t(beam) * q.n_intervals + P.interval] = val;
        }
        if (q.y_out && valid && P.rec >= 0) {
            T* snap = q.y_out + size_t(P.rec) * size_t(p.B) * 2 * plane + xoff;
#pragma unroll
            for (int c = 0; c < 3; ++c) { snap[c] = ys[c]; snap[plane + c] = ys[3 + c]; }
            snap[3] = T(0);
            snap[plane + 3] = T(0);
        }
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p.x[xoff + c] = ys[c];
            p.x[xoff + plane + c] = ys[3 + c];
        }
        p.x[xoff + 3] = T(0);
        p.x[xoff + plane + 3] = T(0);
        mark_nonfinite<T>(p, beam, ys, ys + 3);
    }
    if (valid && tp.j == 0) {
        int32_t* s = q.stats + size_t(beam) * CTRL_STATS;
        s[0] = int32_t(total > 0x7fffffffll ? 0x7fffffffll : total);
        s[1] = doublings;
        s[2] = status;
        s[3] = r;
    }
}

}  // namespace crb
