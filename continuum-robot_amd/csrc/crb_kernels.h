// crb_kernels.h -- gfx950 kernels of the beam stepper (included once by crbeam.hip).
//
// Decomposition: one thread per node ("slot") of a beam.  A thread keeps, in
// registers and for the whole launch: its node's state (3 positions, 3 velocities), the RK4
// accumulators, its element / force coefficients AND its rows of the cyclic-reduction
// multipliers of the mass matrix.  crb_step_rk4 therefore touches HBM once per LAUNCH (state in,
// state out) and nothing else: no coefficient stream, no per-step traffic.
//
// Slot -> thread map.  A beam with S >= 64 slots owns a whole workgroup of NW = NT/64
// wavefronts and slot j lives in wave (j % NW), lane (j / NW).  A cyclic-reduction level of
// stride s = 2^l then needs
//     l >= log2(NW) : the value of lane +/- s/NW of the SAME wave  -> ds_bpermute, no barrier
//     l <  log2(NW) : a value of another wave                      -> LDS + one s_barrier
// so a 256-slot beam (NW = 4) pays barriers only for strides 1 and 2.  LDS is indexed by thread
// id (conflict-free 8-byte accesses); the thread holding slot j is  (j % NW)*64 + j / NW.
// Beams with S < 64 slots pack G = 64/S beams into one wave (thread = g*S + j) and never need
// a barrier for the solve.
//
// Neighbour traffic per RHS evaluation:
//   q of the left node      -> element force of the element left of the node        (stride 1)
//   element force halves    -> nodal internal force; each node sums exactly two     (stride 1)
//   gravity per segment     -> index table (reduced-index quirk, gravity_forces.py:104-146)
//   r at distance 2^l       -> parallel cyclic reduction for Minv (precomputed multipliers)
#pragma once
#include <hip/hip_runtime.h>

#include "crb_math.h"

namespace crb {

// Per-slot constants, loaded once per launch into registers.
struct GravTab {
    int16_t phiA, phiB;   // (slot*4 + dof) of the two rotations averaged by segment <slot>; -1 = absent
    int16_t segA[3];      // per DOF of this slot: segments whose gravity lands on it; -1 = none
    int16_t segB[3];
    int8_t comp[3];       // 0 axial / 1 transverse component of that segment's gravity
    int8_t pad;
};
template <typename T>
struct SlotConst {
    ElemCoef<T> elem;  // element LEFT of this node
    T drag;            // 0.5*rho_f*Cd*A_wet of this node's w DOF (0 when constrained / drag off)
    T half_mass;       // 0.5*rho*A*L of segment <slot> (gravity), 0 when slot >= n_seg
    T mask[3];         // 1 = free DOF, 0 = constrained
    T pad0;
    GravTab grav;
};

template <typename T>
struct KParams {
    const SlotConst<T>* slot;  // [S]
    const T* pcr_levels;       // [levels][S][PCR_LEVEL_VALS]
    const T* pcr_final;        // [S][PCR_FINAL_VALS]
    T* x;                      // [B][2][n_node][4]
    const T* u_held;           // [B][n_node][4] or nullptr
    const T* amp;              // [B] or nullptr
    T* out;                    // rhs / internal force output; MODE_STAGE: next stage state
    const T* xs;               // MODE_STAGE: this stage's state (== x at stage 0)
    T* acc;                    // MODE_STAGE: RK4 accumulator [B][2][n_node][4]
    int stage;                 // MODE_STAGE: 0..3
    const double* t_dev;       // MODE_STAGE: device clock of the step being taken (hipGraph replay: the launch
                               // arguments must not change from step to step); nullptr = t0 holds the stage time
    T* rec_out;                // MODE_STEP: [B][n_rec] strided record of one DOF, or nullptr
    int rec_slot, rec_comp;    // recording thread (slot) and component 0..5 of {q, v}
    int rec_every, rec_n;
    // per-beam coefficient mode (heterogeneous ensembles): table offsets per beam, in elements; 0 = shared
    size_t slot_stride, lv_stride, fin_stride;
    int B, S, G, n_node, off, levels;
    int lognw;                 // log2(wavefronts per beam); 0 when a wave holds whole beams
    uint32_t flags;
    int imp_slot, imp_dof;
    double duration, t0, dt;
    int n_steps;
    T gx, gy;
};

enum : int { MODE_STEP = 0, MODE_RHS = 1, MODE_KQ = 2, MODE_STAGE = 3 };
// stage time of the stage-split stepper: passed by value, or derived from the device clock with the host
// loop's own operations (t, t + dt/2, t + dt/2, t + dt, each a single IEEE addition)
template <typename T>
__device__ __forceinline__ double stage_time(const KParams<T>& p) {
    if (!p.t_dev) return p.t0;
    const double t = *p.t_dev;
    return p.stage == 0 ? t : (p.stage == 3 ? __dadd_rn(t, p.dt) : __dadd_rn(t, 0.5 * p.dt));
}
// (one thread) sets the device clock, or advances it by one step: t <- t + dt
template <int UNUSED>
__global__ void crb_clock_kernel(double* t, double dt, double set_to, int set) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *t = set ? set_to : __dadd_rn(*t, dt);
}
// element kinds over the whole topology: mixed (per-lane branch), all linear, all nonlinear as shipped
// (CRB_CORRECTED_AXIAL plans take the mixed path).  Threads without an element carry zero coefficients,
// for which every formula returns zero forces.
enum : int { EM_MIXED = 0, EM_LINEAR = 1, EM_NONLINEAR = 2 };
constexpr int MAX_LV = 8;

template <typename T>
struct Lds {
    T* q;   // [3][NT]
    T* f;   // [3][NT]
    T* g;   // [2][NT]
    T* r0;  // [3][NT]
    T* r1;  // [3][NT]
    int NT;
};
template <typename T>
__host__ __device__ constexpr size_t lds_bytes(int NT) {
    return size_t(14) * size_t(NT) * sizeof(T);
}

// Where a thread sits: slot j of beam-in-group g, and how to find other slots of its beam.
struct Topo {
    int t, lane, j, S, lognw, nwm1, base;
    bool valid;
    // thread id (LDS index) of slot jj of this thread's beam
    __device__ __forceinline__ int thread_of(int jj) const { return base + ((jj & nwm1) << 6) + (jj >> lognw); }
};

template <typename T>
__device__ __forceinline__ T shfl_from(T v, int src_lane) {
    return __shfl(v, src_lane, 64);
}

// Values of slot j-s ("lo") and j+s ("hi") of a 3-vector r, zero outside the beam.
// CROSS: through LDS (one barrier); else by lane shuffle inside the wave.
template <typename T>
__device__ __forceinline__ void neighbours(const Topo& tp, T* buf, int NT, int s, bool cross, const T r[3], bool want_lo,
                                           bool want_hi, T rlo[3], T rhi[3]) {
    const bool lo_ok = tp.j - s >= 0, hi_ok = tp.j + s < tp.S;
    if (cross) {
        buf[tp.t] = r[0];
        buf[NT + tp.t] = r[1];
        buf[2 * NT + tp.t] = r[2];
        __syncthreads();
        const int tl = lo_ok ? tp.thread_of(tp.j - s) : tp.t, th = hi_ok ? tp.thread_of(tp.j + s) : tp.t;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (want_lo) rlo[c] = lo_ok ? buf[c * NT + tl] : T(0);
            if (want_hi) rhi[c] = hi_ok ? buf[c * NT + th] : T(0);
        }
    } else {
        const int d = s >> tp.lognw;  // lane distance (lognw == 0 when several beams share the wave)
        const int ll = lo_ok ? tp.lane - d : tp.lane, lh = hi_ok ? tp.lane + d : tp.lane;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (want_lo) { const T v = shfl_from<T>(r[c], ll); rlo[c] = lo_ok ? v : T(0); }
            if (want_hi) { const T v = shfl_from<T>(r[c], lh); rhi[c] = hi_ok ? v : T(0); }
        }
    }
}

template <typename T, int LV>
struct SolveCoef {
    T lv[LV > 0 ? LV : 1][PCR_LEVEL_VALS];
    T fin[5];
};

// One evaluation of a = Minv(-k(q) + f_drag + f_grav + u) for this thread's node.
// Returns k(q) in `a` (no solve) when KQ_ONLY.
template <typename T, int LV, bool KQ_ONLY, bool LEAN>
__device__ __forceinline__ void stage_accel(const KParams<T>& p, const Lds<T>& lds, const SlotConst<T>& sc,
                                            const SolveCoef<T, LV>& cf, const Topo& tp, const T q[3], const T v[3],
                                            const T uadd[3], T a[3]) {
    const int NT = lds.NT;
    // LEAN kernels are only launched for plans without gravity (and calls without a held input)
    const bool drag_on = (p.flags & 1u) != 0, grav_on = !LEAN && (p.flags & 2u) != 0, corrected = (p.flags & 4u) != 0;
    const bool cross1 = tp.lognw > 0;  // stride-1 neighbours live in another wave
    const bool q_in_lds = cross1 || (grav_on && !KQ_ONLY);

    // -- 1. the left node's q
    T ql[3], dummy[3];
    if (q_in_lds) {
        neighbours<T>(tp, lds.q, NT, 1, true, q, true, false, ql, dummy);
    } else {
        neighbours<T>(tp, lds.q, NT, 1, false, q, true, false, ql, dummy);
    }
    T fl[3], fr[3];
    elem_force<T>(sc.elem, ql, q, corrected, fl, fr);

    T gseg[2] = {T(0), T(0)};
    if (!KQ_ONLY && grav_on && sc.half_mass != T(0)) {
        const int ia = sc.grav.phiA, ib = sc.grav.phiB;
        T phi = T(0);
        if (ia >= 0) phi = lds.q[(ia & 3) * NT + tp.thread_of(ia >> 2)];
        if (ib >= 0) phi = T(0.5) * (phi + lds.q[(ib & 3) * NT + tp.thread_of(ib >> 2)]);
        gravity_segment<T>(phi, p.gx, p.gy, sc.half_mass, gseg);
    }

    // -- 2. the right neighbour's left-node half of its element force (+ segment gravity via LDS)
    T fnext[3];
    if (!KQ_ONLY && grav_on) {
        lds.g[tp.t] = gseg[0];
        lds.g[NT + tp.t] = gseg[1];
    }
    neighbours<T>(tp, lds.f, NT, 1, cross1 || (grav_on && !KQ_ONLY), fl, false, true, dummy, fnext);
    T r[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = fr[c] + fnext[c];
    if (KQ_ONLY) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = r[c] * sc.mask[c];
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = uadd[c] - r[c];
    if (drag_on) r[1] += drag_force<T>(sc.drag, v[1]);
    if (grav_on) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int sa = sc.grav.segA[c], sb = sc.grav.segB[c];
            const int go = sc.grav.comp[c] * NT;
            if (sa >= 0) r[c] += lds.g[go + tp.thread_of(sa)];
            if (sb >= 0) r[c] += lds.g[go + tp.thread_of(sb)];
        }
    }
    // (no mask multiply: rows/columns of constrained DOFs are zero in every multiplier and in the
    //  final inverse, so whatever sits in r at a constrained DOF never propagates and a = 0 there)

    // -- 3. Minv by parallel cyclic reduction, multipliers resident in registers
#pragma unroll
    for (int lvl = 0; lvl < LV; ++lvl) {
        T rlo[3], rhi[3];
        neighbours<T>(tp, (lvl & 1) ? lds.r1 : lds.r0, NT, 1 << lvl, lvl < tp.lognw, r, true, true, rlo, rhi);
        pcr_apply_level<T>(cf.lv[lvl], rlo, rhi, r);
    }
    pcr_apply_final<T>(cf.fin, r, a);
}

template <typename T>
__device__ __forceinline__ Lds<T> carve_lds(int NT) {
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    T* p = reinterpret_cast<T*>(crb_smem);
    Lds<T> l;
    l.NT = NT;
    l.q = p;
    l.f = p + 3 * NT;
    l.g = p + 6 * NT;
    l.r0 = p + 8 * NT;
    l.r1 = p + 11 * NT;
    return l;
}

// MODE_STEP: n_steps RK4 steps in place.  MODE_RHS: out = [v ; a].  MODE_KQ: out = k(q).
// MODE_STAGE: ONE RK4 stage of the stage-split stepper (the input force changes per stage, e.g. state
// feedback u = K(r - x), lqr_control.py:95-111): k = f(t0, xs, u_held + impulse);
// acc = (stage ? acc : 0) + w k;  stage < 3: out = x + c k;  stage 3: x += dt/6 acc.
template <typename T, int MODE, int LV, int MAXT, int MINW, bool LEAN>
__global__ void __launch_bounds__(MAXT, MINW) crb_beam_kernel(const KParams<T> p) {
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
    if (!tp.valid) {
        // padding thread: an isolated dummy node (no neighbour at any stride, all coefficients 0).
        // lognw stays the launch value: the barrier / shuffle choice must be workgroup-uniform.
        tp.j = 0;
        tp.S = 1;
        tp.base = tp.t;
        tp.nwm1 = 0;
    }
    const bool valid = tp.valid;

    SlotConst<T> sc;
    SolveCoef<T, LV> cf;
    if (valid) {
        sc = p.slot[size_t(beam) * p.slot_stride + tp.j];
#pragma unroll
        for (int l = 0; l < LV; ++l) {
            const T* src = p.pcr_levels + size_t(beam) * p.lv_stride + (size_t(l) * size_t(p.S) + size_t(tp.j)) * PCR_LEVEL_VALS;
#pragma unroll
            for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = src[k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) cf.fin[k] = p.pcr_final[size_t(beam) * p.fin_stride + size_t(tp.j) * PCR_FINAL_VALS + k];
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

    // this thread's node record
    const size_t node = size_t(tp.j + p.off);
    const size_t plane = size_t(p.n_node) * 4;
    const size_t xoff = valid ? (size_t(beam) * 2 * plane + node * 4) : 0;
    T x[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    T uh[3] = {T(0), T(0), T(0)};
    T amp = T(0);
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = p.x[xoff + c] * sc.mask[c];
            x[3 + c] = p.x[xoff + plane + c] * sc.mask[c];
        }
        if (!LEAN && p.u_held) {
            const size_t uoff = size_t(beam) * plane + node * 4;
#pragma unroll
            for (int c = 0; c < 3; ++c) uh[c] = p.u_held[uoff + c];
        }
        if (p.amp && tp.j == p.imp_slot) amp = p.amp[beam];
    }

    if (MODE == MODE_STAGE) {
        T xs[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, acc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                xs[c] = p.xs[xoff + c] * sc.mask[c];
                xs[3 + c] = p.xs[xoff + plane + c] * sc.mask[c];
                if (p.stage > 0) { acc[c] = p.acc[xoff + c]; acc[3 + c] = p.acc[xoff + plane + c]; }
            }
        }
        const T av = (stage_time(p) < p.duration) ? amp : T(0);
        T uadd[3], a[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) uadd[c] = uh[c] + ((c == p.imp_dof) ? av : T(0));
        stage_accel<T, LV, false, LEAN>(p, lds, sc, cf, tp, xs, xs + 3, uadd, a);
        const T w = (p.stage == 0 || p.stage == 3) ? T(1) : T(2);
        const T cs = (p.stage == 2) ? T(p.dt) : T(0.5 * p.dt);
        const T dt6 = T(p.dt / 6.0);
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const T kq = xs[3 + c], kv = a[c];
                acc[c] += w * kq;
                acc[3 + c] += w * kv;
                if (p.stage < 3) {
                    p.out[xoff + c] = x[c] + cs * kq;
                    p.out[xoff + plane + c] = x[3 + c] + cs * kv;
                    p.acc[xoff + c] = acc[c];
                    p.acc[xoff + plane + c] = acc[3 + c];
                } else {
                    p.x[xoff + c] = x[c] + dt6 * acc[c];
                    p.x[xoff + plane + c] = x[3 + c] + dt6 * acc[3 + c];
                }
            }
        }
        return;
    }
    if (MODE != MODE_STEP) {
        T a[3];
        stage_accel<T, LV, MODE == MODE_KQ, LEAN>(p, lds, sc, cf, tp, x, x + 3, uh, a);
        if (valid) {
            if (MODE == MODE_KQ) {
                const size_t ooff = size_t(beam) * plane + node * 4;
#pragma unroll
                for (int c = 0; c < 3; ++c) p.out[ooff + c] = a[c];
                p.out[ooff + 3] = T(0);
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    p.out[xoff + c] = x[3 + c];
                    p.out[xoff + plane + c] = a[c];
                }
                p.out[xoff + 3] = T(0);
                p.out[xoff + plane + 3] = T(0);
            }
        }
        return;
    }

    // ---- classical RK4, state resident in registers across all steps
    const T dt = T(p.dt), hdt = T(0.5 * p.dt), dt6 = T(p.dt / 6.0);
    double tc = p.t0;
    for (int step = 0; step < p.n_steps; ++step) {
        const double t_half = __dadd_rn(tc, 0.5 * p.dt), t_full = __dadd_rn(tc, p.dt);
        T acc[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        T xs[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) xs[c] = x[c];
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            const double ts = (s == 0) ? tc : ((s == 3) ? t_full : t_half);
            const T av = (ts < p.duration) ? amp : T(0);
            T uadd[3], a[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) uadd[c] = (LEAN ? T(0) : uh[c]) + ((c == p.imp_dof) ? av : T(0));
            stage_accel<T, LV, false, LEAN>(p, lds, sc, cf, tp, xs, xs + 3, uadd, a);
            const T w = (s == 0 || s == 3) ? T(1) : T(2);
            const T cs = (s == 2) ? dt : hdt;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const T kq = xs[3 + c], kv = a[c];
                acc[c] += w * kq;
                acc[3 + c] += w * kv;
                xs[c] = x[c] + cs * kq;
                xs[3 + c] = x[3 + c] + cs * kv;
            }
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) x[c] += dt6 * acc[c];
        tc = t_full;
        if (p.rec_out && valid && tp.j == p.rec_slot && (step + 1) % p.rec_every == 0) {
            T val = x[0];
#pragma unroll
            for (int c = 1; c < 6; ++c) val = (c == p.rec_comp) ? x[c] : val;
            p.rec_out[size_t(beam) * p.rec_n + (step + 1) / p.rec_every - 1] = val;
        }
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p.x[xoff + c] = x[c];
            p.x[xoff + plane + c] = x[3 + c];
        }
    }
}

// ------------------------------------------------------------------ adaptive RK45 (f-2)
// crb_rk45_kernel: embedded Dormand-Prince 5(4) with per-beam step-size control, the whole integration
// t0 -> t_end in one launch.  It restates scipy.integrate.solve_ivp(method="RK45") -- the integrator the
// reference's tests hand its RHS to (tests/test_dynamic_beam.py:218-220, test_functional_composition.py:539-546;
// scipy/integrate/_ivp/rk.py, scipy 1.15: rk_step, RungeKutta._step_impl, select_initial_step) -- so a beam
// takes the step sequence scipy would take on the same RHS: same tableau, RMS error norm over the
// REDUCED state with scale = atol + max(|y|,|y_new|) rtol, SAFETY 0.9, factors in [0.2, 10], no growth
// right after a rejection, FSAL, the same initial-step heuristic and end-point clipping.
// One workgroup per beam (or several small beams per wave): every beam has its own clock and step.
// The seven stage derivatives live in LDS ([7][6][NT], thread-private columns: no barrier).
struct Rk45Params {
    double t0, t_end, rtol, atol;
    double* h_io;      // [B] in: first step (<= 0: choose like scipy), out: next step suggestion
    int32_t* stats;    // [B][4] accepted, rejected, nfev, status (0 ok, 1 step too small)
    int n_state;       // 2 * n_free: size of the reference's state vector (the error norm's N)
    int max_steps;     // safety bound on attempted steps
    // dense output of ONE DOF on the uniform grid t_eval[k] = eval_t0 + k*eval_dt, k < n_eval (solve_ivp's
    // t_eval): after every accepted step the grid points in (t_old, t_new] -- plus t_eval[0] == t0 -- are
    // evaluated with scipy's 4th-order interpolant (RkDenseOutput, RK45.P) and stored at eval_out[b][k]
    void* eval_out;    // [B][n_eval] plan dtype, or nullptr
    double eval_t0, eval_dt;
    int n_eval, eval_slot, eval_comp;
};

template <typename T>
__device__ __forceinline__ double block_sum(double v, double* red, int NT, int t, int base, int nthr_beam, bool per_wave_beams) {
    // sum over the threads of ONE beam: whole workgroup (one beam per group) or a segment of the wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (NT > 64) {
        __syncthreads();
        if ((t & 63) == 0) red[t >> 6] = v;
        __syncthreads();
        double s = 0.0;
        for (int w = 0; w < NT / 64; ++w) s += red[w];
        return s;
    }
    return v;
}

// (defined with the lean kernels below) one RHS through the lean machinery: q exchange, element force, merged
// exchange round + level 0, in-wave levels -- for plans without gravity
template <typename T, int LV, int LOGNW, int EM>
__device__ __forceinline__ void lean_rhs(const ElemCoef<T>& ec, T dragc, bool corrected, const SolveCoef<T, LV>& cf, T* lds3, int t,
                                         int lane, int j, int S, bool valid, const T sq[3], const T sv[3], const T uadd[3], T a[3]);

// LNW < 0: the general RHS (stage_accel: any gravity table, any waves-per-beam count, run-time topology).
// LNW = 0..2: the lean RHS with 2^LNW waves per beam and element mode EM (plans without gravity).
template <typename T, int LV, int MAXT, int MINW, int LNW = -1, int EM = 0>
__global__ void __launch_bounds__(MAXT, MINW) crb_rk45_kernel(const KParams<T> p, const Rk45Params q) {
    const int NT = blockDim.x;
    const Lds<T> lds = carve_lds<T>(NT);
    T* const Ks = lds.r1 + 3 * NT;                         // [7][6][NT]
    double* const red = reinterpret_cast<double*>(Ks + 42 * NT);  // [NT/64]
    Topo tp;
    tp.t = threadIdx.x;
    tp.lane = tp.t & 63;
    tp.S = p.S;
    tp.lognw = p.lognw;
    tp.nwm1 = (1 << p.lognw) - 1;
    // one beam per workgroup (G == 1 is enforced by the host for this kernel)
    if (p.lognw == 0) { tp.j = tp.t; tp.base = 0; }
    else { tp.j = (tp.lane << p.lognw) + (tp.t >> 6); tp.base = 0; }
    const int beam = blockIdx.x;
    tp.valid = tp.j < p.S;
    if (!tp.valid) { tp.j = 0; tp.S = 1; tp.base = tp.t; tp.nwm1 = 0; }
    const bool valid = tp.valid;
    const int t = tp.t;

    SlotConst<T> sc;
    SolveCoef<T, LV> cf;
    if (valid) {
        sc = p.slot[size_t(beam) * p.slot_stride + tp.j];
#pragma unroll
        for (int l = 0; l < LV; ++l) {
            const T* src = p.pcr_levels + size_t(beam) * p.lv_stride + (size_t(l) * size_t(p.S) + size_t(tp.j)) * PCR_LEVEL_VALS;
#pragma unroll
            for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = src[k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) cf.fin[k] = p.pcr_final[size_t(beam) * p.fin_stride + size_t(tp.j) * PCR_FINAL_VALS + k];
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
    T y[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, uh[3] = {T(0), T(0), T(0)};
    T amp = T(0);
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            y[c] = p.x[xoff + c] * sc.mask[c];
            y[3 + c] = p.x[xoff + plane + c] * sc.mask[c];
        }
        if (p.u_held) {
            const size_t uoff = size_t(beam) * plane + node * 4;
#pragma unroll
            for (int c = 0; c < 3; ++c) uh[c] = p.u_held[uoff + c];
        }
        if (p.amp && tp.j == p.imp_slot) amp = p.amp[beam];
    }
    // lean RHS: the 14*NT values in front of Ks hold its exchange columns ([3 + 6 + 3][NT + 1] incl. the zero
    // "no neighbour" entries)
    const int lj = (tp.lane << (LNW > 0 ? LNW : 0)) | (LNW > 0 ? (t >> 6) : 0);
    const bool corrected = (p.flags & 4u) != 0;
    const T dragc = (p.flags & 1u) ? sc.drag : T(0);
    if (LNW >= 0) {
        if (t < 12) lds.q[size_t(t) * (NT + 1) + NT] = T(0);
        __syncthreads();
    }
    // f(ts, state) -> derivative d[6] = [v ; a]
    auto deriv = [&](double ts, const T st[6], T d[6]) {
        const T av = (ts < p.duration) ? amp : T(0);
        T uadd[3], a[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) uadd[c] = uh[c] + ((c == p.imp_dof) ? av : T(0));
        if (LNW >= 0) lean_rhs<T, LV, (LNW >= 0 ? LNW : 0), EM>(sc.elem, dragc, corrected, cf, lds.q, t, tp.lane, lj, p.S, valid, st, st + 3, uadd, a);
        else stage_accel<T, LV, false, false>(p, lds, sc, cf, tp, st, st + 3, uadd, a);
#pragma unroll
        for (int c = 0; c < 3; ++c) { d[c] = st[3 + c]; d[3 + c] = a[c]; }
    };
    auto putK = [&](int k, const T d[6]) {
#pragma unroll
        for (int c = 0; c < 6; ++c) Ks[(k * 6 + c) * NT + t] = d[c];
    };
    auto rms = [&](const double v[6]) {  // scipy: norm(x) / sqrt(x.size) over the reduced state
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) s += v[c] * v[c];
        return sqrt(block_sum<T>(s, red, NT, t, 0, 0, false) / double(q.n_state));
    };

    // Dormand-Prince tableau (scipy RK45.A / .B / .C / .E)
    const double C5[6] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0};
    const double A5[6][5] = {{0, 0, 0, 0, 0},
                             {1.0 / 5, 0, 0, 0, 0},
                             {3.0 / 40, 9.0 / 40, 0, 0, 0},
                             {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
                             {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
                             {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
    const double B5[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
    const double E5[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};

    // scipy RK45.P (dense output): y(t_old + x h) = y_old + h * sum_m x^(m+1) * sum_j K_j P[j][m]
    const double P5[7][4] = {{1.0, -2.8535800653862835, 3.0717434641059005, -1.1270175653862835},
                             {0.0, 0.0, 0.0, 0.0},
                             {0.0, 4.023133379230305, -6.249321565289, 2.675424484351598},
                             {0.0, -3.7324019615885042, 10.068970589843675, -5.685526961588504},
                             {0.0, 2.5548038301849423, -6.399112377351017, 3.5219323679207912},
                             {0.0, -1.3744241142186024, 3.272657752246729, -1.7672812570757455},
                             {0.0, 1.3824689317781436, -3.764937863556287, 2.382468931778144}};
    int ie = 0;  // next t_eval index (uniform over the workgroup)
    const bool recorder = q.eval_out && valid && tp.j == q.eval_slot;

    double tc = q.t0;
    int accepted = 0, rejected = 0, nfev = 0, status = 0;
    T k0[6];
    deriv(tc, y, k0);
    putK(0, k0);
    ++nfev;
    double h_abs = q.h_io ? q.h_io[beam] : 0.0;
    if (!(h_abs > 0.0)) {  // scipy select_initial_step (order = 4)
        double a0[6], a1[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double sc0 = q.atol + fabs(double(y[c])) * q.rtol;
            a0[c] = double(y[c]) / sc0;
            a1[c] = double(k0[c]) / sc0;
        }
        const double d0 = rms(a0), d1 = rms(a1);
        const double h0 = fmin((d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1, fabs(q.t_end - q.t0));
        T y1[6], f1[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) y1[c] = y[c] + T(h0) * k0[c];
        deriv(tc + h0, y1, f1);
        ++nfev;
        double a2[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) a2[c] = (double(f1[c]) - double(k0[c])) / (q.atol + fabs(double(y[c])) * q.rtol);
        const double d2 = rms(a2) / h0;
        const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
        h_abs = fmin(fmin(100.0 * h0, h1), fabs(q.t_end - q.t0));
    }

    int attempts = 0;
    while (tc < q.t_end && status == 0) {
        const double min_step = 10.0 * (nextafter(tc, INFINITY) - tc);
        if (h_abs < min_step) h_abs = min_step;
        bool step_rejected = false;
        for (;;) {
            if (h_abs < min_step || ++attempts > q.max_steps) { status = 1; break; }
            double t_new = tc + h_abs;
            if (t_new - q.t_end > 0.0) t_new = q.t_end;
            const double h = t_new - tc;
            h_abs = fabs(h);
            // stages 1..5
#pragma unroll 1
            for (int s = 1; s < 6; ++s) {
                T ys[6], ks[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    double dy = 0.0;
                    for (int jj = 0; jj < s; ++jj) dy += double(Ks[(jj * 6 + c) * NT + t]) * A5[s][jj];
                    ys[c] = T(double(y[c]) + dy * h);
                }
                deriv(tc + C5[s] * h, ys, ks);
                putK(s, ks);
            }
            T yn[6], fn[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double acc = 0.0;
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) acc += double(Ks[(jj * 6 + c) * NT + t]) * B5[jj];
                yn[c] = T(double(y[c]) + h * acc);
            }
            deriv(tc + h, yn, fn);
            putK(6, fn);
            nfev += 6;
            double en[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double e = 0.0;
#pragma unroll
                for (int jj = 0; jj < 7; ++jj) e += double(Ks[(jj * 6 + c) * NT + t]) * E5[jj];
                const double scl = q.atol + fmax(fabs(double(y[c])), fabs(double(yn[c]))) * q.rtol;
                en[c] = e * h / scl;
            }
            const double error_norm = rms(en);
            if (error_norm < 1.0) {
                double factor = (error_norm == 0.0) ? 10.0 : fmin(10.0, 0.9 * pow(error_norm, -0.2));
                if (step_rejected) factor = fmin(1.0, factor);
                h_abs *= factor;
                if (q.eval_out) {  // dense output on the t_eval grid points this step covers
                    while (ie < q.n_eval) {
                        const double te = q.eval_t0 + double(ie) * q.eval_dt;
                        if (te > t_new) break;
                        if (recorder) {
                            const double x = (te - tc) / h;
                            double xp = x, acc = 0.0;
                            const int c = q.eval_comp;
                            for (int m = 0; m < 4; ++m) {
                                double qm = 0.0;
                                for (int jj = 0; jj < 7; ++jj) qm += double(Ks[(jj * 6 + c) * NT + t]) * P5[jj][m];
                                acc += qm * xp;
                                xp *= x;
                            }
                            static_cast<T*>(q.eval_out)[size_t(beam) * q.n_eval + ie] = T(h * acc + double(y[c]));
                        }
                        ++ie;
                    }
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) y[c] = yn[c];
                putK(0, fn);  // FSAL
                tc = t_new;
                ++accepted;
                break;
            }
            h_abs *= fmax(0.2, 0.9 * pow(error_norm, -0.2));
            step_rejected = true;
            ++rejected;
        }
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p.x[xoff + c] = y[c];
            p.x[xoff + plane + c] = y[3 + c];
        }
    }
    if (t == 0) {
        if (q.h_io) q.h_io[beam] = h_abs;
        if (q.stats) {
            q.stats[beam * 4 + 0] = accepted;
            q.stats[beam * 4 + 1] = rejected;
            q.stats[beam * 4 + 2] = nfev;
            q.stats[beam * 4 + 3] = status;
        }
    }
}
template <typename T>
__host__ __device__ constexpr size_t rk45_lds_bytes(int NT) {
    return lds_bytes<T>(NT) + size_t(42) * NT * sizeof(T) + size_t(NT / 64 + 1) * sizeof(double);
}

// ------------------------------------------------------------------ lean fused stepper
// crb_step_lean_kernel: the stepper for plans without gravity and calls without a held input
// (BASELINE configs 3/4), one beam per workgroup of NW = 2^LOGNW waves, everything compile-time:
//   * exchange rounds merged.  Positions of the NEXT stage are known when a stage starts
//     (q_next = x_q + c*v_stage), so they ride on this stage's force exchange instead of costing a
//     round of their own; the force exchange itself is merged with cyclic-reduction level 0: a
//     thread publishes {q_next, p = u - f_right + drag, f_left} once and rebuilds r of both
//     neighbours from what it reads (r_{i-1} = p_{i-1} - f_left_i, r_{i+1} = p_{i+1} - f_left_{i+2}).
//     Rounds per RHS: 1 + (LV-1) instead of 2 + LV; barriers: max(LOGNW,1) [0 for one wave].
//   * round A moves 16-byte LDS words (record = 10 fp64 / 12 fp32 values per thread, padded so
//     that ds_read/write_b128 are bank-conflict free); lane +-1 shifts use DPP wave_shr/wave_shl
//     (no LDS round trip), larger lane shifts ds_bpermute.
template <typename T>
struct LeanRec {                    // [qn0 qn1 qn2 - | p0 p1 p2 fl0 | fl1 fl2 (- -)]
    static constexpr int N = sizeof(T) == 8 ? 10 : 12;   // 80 B / 48 B: conflict-free 16-byte accesses
    static constexpr int V = 16 / sizeof(T);              // values per 16-byte LDS word
};
template <typename T>
__host__ __device__ constexpr size_t lean_lds_bytes(int NT, int lognw) {
    // round A records (+1 all-zero "no neighbour" record; double-buffered when round A is the only
    // barrier round) + SoA buffers (+1 zero column) of the cross-wave levels 1..lognw-1
    return sizeof(T) * (size_t(NT + 1) * LeanRec<T>::N * (lognw == 1 ? 2 : 1) +
                        3 * size_t(NT + 1) * size_t(lognw > 1 ? lognw - 1 : 0));
}

// wave_shr:1 / wave_shl:1 with bound_ctrl: a lane without a source lane reads 0 and no "old" value has
// to be materialised first (update_dpp(0, ..) costs one extra v_mov per DPP).
__device__ __forceinline__ double dpp_from_lower(double x) {  // value held by lane-1 (0 into lane 0)
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_higher(double x) {  // value held by lane+1 (0 into lane 63)
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float dpp_from_lower(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_from_higher(float x) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x130, 0xf, 0xf, true));
}
// Value of lane-D / lane+D of the same wave.  D <= DPP_MAX: chained DPP wave shifts (VALU, no
// LDS round trip; lanes shifted in from outside the wave read 0).  Larger D: ds_bpermute; a lane
// index outside the wave wraps to some lane of the SAME beam -- callers only ever multiply such
// a value by a multiplier that is exactly 0 (no neighbour at that stride).
#ifndef CRB_DPP_MAX
#define CRB_DPP_MAX 4
#endif
// wave priorities per phase of a stage (s_setprio; -1 = leave unchanged)
#ifndef CRB_P_FORCE
#define CRB_P_FORCE 0
#endif
#ifndef CRB_P_XCHG
#define CRB_P_XCHG 2
#endif
#ifndef CRB_P_L1
#define CRB_P_L1 -1
#endif
#ifndef CRB_P_TAIL
#define CRB_P_TAIL 1
#endif
#ifndef CRB_P_FIN
#define CRB_P_FIN -1
#endif
#ifndef CRB_SOA
#define CRB_SOA 1
#endif
#define CRB_SETPRIO(v) do { if ((v) >= 0) __builtin_amdgcn_s_setprio((v) < 0 ? 0 : (v)); } while (0)
template <typename T, int D>
__device__ __forceinline__ T lane_lower(T x, int lane) {
    if (D <= CRB_DPP_MAX) {
#pragma unroll
        for (int i = 0; i < D; ++i) x = dpp_from_lower(x);
        return x;
    }
    return __shfl(x, lane - D, 64);
}
template <typename T, int D>
__device__ __forceinline__ T lane_higher(T x, int lane) {
    if (D <= CRB_DPP_MAX) {
#pragma unroll
        for (int i = 0; i < D; ++i) x = dpp_from_higher(x);
        return x;
    }
    return __shfl(x, lane + D, 64);
}

// 16-byte LDS access of the V values starting at element index I (I % V == 0) of a record
template <typename T>
struct Vec16 {
    typedef T type __attribute__((ext_vector_type(16 / sizeof(T))));
};
template <typename T, int FIRST, int LAST>
__device__ __forceinline__ void rec_load(const T* rec, T* out /*[N]*/) {
    constexpr int V = LeanRec<T>::V;
    typedef typename Vec16<T>::type vec;
#pragma unroll
    for (int w = FIRST / V; w <= LAST / V; ++w) {
        const vec v = *reinterpret_cast<const vec*>(rec + w * V);
#pragma unroll
        for (int k = 0; k < V; ++k) out[w * V + k] = v[k];
    }
}

// Reduction levels 1..LV-1 and the final block inverse of the lean kernels, given r after level 0.
// Levels whose stride stays inside the workgroup's waves-per-beam interleave (l < LOGNW) go through LDS
// columns + a barrier, the others are in-wave lane shifts.  A missing neighbour contributes through a
// multiplier that is exactly 0, so whatever finite value the shift returns there is harmless.
template <typename T, int LV, int LOGNW>
__device__ __forceinline__ void lean_reduce_tail(const SolveCoef<T, LV>& cf, T* ldsB, int t, int lane, int j, int S,
                                                 bool valid, T r[3], T a[3]) {
    constexpr int NW = 1 << LOGNW, NT = 64 << LOGNW, NULLT = NT;
    auto thread_of = [](int jj) { return ((jj & (NW - 1)) << 6) | (jj >> LOGNW); };
    T rlo[3], rhi[3];
#pragma unroll
    for (int l = 1; l < LV; ++l) {
        if (l < LOGNW) {  // another wave holds the neighbour: LDS + barrier
            if (l == 1) CRB_SETPRIO(CRB_P_L1);
            const int st = 1 << l;
            T* buf = ldsB + size_t(l - 1) * 3 * (NT + 1);
            buf[t] = r[0]; buf[(NT + 1) + t] = r[1]; buf[2 * (NT + 1) + t] = r[2];
            __syncthreads();
            const int tl = (valid && j - st >= 0) ? thread_of(j - st) : NULLT;
            const int th = (valid && j + st < S) ? thread_of(j + st) : NULLT;
#pragma unroll
            for (int c = 0; c < 3; ++c) { rlo[c] = buf[c * (NT + 1) + tl]; rhi[c] = buf[c * (NT + 1) + th]; }
        } else {
            if (l == LOGNW) CRB_SETPRIO(CRB_P_TAIL);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                switch (l - LOGNW) {
                    case 0: rlo[c] = lane_lower<T, 1>(r[c], lane); rhi[c] = lane_higher<T, 1>(r[c], lane); break;
                    case 1: rlo[c] = lane_lower<T, 2>(r[c], lane); rhi[c] = lane_higher<T, 2>(r[c], lane); break;
                    case 2: rlo[c] = lane_lower<T, 4>(r[c], lane); rhi[c] = lane_higher<T, 4>(r[c], lane); break;
                    case 3: rlo[c] = lane_lower<T, 8>(r[c], lane); rhi[c] = lane_higher<T, 8>(r[c], lane); break;
                    case 4: rlo[c] = lane_lower<T, 16>(r[c], lane); rhi[c] = lane_higher<T, 16>(r[c], lane); break;
                    default: rlo[c] = lane_lower<T, 32>(r[c], lane); rhi[c] = lane_higher<T, 32>(r[c], lane); break;
                }
            }
        }
        pcr_apply_level<T>(cf.lv[l], rlo, rhi, r);
    }
    CRB_SETPRIO(CRB_P_FIN);
    pcr_apply_final<T>(cf.fin, r, a);
}

template <typename T, int LV, int LOGNW, int EM>
__device__ __forceinline__ void lean_rhs(const ElemCoef<T>& ec, T dragc, bool corrected, const SolveCoef<T, LV>& cf, T* lds3, int t,
                                         int lane, int j, int S, bool valid, const T sq[3], const T sv[3], const T uadd[3], T a[3]) {
    constexpr int NW = 1 << LOGNW, NT = 64 << LOGNW, NULLT = NT;
    T* const ldsQ = lds3;                           // [3][NT+1]  stage positions
    T* const ldsA = lds3 + 3 * size_t(NT + 1);      // [6][NT+1]  p0..2, fl0..2
    T* const ldsB = lds3 + 9 * size_t(NT + 1);      // [LOGNW-1][3][NT+1]
    auto thread_of = [](int jj) { return ((jj & (NW - 1)) << 6) | (jj >> LOGNW); };
    const int t_l1 = (valid && j >= 1) ? thread_of(j - 1) : NULLT;
    const int t_r1 = (valid && j + 1 < S) ? thread_of(j + 1) : NULLT;
    const int t_r2 = (valid && j + 2 < S) ? thread_of(j + 2) : NULLT;
    // -- the left neighbour's position (its own exchange: stage states are arbitrary combinations here)
    T qL[3];
    if (LOGNW == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) qL[c] = lane_lower<T, 1>(sq[c], lane);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) ldsQ[size_t(c) * (NT + 1) + t] = sq[c];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) qL[c] = ldsQ[size_t(c) * (NT + 1) + t_l1];
    }
    T fl[3], fr[3];
    if (EM == EM_NONLINEAR) elem_force_nonlinear<T>(ec.c, qL, sq, false, fl, fr);
    else if (EM == EM_LINEAR) elem_force_linear<T>(ec.c, qL, sq, fl, fr);
    else elem_force<T>(ec, qL, sq, corrected, fl, fr);
    T pp[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) pp[c] = uadd[c] - fr[c];
    pp[1] += drag_force<T>(dragc, sv[1]);
    // -- merged exchange round {p, fl} + level 0.  The barrier of the q exchange above orders the previous
    //    call's reads of these columns before this call's writes (and vice versa for the q columns).
    T r[3], rlo[3], rhi[3];
    if (LOGNW == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            rlo[c] = lane_lower<T, 1>(pp[c], lane) - fl[c];
            r[c] = pp[c] - lane_higher<T, 1>(fl[c], lane);
            rhi[c] = lane_higher<T, 1>(pp[c], lane) - lane_higher<T, 2>(fl[c], lane);
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
    lean_reduce_tail<T, LV, LOGNW>(cf, ldsB, t, lane, j, S, valid, r, a);
}

// EM (EM_*): the element kind when the whole topology has one; the force evaluation is then straight-line
// code that the scheduler interleaves with the tail of the previous stage's reduction (measured +8 %
// over the per-lane branch of EM_MIXED; a wave-uniform run-time branch does not get it).
template <typename T, int LV, int LOGNW, bool GRAV, int EM>
// fp64: 2 waves/SIMD, 256 VGPRs hold the multipliers.  fp32: 3 waves/SIMD (168 VGPRs; 4 waves/SIMD spills)
__global__ void __launch_bounds__(64 << LOGNW, (sizeof(T) == 4 && LOGNW <= 2) ? 3 : 2) crb_step_lean_kernel(const KParams<T> p) {
    static_assert(LV >= 1, "lean stepper needs at least one reduction level");
    constexpr int NW = 1 << LOGNW, NT = 64 << LOGNW, RN = LeanRec<T>::N, RV = LeanRec<T>::V;
    constexpr int NULLT = NT;  // index of the all-zero record / column: "no neighbour"
    // fp64 round A is laid out as 9 columns [qn0..2 p0..2 fl0..2][NT+1] moved by 8-byte accesses (a 16-byte LDS
    // store costs 13 cycles of the store path against 2 x 6 for two 8-byte ones); fp32 keeps 16-byte records
    constexpr bool SOA = CRB_SOA && sizeof(T) == 8;
    auto recA = [](T* base, int th, int k) -> T& { return SOA ? base[size_t(k) * (NT + 1) + th] : base[size_t(th) * RN + k]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    T* const ldsA = reinterpret_cast<T*>(crb_smem);
    T* const ldsB = ldsA + size_t(NT + 1) * RN * (LOGNW == 1 ? 2 : 1);  // [level-1][3][NT+1]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int S = p.S;
    const int j = (lane << LOGNW) | wave;
    const int beam = blockIdx.x;
    const bool valid = j < S;
    auto thread_of = [](int jj) { return ((jj & (NW - 1)) << 6) | (jj >> LOGNW); };
    // LDS positions of the stride-1 / stride-2 neighbours (NULLT outside the beam)
    const int t_l1 = (valid && j >= 1) ? thread_of(j - 1) : NULLT;
    const int t_r1 = (valid && j + 1 < S) ? thread_of(j + 1) : NULLT;
    const int t_r2 = (valid && j + 2 < S) ? thread_of(j + 2) : NULLT;

    if (LOGNW > 0 && t == 0) {  // zero the "no neighbour" slots once
#pragma unroll
        for (int k = 0; k < RN; ++k) {
            recA(ldsA, NULLT, k) = T(0);
            if (LOGNW == 1) recA(ldsA + size_t(NT + 1) * RN, NULLT, k) = T(0);
        }
#pragma unroll
        for (int l = 1; l < LOGNW; ++l)
#pragma unroll
            for (int c = 0; c < 3; ++c) ldsB[(size_t(l - 1) * 3 + c) * (NT + 1) + NULLT] = T(0);
    }

    // ---- per-thread constants
    ElemCoef<T> ec;
    T dragc = T(0);
    SolveCoef<T, LV> cf;
    if (valid) {
        const SlotConst<T>& sc = p.slot[size_t(beam) * p.slot_stride + j];
        ec = sc.elem;
        dragc = (p.flags & 1u) ? sc.drag : T(0);
#pragma unroll
        for (int l = 0; l < LV; ++l) {
            const T* src = p.pcr_levels + size_t(beam) * p.lv_stride + (size_t(l) * size_t(S) + size_t(j)) * PCR_LEVEL_VALS;
#pragma unroll
            for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = src[k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) cf.fin[k] = p.pcr_final[size_t(beam) * p.fin_stride + size_t(j) * PCR_FINAL_VALS + k];
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
    const bool corrected = (p.flags & 4u) != 0;
    // GRAV: gravity on the canonical cantilever (only node 0 constrained), where the reference's reduced-index
    // addressing (gravity_forces.py:104-146) is nearest-neighbour: segment j averages the rotations of slots j
    // and j+1 (slot j alone at the tip) and loads slots j and j+1.  A thread evaluates segment j AND segment
    // j-1 itself (it knows phi of slots j-1, j, j+1), so gravity needs no exchange of its own.
    T hm_own = T(0), hm_left = T(0), phiR = T(0);
    const bool has_right = valid && j + 1 < S;
    if (GRAV && valid) {
        hm_own = p.slot[size_t(beam) * p.slot_stride + j].half_mass;
        if (j >= 1) hm_left = p.slot[size_t(beam) * p.slot_stride + j - 1].half_mass;
    }

    // ---- state
    const size_t node = size_t(valid ? j + p.off : 0);
    const size_t plane = size_t(p.n_node) * 4;
    const size_t xoff = size_t(beam) * 2 * plane + node * 4;
    T xq[3] = {T(0), T(0), T(0)}, xv[3] = {T(0), T(0), T(0)};
    T amp = T(0);
    if (valid) {
        const SlotConst<T>& sc = p.slot[size_t(beam) * p.slot_stride + j];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xq[c] = p.x[xoff + c] * sc.mask[c];
            xv[c] = p.x[xoff + plane + c] * sc.mask[c];
        }
        if (p.amp && j == p.imp_slot) amp = p.amp[beam];
    }

    // ---- left neighbour's q for the very first stage
    T qL[3];
    if (LOGNW == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) qL[c] = lane_lower<T, 1>(xq[c], lane);  // lane 0 reads 0 = clamped / absent root
        if (GRAV) phiR = lane_higher<T, 1>(xq[2], lane);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) recA(ldsA, t, c) = xq[c];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) qL[c] = recA(ldsA, t_l1, c);
        if (GRAV) phiR = recA(ldsA, t_r1, 2);
        __syncthreads();
    }

    const T dt = T(p.dt), hdt = T(0.5 * p.dt), dt6 = T(p.dt / 6.0);
    double tc = p.t0;
    T accq[3], accv[3], sq[3], sv[3];  // RK4 accumulators and stage state
    for (int step = 0; step < p.n_steps; ++step) {
        const double t_half = __dadd_rn(tc, 0.5 * p.dt), t_full = __dadd_rn(tc, p.dt);
#pragma unroll
        for (int c = 0; c < 3; ++c) { accq[c] = T(0); accv[c] = T(0); sq[c] = xq[c]; sv[c] = xv[c]; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double ts = (s == 0) ? tc : ((s == 3) ? t_full : t_half);
            const bool imp_on = ts < p.duration;  // wave-uniform: the impulse selector stays on the scalar unit
            const T w = (s == 0 || s == 3) ? T(1) : T(2);
            const T cs = (s == 2) ? dt : hdt;

            // -- positions of the next stage (or of the next step after stage 3)
            T qn[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                accq[c] += w * sv[c];
                qn[c] = (s == 3) ? (xq[c] + dt6 * accq[c]) : (xq[c] + cs * sv[c]);
            }
            // -- element force of the element left of this node
            T fl[3], fr[3];
            CRB_SETPRIO(CRB_P_FORCE);
            if (EM == EM_NONLINEAR) elem_force_nonlinear<T>(ec.c, qL, sq, false, fl, fr);
            else if (EM == EM_LINEAR) elem_force_linear<T>(ec.c, qL, sq, fl, fr);
            else elem_force<T>(ec, qL, sq, corrected, fl, fr);
            CRB_SETPRIO(CRB_P_XCHG);
            T pp[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) pp[c] = ((imp_on && c == p.imp_dof) ? T(1) : T(0)) * amp - fr[c];
            pp[1] += drag_force<T>(dragc, sv[1]);
            if (GRAV) {
                T g_own[2], g_left[2];
                gravity_segment<T>(has_right ? T(0.5) * (sq[2] + phiR) : sq[2], p.gx, p.gy, hm_own, g_own);
                if (LOGNW == 0) {  // segment j-1 IS the left lane's own segment: take its result (bit-identical), one sincos less
                    g_left[0] = lane_lower<T, 1>(g_own[0], lane);
                    g_left[1] = lane_lower<T, 1>(g_own[1], lane);
                } else {
                    gravity_segment<T>(T(0.5) * (qL[2] + sq[2]), p.gx, p.gy, hm_left, g_left);
                }
                pp[0] += g_own[0] + g_left[0];
                pp[1] += g_own[1] + g_left[1];
            }

            // -- round A: publish {qn, p, fl}; rebuild r of this node and of both stride-1 neighbours.
            // Outside the beam a neighbour reads as zeros (DPP edge / all-zero LDS record).
            T r[3], rlo[3], rhi[3];
            if (LOGNW == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // (shuffles stay outside any condition: every lane must take part.  Lanes past
                    //  the last slot are padding threads whose p and fl are 0, wave edges shift in 0.)
                    if (GRAV && c == 2) phiR = lane_higher<T, 1>(qn[2], lane);
                    qL[c] = lane_lower<T, 1>(qn[c], lane);
                    rlo[c] = lane_lower<T, 1>(pp[c], lane) - fl[c];
                    r[c] = pp[c] - lane_higher<T, 1>(fl[c], lane);
                    rhi[c] = lane_higher<T, 1>(pp[c], lane) - lane_higher<T, 2>(fl[c], lane);
                }
            } else {
                T* bufA = ldsA + ((LOGNW == 1 && (s & 1)) ? size_t(NT + 1) * RN : 0);
                if (SOA) {
                    recA(bufA, t, 0) = qn[0]; recA(bufA, t, 1) = qn[1]; recA(bufA, t, 2) = qn[2];
                    recA(bufA, t, 3) = pp[0]; recA(bufA, t, 4) = pp[1]; recA(bufA, t, 5) = pp[2];
                    recA(bufA, t, 6) = fl[0]; recA(bufA, t, 7) = fl[1]; recA(bufA, t, 8) = fl[2];
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < 3; ++c) {   // what level 0 needs first, the next stage's qL last
                        rlo[c] = recA(bufA, t_l1, 3 + c) - fl[c];
                        r[c] = pp[c] - recA(bufA, t_r1, 6 + c);
                        rhi[c] = recA(bufA, t_r1, 3 + c) - recA(bufA, t_r2, 6 + c);
                    }
                    if (GRAV) phiR = recA(bufA, t_r1, 2);
#pragma unroll
                    for (int c = 0; c < 3; ++c) qL[c] = recA(bufA, t_l1, c);
                } else {
                typedef typename Vec16<T>::type vec;
                T out[RN];
                out[0] = qn[0]; out[1] = qn[1]; out[2] = qn[2]; out[3] = T(0);
                out[4] = pp[0]; out[5] = pp[1]; out[6] = pp[2];
                out[7] = fl[0]; out[8] = fl[1]; out[9] = fl[2];
#pragma unroll
                for (int k = 10; k < RN; ++k) out[k] = T(0);
                vec* rec = reinterpret_cast<vec*>(bufA + size_t(t) * RN);
#pragma unroll
                for (int wv = 0; wv < RN / RV; ++wv) {
                    vec v;
#pragma unroll
                    for (int k = 0; k < RV; ++k) v[k] = out[wv * RV + k];
                    rec[wv] = v;
                }
                __syncthreads();
                T L[RN], R1[RN], R2[RN];
                rec_load<T, 0, 6>(bufA + size_t(t_l1) * RN, L);
                rec_load<T, GRAV ? 2 : 4, 9>(bufA + size_t(t_r1) * RN, R1);
                rec_load<T, 7, 9>(bufA + size_t(t_r2) * RN, R2);
                if (GRAV) phiR = R1[2];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    qL[c] = L[c];
                    rlo[c] = L[4 + c] - fl[c];
                    r[c] = pp[c] - R1[7 + c];
                    rhi[c] = R1[4 + c] - R2[7 + c];
                }
                }
            }
            pcr_apply_level<T>(cf.lv[0], rlo, rhi, r);

            // -- remaining reduction levels and the final block inverse
            T a[3];
            lean_reduce_tail<T, LV, LOGNW>(cf, ldsB, t, lane, j, S, valid, r, a);

            // -- RK4 bookkeeping
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                accv[c] += w * a[c];
                sq[c] = qn[c];
                sv[c] = (s == 3) ? (xv[c] + dt6 * accv[c]) : (xv[c] + cs * a[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) { xq[c] = sq[c]; xv[c] = sv[c]; }
        tc = t_full;
        if (p.rec_out && valid && j == p.rec_slot && (step + 1) % p.rec_every == 0) {
            T val = xq[0];
#pragma unroll
            for (int c = 1; c < 6; ++c) val = (c == p.rec_comp) ? (c < 3 ? xq[c] : xv[c - 3]) : val;
            p.rec_out[size_t(beam) * p.rec_n + (step + 1) / p.rec_every - 1] = val;
        }
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p.x[xoff + c] = xq[c];
            p.x[xoff + plane + c] = xv[c];
        }
    }
}

// ------------------------------------------------------------------ lean stage kernel
// crb_stage_lean_kernel: ONE RK4 stage of the stage-split stepper (crb_rk4_stage: the input force changes per
// stage, e.g. LQR feedback u = K(r - x) evaluated by crb_feedback_force) with the lean stepper's machinery:
// register-resident multipliers, one merged exchange round {p, f_left} + level 0, in-wave levels by DPP.
// A launch is one RHS per beam, so what the generic stage kernel pays most for is re-reading the solve
// tables (440 B per node) for every beam: here a workgroup keeps them in registers and walks over several
// beams (shared-table plans; per-beam tables reload).  The left neighbour's q (and the right neighbour's
// rotation for gravity) are plain global loads of the stage state: no exchange round for them.
//   k = f(t_stage, xs, u_stage + impulse);  acc = (stage ? acc : 0) + w k;
//   stage < 3: out = x + c k;   stage 3: x += dt/6 acc           (same contract as MODE_STAGE)
template <typename T>
__host__ __device__ constexpr size_t stage_lean_lds_bytes(int NT, int lognw) {
    return sizeof(T) * (size_t(NT + 1) * 6 * (lognw == 1 ? 2 : 1) + 3 * size_t(NT + 1) * size_t(lognw > 1 ? lognw - 1 : 0));
}
template <typename T, int LV, int LOGNW, bool GRAV, int EM>
__global__ void __launch_bounds__(64 << LOGNW, (sizeof(T) == 4 && LOGNW <= 2) ? 3 : 2) crb_stage_lean_kernel(const KParams<T> p) {
    static_assert(LV >= 1, "lean stage kernel needs at least one reduction level");
    constexpr int NW = 1 << LOGNW, NT = 64 << LOGNW, NULLT = NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    T* const ldsA = reinterpret_cast<T*>(crb_smem);                         // [1 or 2][6][NT+1]: p0..2, fl0..2
    T* const ldsB = ldsA + size_t(NT + 1) * 6 * (LOGNW == 1 ? 2 : 1);       // [level-1][3][NT+1]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int S = p.S;
    const int j = (lane << LOGNW) | wave;
    const bool valid = j < S;
    auto thread_of = [](int jj) { return ((jj & (NW - 1)) << 6) | (jj >> LOGNW); };
    const int t_l1 = (valid && j >= 1) ? thread_of(j - 1) : NULLT;
    const int t_r1 = (valid && j + 1 < S) ? thread_of(j + 1) : NULLT;
    const int t_r2 = (valid && j + 2 < S) ? thread_of(j + 2) : NULLT;
    if (LOGNW > 0 && t == 0) {
#pragma unroll
        for (int k = 0; k < 6 * (LOGNW == 1 ? 2 : 1); ++k) ldsA[size_t(k) * (NT + 1) + NULLT] = T(0);
#pragma unroll
        for (int l = 1; l < LOGNW; ++l)
#pragma unroll
            for (int c = 0; c < 3; ++c) ldsB[(size_t(l - 1) * 3 + c) * (NT + 1) + NULLT] = T(0);
    }
    const bool shared_tables = p.slot_stride == 0 && p.lv_stride == 0 && p.fin_stride == 0;
    const bool corrected = (p.flags & 4u) != 0;
    const bool has_right = valid && j + 1 < S, has_left = valid && j >= 1;
    const size_t node = size_t(valid ? j + p.off : 0);
    const size_t plane = size_t(p.n_node) * 4;
    const T w = (p.stage == 0 || p.stage == 3) ? T(1) : T(2);
    const T cs = (p.stage == 2) ? T(p.dt) : T(0.5 * p.dt);
    const T dt6 = T(p.dt / 6.0);
    const bool imp_on = stage_time(p) < p.duration;

    ElemCoef<T> ec;
    T dragc = T(0), hm_own = T(0), hm_left = T(0);
    T mask[3] = {T(0), T(0), T(0)}, maskL[3] = {T(0), T(0), T(0)};
    SolveCoef<T, LV> cf;
    auto load_tables = [&](int beam) {
        if (valid) {
            const SlotConst<T>* st = p.slot + size_t(beam) * p.slot_stride;
            const SlotConst<T>& sc = st[j];
            ec = sc.elem;
            dragc = (p.flags & 1u) ? sc.drag : T(0);
#pragma unroll
            for (int c = 0; c < 3; ++c) { mask[c] = sc.mask[c]; maskL[c] = has_left ? st[j - 1].mask[c] : T(0); }
            if (GRAV) { hm_own = sc.half_mass; hm_left = has_left ? st[j - 1].half_mass : T(0); }
#pragma unroll
            for (int l = 0; l < LV; ++l) {
                const T* src = p.pcr_levels + size_t(beam) * p.lv_stride + (size_t(l) * size_t(S) + size_t(j)) * PCR_LEVEL_VALS;
#pragma unroll
                for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf.lv[l][k] = src[k];
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) cf.fin[k] = p.pcr_final[size_t(beam) * p.fin_stride + size_t(j) * PCR_FINAL_VALS + k];
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

    int it = 0;
    for (int beam = blockIdx.x; beam < p.B; beam += gridDim.x, ++it) {
        if (!shared_tables) load_tables(beam);
        // ---- this stage's state, the neighbours' pieces of it, the input force
        const size_t xoff = size_t(beam) * 2 * plane + node * 4;
        T sq[3] = {T(0), T(0), T(0)}, sv[3] = {T(0), T(0), T(0)}, qL[3] = {T(0), T(0), T(0)}, uin[3] = {T(0), T(0), T(0)};
        T x0q[3] = {T(0), T(0), T(0)}, x0v[3] = {T(0), T(0), T(0)}, aq[3] = {T(0), T(0), T(0)}, av[3] = {T(0), T(0), T(0)};
        T phiR = T(0), amp = T(0);
        typedef T rec4 __attribute__((ext_vector_type(4)));
        if (valid) {
            // a node record is 4 values = one aligned 32-byte (fp64) / 16-byte (fp32) vector: whole-record
            // loads instead of three scalar ones (the slots of a wave are every NW-th node, so scalar loads
            // would pull each 128-byte line through L1 once per component: measured 31.7 -> 24 us)
            auto ldrec = [](const T* ptr) { return *reinterpret_cast<const rec4*>(ptr); };
            const rec4 rq = ldrec(p.xs + xoff), rv = ldrec(p.xs + xoff + plane);
            const rec4 bq = ldrec(p.x + xoff), bv = ldrec(p.x + xoff + plane);
            rec4 rl = rec4{T(0), T(0), T(0), T(0)}, cq = rl, cv = rl, ru = rl;
            if (has_left) rl = ldrec(p.xs + xoff - 4);
            if (p.stage > 0) { cq = ldrec(p.acc + xoff); cv = ldrec(p.acc + xoff + plane); }
            if (p.u_held) ru = ldrec(p.u_held + size_t(beam) * plane + node * 4);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sq[c] = rq[c] * mask[c];
                sv[c] = rv[c] * mask[c];
                qL[c] = rl[c] * maskL[c];
                x0q[c] = bq[c] * mask[c];
                x0v[c] = bv[c] * mask[c];
                aq[c] = cq[c];
                av[c] = cv[c];
                uin[c] = ru[c];
            }
            if (GRAV && has_right) phiR = p.xs[xoff + 4 + 2];
            if (p.amp && j == p.imp_slot) amp = p.amp[beam];
        }
        // ---- forces on this node from its own element, drag, gravity, inputs
        T fl[3], fr[3];
        CRB_SETPRIO(CRB_P_FORCE);
        if (EM == EM_NONLINEAR) elem_force_nonlinear<T>(ec.c, qL, sq, false, fl, fr);
        else if (EM == EM_LINEAR) elem_force_linear<T>(ec.c, qL, sq, fl, fr);
        else elem_force<T>(ec, qL, sq, corrected, fl, fr);
        CRB_SETPRIO(CRB_P_XCHG);
        T pp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) pp[c] = uin[c] + ((imp_on && c == p.imp_dof) ? T(1) : T(0)) * amp - fr[c];
        pp[1] += drag_force<T>(dragc, sv[1]);
        if (GRAV) {
            T g_own[2], g_left[2];
            gravity_segment<T>(has_right ? T(0.5) * (sq[2] + phiR) : sq[2], p.gx, p.gy, hm_own, g_own);
            if (LOGNW == 0) {  // (as in the stepper: the left lane's own segment)
                g_left[0] = lane_lower<T, 1>(g_own[0], lane);
                g_left[1] = lane_lower<T, 1>(g_own[1], lane);
            } else {
                gravity_segment<T>(T(0.5) * (qL[2] + sq[2]), p.gx, p.gy, hm_left, g_left);
            }
            pp[0] += g_own[0] + g_left[0];
            pp[1] += g_own[1] + g_left[1];
        }
        // ---- round A: publish {p, fl}, rebuild r of this node and of both stride-1 neighbours, level 0
        T r[3], rlo[3], rhi[3];
        if (LOGNW == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                rlo[c] = lane_lower<T, 1>(pp[c], lane) - fl[c];
                r[c] = pp[c] - lane_higher<T, 1>(fl[c], lane);
                rhi[c] = lane_higher<T, 1>(pp[c], lane) - lane_higher<T, 2>(fl[c], lane);
            }
        } else {
            T* bufA = ldsA + ((LOGNW == 1 && (it & 1)) ? size_t(NT + 1) * 6 : 0);
            auto col = [&](int k, int th) -> T& { return bufA[size_t(k) * (NT + 1) + th]; };
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
        T a[3];
        lean_reduce_tail<T, LV, LOGNW>(cf, ldsB, t, lane, j, S, valid, r, a);
        // ---- RK4 bookkeeping of this stage
        if (valid) {
            auto strec = [](T* ptr, const T v[3]) { *reinterpret_cast<rec4*>(ptr) = rec4{v[0], v[1], v[2], T(0)}; };
            T nq[3], nv[3], oq[3], ov[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                nq[c] = (p.stage ? aq[c] : T(0)) + w * sv[c];
                nv[c] = (p.stage ? av[c] : T(0)) + w * a[c];
                oq[c] = (p.stage < 3) ? x0q[c] + cs * sv[c] : x0q[c] + dt6 * nq[c];
                ov[c] = (p.stage < 3) ? x0v[c] + cs * a[c] : x0v[c] + dt6 * nv[c];
            }
            if (p.stage < 3) {   // (whole records: the pad value is written as 0)
                strec(p.out + xoff, oq); strec(p.out + xoff + plane, ov);
                strec(p.acc + xoff, nq); strec(p.acc + xoff + plane, nv);
            } else {
                strec(p.x + xoff, oq); strec(p.x + xoff + plane, ov);
            }
        }
        // LOGNW >= 2: the level-1 barrier above orders this beam's round-A reads before the next beam's
        // round-A writes; LOGNW == 1 alternates two round-A buffers; LOGNW == 0 uses no LDS
    }
}

// ------------------------------------------------------------------ assembly / factorisation
// crb_assemble_kernel: everything DynamicEulerBernoulliBeam.__init__ computes in floating point
// (dynamic_beam_model.py:25-74), one thread per slot, one workgroup per beam topology:
//   per-element coefficient packs (segments.py:32-62, 128-130), consistent mass in node-block form
//   (segments.py:64-78 assembled as euler_bernoulli_beam.py:139-161), boundary-condition masks
//   (:240-265), drag factors (fluid_forces.py:87-90), segment masses (gravity_forces.py:59-63), and the
//   cyclic-reduction factorisation of M that replaces inv(M) (dynamic_beam_model.py:60).
// Integer topology (free-DOF masks, gravity index table) comes from the host; all arithmetic is fp64,
// tables are written both in fp64 (inspection) and in the plan dtype T (what the steppers load).
struct AsmParams {
    const double* L; const double* E; const double* I; const double* rho; const double* A;  // [n_elem]
    const uint8_t* nonlinear;   // [n_elem]
    const uint8_t* free_dof;    // [3*n_node]
    const double* wet; const double* cd;  // [n_elem] or null
    const GravTab* grav;        // [S] index table (host-built topology)
    double fluid_density;
    uint32_t flags;
    int n_elem, n_node, off, S, levels_full;
    void* slot_out;             // SlotConst<T>[nb][S]
    double* lv64; void* lvT;    // [nb][levels_full][S][10]   (lv64 may be null)
    double* fin64_all;          // [levels_full+1][S][6] final inverse after k levels (beam 0 only; may be null)
    void* finT;                 // [nb][S][6] final inverse after `fin_level` levels, plan dtype (may be null)
    int fin_level;
    double* norms;              // [levels_full]  max over beams (must be zeroed before launch)
    double* blocks0;            // [S][15] node blocks before reduction (beam 0 only; may be null)
    // one workgroup per beam; element columns are [nb][n_elem] with this stride (0 = one shared beam)
    size_t elem_stride;
};

__device__ __forceinline__ void atomic_max_nonneg(double* addr, double v) {
    // v >= 0: the IEEE bit pattern orders like an unsigned integer
    atomicMax(reinterpret_cast<unsigned long long*>(addr), static_cast<unsigned long long>(__double_as_longlong(v)));
}

template <typename T>
__global__ void __launch_bounds__(1024) crb_assemble_kernel(const AsmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    NodeBlocks* sh = reinterpret_cast<NodeBlocks*>(crb_smem);  // [S]
    const int j = threadIdx.x;
    const int beam = blockIdx.x;
    const bool valid = j < p.S;
    const int node = j + p.off, ne = p.n_elem, nn = p.n_node;
    auto fm = [&](int nd, int c) { return nd >= 0 && nd < nn && p.free_dof[3 * nd + c] != 0; };
    const size_t eo = size_t(beam) * p.elem_stride;
    const double *pL = p.L + eo, *pE = p.E + eo, *pI = p.I + eo, *pRho = p.rho + eo, *pA = p.A + eo;
    const uint8_t* pNl = p.nonlinear + eo;
    const double *pWet = p.wet ? p.wet + eo : nullptr, *pCd = p.cd ? p.cd + eo : nullptr;
    const size_t tab = size_t(beam) * size_t(p.S);  // this beam's first slot in the per-beam tables

    NodeBlocks cur;
    for (int k = 0; k < 4; ++k) cur.A[k] = cur.B[k] = cur.C[k] = 0.0;
    cur.a_ax = cur.b_ax = cur.c_ax = 0.0;
    double Lc = 1.0;
    if (valid) {
        // ---- per-slot constants
        SlotConst<T> sc;
        const int e = node - 1;
        int kind = KIND_NONE;
        if (e >= 0) {
            kind = pNl[e] ? KIND_NONLINEAR : KIND_LINEAR;
            elem_coef_build<T>(sc.elem, kind, pL[e], pE[e], pI[e], pA[e]);
        } else {
            elem_coef_build<T>(sc.elem, KIND_NONE, 1.0, 1.0, 1.0, 1.0);
        }
        for (int c = 0; c < 3; ++c) sc.mask[c] = fm(node, c) ? T(1) : T(0);
        sc.pad0 = T(0);
        sc.drag = T(0);
        if ((p.flags & 1u) && fm(node, 1)) {
            const int row = node < ne ? node : ne - 1;
            sc.drag = T(0.5 * p.fluid_density * pCd[row] * pWet[row]);
        }
        sc.half_mass = ((p.flags & 2u) && j < ne) ? T(0.5 * (pRho[j] * pA[j] * pL[j])) : T(0);
        sc.grav = p.grav[j];
        static_cast<SlotConst<T>*>(p.slot_out)[tab + j] = sc;

        // ---- mass matrix, node-block form, with the boundary-condition masks
        const int el = node - 1, er = node;
        if (el >= 0) mass_add_as_left_elem(cur, pL[el], pRho[el] * pA[el], j >= 1);
        if (er < ne) mass_add_as_right_elem(cur, pL[er], pRho[er] * pA[er]);
        const bool hl = j >= 1, hr = j + 1 < p.S;
        mass_apply_masks(cur, fm(node, 0), fm(node, 1), fm(node, 2), hl && fm(node - 1, 0), hl && fm(node - 1, 1),
                         hl && fm(node - 1, 2), hr && fm(node + 1, 0), hr && fm(node + 1, 1), hr && fm(node + 1, 2));
        if (p.blocks0 && beam == 0) {
            double* b0 = p.blocks0 + size_t(j) * 15;
            b0[0] = cur.a_ax; b0[1] = cur.b_ax; b0[2] = cur.c_ax;
            for (int k = 0; k < 4; ++k) { b0[3 + k] = cur.A[k]; b0[7 + k] = cur.B[k]; b0[11 + k] = cur.C[k]; }
        }
        const int elc = node - 1 >= 0 ? node - 1 : 0;
        Lc = pL[elc < ne ? elc : ne - 1];
    }

    // ---- cyclic-reduction factorisation, one level per iteration
    for (int l = 0; l <= p.levels_full; ++l) {
        if (valid) {
            double Bi[4];
            inv2(cur.B, Bi);
            const double mu = fm(node, 0) ? 1.0 : 0.0, mw = fm(node, 1) ? 1.0 : 0.0, mp = fm(node, 2) ? 1.0 : 0.0;
            const double fin[PCR_FINAL_VALS] = {mu / cur.b_ax, mw * mw * Bi[0], mw * mp * Bi[1], mp * mw * Bi[2],
                                                mp * mp * Bi[3], 0.0};
            if (p.fin64_all && beam == 0)
                for (int k = 0; k < PCR_FINAL_VALS; ++k) p.fin64_all[(size_t(l) * p.S + j) * PCR_FINAL_VALS + k] = fin[k];
            if (p.finT && l == p.fin_level)
                for (int k = 0; k < PCR_FINAL_VALS; ++k) static_cast<T*>(p.finT)[(tab + j) * PCR_FINAL_VALS + k] = T(fin[k]);
            sh[j] = cur;
        }
        if (l == p.levels_full) break;
        __syncthreads();
        const int s = 1 << l;
        NodeBlocks nxt = cur;
        if (valid) {
            PcrLevel lv;
            const bool hl = j - s >= 0, hh = j + s < p.S;
            const NodeBlocks lo = sh[hl ? j - s : j], hi = sh[hh ? j + s : j];
            pcr_factor_level(cur, lo, hl, hi, hh, lv, nxt);
            const size_t o = ((size_t(beam) * p.levels_full + l) * p.S + j) * PCR_LEVEL_VALS;
            double vals[PCR_LEVEL_VALS] = {lv.al_ax, lv.ga_ax, lv.al[0], lv.al[1], lv.al[2], lv.al[3],
                                           lv.ga[0], lv.ga[1], lv.ga[2], lv.ga[3]};
            for (int k = 0; k < PCR_LEVEL_VALS; ++k) {
                if (p.lv64 && beam == 0) p.lv64[(size_t(l) * p.S + j) * PCR_LEVEL_VALS + k] = vals[k];
                static_cast<T*>(p.lvT)[o + k] = T(vals[k]);
            }
            atomic_max_nonneg(p.norms + l, pcr_level_norm(lv, Lc));
        }
        __syncthreads();
        cur = nxt;
    }
}

// ------------------------------------------------------------------ feedback force (f-1)
// crb_feedback_kernel: U = (R - X) K^T for the whole ensemble, the controller of
// examples/lqr_control.py:95-111 / control/full_state_linear.py:81 (u = K (r - x)) as ONE fp64 GEMM
//     [B x 2n] . [2n x n]    on v_mfma_f64_16x16x4_f64,
// with the gather from the device state layout fused into the A-operand load and the scatter into the
// device force layout fused into the epilogue (no reduced-order copies of the state, no library call).
// Tile: BM beams x BN outputs per 256-thread workgroup (64 x 64, or 32 x 32 when that is needed to fill the
// chip), each wave a quarter of it in 16 x 16 MFMA tiles,
// K step 32 through LDS with the next step's loads in flight.  LDS rows are [row][32 + 2 pad] doubles: the fragment reads
// (lane l -> row l&15, k = kk + (l>>4)) then touch every bank exactly once.
// MFMA lane maps (MI355X guide): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D: col j = l&15, row i = (l>>4) + 4*reg.
typedef double crb_d4 __attribute__((ext_vector_type(4)));
struct FeedbackParams {
    const double* xs;       // [B][2][n_node][4]
    const double* ref;      // [B][2n] reduced, or nullptr (= 0)
    const double* gain;     // [n][2n] row-major
    double* u;              // [B][n_node][4]; only free-DOF entries are written
    const int32_t* col_off; // [2n] offset of reduced state index j inside a beam's state record
    const int32_t* row_off; // [n]  offset of reduced position index i inside a beam's force record
    int B, n, n2;           // n2 = 2n
    size_t x_stride, u_stride;
};
// BM x BN outputs per 256-thread workgroup; the 4 waves form a WR x (4/WR) grid, each wave owning
// (BM/WR) x (BN/WC) outputs = TM x TN MFMA tiles of 16 x 16.  K advances in steps of BK through two LDS
// stages (one barrier per step).  The global loads of step s+1 are issued BEFORE the MFMAs of step s and
// only touched (negated, masked, stored to LDS) AFTER them, so they fly during the matrix work; every load is
// unconditional (rows / columns out of range read a clamped address and are zeroed by a select) -- a load
// under a per-lane branch would be followed by s_waitcnt vmcnt(0) inside the branch.  The reduced-index ->
// state-offset table sits in LDS.
#ifndef CRB_GEMM_SCHED
#define CRB_GEMM_SCHED 0
#endif
template <int BM, int BN, int BK, int WR, bool HAS_REF>
__global__ void __launch_bounds__(256) crb_feedback_kernel(const FeedbackParams p) {
    constexpr int WC = 4 / WR, TM = BM / (16 * WR), TN = BN / (16 * WC), LD = BK + 2;
    constexpr int QA = BM * BK / 256, QB = BN * BK / 256, RSTEP = 256 / BK;
    static_assert(BM % (16 * WR) == 0 && BN % (16 * WC) == 0 && 256 % BK == 0 && BM % RSTEP == 0 && BN % RSTEP == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    double* const As = reinterpret_cast<double*>(crb_smem);          // [2][BM * LD]
    double* const Bs = As + 2 * BM * LD;                             // [2][BN * LD]
    int32_t* const coff_s = reinterpret_cast<int32_t*>(Bs + 2 * BN * LD);  // [n2]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = (wave / WC) * (BM / WR), wn = (wave % WC) * (BN / WC);
    for (int k = t; k < p.n2; k += 256) coff_s[k] = p.col_off[k];
    crb_d4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = crb_d4{0.0, 0.0, 0.0, 0.0};

    // loader: this thread fetches column lk of rows lr + RSTEP*q of both tiles
    const int lk = t & (BK - 1), lr = t / BK;
    const double* xrow[QA];
    const double* rrow[QA];
    const double* grow[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int b = m0 + lr + RSTEP * q;
        const int bc = b < p.B ? b : p.B - 1;
        xrow[q] = p.xs + size_t(bc) * p.x_stride;
        rrow[q] = HAS_REF ? p.ref + size_t(bc) * p.n2 : nullptr;
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int i = n0 + lr + RSTEP * q;
        grow[q] = p.gain + size_t(i < p.n ? i : p.n - 1) * p.n2;
    }
    __syncthreads();  // coff_s
    // Three-deep pipeline over K steps: while the MFMAs of step s run out of LDS stage s&1, the values of
    // step s+1 (loaded one iteration ago, now in registers) are negated / masked / stored into stage (s+1)&1
    // piecewise BETWEEN the MFMA groups, and the global loads of step s+2 are in flight.
    // Out-of-range ROWS of either tile need no masking (their outputs are never stored); the K tail is
    // zeroed on the gain side only, by a multiplication (a per-lane select on k would come back as a
    // branch and split the scheduling region).  Without a reference the A tile holds +x and the sign goes
    // into the epilogue: the state goes from global memory to LDS untouched.
    struct Regs { double xa[QA], ra[QA], gb[QB]; double kmask; };
    Regs R0, R1;
    auto fetch = [&](Regs& R, int k0) {
        const int k = k0 + lk;
        const bool kok = k < p.n2;
        R.kmask = kok ? 1.0 : 0.0;
        const int kc = kok ? k : 0;
        const int coff = coff_s[kc];
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            R.xa[q] = xrow[q][coff];
            if (HAS_REF) R.ra[q] = rrow[q][kc];
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) R.gb[q] = grow[q][kc];
    };
    auto stash_piece = [&](const Regs& R, int st, int piece) {   // element `piece` of the QA + QB this thread stores
        double* A = As + st * BM * LD;
        double* Bt = Bs + st * BN * LD;
        if (piece < QA) {
            const int q = piece;
            A[(lr + RSTEP * q) * LD + lk] = HAS_REF ? R.ra[q] - R.xa[q] : R.xa[q];
        } else if (piece < QA + QB) {
            const int q = piece - QA;
            Bt[(lr + RSTEP * q) * LD + lk] = R.gb[q] * R.kmask;
        }
    };
    constexpr int NSUB = BK / 4, NPIECE = QA + QB, PPS = (NPIECE + NSUB - 1) / NSUB;   // pieces per MFMA sub-step
    // split K: slice blockIdx.z of gridDim.z takes a contiguous range of K steps; with two slices the partial
    // sums are added into a zeroed U by fp64 atomics (0 + a + b == 0 + b + a bit for bit: still deterministic)
    const int all_steps = (p.n2 + BK - 1) / BK;
    const int per_slice = (all_steps + int(gridDim.z) - 1) / int(gridDim.z);
    const int kbase = int(blockIdx.z) * per_slice * BK;
    const int nsteps = min(per_slice, all_steps - int(blockIdx.z) * per_slice);
    fetch(R0, kbase);
#pragma unroll
    for (int piece = 0; piece < NPIECE; ++piece) stash_piece(R0, 0, piece);
    fetch(R1, kbase + BK);
    __syncthreads();
    auto step = [&](Regs& Rcur /* step s+1 */, Regs& Rnxt /* receives step s+2 */, int sidx) {
        const int st = sidx & 1;
        // (no conditions here: past the end fetch() reads clamped addresses and the stash fills an LDS stage
        //  that nobody reads any more -- one basic block, so that the interleave below can be enforced)
        fetch(Rnxt, kbase + (sidx + 2) * BK);
        const double* Aw = As + st * BM * LD + (wm + (lane & 15)) * LD + (lane >> 4);
        const double* Bw = Bs + st * BN * LD + (wn + (lane & 15)) * LD + (lane >> 4);
        double af[2][TM], bf[2][TN];   // fragments of sub-step kk+4 are read while the MFMAs of sub-step kk run
#pragma unroll
        for (int a = 0; a < TM; ++a) af[0][a] = Aw[16 * a * LD];
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = Bw[16 * b * LD];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const int cur = (kk >> 2) & 1, nxt = cur ^ 1;
            if (kk + 4 < BK) {
#pragma unroll
                for (int a = 0; a < TM; ++a) af[nxt][a] = Aw[16 * a * LD + kk + 4];
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[nxt][b] = Bw[16 * b * LD + kk + 4];
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur][a], bf[cur][b], acc[a][b], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < PPS; ++i) stash_piece(Rcur, st ^ 1, (kk >> 2) * PPS + i);
        }
#if CRB_GEMM_SCHED
        // a wave issues in order: an MFMA holds the matrix pipe for 64 cycles, and whatever follows it in
        // program order can issue in that shadow only if it is not another MFMA.  Ask the scheduler for
        // MFMA / LDS read / VALU / global load / LDS write round-robin instead of MFMA clusters.
#pragma unroll
        for (int i = 0; i < NSUB * TM * TN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#endif
        __syncthreads();
    };
    for (int sidx = 0; sidx < nsteps; sidx += 2) {
        step(R1, R0, sidx);
        if (sidx + 1 < nsteps) step(R0, R1, sidx + 1);
    }
    // epilogue: D row (beam) = (lane>>4) + 4*reg, D col (output) = lane&15; scatter into the force layout
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int i = n0 + wn + 16 * b + (lane & 15);
            if (i >= p.n) continue;
            const int roff = p.row_off[i];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int beam = m0 + wm + 16 * a + (lane >> 4) + 4 * reg;
                if (beam < p.B) {
                    const double v = HAS_REF ? acc[a][b][reg] : -acc[a][b][reg];
                    double* dst = p.u + size_t(beam) * p.u_stride + roff;
                    if (gridDim.z > 1) unsafeAtomicAdd(dst, v);
                    else *dst = v;
                }
            }
        }
}
// crb_feedback_ws_kernel: the same product, wave-specialised.  512 threads: waves 0..3 only read fragments
// from LDS and issue MFMAs (wave w: rows 16w..16w+15 of the 64-row tile x all BN columns), waves 4..7 only
// move data (global -> registers -> LDS stage of the NEXT K step, loads of the step after that in flight).
// Each SIMD then holds one matrix wave and one loader wave: the loader's address arithmetic, stores and
// memory waits issue in the shadow of the other wave's 64-cycle MFMAs instead of in front of them.
// One barrier per K step, taken by both roles.  In-kernel cycle stamps at 2048 x 768 x 384 (64 x 48 tiles,
// BK 64, 12 K steps): prologue 7.3k cycles (offset table, first tile), per step 3.6k cycles for 48 MFMAs
// (75 each, 64 = pipe-bound) + 0.4-0.7k at the barrier; 32.7 us against 34.1 us for crb_feedback_kernel.
template <int BN, int BK, bool HAS_REF>
__global__ void __launch_bounds__(512) crb_feedback_ws_kernel(const FeedbackParams p) {
    constexpr int BM = 64, TN = BN / 16, LD = BK + 2;
    constexpr int QA = BM * BK / 256, QB = BN * BK / 256, RSTEP = 256 / BK;
    static_assert(BN % 16 == 0 && 256 % BK == 0 && BM % RSTEP == 0 && BN % RSTEP == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    double* const As = reinterpret_cast<double*>(crb_smem);          // [2][BM * LD]
    double* const Bs = As + 2 * BM * LD;                             // [2][BN * LD]
    int32_t* const coff_s = reinterpret_cast<int32_t*>(Bs + 2 * BN * LD);  // [n2]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    for (int k = t; k < p.n2; k += 512) coff_s[k] = p.col_off[k];
    const int nsteps = (p.n2 + BK - 1) / BK;
    __syncthreads();  // coff_s

    if (wave >= 4) {
        // ------------------------------------------------ loader role
        __builtin_amdgcn_s_setprio(0);
        const int lt = t - 256;
        const int lk = lt & (BK - 1), lr = lt / BK;
        const double* xrow[QA];
        const double* rrow[QA];
        const double* grow[QB];
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const int b = m0 + lr + RSTEP * q;
            const int bc = b < p.B ? b : p.B - 1;
            xrow[q] = p.xs + size_t(bc) * p.x_stride;
            rrow[q] = HAS_REF ? p.ref + size_t(bc) * p.n2 : nullptr;
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int i = n0 + lr + RSTEP * q;
            grow[q] = p.gain + size_t(i < p.n ? i : p.n - 1) * p.n2;
        }
        struct Regs { double xa[QA], ra[QA], gb[QB]; double kmask; };
        Regs R0, R1;
        auto fetch = [&](Regs& R, int k0) {   // unconditional loads (clamped), K tail zeroed on the gain side
            const int k = k0 + lk;
            const bool kok = k < p.n2;
            R.kmask = kok ? 1.0 : 0.0;
            const int kc = kok ? k : 0;
            const int coff = coff_s[kc];
#pragma unroll
            for (int q = 0; q < QA; ++q) {
                R.xa[q] = xrow[q][coff];
                if (HAS_REF) R.ra[q] = rrow[q][kc];
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) R.gb[q] = grow[q][kc];
        };
        auto stash = [&](const Regs& R, int st) {
            double* A = As + st * BM * LD;
            double* Bt = Bs + st * BN * LD;
#pragma unroll
            for (int q = 0; q < QA; ++q) A[(lr + RSTEP * q) * LD + lk] = HAS_REF ? R.ra[q] - R.xa[q] : R.xa[q];
#pragma unroll
            for (int q = 0; q < QB; ++q) Bt[(lr + RSTEP * q) * LD + lk] = R.gb[q] * R.kmask;
        };
        fetch(R0, 0);
        fetch(R1, BK);
        stash(R0, 0);
        __syncthreads();                       // stage 0 ready
        for (int sidx = 0; sidx < nsteps; sidx += 2) {
            fetch(R0, (sidx + 2) * BK);
            stash(R1, 1);                      // step sidx+1 -> stage 1 while the matrix waves work on stage 0
            __syncthreads();
            if (sidx + 1 < nsteps) {
                fetch(R1, (sidx + 3) * BK);
                stash(R0, 0);                  // step sidx+2 -> stage 0 while they work on stage 1
                __syncthreads();
            }
        }
        return;
    }
    // ---------------------------------------------------- matrix role
    __builtin_amdgcn_s_setprio(2);
    crb_d4 acc[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[b] = crb_d4{0.0, 0.0, 0.0, 0.0};
    __syncthreads();                           // stage 0 ready
    for (int sidx = 0; sidx < nsteps; ++sidx) {
        const int st = sidx & 1;
        const double* Aw = As + st * BM * LD + (16 * wave + (lane & 15)) * LD + (lane >> 4);
        const double* Bw = Bs + st * BN * LD + (lane & 15) * LD + (lane >> 4);
        double af[2], bf[2][TN];               // fragments of sub-step kk+4 are read while the MFMAs of sub-step kk run
        af[0] = Aw[0];
#pragma unroll
        for (int b = 0; b < TN; ++b) bf[0][b] = Bw[16 * b * LD];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const int cur = (kk >> 2) & 1, nxt = cur ^ 1;
            if (kk + 4 < BK) {
                af[nxt] = Aw[kk + 4];
#pragma unroll
                for (int b = 0; b < TN; ++b) bf[nxt][b] = Bw[16 * b * LD + kk + 4];
            }
#pragma unroll
            for (int b = 0; b < TN; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[cur], bf[cur][b], acc[b], 0, 0, 0);
        }
        __syncthreads();
    }
    // epilogue: D row (beam) = (lane>>4) + 4*reg, D col (output) = lane&15; scatter into the force layout
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int i = n0 + 16 * b + (lane & 15);
        if (i >= p.n) continue;
        const int roff = p.row_off[i];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int beam = m0 + 16 * wave + (lane >> 4) + 4 * reg;
            if (beam < p.B) p.u[size_t(beam) * p.u_stride + roff] = HAS_REF ? acc[b][reg] : -acc[b][reg];
        }
    }
}
template <int BM, int BN, int BK>
__host__ __device__ constexpr size_t feedback_lds_bytes(int n2) {
    return size_t(2) * (BM + BN) * (BK + 2) * sizeof(double) + size_t(n2) * sizeof(int32_t);
}

// ------------------------------------------------------------------ layout conversion
// reduced [B][rows*n_free] <-> device [B][rows][n_node][4]; free_index[r] = 3*node + dof
template <typename T, bool PACK>
__global__ void crb_pack_kernel(const int32_t* free_index, int n_free, int n_node, int rows, int B, const T* src_red,
                                T* dev, T* dst_red) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t per_beam = size_t(rows) * n_free;
    if (i >= size_t(B) * per_beam) return;
    const size_t b = i / per_beam, rem = i - b * per_beam;
    const int row = int(rem / n_free), r = int(rem - size_t(row) * n_free);
    const int fi = free_index[r];
    const size_t d = (b * rows + row) * size_t(n_node) * 4 + size_t(fi / 3) * 4 + (fi % 3);
    if (PACK) dev[d] = src_red[i];
    else dst_red[i] = dev[d];
}

template <typename T>
__global__ void crb_gather_kernel(const T* x, size_t beam_stride, size_t offset, int B, T* out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = x[size_t(b) * beam_stride + offset];
}

}  // namespace crb
