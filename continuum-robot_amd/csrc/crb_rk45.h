// crb_rk45.h -- adaptive Dormand-Prince 5(4) with per-beam step control (crb_solve_rk45).
#pragma once
#include <hip/hip_runtime.h>

#include "crb_generic.h"

namespace crb {

// ------------------------------------------------------------------ adaptive RK45 (f-2)
// crb_rk45_kernel: embedded Dormand-Prince 5(4) with per-beam step-size control, the whole integration
// t0 -> t_end in one launch.  It restates scipy.integrate.solve_ivp(method="RK45") -- the integrator the
// reference's tests hand its RHS to (tests/test_dynamic_beam.py:218-220, test_functional_composition.py:539-546;
// scipy/integrate/_ivp/rk.py, scipy 1.15: rk_step, RungeKutta._step_impl, select_initial_step) -- so a beam
// takes the step sequence scipy would take on the same RHS: same tableau, RMS error norm over the
// REDUCED state with scale = atol + max(|y|,|y_new|) rtol, SAFETY 0.9, factors in [0.2, 10], no growth
// right after a rejection, FSAL, the same initial-step heuristic and end-point clipping.
// One workgroup per beam (or several small beams per wave): every beam has its own clock and step.
// The stage derivatives live in LDS (thread-private columns: no barrier).  General RHS: all of [7][6][NT].
// Lean RHS (SLIM): 27 columns instead of 42 -- the accelerations of K0..K5 and the velocity halves of K3..K5; the
// velocity half of K0 IS the state's velocity, those of K1 / K2 are one / two multiply-adds of K0 / K1's
// accelerations (re-evaluated with the very expression that produced them), K6 stays in registers until the step is
// decided.  With the lean RHS's 12 exchange columns that is 80 KB for a 256-node beam: two workgroups per CU.
struct Rk45Params {
    double t0, t_end, rtol, atol;
    double* h_io;      // [B] in: first step (<= 0: choose like scipy), out: next step suggestion
    int32_t* stats;    // [B][4] accepted, rejected, nfev, status (0 ok, 1 step too small)
    int n_state;       // 2 * n_free: size of the reference's state vector (the error norm's N)
    const int32_t* n_state_b;   // [B] per-beam sizes (mixed ensembles) or nullptr
    int max_steps;     // safety bound on attempted steps
    // dense output of ONE DOF on the uniform grid t_eval[k] = eval_t0 + k*eval_dt, k < n_eval (solve_ivp's
    // t_eval): after every accepted step the grid points in (t_old, t_new] -- plus t_eval[0] == t0 -- are
    // evaluated with scipy's 4th-order interpolant (RkDenseOutput, RK45.P) and stored at eval_out[b][k]
    void* eval_out;    // [B][n_eval] plan dtype, or nullptr
    double eval_t0, eval_dt;
    int n_eval, eval_slot, eval_comp;
};

// scipy RK45.A with RK45.B as row 6 (the state of stage 6 is y_new); RK45.C with C[6] = 1
__device__ const double RK45_W[7][6] = {{0, 0, 0, 0, 0, 0},
                                        {1.0 / 5, 0, 0, 0, 0, 0},
                                        {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
                                        {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
                                        {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
                                        {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
                                        {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
__device__ const double RK45_C[7] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};

// LDS values the lean RHS exchanges through: [3 + 6 + 3][NT + 1] (lean_rhs), rounded up to an even count
__host__ __device__ constexpr int rk45_exchange_vals(int NT) { return (12 * (NT + 1) + 1) & ~1; }

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
    constexpr bool SLIM = LNW >= 0;
    const Lds<T> lds = carve_lds<T>(NT);
    T* const Ks = SLIM ? lds.q + rk45_exchange_vals(NT) : lds.r1 + 3 * NT;   // SLIM: [27][NT], else [7][6][NT]
    double* const red = reinterpret_cast<double*>(Ks + (SLIM ? 27 : 42) * NT);  // [NT/64]
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
    tp.beam = beam;
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
        if (p.amp && tp.j == (p.imp_node_b ? p.imp_node_b[beam] - p.off : p.imp_slot)) amp = p.amp[beam];
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
    // K_k[c]: c < 3 the velocity half, c >= 3 the acceleration half.  SLIM keeps in LDS: columns 3k + (c-3) for the
    // accelerations of k < 6, columns 18 + 3(k-3) + c for the velocity halves of k = 3..5
    auto putK = [&](int k, const T d[6]) {
        if (SLIM) {
            if (k < 6) {
#pragma unroll
                for (int c = 0; c < 3; ++c) Ks[(k * 3 + c) * NT + t] = d[3 + c];
            }
            if (k >= 3 && k < 6) {
#pragma unroll
                for (int c = 0; c < 3; ++c) Ks[(18 + (k - 3) * 3 + c) * NT + t] = d[c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < 6; ++c) Ks[(k * 6 + c) * NT + t] = d[c];
        }
    };
    auto rms = [&](const double v[6]) {  // scipy: norm(x) / sqrt(x.size) over the reduced state
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < 6; ++c) s += v[c] * v[c];
        return sqrt(block_sum<T>(s, red, NT, t, 0, 0, false) / double(q.n_state_b ? q.n_state_b[beam] : q.n_state));
    };

    // Dormand-Prince tableau: RK45_W / RK45_C above (scipy RK45.A, .B, .C), error weights RK45.E
    const double E5[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};

    // scipy RK45.P (dense output): y(t_old + x h) = y_old + h * sum_m x^(m+1) * sum_j K_j P[j][m]
    const double P5[7][4] = {{1.0, -2.8535800653862835, 3.0717434641059005, -1.1270175653862835},
                             {0.0, 0.0, 0.0, 0.0},
                             {0.0, 4.023133379230305, -6.249321565289, 2.675424484351598},
                             {0.0, -3.7324019615885042, 10.068970589843675, -5.685526961588504},
                             {0.0, 2.5548038301849423, -6.399112377351017, 3.5219323679207912},
                             {0.0, -1.3744241142186024, 3.272657752246729, -1.7672812570757455},
                             {0.0, 1.3824689317781436, -3.764937863556287, 2.382468931778144}};
    // K_k[c] as a double, for k and c known at compile time after unrolling; `last` = K6 (registers), `hh` = the step
    // the stage states were formed with (the velocity halves of K1 / K2 are those stage states' velocities)
    auto getK = [&](int k, int c, const T* last, double hh) -> double {
        if (!SLIM) return double(Ks[(k * 6 + c) * NT + t]);
        if (k == 6) return double(last[c]);
        if (c >= 3) return double(Ks[(k * 3 + (c - 3)) * NT + t]);
        if (k == 0) return double(y[3 + c]);
        if (k >= 3) return double(Ks[(18 + (k - 3) * 3 + c) * NT + t]);
        double dy = 0.0;      // k = 1, 2: exactly the stage state's velocity (the expression of the stage loop below)
        for (int jj = 0; jj < k; ++jj) dy += double(Ks[(jj * 3 + c) * NT + t]) * RK45_W[k][jj];
        return double(T(double(y[3 + c]) + dy * hh));
    };
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
            // stages 1..6 as ONE loop (not unrolled: a single copy of the RHS, and no interleaving of stages that would
            // cost registers); stage 6 forms y_new with the B row and its derivative is K6 (FSAL).  Only the K of
            // earlier stages enter (a wave-uniform test: what a later column still holds from a rejected attempt may
            // be Inf / NaN and must not be touched, not even with a zero weight).
            T yn[6], fn[6];
#pragma nounroll
            for (int s = 1; s <= 6; ++s) {
                double dy[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) {
                    if (jj < s) {
                        const double wj = RK45_W[s][jj];
#pragma unroll
                        for (int c = 0; c < 6; ++c) dy[c] += getK(jj, c, nullptr, h) * wj;
                    }
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) yn[c] = T(double(y[c]) + dy[c] * h);
                deriv(tc + RK45_C[s] * h, yn, fn);
                putK(s, fn);
            }
            nfev += 6;
            double en[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double e = 0.0;
#pragma unroll
                for (int jj = 0; jj < 7; ++jj) e += getK(jj, c, fn, h) * E5[jj];
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
                        auto dense = [&](int c) {   // scipy RkDenseOutput for component c of this thread's node
                            const double x = (te - tc) / h;
                            double xp = x, acc = 0.0;
#pragma unroll
                            for (int m = 0; m < 4; ++m) {
                                double qm = 0.0;
#pragma unroll
                                for (int jj = 0; jj < 7; ++jj) qm += getK(jj, c, fn, h) * P5[jj][m];
                                acc += qm * xp;
                                xp *= x;
                            }
                            return T(h * acc + double(y[c]));
                        };
                        if (q.eval_slot == REC_ALL_SLOTS) {   // whole-state snapshot ie: [n_eval][B][2][n_node][4]
                            if (valid) {
                                T* snap = static_cast<T*>(q.eval_out) + size_t(ie) * size_t(p.B) * 2 * plane + xoff;
#pragma unroll
                                for (int c = 0; c < 3; ++c) { snap[c] = dense(c); snap[plane + c] = dense(3 + c); }
                                snap[3] = T(0);
                                snap[plane + 3] = T(0);
                            }
                        } else if (recorder) {   // (the component is a run-time value: select among compile-time ones)
#pragma unroll
                            for (int c = 0; c < 6; ++c)
                                if (c == q.eval_comp) static_cast<T*>(q.eval_out)[size_t(beam) * q.n_eval + ie] = dense(c);
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
__host__ __device__ constexpr size_t rk45_lds_bytes(int NT, bool slim = false) {
    return (slim ? size_t(rk45_exchange_vals(NT)) * sizeof(T) : lds_bytes<T>(NT)) + size_t(slim ? 27 : 42) * NT * sizeof(T) +
           size_t(NT / 64 + 1) * sizeof(double);
}

}  // namespace crb
