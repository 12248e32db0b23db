// crb_generic.h -- parameter blocks, slot topology and the general kernels (any gravity table, any
// waves-per-beam count, several small beams per wave): crb_beam_kernel<MODE_STEP|RHS|KQ|STAGE>.
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
    int rec_slot, rec_comp;    // recording thread (slot; REC_ALL_SLOTS = whole-state snapshots
                               // [n_rec][B][2][n_node][4]) and component 0..5 of {q, v}
    int rec_every, rec_n;
    // per-beam coefficient mode (heterogeneous ensembles): table offsets per beam, in elements; 0 = shared
    size_t slot_stride, lv_stride, fin_stride;
    int B, S, G, n_node, off, levels;
    int lognw;                 // log2(wavefronts per beam); 0 when a wave holds whole beams
    uint32_t flags;
    int imp_slot, imp_dof;
    const int32_t* imp_node_b; // [B] per-beam impulse node (mixed ensembles) or nullptr: imp_slot for every beam
    double duration, t0, dt;
    int n_steps;
    T gx, gy;
    const T* gvec;             // [B][2] per-beam gravity vector (mixed ensembles), or nullptr: gx, gy for every beam
    // reduced I/O (MODE_RHS / MODE_KQ, crb_rhs_host): when red_map != nullptr, x / u_held / out are vectors in the
    // reference's REDUCED ordering ([B][2 n_red] states, [B][n_red] forces) and red_map[3 node + dof] is the reduced
    // index of a DOF or -1 -- the (un)packing of crb_pack_* fused into the kernel's own loads and stores
    const int32_t* red_map;
    int n_red;
    // fused state feedback (MODE_STEP, FB instantiation; crb_step_rk4_feedback for beams that live in one wave):
    // u = K (r - x) re-evaluated in every stage from an LDS copy of the gain.  red_map / n_red as above.
    const T* fb_gain;          // [n_red][2 n_red] row-major, reduced ordering
    const T* fb_ref;           // [B][2 n_red] or nullptr (= 0)
    // completion flag of a ONE-workgroup launch whose output lands in host-mapped memory (crb_rhs_host): written with
    // system scope after the output stores, so that the host can spin on it instead of waiting for the stream
    unsigned long long* done_flag;
    unsigned long long done_seq;
    // per-beam status (optional): status[b] stays 0 while beam b is finite; the first launch that leaves a non-finite value in
    // the beam's state writes `status_value` there (the number of steps the ensemble has taken by the end of that launch)
    int32_t* status;
    int32_t status_value;
};

// (end of a stepper launch) a thread whose node came out non-finite marks its beam, once: NaN / Inf never turn finite again,
// so the first launch that finds them is the launch in which they appeared
template <typename T>
__device__ __forceinline__ void mark_nonfinite(const KParams<T>& p, int beam, const T q[3], const T v[3]) {
    if (!p.status) return;
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) bad = bad || !isfinite(q[c]) || !isfinite(v[c]);
    if (bad) atomicCAS(p.status + beam, 0, p.status_value);
}

enum : int { MODE_STEP = 0, MODE_RHS = 1, MODE_KQ = 2, MODE_STAGE = 3 };
constexpr int REC_ALL_SLOTS = -2;
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
    int t, lane, j, S, lognw, nwm1, base, beam;
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
// STIFF (implicit stepper, crb_stiff.h): the solve tables in `cf` are those of A = M + alpha K0 and the right-hand side
// gains + alpha K0 z (K0 = the linear element stiffness, z = the current acceleration iterate), so that the result is
// the next iterate of  a <- Ainv (F(q, v) + alpha K0 a).
template <typename T, int LV, bool KQ_ONLY, bool LEAN, bool STIFF = false>
__device__ __forceinline__ void stage_accel(const KParams<T>& p, const Lds<T>& lds, const SlotConst<T>& sc,
                                            const SolveCoef<T, LV>& cf, const Topo& tp, const T q[3], const T v[3],
                                            const T uadd[3], T a[3], const T* z = nullptr, T alpha = T(0)) {
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
    if (STIFF) {   // - alpha K0 z joins the internal force (it is subtracted from the right-hand side below)
        T zl[3], lin[5] = {T(0), T(0), T(0), T(0), T(0)}, kl[3], kr[3];
        const T zz[3] = {z[0], z[1], z[2]};
        neighbours<T>(tp, lds.r1, NT, 1, cross1, zz, true, false, zl, dummy);
        elem_linear_coefs<T>(sc.elem.c, sc.elem.kind, lin);
        elem_force_linear<T>(lin, zl, zz, kl, kr);
        if (sc.elem.kind == KIND_NONLINEAR && !corrected) kl[0] = lin[0] * zl[0];   // tangent of the shipped f1: no u2 term
#pragma unroll
        for (int c = 0; c < 3; ++c) { fl[c] -= alpha * kl[c]; fr[c] -= alpha * kr[c]; }
    }

    T gseg[2] = {T(0), T(0)};
    if (!KQ_ONLY && grav_on && sc.half_mass != T(0)) {
        const int ia = sc.grav.phiA, ib = sc.grav.phiB;
        T phi = T(0);
        if (ia >= 0) phi = lds.q[(ia & 3) * NT + tp.thread_of(ia >> 2)];
        if (ib >= 0) phi = T(0.5) * (phi + lds.q[(ib & 3) * NT + tp.thread_of(ib >> 2)]);
        T gx = p.gx, gy = p.gy;
        if (p.gvec) { gx = p.gvec[2 * size_t(tp.beam)]; gy = p.gvec[2 * size_t(tp.beam) + 1]; }   // (per-beam ForceParams)
        gravity_segment<T>(phi, gx, gy, sc.half_mass, gseg);
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
// FB (MODE_STEP): state feedback inside the stages (lqr_control.py:95-111: c = K (0 - x) added to the input of every
// RHS evaluation).  The gain sits in LDS, transposed ([2n][n]: the threads of a beam read consecutive addresses), the
// stage's error vector r - x of every beam of the workgroup next to it in the reference's reduced ordering; a thread
// forms the three entries of K e that belong to its node: 6n multiply-adds per stage.  For beams whose gain fits LDS
// (<= ~28 elements); larger ensembles take the stage-split path (one MFMA GEMM per stage, crb_feedback.h).
constexpr int FB_BATCH = 8;
__host__ __device__ constexpr int fb_padded(int n2) { return (n2 + FB_BATCH - 1) / FB_BATCH * FB_BATCH; }
// Gains of 21 .. FBM_N rows (beams of 7 .. 10 elements: the reference's examples) take the product K e to the matrix cores:
// U^T = K E^T with the GAIN as the A operand -- its fragments (2 row tiles x 16 k-steps = 32 values per lane) are loaded into
// registers once per launch -- and the stage's error vectors of the wave's beams as the B operand (16 LDS reads per lane and
// stage instead of 240 per thread); 32 matrix instructions per stage whatever the number of beams in the wave (<= 16).
constexpr int FBM_N = 32, FBM_MT = 2, FBM_KS = 16, FBM_UPAD = 32;
// (the fixed 32 instructions beat the batched LDS product from ~20 rows on: 64 x 10 elements 14.1 against 17.0 us per step,
//  64 x 6 elements 13.7 against 12.8)
__host__ __device__ constexpr bool fb_on_matrix_cores(int G, int n) { return n > 20 && n <= FBM_N && G <= 16; }
template <typename T>
__host__ __device__ constexpr size_t fb_lds_bytes(int NT, int G, int n) {
    return lds_bytes<T>(NT) + (size_t(G) * fb_padded(2 * n) + size_t(fb_padded(2 * n)) * n + (fb_on_matrix_cores(G, n) ? size_t(G) * FBM_UPAD : 0)) * sizeof(T);
}
template <typename T>
__host__ __device__ constexpr size_t fb_lean_lds_bytes(int G, int n) {   // the feedback form of the packed lean stepper: no exchange columns
    return (size_t(G) * fb_padded(2 * n) + size_t(fb_padded(2 * n)) * n + size_t(G) * FBM_UPAD) * sizeof(T);
}
// the MFMA of each dtype: A / B fragments are one value per lane (A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15])
// for both, the C/D row of accumulator register `reg` differs (cdna_hip_programming.md, 'Fragment layout')
template <typename T> struct MfmaOps;
template <> struct MfmaOps<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};
template <> struct MfmaOps<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};
template <typename T>
__device__ __forceinline__ void fbm_gain_fragments(T (&af)[FBM_MT][FBM_KS], const T* fbK, int n, int n2p, int lane) {
    const int fi = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < FBM_MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < FBM_KS; ++ks) {
            const int i = 16 * mt + fi, k = 4 * ks + kq;
            af[mt][ks] = (i < n && k < n2p) ? fbK[size_t(k) * n + i] : T(0);     // (fbK is the transposed gain, its padding rows are 0)
        }
}
// fbu[g][i] = (K e_g)[i] for the G beams of the wave (e_g = fbx + g n2p); whole wave, the caller puts barriers around it
template <typename T>
__device__ __forceinline__ void fbm_product(const T (&af)[FBM_MT][FBM_KS], const T* fbx, T* fbu, int G, int n, int n2p, int lane) {
    typedef typename MfmaOps<T>::acc_t acc4;
    const int fj = lane & 15, kq = lane >> 4;
    const T* ecol = fbx + size_t(fj < G ? fj : 0) * n2p + kq;
    T ef[FBM_KS];
#pragma unroll
    for (int ks = 0; ks < FBM_KS; ++ks) ef[ks] = (fj < G && 4 * ks + kq < n2p) ? ecol[4 * ks] : T(0);
    acc4 acc[FBM_MT];
#pragma unroll
    for (int mt = 0; mt < FBM_MT; ++mt) acc[mt] = acc4{T(0), T(0), T(0), T(0)};
    // (all FBM_KS x FBM_MT products, also where the padding makes them products of zeros: skipping those under wave-uniform
    //  branches breaks the two accumulator chains apart -- measured 17.8 instead of 14.1 us per step for a 10-element beam)
#pragma unroll
    for (int ks = 0; ks < FBM_KS; ++ks)
#pragma unroll
        for (int mt = 0; mt < FBM_MT; ++mt) acc[mt] = MfmaOps<T>::run(af[mt][ks], ef[ks], acc[mt]);
    if (fj < G) {
#pragma unroll
        for (int mt = 0; mt < FBM_MT; ++mt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = 16 * mt + MfmaOps<T>::row(lane, reg);
                if (i < n) fbu[size_t(fj) * FBM_UPAD + i] = acc[mt][reg];
            }
    }
}
// The three entries of K e that belong to one node, rows i0 / i1 / i2 of the gain (fbK: its transpose in LDS), in batches of
// FB_BATCH columns: all LDS loads of a batch are issued before its first multiply-add (one wave per SIMD here, nothing else
// hides the LDS latency; left to itself the scheduler alternates load / wait / multiply-add: 200 cycles per column).  The
// column count is padded to a multiple of the batch with zero columns.
template <typename T>
__device__ __forceinline__ void fb_product_lds(const T* e, const T* fbK, int n, int n2p, int i0, int i1, int i2, T& u0, T& u1, T& u2) {
    for (int k = 0; k < n2p; k += FB_BATCH) {
        T ek[FB_BATCH], r0[FB_BATCH], r1[FB_BATCH], r2[FB_BATCH];
#pragma unroll
        for (int qq = 0; qq < FB_BATCH; ++qq) {
            const T* row = fbK + size_t(k + qq) * n;
            ek[qq] = e[k + qq]; r0[qq] = row[i0]; r1[qq] = row[i1]; r2[qq] = row[i2];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * FB_BATCH, 0);   // the DS reads first ...
#pragma unroll
        for (int qq = 0; qq < FB_BATCH; ++qq) { u0 += r0[qq] * ek[qq]; u1 += r1[qq] * ek[qq]; u2 += r2[qq] * ek[qq]; }
        __builtin_amdgcn_sched_group_barrier(0x002, 3 * FB_BATCH, 0);   // ... then the arithmetic
    }
}
// K e of the stage for this thread's node, through whichever form the gain's size selects (both read e = fbx + g n2p of every beam
// of the wave, written by the caller before the first barrier here).  Whole wave.
template <typename T>
__device__ __forceinline__ void fb_feedback(bool on_matrix_cores, const T (&af)[FBM_MT][FBM_KS], const T* fbx, const T* fbK, T* fbu, int G,
                                            int g, int n, int n2p, int lane, bool valid, const int (&red)[3], T (&u)[3]) {
    __syncthreads();
    u[0] = u[1] = u[2] = T(0);
    if (on_matrix_cores) {
        fbm_product<T>(af, fbx, fbu, G, n, n2p, lane);
        __syncthreads();
        if (valid) {
            const T* ub = fbu + size_t(g) * FBM_UPAD;
#pragma unroll
            for (int c = 0; c < 3; ++c) u[c] = red[c] >= 0 ? ub[red[c]] : T(0);
        }
    } else if (valid) {
        const int i0 = red[0] >= 0 ? red[0] : 0, i1 = red[1] >= 0 ? red[1] : 0, i2 = red[2] >= 0 ? red[2] : 0;
        T u0 = T(0), u1 = T(0), u2 = T(0);
        fb_product_lds<T>(fbx + size_t(g) * n2p, fbK, n, n2p, i0, i1, i2, u0, u1, u2);
        u[0] = red[0] >= 0 ? u0 : T(0);
        u[1] = red[1] >= 0 ? u1 : T(0);
        u[2] = red[2] >= 0 ? u2 : T(0);
    }
}

template <typename T, int MODE, int LV, int MAXT, int MINW, bool LEAN, bool FB = false>
__global__ void __launch_bounds__(MAXT, MINW) crb_beam_kernel(const KParams<T> p) {
    static_assert(!FB || (MODE == MODE_STEP && !LEAN), "feedback lives in the general stepper");
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
    int red[3] = {-1, -1, -1};   // reduced indices of this node's DOFs (reduced I/O only)
    if (valid && (MODE == MODE_RHS || MODE == MODE_KQ) && p.red_map) {
        const size_t rb = size_t(beam) * 2 * size_t(p.n_red), ub = size_t(beam) * size_t(p.n_red);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            red[c] = p.red_map[3 * node + c];
            if (red[c] >= 0) {
                x[c] = p.x[rb + red[c]];
                if (MODE == MODE_RHS) x[3 + c] = p.x[rb + p.n_red + red[c]];
                if (p.u_held) uh[c] = p.u_held[ub + red[c]];
            }
        }
    } else if (valid) {
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
        if (p.amp && tp.j == (p.imp_node_b ? p.imp_node_b[beam] - p.off : p.imp_slot)) amp = p.amp[beam];
    }

    // FB: reduced indices and reference of this node, the gain into LDS
    const int fb_n = p.n_red, fb_n2 = 2 * p.n_red, fb_n2p = fb_padded(fb_n2);
    T* const fbx = lds.r1 + 3 * NT;                       // [G][2n padded]  r - x of the stage, per beam of the workgroup
    T* const fbK = fbx + size_t(p.G) * fb_n2p;            // [2n padded][n]  gain, transposed
    T* const fbu = fbK + size_t(fb_n2p) * fb_n;           // [G][FBM_UPAD]   K e of the stage (matrix-core form)
    const bool fb_mfma = FB && fb_on_matrix_cores(p.G, fb_n);
    T fb_af[FBM_MT][FBM_KS];
    T rq[3] = {T(0), T(0), T(0)}, rv[3] = {T(0), T(0), T(0)};
    if (FB) {
        if (valid) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                red[c] = p.red_map[3 * node + c];
                if (red[c] >= 0 && p.fb_ref) {
                    rq[c] = p.fb_ref[size_t(beam) * fb_n2 + red[c]];
                    rv[c] = p.fb_ref[size_t(beam) * fb_n2 + fb_n + red[c]];
                }
            }
        }
        for (int idx = tp.t; idx < fb_n * fb_n2; idx += NT) {
            const int i = idx / fb_n2, k = idx - i * fb_n2;
            fbK[size_t(k) * fb_n + i] = p.fb_gain[idx];
        }
        for (int idx = tp.t; idx < (fb_n2p - fb_n2) * fb_n; idx += NT) fbK[size_t(fb_n2) * fb_n + idx] = T(0);   // zero columns
        for (int idx = tp.t; idx < p.G * (fb_n2p - fb_n2); idx += NT)
            fbx[size_t(idx / (fb_n2p - fb_n2)) * fb_n2p + fb_n2 + idx % (fb_n2p - fb_n2)] = T(0);
        __syncthreads();
        if (fb_mfma) fbm_gain_fragments<T>(fb_af, fbK, fb_n, fb_n2p, tp.lane);
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
            if (p.stage == 3) {
                T fq[3], fv[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) { fq[c] = x[c] + dt6 * acc[c]; fv[c] = x[3 + c] + dt6 * acc[3 + c]; }
                mark_nonfinite<T>(p, beam, fq, fv);
            }
        }
        return;
    }
    if (MODE != MODE_STEP) {
        T a[3];
        if (MODE == MODE_RHS) {   // (the impulse of a stepper's input at t0; amp = 0 without one)
#pragma unroll
            for (int c = 0; c < 3; ++c) uh[c] += (c == p.imp_dof && p.t0 < p.duration) ? amp : T(0);
        }
        stage_accel<T, LV, MODE == MODE_KQ, LEAN>(p, lds, sc, cf, tp, x, x + 3, uh, a);
        if (valid && p.red_map) {
            const size_t rb = size_t(beam) * (MODE == MODE_KQ ? 1 : 2) * size_t(p.n_red);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (red[c] >= 0) {
                    if (MODE == MODE_KQ) p.out[rb + red[c]] = a[c];
                    else { p.out[rb + red[c]] = x[3 + c]; p.out[rb + p.n_red + red[c]] = a[c]; }
                }
        } else if (valid) {
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
        if (p.done_flag) {   // (one workgroup: every thread's stores are fenced to the system, then one thread raises the flag)
            __threadfence_system();
            __syncthreads();
            if (tp.t == 0) __hip_atomic_store(p.done_flag, p.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
            if (FB) {
                // (a beam lives in ONE wave here, whose LDS operations execute in order: the barrier only keeps the
                //  compiler from moving the loads above the stores)
                T* const e = fbx + size_t(valid ? g : 0) * fb_n2p;
                if (valid) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (red[c] >= 0) { e[red[c]] = rq[c] - xs[c]; e[fb_n + red[c]] = rv[c] - xs[3 + c]; }
                }
                T ufb[3];
                fb_feedback<T>(fb_mfma, fb_af, fbx, fbK, fbu, p.G, valid ? g : 0, fb_n, fb_n2p, tp.lane, valid, red, ufb);
#pragma unroll
                for (int c = 0; c < 3; ++c) uadd[c] += ufb[c];
            }
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
        if (p.rec_out && valid && (step + 1) % p.rec_every == 0) {
            const size_t k = size_t((step + 1) / p.rec_every - 1);
            if (p.rec_slot == REC_ALL_SLOTS) {   // whole-state snapshot k
                T* snap = p.rec_out + k * size_t(p.B) * 2 * plane + xoff;
#pragma unroll
                for (int c = 0; c < 3; ++c) { snap[c] = x[c]; snap[plane + c] = x[3 + c]; }
                snap[3] = T(0);
                snap[plane + 3] = T(0);
            } else if (tp.j == p.rec_slot) {
                T val = x[0];
#pragma unroll
                for (int c = 1; c < 6; ++c) val = (c == p.rec_comp) ? x[c] : val;
                p.rec_out[size_t(beam) * p.rec_n + k] = val;
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p.x[xoff + c] = x[c];
            p.x[xoff + plane + c] = x[3 + c];
        }
        mark_nonfinite<T>(p, beam, x, x + 3);
    }
}

}  // namespace crb
