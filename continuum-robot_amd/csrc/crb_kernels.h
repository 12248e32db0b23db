// crb_kernels.h -- gfx950 kernels of the beam stepper (included once by crbeam.hip).
//
// Decomposition: one thread per node ("slot") of a beam, G = threads / n_slots
// beams per workgroup.  A thread keeps its node's state (3 positions, 3 velocities), the RK4
// accumulators and its element/force coefficients in registers for the whole launch; the only
// HBM traffic of crb_step_rk4 is one read and one write of the state per LAUNCH.  Neighbour
// data moves through LDS (SoA, conflict-free 8-byte accesses):
//   q of the left node      -> element force of the element left of the node
//   element force halves    -> nodal internal force (no atomics: each node sums exactly two)
//   gravity per segment     -> index table (reduced-index quirk of gravity_forces.py:104-146)
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
    T* out;                    // rhs / internal force output
    int B, S, G, n_node, off, levels;
    uint32_t flags;
    int imp_slot, imp_dof;
    double duration, t0, dt;
    int n_steps;
    T gx, gy;
};

enum : int { MODE_STEP = 0, MODE_RHS = 1, MODE_KQ = 2 };

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

// One evaluation of a = Minv(-k(q) + f_drag + f_grav + u) for this thread's node.
// Returns k(q) in `a` (no solve) when KQ_ONLY.
template <typename T, bool KQ_ONLY>
__device__ __forceinline__ void stage_accel(const KParams<T>& p, const Lds<T>& lds, const SlotConst<T>& sc, bool active,
                                            int t, int j, int base, const T q[3], const T v[3], const T uadd[3], T a[3]) {
    const int NT = lds.NT;
    const bool drag_on = (p.flags & 1u) != 0, grav_on = (p.flags & 2u) != 0, corrected = (p.flags & 4u) != 0;

    // -- 1. publish q, fetch the left node's q
    if (active) {
        lds.q[t] = q[0];
        lds.q[NT + t] = q[1];
        lds.q[2 * NT + t] = q[2];
    }
    __syncthreads();
    T ql[3] = {T(0), T(0), T(0)};
    if (active && j > 0) {
        ql[0] = lds.q[t - 1];
        ql[1] = lds.q[NT + t - 1];
        ql[2] = lds.q[2 * NT + t - 1];
    }
    T fl[3], fr[3];
    elem_force<T>(sc.elem, ql, q, corrected, fl, fr);

    T gseg[2] = {T(0), T(0)};
    if (!KQ_ONLY && grav_on && active && sc.half_mass != T(0)) {
        const int ia = sc.grav.phiA, ib = sc.grav.phiB;
        T phi = T(0);
        if (ia >= 0) phi = lds.q[(ia & 3) * NT + base + (ia >> 2)];
        if (ib >= 0) phi = T(0.5) * (phi + lds.q[(ib & 3) * NT + base + (ib >> 2)]);
        gravity_segment<T>(phi, p.gx, p.gy, sc.half_mass, gseg);
    }

    // -- 2. publish the left-node half of the element force (+ segment gravity)
    if (active) {
        lds.f[t] = fl[0];
        lds.f[NT + t] = fl[1];
        lds.f[2 * NT + t] = fl[2];
        if (!KQ_ONLY && grav_on) {
            lds.g[t] = gseg[0];
            lds.g[NT + t] = gseg[1];
        }
    }
    __syncthreads();
    T r[3];
    {
        const bool has_right = active && (j + 1 < p.S);
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = fr[c] + (has_right ? lds.f[c * NT + t + 1] : T(0));
    }
    if (KQ_ONLY) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a[c] = r[c] * sc.mask[c];
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] = uadd[c] - r[c];
    if (drag_on) r[1] += drag_force<T>(sc.drag, v[1]);
    if (grav_on && active) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int sa = sc.grav.segA[c], sb = sc.grav.segB[c];
            const int off = sc.grav.comp[c] * NT + base;
            if (sa >= 0) r[c] += lds.g[off + sa];
            if (sb >= 0) r[c] += lds.g[off + sb];
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) r[c] *= sc.mask[c];

    // -- 3. Minv by parallel cyclic reduction with precomputed multipliers
    const T* cf_base = p.pcr_levels + size_t(j) * PCR_LEVEL_VALS;
    for (int lvl = 0; lvl < p.levels; ++lvl) {
        const int s = 1 << lvl;
        T* buf = (lvl & 1) ? lds.r1 : lds.r0;
        T cf[PCR_LEVEL_VALS];
        if (active) {
            const T* src = cf_base + size_t(lvl) * size_t(p.S) * PCR_LEVEL_VALS;
#pragma unroll
            for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf[k] = src[k];
            buf[t] = r[0];
            buf[NT + t] = r[1];
            buf[2 * NT + t] = r[2];
        } else {
#pragma unroll
            for (int k = 0; k < PCR_LEVEL_VALS; ++k) cf[k] = T(0);
        }
        __syncthreads();
        T rlo[3] = {T(0), T(0), T(0)}, rhi[3] = {T(0), T(0), T(0)};
        if (active && j - s >= 0) {
            rlo[0] = buf[t - s];
            rlo[1] = buf[NT + t - s];
            rlo[2] = buf[2 * NT + t - s];
        }
        if (active && j + s < p.S) {
            rhi[0] = buf[t + s];
            rhi[1] = buf[NT + t + s];
            rhi[2] = buf[2 * NT + t + s];
        }
        pcr_apply_level<T>(cf, rlo, rhi, r);
    }
    {
        T cf[PCR_FINAL_VALS];
#pragma unroll
        for (int k = 0; k < PCR_FINAL_VALS; ++k) cf[k] = active ? p.pcr_final[size_t(j) * PCR_FINAL_VALS + k] : T(0);
        pcr_apply_final<T>(cf, r, a);
    }
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
template <typename T, int MODE, int MAXT>
__global__ void __launch_bounds__(MAXT) crb_beam_kernel(const KParams<T> p) {
    const int NT = blockDim.x;
    const Lds<T> lds = carve_lds<T>(NT);
    const int t = threadIdx.x;
    const int g = t / p.S, j = t - g * p.S;
    const int beam = blockIdx.x * p.G + g;
    const bool active = (g < p.G) && (beam < p.B);
    const int base = g * p.S;

    SlotConst<T> sc;
    if (active) {
        sc = p.slot[j];
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

    // this thread's node record
    const size_t node = size_t(j + p.off);
    const size_t plane = size_t(p.n_node) * 4;
    const size_t xoff = active ? (size_t(beam) * 2 * plane + node * 4) : 0;
    T x[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    T uh[3] = {T(0), T(0), T(0)};
    T amp[3] = {T(0), T(0), T(0)};
    if (active) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            x[c] = p.x[xoff + c] * sc.mask[c];
            x[3 + c] = p.x[xoff + plane + c] * sc.mask[c];
        }
        if (p.u_held) {
            const size_t uoff = size_t(beam) * plane + node * 4;
#pragma unroll
            for (int c = 0; c < 3; ++c) uh[c] = p.u_held[uoff + c];
        }
        if (p.amp && j == p.imp_slot) {
            const T av = p.amp[beam];
#pragma unroll
            for (int c = 0; c < 3; ++c) amp[c] = (c == p.imp_dof) ? av : T(0);
        }
    }

    if (MODE != MODE_STEP) {
        T a[3];
        stage_accel<T, MODE == MODE_KQ>(p, lds, sc, active, t, j, base, x, x + 3, uh, a);
        if (active) {
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
            const bool on = ts < p.duration;
            T uadd[3], a[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) uadd[c] = uh[c] + (on ? amp[c] : T(0));
            stage_accel<T, false>(p, lds, sc, active, t, j, base, xs, xs + 3, uadd, a);
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
    }
    if (active) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p.x[xoff + c] = x[c];
            p.x[xoff + plane + c] = x[3 + c];
        }
    }
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
