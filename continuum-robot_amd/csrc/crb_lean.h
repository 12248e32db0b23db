// crb_lean.h -- the lean kernels: fused multi-step RK4 stepper (the headline path), lean RHS, one-stage kernel
// of the stage-split stepper.  Register-resident solve tables, merged exchange rounds, DPP lane shifts.
#pragma once
#include <hip/hip_runtime.h>

#include "crb_generic.h"

namespace crb {

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
struct LeanRec {                    // fp32: [qn0 qn1 qn2 - | p0 p1 p2 - | fl0 fl1 fl2 -]   (fp64: columns, see the stepper)
    static constexpr int N = sizeof(T) == 8 ? 10 : 12;   // 80 B / 48 B: conflict-free 16-byte accesses
    static constexpr int V = 16 / sizeof(T);              // values per 16-byte LDS word
};
template <typename T>
__host__ __device__ constexpr size_t lean_lds_bytes(int NT, int lognw) {
    // round A records (+1 all-zero "no neighbour" record; double-buffered when round A is the only
    // barrier round) + SoA buffers (+1 zero column) of the cross-wave levels 1..lognw-1
    // (fp32: the cross-wave levels move one 16-byte record [r0 r1 r2 -] per thread instead of three 4-byte columns)
    return sizeof(T) * (size_t(NT + 1) * LeanRec<T>::N * (lognw == 1 ? 2 : 1) +
                        (sizeof(T) == 4 ? 4 : 3) * size_t(NT + 1) * size_t(lognw > 1 ? lognw - 1 : 0));
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
constexpr int LEAN_DPP_MAX = 4;   // lane shifts up to this distance are chained DPP moves, larger ones ds_bpermute
// wave priorities per phase of a stage (s_setprio; -1 = leave unchanged): element force 0, exchanges and the cross-wave
// level 2, in-wave levels 1 -- the two workgroups of a CU then interleave force arithmetic with the other's exchange
// latency (+10 %).  fp32 plans (four waves per SIMD, LDS-issue-limited) want the priority to RISE through the stage and
// drop at its end (config 4: +3.7 % over the fp64 values; the fp64 stepper is indifferent to those or slower).
constexpr int P_FORCE = 0, P_XCHG = 2, P_L1 = -1, P_TAIL = 1, P_FIN = -1;
constexpr int P32_L1 = 3, P32_TAIL = 3, P32_FIN = 0;
#define CRB_SETPRIO(v) do { if ((v) >= 0) __builtin_amdgcn_s_setprio((v) < 0 ? 0 : (v)); } while (0)
template <typename T, int D>
__device__ __forceinline__ T lane_lower(T x, int lane) {
    if (D <= LEAN_DPP_MAX) {
#pragma unroll
        for (int i = 0; i < D; ++i) x = dpp_from_lower(x);
        return x;
    }
    return __shfl(x, lane - D, 64);
}
template <typename T, int D>
__device__ __forceinline__ T lane_higher(T x, int lane) {
    if (D <= LEAN_DPP_MAX) {
#pragma unroll
        for (int i = 0; i < D; ++i) x = dpp_from_higher(x);
        return x;
    }
    return __shfl(x, lane + D, 64);
}

// Reduction levels 1..LV-1 and the final block inverse of the lean kernels, given r after level 0.
// Levels whose stride stays inside the workgroup's waves-per-beam interleave (l < LOGNW) go through LDS
// columns + a barrier, the others are in-wave lane shifts.  A missing neighbour contributes through a
// multiplier that is exactly 0, so whatever finite value the shift returns there is harmless.
// ISOLATE (packed beams: several beams share the wave): what a lane shift drags across a beam boundary is replaced
// by 0 with a select -- a multiplier of exactly 0 would turn a neighbouring beam's Inf/NaN into NaN here, and
// the reference's beams are independent (a diverged beam must not take its wave-mates with it).
// RECB (fp32 stepper): a cross-wave level is ONE 16-byte store and two 16-byte loads per thread instead of 3 + 6
// 4-byte ones -- the fp32 stepper runs four waves per SIMD and is limited by LDS instruction issue, not by bytes.
template <typename T, int LV, int LOGNW, bool ISOLATE = false, bool RECB = false>
__device__ __forceinline__ void lean_reduce_tail(const SolveCoef<T, LV>& cf, T* ldsB, int t, int lane, int j, int S,
                                                 bool valid, T r[3], T a[3]) {
    constexpr int NW = 1 << LOGNW, NT = 64 << LOGNW, NULLT = NT;
    auto thread_of = [](int jj) { return ((jj & (NW - 1)) << 6) | (jj >> LOGNW); };
    T rlo[3], rhi[3];
#pragma unroll
    for (int l = 1; l < LV; ++l) {
        if (l < LOGNW) {  // another wave holds the neighbour: LDS + barrier
            if (l == 1) CRB_SETPRIO(sizeof(T) == 4 ? P32_L1 : P_L1);
            const int st = 1 << l;
            const int tl = (valid && j - st >= 0) ? thread_of(j - st) : NULLT;
            const int th = (valid && j + st < S) ? thread_of(j + st) : NULLT;
            if constexpr (RECB) {
                typedef T rec4 __attribute__((ext_vector_type(4)));
                rec4* buf = reinterpret_cast<rec4*>(ldsB) + size_t(l - 1) * (NT + 1);
                buf[t] = rec4{r[0], r[1], r[2], T(0)};
                __syncthreads();
                const rec4 lo = buf[tl], hi = buf[th];
#pragma unroll
                for (int c = 0; c < 3; ++c) { rlo[c] = lo[c]; rhi[c] = hi[c]; }
            } else {
                T* buf = ldsB + size_t(l - 1) * 3 * (NT + 1);
                buf[t] = r[0]; buf[(NT + 1) + t] = r[1]; buf[2 * (NT + 1) + t] = r[2];
                __syncthreads();
#pragma unroll
                for (int c = 0; c < 3; ++c) { rlo[c] = buf[c * (NT + 1) + tl]; rhi[c] = buf[c * (NT + 1) + th]; }
            }
        } else {
            if (l == LOGNW) CRB_SETPRIO(sizeof(T) == 4 ? P32_TAIL : P_TAIL);
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
            if (ISOLATE) {
                const int st = 1 << l;
                const bool lo_ok = valid && j - st >= 0, hi_ok = valid && j + st < S;
#pragma unroll
                for (int c = 0; c < 3; ++c) { rlo[c] = lo_ok ? rlo[c] : T(0); rhi[c] = hi_ok ? rhi[c] : T(0); }
            }
        }
        pcr_apply_level<T>(cf.lv[l], rlo, rhi, r);
    }
    CRB_SETPRIO(sizeof(T) == 4 ? P32_FIN : P_FIN);
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
            r[c] = pp[c] - lane_higher<T, 1>(fl[c], lane);   // (one wave: r first, then ITS neighbours -- see the stepper)
            rlo[c] = lane_lower<T, 1>(r[c], lane);
            rhi[c] = lane_higher<T, 1>(r[c], lane);
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

// Waves per SIMD the fp64 lean kernels are register-allocated for.  Two, except single-wave beams with gravity and 5 or
// more reduction levels: their working set does not fit 256 registers (18 .. 140 spilled VGPRs), and a beam that lives in
// ONE wave gains nothing from a second resident wave until the ensemble exceeds 1024 beams -- measured, spill-free at one
// wave per SIMD against two with spills: 1024 x 64 + gravity (BASELINE config 2) 3.27 against 4.13 us per step, 4096 x 64
// 11.4 against 11.6.  Beams of several waves keep two (their cross-wave exchanges need the second workgroup to hide the
// barrier latency: 4096 x 128 / x 256 + gravity take 32 / 70 us per step at one wave per SIMD against 21 / 39 at two, spills
// included).
__host__ __device__ constexpr int lean_minw_f64(int lv, int lognw, bool grav) { return (grav && lv >= 5 && lognw == 0) ? 1 : 2; }
// EM (EM_*): the element kind when the whole topology has one; the force evaluation is then straight-line
// code that the scheduler interleaves with the tail of the previous stage's reduction (measured +8 %
// over the per-lane branch of EM_MIXED; a wave-uniform run-time branch does not get it).
// HELD: a per-node input force held over the launch (zero-order-hold control, `u` of dynamic_system(t, x, u)
// as an array): three more registers and three more additions per stage, so it is its own instantiation.
// PACK (one-wave form only): beams with fewer than 64 slots, G = 64/S of them per wave (lane = g*S + j).  All
// exchanges stay DPP lane shifts; whatever a shift drags across a beam boundary is replaced by 0 with a SELECT (in
// round A and in every reduction level, lean_reduce_tail<..., ISOLATE>): a diverged wave-mate's Inf / NaN must not
// reach its neighbours through a 0 * NaN (the reference's beams are independent).
// FB (packed one-wave form): the state feedback u = K (r - x) inside every stage, as in crb_beam_kernel<..., FB> (crb_generic.h: gain
// in LDS, the product K e on the matrix cores for gains of 21 .. 32 rows) but around THIS kernel's right-hand side -- the
// general kernel's costs 4000 cycles per stage for a 10-element beam, this one's 1900.  One wave per SIMD (the gain's
// fragments take 64 more registers).
template <typename T, int LV, int LOGNW, bool GRAV, int EM, bool HELD = false, bool PACK = false, bool FB = false>
// fp64: 2 waves/SIMD, 256 VGPRs hold the multipliers.  fp32: the headline shape (<= 4 levels, no gravity, no held
// input) fits 4 waves/SIMD (128 VGPRs; three 8-byte addresses spill, outside the step loop: config 4 runs 8.1e10
// element-steps/s at 4 waves against 6.6e10 at 3), the other fp32 instantiations keep 3 waves/SIMD (168 VGPRs)
__global__ void __launch_bounds__(64 << LOGNW, FB ? 1 : ((sizeof(T) == 4 && LOGNW <= 2) ? ((LV <= 4 && !GRAV && !HELD) ? 4 : 3) : lean_minw_f64(LV, LOGNW, GRAV)))
crb_step_lean_kernel(const KParams<T> p_formal) {
    // The launch parameters are read through the kernarg pointer, which is "laundered" (CRB_FRESH) at the top of
    // every beam and again after the step loop: what the beam prologue / epilogue need (pointers, strides, sizes) is
    // then re-read from the kernarg segment by a few scalar loads instead of staying live in SGPRs across the step
    // loop.  That loop keeps ~36 SGPRs of fp64 polynomial constants; with the walk-over-beams loop around it the
    // scalar file overflowed (28 spills) and the scheduler fell back to a serialised LDS schedule: 28.4 -> 32.4 us/step.
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) KParams<T>* KP;
    KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();   // the explicit arguments start at offset 0 of the segment
    (void)p_formal;
#define CRB_FRESH(ptr) asm volatile("" : "+s"(ptr))
#define CRB_PARAMS(ptr) (*(const KParams<T>*)(ptr))
#else   // (host pass: only the stub is emitted; keep the body well-formed)
    const KParams<T>* kp = &p_formal;
#define CRB_FRESH(ptr) (void)(ptr)
#define CRB_PARAMS(ptr) (*(ptr))
#endif
    KParams<T> p = CRB_PARAMS(kp);
    static_assert(!PACK || LOGNW == 0, "packed beams live inside one wave");
    static_assert(!FB || (PACK && !HELD), "the feedback form is the packed one-wave stepper");
    static_assert(LV >= 1, "lean stepper needs at least one reduction level");
    constexpr int NW = 1 << LOGNW, NT = 64 << LOGNW, RN = LeanRec<T>::N;
    constexpr int NULLT = NT;  // index of the all-zero record / column: "no neighbour"
    // fp64 round A is laid out as 9 columns [qn0..2 p0..2 fl0..2][NT+1] moved by 8-byte accesses (a 16-byte LDS
    // store costs 13 cycles of the store path against 2 x 6 for two 8-byte ones); fp32 keeps 16-byte records
    constexpr bool SOA = sizeof(T) == 8;
    constexpr bool RECB = sizeof(T) == 4;   // cross-wave levels as 16-byte records (fp64: 32-byte records measured 4 % slower)
    auto recA = [](T* base, int th, int k) -> T& { return SOA ? base[size_t(k) * (NT + 1) + th] : base[size_t(th) * RN + k]; };
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    T* const ldsA = reinterpret_cast<T*>(crb_smem);
    T* const ldsB = ldsA + size_t(NT + 1) * RN * (LOGNW == 1 ? 2 : 1);  // [level-1][3][NT+1]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int S = p.S;
    const int pg = PACK ? lane / S : 0;                         // beam of this lane inside the wave
    const int j = PACK ? lane - pg * S : ((lane << LOGNW) | wave);
    const bool has_slot = PACK ? (pg < p.G) : (j < S);          // this thread carries a slot (of some beam)
    auto thread_of = [](int jj) { return ((jj & (NW - 1)) << 6) | (jj >> LOGNW); };
    // LDS positions of the stride-1 / stride-2 neighbours (NULLT outside the beam)
    const int t_l1 = (has_slot && j >= 1) ? thread_of(j - 1) : NULLT;
    const int t_r1 = (has_slot && j + 1 < S) ? thread_of(j + 1) : NULLT;
    if (LOGNW > 0 && t == 0) {  // zero the "no neighbour" slots once
#pragma unroll
        for (int k = 0; k < RN; ++k) {
            recA(ldsA, NULLT, k) = T(0);
            if (LOGNW == 1) recA(ldsA + size_t(NT + 1) * RN, NULLT, k) = T(0);
        }
#pragma unroll
        for (int l = 1; l < LOGNW; ++l)
#pragma unroll
            for (int c = 0; c < (RECB ? 4 : 3); ++c)
                ldsB[RECB ? (size_t(l - 1) * (NT + 1) + NULLT) * 4 + c : (size_t(l - 1) * 3 + c) * (NT + 1) + NULLT] = T(0);
    }

    // ---- per-thread constants: element pack, drag factor, the thread's rows of the solve tables.  Plans whose
    // beams share one table set load them ONCE per workgroup; the workgroup then walks over its beams (the grid is
    // sized to what is resident, crb_lean.hip): at 4096 x 256 the table prologue is 112 KB per workgroup from L2,
    // 460 MB per launch if every beam's workgroup repeats it -- 43 us, the whole fixed cost of a launch.
    const bool shared_tables = p.slot_stride == 0 && p.lv_stride == 0 && p.fin_stride == 0;
    ElemCoef<T> ec;
    T dragc = T(0);
    SolveCoef<T, LV> cf;
    // GRAV: gravity on the canonical cantilever (only node 0 constrained), where the reference's reduced-index
    // addressing (gravity_forces.py:104-146) is nearest-neighbour: segment j averages the rotations of slots j
    // and j+1 (slot j alone at the tip) and loads slots j and j+1.  A thread evaluates segment j AND segment
    // j-1 itself (it knows phi of slots j-1, j, j+1), so gravity needs no exchange of its own.
    T hm_own = T(0), hm_left = T(0);
    auto load_tables = [&](int beam) {
        if (has_slot) {
            const SlotConst<T>* st = p.slot + size_t(beam) * p.slot_stride;
            const SlotConst<T>& sc = st[j];
            ec = sc.elem;
            dragc = (p.flags & 1u) ? sc.drag : T(0);
            if (GRAV) { hm_own = sc.half_mass; hm_left = j >= 1 ? st[j - 1].half_mass : T(0); }
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
    // FB: the gain (transposed) and the stage's error vectors of the wave's beams live in LDS (the one-wave form uses none otherwise)
    const int fb_n = FB ? p.n_red : 0, fb_n2 = 2 * fb_n, fb_n2p = fb_padded(fb_n2), fb_G = FB ? p.G : 0;
    T* const fbx = reinterpret_cast<T*>(crb_smem);        // [G][2n padded]
    T* const fbK = fbx + size_t(fb_G) * fb_n2p;           // [2n padded][n]
    T* const fbu = fbK + size_t(fb_n2p) * fb_n;           // [G][FBM_UPAD]
    const bool fb_mfma = FB && fb_on_matrix_cores(fb_G, fb_n);
    T fb_af[FBM_MT][FBM_KS];
    if (FB) {
        for (int idx = t; idx < fb_n * fb_n2; idx += NT) {
            const int i = idx / fb_n2, k = idx - i * fb_n2;
            fbK[size_t(k) * fb_n + i] = p.fb_gain[idx];
        }
        for (int idx = t; idx < (fb_n2p - fb_n2) * fb_n; idx += NT) fbK[size_t(fb_n2) * fb_n + idx] = T(0);
        for (int idx = t; idx < fb_G * (fb_n2p - fb_n2); idx += NT)
            fbx[size_t(idx / (fb_n2p - fb_n2)) * fb_n2p + fb_n2 + idx % (fb_n2p - fb_n2)] = T(0);
        __syncthreads();
        if (fb_mfma) fbm_gain_fragments<T>(fb_af, fbK, fb_n, fb_n2p, lane);
    }
    const bool corrected = (p.flags & 4u) != 0;
    const T dt = T(p.dt), hdt = T(0.5 * p.dt), dt6 = T(p.dt / 6.0);
    const size_t plane = size_t(p.n_node) * 4;
    const int n_groups = PACK ? (p.B + p.G - 1) / p.G : p.B;

    // Beam switch without a dependent global load (plans with one table set, one beam per workgroup at a time): while the current
    // beam steps, the NEXT beam's state records and impulse amplitude are already in flight, and the boundary-condition masks
    // are three bits of one register instead of three loads per beam -- in-kernel stamps put those loads at 1.5 of the 2.2 us a
    // beam switch takes (kernarg re-read 0.2, first exchange 0.15, epilogue 0.35).
    constexpr bool PREFETCH = !PACK && !FB;
    T nq[3] = {T(0), T(0), T(0)}, nv[3] = {T(0), T(0), T(0)}, namp = T(0);
    bool have_next = false;
    int mask_bits = 7;
    if (PREFETCH && shared_tables && has_slot) {
        const SlotConst<T>& sc0 = p.slot[j];
        mask_bits = (sc0.mask[0] != T(0) ? 1 : 0) | (sc0.mask[1] != T(0) ? 2 : 0) | (sc0.mask[2] != T(0) ? 4 : 0);
    }
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    CRB_FRESH(kp);
    p = CRB_PARAMS(kp);
    const int beam = PACK ? grp * p.G + pg : grp;
    const bool valid = has_slot && (!PACK || beam < p.B);
    const bool has_left = valid && j >= 1, has_right = valid && j + 1 < S;
    if (!shared_tables) load_tables(valid ? beam : 0);
    T phiR = T(0);
    T gx = p.gx, gy = p.gy;
    if (GRAV && p.gvec && valid) { gx = p.gvec[2 * size_t(beam)]; gy = p.gvec[2 * size_t(beam) + 1]; }   // per-beam ForceParams

    // ---- state
    const size_t node = size_t(valid ? j + p.off : 0);
    const size_t xoff = size_t(valid ? beam : 0) * 2 * plane + node * 4;
    T xq[3] = {T(0), T(0), T(0)}, xv[3] = {T(0), T(0), T(0)};
    T amp = T(0);
    if (PREFETCH && have_next) {                  // (wave-uniform: a real branch, the loads below are not issued)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool on = (mask_bits >> c) & 1;
            xq[c] = on ? nq[c] : T(0);
            xv[c] = on ? nv[c] : T(0);
        }
        amp = namp;
    } else if (valid) {
        const SlotConst<T>& sc = p.slot[size_t(beam) * p.slot_stride + j];   // (per-beam masks are not kept in registers)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xq[c] = p.x[xoff + c] * sc.mask[c];
            xv[c] = p.x[xoff + plane + c] * sc.mask[c];
        }
        if (p.amp && j == (p.imp_node_b ? p.imp_node_b[beam] - p.off : p.imp_slot)) amp = p.amp[beam];
    }
    have_next = false;
    if (PREFETCH && shared_tables && grp + int(gridDim.x) < n_groups) {   // (wave-uniform)
        have_next = true;
        if (has_slot) {
            const int nb = grp + int(gridDim.x);
            const size_t noff = size_t(nb) * 2 * plane + size_t(j + p.off) * 4;
#pragma unroll
            for (int c = 0; c < 3; ++c) { nq[c] = p.x[noff + c]; nv[c] = p.x[noff + plane + c]; }
            namp = T(0);
            if (p.amp && j == (p.imp_node_b ? p.imp_node_b[nb] - p.off : p.imp_slot)) namp = p.amp[nb];
        }
    }
    T uh[3] = {T(0), T(0), T(0)};
    if (HELD && valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) uh[c] = p.u_held[size_t(beam) * plane + node * 4 + c];
    }
    int red[3] = {-1, -1, -1};            // FB: reduced indices and reference of this node
    T rq[3] = {T(0), T(0), T(0)}, rv[3] = {T(0), T(0), T(0)};
    if (FB && valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            red[c] = p.red_map[3 * node + c];
            if (red[c] >= 0 && p.fb_ref) {
                rq[c] = p.fb_ref[size_t(beam) * fb_n2 + red[c]];
                rv[c] = p.fb_ref[size_t(beam) * fb_n2 + fb_n + red[c]];
            }
        }
    }

    // ---- left neighbour's q for the very first stage
    T qL[3];
    if (LOGNW == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            qL[c] = lane_lower<T, 1>(xq[c], lane);  // lane 0 reads 0 = clamped / absent root
            if (PACK) qL[c] = has_left ? qL[c] : T(0);   // (a select: the neighbouring beam may hold Inf/NaN)
        }
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
            CRB_SETPRIO(P_FORCE);
            if (EM == EM_NONLINEAR) elem_force_nonlinear<T>(ec.c, qL, sq, false, fl, fr);
            else if (EM == EM_LINEAR) elem_force_linear<T>(ec.c, qL, sq, fl, fr);
            else elem_force<T>(ec, qL, sq, corrected, fl, fr);
            CRB_SETPRIO(P_XCHG);
            T pp[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                pp[c] = ((imp_on && c == p.imp_dof) ? T(1) : T(0)) * amp - fr[c];
                if (HELD) pp[c] += uh[c];
            }
            if (FB) {   // u = K (r - x) of the stage state (lqr_control.py:95-111)
                T* const e = fbx + size_t(valid ? pg : 0) * fb_n2p;
                if (valid) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (red[c] >= 0) { e[red[c]] = rq[c] - sq[c]; e[fb_n + red[c]] = rv[c] - sv[c]; }
                }
                T ufb[3];
                fb_feedback<T>(fb_mfma, fb_af, fbx, fbK, fbu, fb_G, valid ? pg : 0, fb_n, fb_n2p, lane, valid, red, ufb);
#pragma unroll
                for (int c = 0; c < 3; ++c) pp[c] += ufb[c];
            }
            pp[1] += drag_force<T>(dragc, sv[1]);
            if (GRAV) {
                T g_own[2], g_left[2];
                gravity_segment<T>(has_right ? T(0.5) * (sq[2] + phiR) : sq[2], gx, gy, hm_own, g_own);
                if (LOGNW == 0) {  // segment j-1 IS the left lane's own segment: take its result (bit-identical), one sincos less
                    g_left[0] = lane_lower<T, 1>(g_own[0], lane);
                    g_left[1] = lane_lower<T, 1>(g_own[1], lane);
                    if (PACK) { g_left[0] = has_left ? g_left[0] : T(0); g_left[1] = has_left ? g_left[1] : T(0); }
                } else {
                    gravity_segment<T>(T(0.5) * (qL[2] + sq[2]), gx, gy, hm_left, g_left);
                }
                pp[0] += g_own[0] + g_left[0];
                pp[1] += g_own[1] + g_left[1];
            }

            // -- round A: publish {qn, p, fl}; rebuild r of this node and of both stride-1 neighbours.
            // Outside the beam a neighbour reads as zeros (DPP edge / all-zero LDS record).
            T r[3], rlo[3], rhi[3];
            if (LOGNW == 0) {
                // one wave: no barrier to save by merging, so r is formed first and ITS neighbours are shifted in
                // (4 lane shifts per component instead of the merged round's 6, and 3 subtractions instead of 9;
                //  r of a neighbour is the same subtraction of the same operands, evaluated in the neighbour's lane)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // (shuffles stay outside any condition: every lane must take part.  Lanes past
                    //  the last slot are padding threads whose p and fl are 0, wave edges shift in 0.)
                    if (GRAV && c == 2) phiR = lane_higher<T, 1>(qn[2], lane);
                    qL[c] = lane_lower<T, 1>(qn[c], lane);
                    const T fl_r1 = lane_higher<T, 1>(fl[c], lane);
                    if (PACK) {   // selects, not 0/1 factors: a neighbouring beam's Inf/NaN must not leak (0 * NaN = NaN)
                        qL[c] = has_left ? qL[c] : T(0);
                        r[c] = pp[c] - (has_right ? fl_r1 : T(0));
                    } else {
                        r[c] = pp[c] - fl_r1;
                    }
                    rlo[c] = lane_lower<T, 1>(r[c], lane);
                    rhi[c] = lane_higher<T, 1>(r[c], lane);
                    if (PACK) { rlo[c] = has_left ? rlo[c] : T(0); rhi[c] = has_right ? rhi[c] : T(0); }
                }
            } else {
                T* bufA = ldsA + ((LOGNW == 1 && (s & 1)) ? size_t(NT + 1) * RN : 0);
                if (SOA) {
                    // r first (one exchange of {qn, fl}), then ITS stride-1 neighbours (a second one): the same 9 stores +
                    // 12 loads as the merged round {qn, p, fl}, 6 subtractions less, one barrier more -- the fp64 stepper is
                    // bound by vector-ALU issue, not by its barriers: 27.72 -> 27.18 us per step at 4096 x 256
                    recA(bufA, t, 0) = qn[0]; recA(bufA, t, 1) = qn[1]; recA(bufA, t, 2) = qn[2];
                    recA(bufA, t, 6) = fl[0]; recA(bufA, t, 7) = fl[1]; recA(bufA, t, 8) = fl[2];
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < 3; ++c) r[c] = pp[c] - recA(bufA, t_r1, 6 + c);
                    if (GRAV) phiR = recA(bufA, t_r1, 2);
#pragma unroll
                    for (int c = 0; c < 3; ++c) qL[c] = recA(bufA, t_l1, c);
                    recA(bufA, t, 3) = r[0]; recA(bufA, t, 4) = r[1]; recA(bufA, t, 5) = r[2];
                    __syncthreads();
#pragma unroll
                    for (int c = 0; c < 3; ++c) { rlo[c] = recA(bufA, t_l1, 3 + c); rhi[c] = recA(bufA, t_r1, 3 + c); }
                } else {
                    // fp32 records, r first: [qn0 qn1 qn2 - | fl0 fl1 fl2 -] out, the left neighbour's word 0 and the right
                    // one's word 1 (and its phi) in; then [r0 r1 r2 -] (word 2) out and both neighbours' in: 3 stores + 4 loads
                    typedef T rec4 __attribute__((ext_vector_type(4)));
                    rec4* recs = reinterpret_cast<rec4*>(bufA);
                    constexpr int W = RN / 4;   // 16-byte words per record
                    recs[size_t(t) * W + 0] = rec4{qn[0], qn[1], qn[2], T(0)};
                    recs[size_t(t) * W + 1] = rec4{fl[0], fl[1], fl[2], T(0)};
                    __syncthreads();
                    const rec4 lq = recs[size_t(t_l1) * W + 0], rf = recs[size_t(t_r1) * W + 1];
                    if (GRAV) phiR = recs[size_t(t_r1) * W + 0][2];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { qL[c] = lq[c]; r[c] = pp[c] - rf[c]; }
                    recs[size_t(t) * W + 2] = rec4{r[0], r[1], r[2], T(0)};
                    __syncthreads();
                    const rec4 rl = recs[size_t(t_l1) * W + 2], rr = recs[size_t(t_r1) * W + 2];
#pragma unroll
                    for (int c = 0; c < 3; ++c) { rlo[c] = rl[c]; rhi[c] = rr[c]; }
                }
            }
            pcr_apply_level<T>(cf.lv[0], rlo, rhi, r);

            // -- remaining reduction levels and the final block inverse
            T a[3];
            lean_reduce_tail<T, LV, LOGNW, PACK, RECB>(cf, ldsB, t, lane, j, S, valid, r, a);

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
        if (p.rec_out && valid && (step + 1) % p.rec_every == 0) {
            const size_t k = size_t((step + 1) / p.rec_every - 1);
            if (p.rec_slot == REC_ALL_SLOTS) {   // whole-state snapshot k: every thread stores its node's two records
                typedef T rec4 __attribute__((ext_vector_type(4)));
                T* snap = p.rec_out + k * size_t(p.B) * 2 * plane + xoff;
                *reinterpret_cast<rec4*>(snap) = rec4{xq[0], xq[1], xq[2], T(0)};
                *reinterpret_cast<rec4*>(snap + plane) = rec4{xv[0], xv[1], xv[2], T(0)};
            } else if (j == p.rec_slot) {
                T val = xq[0];
#pragma unroll
                for (int c = 1; c < 6; ++c) val = (c == p.rec_comp) ? (c < 3 ? xq[c] : xv[c - 3]) : val;
                p.rec_out[size_t(beam) * p.rec_n + k] = val;
            }
        }
    }
    CRB_FRESH(kp);
    p = CRB_PARAMS(kp);
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p.x[xoff + c] = xq[c];
            p.x[xoff + plane + c] = xv[c];
        }
        mark_nonfinite<T>(p, beam, xq, xv);
    }
    // (the next beam's first LDS writes come after barriers that follow this beam's last LDS reads: LOGNW >= 2 the
    //  level-1 exchange, LOGNW == 1 the alternating round-A buffers + the two barriers of the first-stage q exchange)
    }
#undef CRB_FRESH
#undef CRB_PARAMS
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
// Beams of two / four waves move their node records through LDS in memory order (see the kernel); loading the NEXT beam's
// records one beam ahead was measured and is not taken (21.6 against 21.4 us at 2048 x 128, with 56 more registers).
constexpr int STAGE_IO_STREAMS = 7;   // xs q / v, x q / v, acc q / v, u
template <typename T>
__host__ __device__ constexpr size_t stage_lean_lds_bytes(int NT, int lognw) {
    // exchange columns (16-byte multiple) + the record staging of the transposed I/O (beams of more than one wave)
    const size_t cols = sizeof(T) * (size_t(NT + 1) * 6 * (lognw == 1 ? 2 : 1) + 3 * size_t(NT + 1) * size_t(lognw > 1 ? lognw - 1 : 0));
    const size_t io = (lognw > 0 && lognw <= 2) ? size_t(STAGE_IO_STREAMS) * size_t(NT + 1) * 4 * sizeof(T) : 0;   // (eight waves: no room)
    return ((cols + 31) / 32) * 32 + io;
}
template <typename T, int LV, int LOGNW, bool GRAV, int EM>
__global__ void __launch_bounds__(64 << LOGNW, (sizeof(T) == 4 && LOGNW <= 2) ? 3 : lean_minw_f64(LV, LOGNW, GRAV)) crb_stage_lean_kernel(const KParams<T> p) {
    static_assert(LV >= 1, "lean stage kernel needs at least one reduction level");
    constexpr int NW = 1 << LOGNW, NT = 64 << LOGNW, NULLT = NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    T* const ldsA = reinterpret_cast<T*>(crb_smem);                         // [1 or 2][6][NT+1]: p0..2, fl0..2
    T* const ldsB = ldsA + size_t(NT + 1) * 6 * (LOGNW == 1 ? 2 : 1);       // [level-1][3][NT+1]
    // Transposed record I/O (beams of more than one wave).  A wave's slots are every NW-th node, so a thread that loads /
    // stores ITS OWN slot's 32-byte records makes every 16-byte half a memory request of its own (measured at equal
    // node count: 18 us per launch with one wave per beam, 24 with two, 31 with four).  Instead thread t moves the records
    // of node t -- consecutive lanes, consecutive records -- and the records change hands in LDS: staged at the position
    // of the OWNER thread (so that the owner and its stride-1 neighbours read consecutive positions), which also hands
    // the left neighbour's q and the right neighbour's rotation over without loads of their own.
    constexpr bool IO = LOGNW > 0 && LOGNW <= 2;
    typedef T rec4 __attribute__((ext_vector_type(4)));
    constexpr size_t COLS_BYTES = ((sizeof(T) * (size_t(NT + 1) * 6 * (LOGNW == 1 ? 2 : 1) + 3 * size_t(NT + 1) * size_t(LOGNW > 1 ? LOGNW - 1 : 0)) + 31) / 32) * 32;
    rec4* const io = reinterpret_cast<rec4*>(crb_smem + COLS_BYTES);        // [7][NT+1] records, position NT = zeros
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
    if (IO && t < STAGE_IO_STREAMS) io[size_t(t) * (NT + 1) + NULLT] = rec4{T(0), T(0), T(0), T(0)};
    const bool shared_tables = p.slot_stride == 0 && p.lv_stride == 0 && p.fin_stride == 0;
    const bool corrected = (p.flags & 4u) != 0;
    const bool has_right = valid && j + 1 < S, has_left = valid && j >= 1;
    const size_t node = size_t(valid ? j + p.off : 0);
    // memory side of the transposed I/O: this thread moves the records of slot t (node t + off), whose owner is thread_of(t)
    const bool mvalid = t < S;
    const size_t mnode = size_t(mvalid ? t + p.off : 0);
    const int mpos = mvalid ? thread_of(t) : NULLT;
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

    rec4 pre[STAGE_IO_STREAMS];   // (IO) the records of node t of the beam about to be staged
    auto fetch_io = [&](int beam) {
        const rec4 z = rec4{T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int k = 0; k < STAGE_IO_STREAMS; ++k) pre[k] = z;
        if (mvalid && beam < p.B) {
            auto ldrec = [](const T* ptr) { return *reinterpret_cast<const rec4*>(ptr); };
            const size_t moff = size_t(beam) * 2 * plane + mnode * 4;
            pre[0] = ldrec(p.xs + moff); pre[1] = ldrec(p.xs + moff + plane);
            pre[2] = ldrec(p.x + moff); pre[3] = ldrec(p.x + moff + plane);
            if (p.stage > 0) { pre[4] = ldrec(p.acc + moff); pre[5] = ldrec(p.acc + moff + plane); }
            if (p.u_held) pre[6] = ldrec(p.u_held + size_t(beam) * plane + mnode * 4);
        }
    };
    int it = 0;
    for (int beam = blockIdx.x; beam < p.B; beam += gridDim.x, ++it) {
        if (!shared_tables) load_tables(beam);
        if (IO) fetch_io(beam);
        // ---- this stage's state, the neighbours' pieces of it, the input force
        const size_t xoff = size_t(beam) * 2 * plane + node * 4;
        T sq[3] = {T(0), T(0), T(0)}, sv[3] = {T(0), T(0), T(0)}, qL[3] = {T(0), T(0), T(0)}, uin[3] = {T(0), T(0), T(0)};
        T x0q[3] = {T(0), T(0), T(0)}, x0v[3] = {T(0), T(0), T(0)}, aq[3] = {T(0), T(0), T(0)}, av[3] = {T(0), T(0), T(0)};
        T phiR = T(0), amp = T(0);
        if (IO) {
            auto at = [&](int k, int pos) -> rec4& { return io[size_t(k) * (NT + 1) + pos]; };
            if (mvalid) {
#pragma unroll
                for (int k = 0; k < STAGE_IO_STREAMS; ++k) at(k, mpos) = pre[k];
            }
            __syncthreads();
            const int own = valid ? t : NULLT;
            const rec4 rq = at(0, own), rv = at(1, own), bq = at(2, own), bv = at(3, own), cq = at(4, own), cv = at(5, own), ru = at(6, own);
            const rec4 rl = at(0, t_l1);
            if (GRAV) phiR = at(0, t_r1)[2];
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
            if (valid && p.amp && j == (p.imp_node_b ? p.imp_node_b[beam] - p.off : p.imp_slot)) amp = p.amp[beam];
        } else if (valid) {
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
            if (p.amp && j == (p.imp_node_b ? p.imp_node_b[beam] - p.off : p.imp_slot)) amp = p.amp[beam];
        }
        // ---- forces on this node from its own element, drag, gravity, inputs
        T fl[3], fr[3];
        CRB_SETPRIO(P_FORCE);
        if (EM == EM_NONLINEAR) elem_force_nonlinear<T>(ec.c, qL, sq, false, fl, fr);
        else if (EM == EM_LINEAR) elem_force_linear<T>(ec.c, qL, sq, fl, fr);
        else elem_force<T>(ec, qL, sq, corrected, fl, fr);
        CRB_SETPRIO(P_XCHG);
        T pp[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) pp[c] = uin[c] + ((imp_on && c == p.imp_dof) ? T(1) : T(0)) * amp - fr[c];
        pp[1] += drag_force<T>(dragc, sv[1]);
        if (GRAV) {
            T g_own[2], g_left[2];
            T gx = p.gx, gy = p.gy;
            if (p.gvec) { gx = p.gvec[2 * size_t(beam)]; gy = p.gvec[2 * size_t(beam) + 1]; }   // per-beam ForceParams
            gravity_segment<T>(has_right ? T(0.5) * (sq[2] + phiR) : sq[2], gx, gy, hm_own, g_own);
            if (LOGNW == 0) {  // (as in the stepper: the left lane's own segment)
                g_left[0] = lane_lower<T, 1>(g_own[0], lane);
                g_left[1] = lane_lower<T, 1>(g_own[1], lane);
            } else {
                gravity_segment<T>(T(0.5) * (qL[2] + sq[2]), gx, gy, hm_left, g_left);
            }
            pp[0] += g_own[0] + g_left[0];
            pp[1] += g_own[1] + g_left[1];
        }
        // ---- round A: publish {p, fl}, rebuild r of this node and of both stride-1 neighbours, level 0
        T r[3], rlo[3], rhi[3];
        if (LOGNW == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                r[c] = pp[c] - lane_higher<T, 1>(fl[c], lane);   // (one wave: r first, then ITS neighbours -- see the stepper)
                rlo[c] = lane_lower<T, 1>(r[c], lane);
                rhi[c] = lane_higher<T, 1>(r[c], lane);
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
        if (IO) {
            auto at = [&](int k, int pos) -> rec4& { return io[size_t(k) * (NT + 1) + pos]; };
            T nq[3], nv[3], oq[3], ov[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                nq[c] = (p.stage ? aq[c] : T(0)) + w * sv[c];
                nv[c] = (p.stage ? av[c] : T(0)) + w * a[c];
                oq[c] = (p.stage < 3) ? x0q[c] + cs * sv[c] : x0q[c] + dt6 * nq[c];
                ov[c] = (p.stage < 3) ? x0v[c] + cs * a[c] : x0v[c] + dt6 * nv[c];
            }
            if (valid && p.stage == 3) mark_nonfinite<T>(p, beam, oq, ov);
            // (the staging is free again: every thread read its inputs before the barrier of round A)
            if (valid) {
                at(0, t) = rec4{oq[0], oq[1], oq[2], T(0)}; at(1, t) = rec4{ov[0], ov[1], ov[2], T(0)};
                if (p.stage < 3) { at(2, t) = rec4{nq[0], nq[1], nq[2], T(0)}; at(3, t) = rec4{nv[0], nv[1], nv[2], T(0)}; }
            }
            __syncthreads();
            if (mvalid) {
                const size_t moff = size_t(beam) * 2 * plane + mnode * 4;
                auto strec = [](T* ptr, const rec4 v) { *reinterpret_cast<rec4*>(ptr) = v; };
                if (p.stage < 3) {   // (whole records: the pad value is written as 0)
                    strec(p.out + moff, at(0, mpos)); strec(p.out + moff + plane, at(1, mpos));
                    strec(p.acc + moff, at(2, mpos)); strec(p.acc + moff + plane, at(3, mpos));
                } else {
                    strec(p.x + moff, at(0, mpos)); strec(p.x + moff + plane, at(1, mpos));
                }
            }
            __syncthreads();   // the next beam's records may be staged only after these reads
        } else if (valid) {
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
                mark_nonfinite<T>(p, beam, oq, ov);
            }
        }
        // LOGNW >= 2: the level-1 barrier above orders this beam's round-A reads before the next beam's
        // round-A writes; LOGNW == 1 alternates two round-A buffers; LOGNW == 0 uses no LDS
    }
}

}  // namespace crb
