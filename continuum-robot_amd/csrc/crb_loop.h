// crb_loop.h -- the closed-loop (LQR) RK4 rollout of an ensemble as ONE persistent launch.
//
// examples/lqr_control.py:95-125 integrates  x' = f(t, x, K (r - x))  (control/full_state_linear.py:81): the
// controller is part of the right-hand side, so a classical RK4 step needs the product U = (R - X) K^T -- a
// [B x 2n] . [2n x n] GEMM -- at each of its four stages, between two halves that want opposite decompositions:
//   * the GEMM wants the gain split by OUTPUT COLUMNS so that a workgroup's slice of K never moves again,
//   * the mass solve (cyclic reduction along the beam) wants WHOLE BEAMS in one workgroup.
// The stage-split path (crb_feedback_ws_kernel + crb_stage_lean_kernel, 8 launches per step) pays for that
// transposition with ~38 node records per node-step through HBM and re-streams the gain through LDS in every
// launch.  Here a GROUP of NB workgroups owns a row block of 64 beams for the whole rollout:
//   GEMM phase   workgroup y of the group computes U[64 beams][48 columns = nodes 16y .. 16y+15] on
//                v_mfma_f64_16x16x4_f64.  Its gain slice [48 x 2n] is loaded into REGISTERS once per launch (four
//                waves, K split four ways: 18 NB fragments of one double per lane and wave = 288 VGPRs at NB = 8);
//                only the stage states (the A operand) stream, straight from L2 into MFMA fragments: the buffer that
//                holds them is laid out fragment-major, so every load is a contiguous 1.5 KB per wave and needs no
//                LDS staging, no loader waves and no offset tables.
//   hand-off     split-K partials are summed through LDS, the tile is published (write-through stores, one counter
//                add per workgroup), the group waits for its NB tiles.
//   stage phase  workgroup y takes beams 64/NB * y ... of the row block: element forces, gravity, drag, the mass solve
//                (the lean exchange structure of crb_lean.h) and the RK4 bookkeeping.  The RK4 accumulator and the solve
//                tables stay in LDS (the register file belongs to the gain), the step's start state in a buffer of the
//                workgroup's own that no other workgroup touches (L2); the new stage state goes back to the fragment
//                buffer; hand-off.
// HBM sees the state once at the start and once at the end of the rollout.  Workgroups of different groups never
// wait for each other.
//
// Hand-off protocol (MI355X_MICROARCH.md, 'inter-workgroup visibility', first row of the table of valid forms): every
// byte that changes hands is stored write-through (sc1) and loaded with sc1 (never from this CU's L1); every storing
// wave drains its stores (s_waitcnt vmcnt(0)), the workgroup's barrier follows, ONE lane adds to the group's
// monotonic counter (agent scope) and polls it (relaxed, s_sleep) until all NB workgroups of the group have arrived;
// the other waves load behind the barrier that lane then joins.  Results do not depend on dispatch order or
// workgroup -> XCD placement: roles are handed out by a ticket counter in START order, so the members of every
// group but the last one started are running, and a wait gives up after `timeout` ticks of the 100 MHz clock
// (error word set, every workgroup leaves) instead of hanging the device.
#pragma once
#include <hip/hip_runtime.h>

#include "crb_feedback.h"
#include "crb_lean.h"

namespace crb {

constexpr int LOOP_SYNC_WORDS = 2048;   // 8 KB at the start of the work buffer: [0] ticket, [1] error, counters at [32 + 32 g]
constexpr int LOOP_MAX_GROUPS = (LOOP_SYNC_WORDS - 32) / 32;
constexpr int LOOP_SC1 = 16;            // aux bit of the raw buffer intrinsics: sc1 (write-through store / L1-bypassing load)
#ifndef CRB_LOOP_PROF                   // 1: every workgroup adds its time per phase (100 MHz ticks) to sync words [2 .. 13] (tuning builds)
#define CRB_LOOP_PROF 0
#endif
constexpr int LOOP_DEPTH = 4;           // register sets of the A operand's ring: the loads of three slot pairs fly ahead of the MFMAs (24 + 6 VGPRs each)

template <typename T>
struct LoopParams {
    KParams<T> k;              // tables, state, impulse, clock (as for the stage kernel)
    const T* kfrag;            // [NB][4 waves][3][6 NB][64 lanes]: the gain as the workgroups' MFMA B fragments (crb_loop_gain_kernel)
    const T* ref;              // [B][2n] reduced, or nullptr (= regulation to 0)
    const int32_t* red_map;    // [3 n_node]: reduced index of a full DOF or -1 (reference vectors, HAS_REF)
    int n_red;                 // n
    T* ebuf;                   // [groups][8 NB slot pairs][64 beams][12]  stage state (HAS_REF: r - state) in fragment order: the GEMM's A operand
    T* ubuf;                   // [groups][NB][64 beams][48]  feedback force, one contiguous tile per workgroup (column block)
    T* ownbuf;                 // [groups][NB][passes][6 items][256 threads][16 B] {step-start state, stage state} of every node, private to the workgroup
                               // that owns the beam (plain accesses through L2; what is handed on goes to ebuf write-through)
    unsigned* sync;            // LOOP_SYNC_WORDS words, zeroed before the launch
    int n_rb, n_groups;        // row blocks of 64 beams; groups of NB workgroups in the grid
    int fences;                // 1: agent-scope release / acquire fences around every hand-off as well (debugging aid)
    unsigned long long timeout;
};

template <int NB>
__host__ __device__ constexpr size_t loop_ebuf_elems() { return size_t(8 * NB) * 64 * 12; }
template <int NB>
__host__ __device__ constexpr size_t loop_ubuf_elems() { return size_t(64) * 48 * NB; }
// per-slot table record in LDS, in 16-byte items: [LV levels x 10 multipliers][final 5, drag][element c0..c5][half masses own, left]
template <int LV>
__host__ __device__ constexpr int loop_tab_items() { return (LV * 10 + 14) / 2; }
// beams a thread of the stage phase carries at once (loop_rhs): two where the exchange regions of four beams in flight
// still fit the CU's 160 KB of LDS next to accumulator, tables and sync words (LV = 5), else one
template <int LV, int LOGNW>
__host__ __device__ constexpr int loop_nbi() { return (LOGNW == 0 || LV <= 5) ? 2 : 1; }
template <typename T, int LV, int LOGNW, int NB>
__host__ __device__ constexpr size_t loop_lds_bytes() {
    constexpr int NTB = 64 << LOGNW, BPW = 64 / NB, BPP = 256 / NTB;
    // RK4 accumulator of the workgroup's beams + solve tables + element kinds + max(split-K scratch of six 16 x 48 partial
    // tiles, exchange columns) + 16 B
    const size_t acc = size_t(6) * BPW * NTB * sizeof(T);
    const size_t tab = size_t(loop_tab_items<LV>()) * NTB * 2 * sizeof(T) + size_t(NTB) * sizeof(int32_t);
    const size_t red = size_t(6) * 3 * 4 * 64 * sizeof(T);
    // exchange columns of the beams in flight: q (3), then round A (8) -- over which a round's outgoing records are gathered
    const size_t xch_a = LOGNW == 0 ? 0 : size_t(BPP) * loop_nbi<LV, LOGNW>() * 8 * (NTB + 1) * sizeof(T);
    const size_t stg = size_t(256) * loop_nbi<LV, LOGNW>() * 6 * sizeof(T);
    const size_t xch = size_t(BPP) * loop_nbi<LV, LOGNW>() * 3 * (NTB + 1) * sizeof(T) + (xch_a > stg ? xch_a : stg);
    return acc + tab + ((red > xch ? red : xch) + 15) / 16 * 16 + 16;
}

typedef unsigned loop_u4 __attribute__((ext_vector_type(4)));
typedef unsigned loop_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double loop_d(unsigned lo, unsigned hi) { return __hiloint2double(int(hi), int(lo)); }
__device__ __forceinline__ loop_u4 loop_pack(double a, double b) {
    return loop_u4{unsigned(__double2loint(a)), unsigned(__double2hiint(a)), unsigned(__double2loint(b)), unsigned(__double2hiint(b))};
}
// three / six consecutive doubles at byte offset `off` of a buffer, L1-bypassing
__device__ __forceinline__ void loop_ld3(__amdgpu_buffer_rsrc_t r, unsigned off, double o[3]) {
    const loop_u4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, LOOP_SC1);
    const loop_u2 b = __builtin_amdgcn_raw_buffer_load_b64(r, off + 16, 0, LOOP_SC1);
    o[0] = loop_d(a[0], a[1]); o[1] = loop_d(a[2], a[3]); o[2] = loop_d(b[0], b[1]);
}
__device__ __forceinline__ void loop_ld6(__amdgpu_buffer_rsrc_t r, unsigned off, double q[3], double v[3]) {
    const loop_u4 a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, LOOP_SC1);
    const loop_u4 b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, LOOP_SC1);
    const loop_u4 c = __builtin_amdgcn_raw_buffer_load_b128(r, off + 32, 0, LOOP_SC1);
    q[0] = loop_d(a[0], a[1]); q[1] = loop_d(a[2], a[3]); q[2] = loop_d(b[0], b[1]);
    v[0] = loop_d(b[2], b[3]); v[1] = loop_d(c[0], c[1]); v[2] = loop_d(c[2], c[3]);
}
__device__ __forceinline__ void loop_st6(__amdgpu_buffer_rsrc_t r, unsigned off, const double q[3], const double v[3]) {
    __builtin_amdgcn_raw_buffer_store_b128(loop_pack(q[0], q[1]), r, off, 0, LOOP_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(loop_pack(q[2], v[0]), r, off + 16, 0, LOOP_SC1);
    __builtin_amdgcn_raw_buffer_store_b128(loop_pack(v[1], v[2]), r, off + 32, 0, LOOP_SC1);
}

// The gain slice of a workgroup is 18 NB doubles per lane: 288 registers at NB = 8, more than the 256 accumulator
// registers.  The matrix instruction is written out so that the register CLASS of every fragment is fixed: the resident
// fragments sit in accumulator registers (which vector-ALU code cannot touch, so the stage phase cannot evict them), the
// sums and the streamed operands in vector registers.  Left to the compiler the sums take the accumulator file, a third
// of the slice spills to scratch and the A loads sink to their first use.
// Not all of it is resident: the last LOOP_BSTREAM (36) k-steps of the third column tile are re-read from L2 in every GEMM
// phase, riding in the A operand's register ring (coalesced 512-byte loads, +18 % on the A stream at 36 of 144) -- the
// accumulator file then keeps room for what the stage phase parks there, instead of the compiler evicting fragments to
// scratch.
constexpr int LOOP_BSTREAM = 36;
template <int NB>
__host__ __device__ constexpr int loop_n_streamed() {   // k-steps of column tile 2 that are streamed (a multiple of 3: whole slot pairs)
    return 18 * NB > 128 ? (LOOP_BSTREAM > 18 * NB - 126 ? LOOP_BSTREAM : 18 * NB - 126) : 0;
}
template <int NB>
__host__ __device__ constexpr bool loop_b_streamed(int n, int ks) { return n == 2 && ks >= 6 * NB - loop_n_streamed<NB>(); }
__device__ __forceinline__ void loop_mfma(bool B_IN_VGPR /* a constant once the loops are unrolled */, typename MfmaOps<double>::acc_t& c, double a, double b) {
    // (s_nop 1: a freshly copied operand may not feed the matrix pipe at once; the sum chains on itself without a wait)
    if (B_IN_VGPR) asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}

// The gain [n][2n] (reduced ordering, linear_quadratic_regulator.py:84-191) re-laid as the MFMA B fragments of the
// persistent kernel's workgroups: out[y][wave][n][ks][lane] = K[row][col] with
//   row = DOF (48 y + 16 n + (lane & 15)) of the padded slot-major order (slot = row / 3, dof = row % 3),
//   col = (slot 2 (2 NB wave + ks / 3) + (lane >> 5), plane (lane >> 4) & 1, dof ks % 3),
// zero where either DOF is constrained or lies in the padding.  One thread per value.
template <typename T, int NB>
__global__ void crb_loop_gain_kernel(const T* gain, const int32_t* red_map, int n_red, int S, int off, T* out) {
    constexpr int SPW = 2 * NB, KSW = 3 * SPW;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NB * 4 * 3 * KSW * 64) return;
    const int lane = idx & 63, ks = (idx >> 6) % KSW, n = (idx >> 6) / KSW % 3, wave = (idx >> 6) / (3 * KSW) % 4, y = (idx >> 6) / (12 * KSW);
    const int fi = lane & 15, kq = lane >> 4;
    const int r = 48 * y + 16 * n + fi, slot_r = r / 3, dof_r = r - 3 * slot_r;
    const int spl = ks / 3, tt = ks - 3 * spl;
    const int slot_c = 2 * (SPW * wave + spl) + (kq >> 1);
    const int redr = slot_r < S ? red_map[3 * (slot_r + off) + dof_r] : -1;
    const int redc = slot_c < S ? red_map[3 * (slot_c + off) + tt] : -1;
    out[idx] = (redr >= 0 && redc >= 0) ? gain[size_t(redr) * 2 * n_red + size_t(kq & 1) * n_red + redc] : T(0);
}

// NBI right-hand sides  a = Minv (uadd - k(q) + drag + gravity)  at once: the NTB = 64 << LOGNW threads that carry a
// beam each evaluate the same node of NBI beams (`lds_q` / `lds_a`: their exchange columns, beam after beam; `tl`: the thread's
// index inside the beam).  The exchange structure is lean_rhs's (crb_lean.h) plus the plain cantilever's nearest-neighbour gravity, as
// crb_stage_lean_kernel evaluates it -- with every table value read from LDS where it is used (`tab`: this thread's record,
// 16-byte item k at tab[2 k NTB]) instead of living in registers.  Why several beams per thread: the workgroup runs ONE wave
// per SIMD (the register file belongs to the gain), so nothing hides an instruction's latency but independent work of the
// same wave; the beams share every table read and every barrier.  Barriers are workgroup-wide: all beams of the workgroup
// run through here in lockstep.
template <typename T, int LV, int LOGNW, bool GRAV, int EM, int NBI>
__device__ __forceinline__ void loop_rhs(const T* tab, int kind, const T gx[NBI], const T gy[NBI], bool corrected, bool drag_on, T* lds_q, T* lds_a,
                                         int tl, int lane, int j, int S, bool valid, const T sq[NBI][3], const T sv[NBI][3],
                                         const T uadd[NBI][3], T a[NBI][3]) {
    static_assert(LOGNW <= 1, "levels above 0 are lane shifts");
    constexpr int NW = 1 << LOGNW, NT = 64 << LOGNW, NULLT = NT;
    typedef T t2 __attribute__((ext_vector_type(2)));
    auto item = [&](int k) { return *reinterpret_cast<const t2*>(tab + size_t(2 * k) * NT); };
    auto ldsQ = [&](int i) { return lds_q + size_t(i) * 3 * (NT + 1); };     // [3][NT+1]  stage positions of beam i
    auto ldsA = [&](int i) { return lds_a + size_t(i) * 8 * (NT + 1); };     // [8][NT+1]  p0..2, fl0..2, g0..1
    auto thread_of = [](int jj) { return ((jj & (NW - 1)) << 6) | (jj >> LOGNW); };
    const bool has_left = valid && j >= 1, has_right = valid && j + 1 < S;
    const int t_l1 = has_left ? thread_of(j - 1) : NULLT;
    const int t_r1 = has_right ? thread_of(j + 1) : NULLT;
    const int t_r2 = (valid && j + 2 < S) ? thread_of(j + 2) : NULLT;
    T qL[NBI][3], phiR[NBI];
    if (LOGNW == 0) {
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) qL[i][c] = lane_lower<T, 1>(sq[i][c], lane);
            phiR[i] = GRAV ? lane_higher<T, 1>(sq[i][2], lane) : T(0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NBI; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) ldsQ(i)[size_t(c) * (NT + 1) + tl] = sq[i][c];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) qL[i][c] = ldsQ(i)[size_t(c) * (NT + 1) + t_l1];
            phiR[i] = GRAV ? ldsQ(i)[size_t(2) * (NT + 1) + t_r1] : T(0);
        }
    }
    ElemCoef<T> ec;
    {
        const t2 c01 = item(5 * LV + 3), c23 = item(5 * LV + 4), c45 = item(5 * LV + 5);
        ec.c[0] = c01[0]; ec.c[1] = c01[1]; ec.c[2] = c23[0]; ec.c[3] = c23[1]; ec.c[4] = c45[0]; ec.c[5] = c45[1];
        ec.kind = kind;
    }
    const t2 f4d = item(5 * LV + 2);   // [final 4, drag]
    T fl[NBI][3], pp[NBI][3], g_own[NBI][2];
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        T fr[3];
        if (EM == EM_NONLINEAR) elem_force_nonlinear<T>(ec.c, qL[i], sq[i], false, fl[i], fr);
        else if (EM == EM_LINEAR) elem_force_linear<T>(ec.c, qL[i], sq[i], fl[i], fr);
        else elem_force<T>(ec, qL[i], sq[i], corrected, fl[i], fr);
#pragma unroll
        for (int c = 0; c < 3; ++c) pp[i][c] = uadd[i][c] - fr[c];
        pp[i][1] += drag_force<T>(drag_on ? f4d[1] : T(0), sv[i][1]);
        g_own[i][0] = g_own[i][1] = T(0);
    }
    // gravity_forces.py:104-146 on the plain cantilever: segment j averages the rotations of slots j, j+1 and loads slots j
    // and j+1, so node j carries g(segment j) + g(segment j-1).  One sincos per node: a thread evaluates ITS segment and
    // takes the left one's from the left neighbour (lane shift, or a column of the exchange round below).
    if (GRAV) {
        const t2 hm = item(5 * LV + 6);   // [half mass of segment j, of segment j-1]
#pragma unroll
        for (int i = 0; i < NBI; ++i) gravity_segment<T>(has_right ? T(0.5) * (sq[i][2] + phiR[i]) : sq[i][2], gx[i], gy[i], hm[0], g_own[i]);
    }
    T r[NBI][3], rlo[NBI][3], rhi[NBI][3];
    if (LOGNW == 0) {
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            if (GRAV) {
                pp[i][0] += g_own[i][0] + lane_lower<T, 1>(g_own[i][0], lane);
                pp[i][1] += g_own[i][1] + lane_lower<T, 1>(g_own[i][1], lane);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                r[i][c] = pp[i][c] - lane_higher<T, 1>(fl[i][c], lane);
                rlo[i][c] = lane_lower<T, 1>(r[i][c], lane);
                rhi[i][c] = lane_higher<T, 1>(r[i][c], lane);
            }
        }
    } else {
        // round A: {p without gravity, f_left, own segment's gravity}; a node's p is completed from its own and its left
        // neighbour's segment wherever it is needed (this node, and both stride-1 neighbours for level 0)
        const int t_l2 = (valid && j >= 2) ? thread_of(j - 2) : NULLT;
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            T* A = ldsA(i);
#pragma unroll
            for (int c = 0; c < 3; ++c) { A[size_t(c) * (NT + 1) + tl] = pp[i][c]; A[size_t(3 + c) * (NT + 1) + tl] = fl[i][c]; }
            if (GRAV) { A[size_t(6) * (NT + 1) + tl] = g_own[i][0]; A[size_t(7) * (NT + 1) + tl] = g_own[i][1]; }
            // (the caller gathers outgoing records over these columns between two calls: the "no neighbour" entries are
            //  zeroed again -- after the barrier above, which every wave passes once it has read what was gathered)
            if (tl == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) A[size_t(k) * (NT + 1) + NULLT] = T(0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const T* A = ldsA(i);
            auto col = [&](int k, int th) { return A[size_t(k) * (NT + 1) + th]; };
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                T p_l = col(c, t_l1), p_o = pp[i][c], p_r = col(c, t_r1);
                if (GRAV && c < 2) {
                    const T g_l1 = col(6 + c, t_l1);
                    p_l += g_l1 + col(6 + c, t_l2);
                    p_o += g_own[i][c] + g_l1;
                    p_r += col(6 + c, t_r1) + g_own[i][c];
                }
                rlo[i][c] = p_l - fl[i][c];
                r[i][c] = p_o - col(3 + c, t_r1);
                rhi[i][c] = p_r - col(3 + c, t_r2);
            }
        }
    }
#pragma unroll
    for (int l = 0; l < LV; ++l) {
        T cfl[PCR_LEVEL_VALS];
#pragma unroll
        for (int k = 0; k < 5; ++k) { const t2 v = item(5 * l + k); cfl[2 * k] = v[0]; cfl[2 * k + 1] = v[1]; }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            if (l > 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    switch (l - LOGNW) {
                        case 0: rlo[i][c] = lane_lower<T, 1>(r[i][c], lane); rhi[i][c] = lane_higher<T, 1>(r[i][c], lane); break;
                        case 1: rlo[i][c] = lane_lower<T, 2>(r[i][c], lane); rhi[i][c] = lane_higher<T, 2>(r[i][c], lane); break;
                        case 2: rlo[i][c] = lane_lower<T, 4>(r[i][c], lane); rhi[i][c] = lane_higher<T, 4>(r[i][c], lane); break;
                        case 3: rlo[i][c] = lane_lower<T, 8>(r[i][c], lane); rhi[i][c] = lane_higher<T, 8>(r[i][c], lane); break;
                        case 4: rlo[i][c] = lane_lower<T, 16>(r[i][c], lane); rhi[i][c] = lane_higher<T, 16>(r[i][c], lane); break;
                        default: rlo[i][c] = lane_lower<T, 32>(r[i][c], lane); rhi[i][c] = lane_higher<T, 32>(r[i][c], lane); break;
                    }
                }
            }
            pcr_apply_level<T>(cfl, rlo[i], rhi[i], r[i]);
        }
    }
    const t2 f01 = item(5 * LV), f23 = item(5 * LV + 1);
    const T fin[5] = {f01[0], f01[1], f23[0], f23[1], f4d[0]};
#pragma unroll
    for (int i = 0; i < NBI; ++i) pcr_apply_final<T>(fin, r[i], a[i]);
}

// NB workgroups per group (= column blocks of 48 = 16 slots each; slots padded to 16 NB = the threads that carry a beam),
// LOGNW as in the lean kernels (waves per beam).  256 threads, one workgroup per CU (the gain slice wants the
// whole register file).
template <typename T, int LV, int LOGNW, int NB, bool GRAV, int EM, bool HAS_REF>
__global__ void __launch_bounds__(256, 1) crb_loop_kernel(const LoopParams<T> P) {
    static_assert(sizeof(T) == 8, "the persistent closed-loop stepper is built for fp64 plans");
    typedef typename MfmaOps<T>::acc_t acc4;
    typedef T t2 __attribute__((ext_vector_type(2)));
    constexpr int SPAD = 16 * NB, SPW = 2 * NB, KSW = 3 * SPW, D = LOOP_DEPTH;
    constexpr int NTB = 64 << LOGNW, BPP = 256 / NTB, BPW = 64 / NB, NPASS = BPW / BPP, NLI = BPW * NTB;
    static_assert(SPAD == NTB, "one thread per padded slot");
    static_assert(NPASS * BPP == BPW && BPW * NB == 64, "beams of a row block divide evenly");
    static_assert(D >= 2 && D <= SPW, "pipeline depth");
    constexpr int NBI = loop_nbi<LV, LOGNW>(), NDP = NPASS / NBI;   // beams a thread carries at once; such rounds per stage
    static_assert(NDP * NBI == NPASS, "passes divide into rounds");
    constexpr int ITEMS = loop_tab_items<LV>();
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    T* const accs = reinterpret_cast<T*>(crb_smem);                 // [6][NLI]: RK4 accumulator q, v
    T* const tabs = accs + size_t(6) * NLI;                         // [ITEMS][NTB][2]
    int32_t* const kinds = reinterpret_cast<int32_t*>(tabs + size_t(ITEMS) * NTB * 2);   // [NTB]
    T* const scratch = reinterpret_cast<T*>(kinds + NTB);
    unsigned* const sh = reinterpret_cast<unsigned*>(crb_smem + loop_lds_bytes<T, LV, LOGNW, NB>() - 16);
    const KParams<T>& p = P.k;
    // (the wave index is made provably uniform: everything derived from it -- roles, LDS bases, branches -- stays scalar)
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);

    // ---- role: tickets in start order
    if (t == 0) sh[0] = __hip_atomic_fetch_add(P.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = sh[0];
    const int group = __builtin_amdgcn_readfirstlane(int(ticket / NB)), y = __builtin_amdgcn_readfirstlane(int(ticket % NB));
    if (group >= P.n_groups) return;
    unsigned* const counter = P.sync + 32 + 32 * group;
    unsigned* const errw = P.sync + 1;
    unsigned arrivals = 0;   // hand-offs this group has been through
#if CRB_LOOP_PROF
    unsigned long long prof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memrealtime();
#define CRB_LOOP_STAMP(k) do { const unsigned long long now__ = __builtin_amdgcn_s_memrealtime(); prof[k] += now__ - prof_t; prof_t = now__; } while (0)
#else
#define CRB_LOOP_STAMP(k) do { } while (0)
#endif
    // hand-off in two halves.  arrive: this workgroup's stores are out, one lane says so.  wait: true = every workgroup of
    // the group has arrived; false = give up (timeout, or another group's error).  What a workgroup loads between the
    // two must be its own data.
    auto arrive = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
        __syncthreads();
        ++arrivals;
        if (t == 0) {
            if (P.fences) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto wait = [&]() -> bool {
        if (t == 0) {
            const unsigned target = arrivals * NB;
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            unsigned ok = 1;
            for (unsigned spins = 0;; ++spins) {
                if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
                __builtin_amdgcn_s_sleep(2);
                if ((spins & 255u) == 255u) {
                    if (__hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t_start > P.timeout) {
                        __hip_atomic_store(errw, 1u + unsigned(group), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = 0;
                        break;
                    }
                }
            }
            if (P.fences) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            sh[1] = ok;
        }
        __syncthreads();
        return sh[1] != 0u;
    };
    auto handoff = [&]() -> bool { arrive(); return wait(); };

    // ---- GEMM role of this thread: wave = K quarter (slot pairs SPW wave ...), lane = (row / column fi, k quarter kq)
    const int fi = lane & 15, kq = lane >> 4;
    // gain slice: Bf[n][3 spl + tt] = K[row 48 y + 16 n + fi][column of (slot 2 sp + (kq >> 1), plane kq & 1, dof tt)], prepared
    // in exactly this order by crb_loop_gain_kernel (coalesced loads).  Resident fragments: accumulator registers, for the
    // whole launch; streamed ones (loop_b_streamed): re-read with the A operand in every GEMM phase.
    const T* const bsrc = P.kfrag + (size_t(y) * 4 + wave) * 3 * KSW * 64 + lane;
    T Bf[3][KSW];
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) Bf[n][ks] = loop_b_streamed<NB>(n, ks) ? T(0) : bsrc[size_t(n * KSW + ks) * 64];
    // ---- stage role: beam `sub` of the pass, slot j
    const int sub = wave >> LOGNW, wib = wave & ((1 << LOGNW) - 1), tl = (wib << 6) | lane;
    const int S = p.S;
    const int j = (lane << LOGNW) | wib;
    const bool slot_ok = j < S;
    // exchange columns of the beams in flight: [BPP NBI beams][q: 3 columns], then [BPP NBI beams][round A: 8 columns: p, f_left,
    // segment gravity]; this thread's beams are sub NBI + i
    T* const lds_q = scratch + size_t(sub) * NBI * 3 * (NTB + 1);
    T* const lds_a = scratch + size_t(BPP) * NBI * 3 * (NTB + 1) + size_t(sub) * NBI * 8 * (NTB + 1);
    const T* const tab = tabs + size_t(2) * tl;
    const bool corrected = (p.flags & 4u) != 0, drag_on = (p.flags & 1u) != 0;
    const size_t plane = size_t(p.n_node) * 4;
    const T dt = T(p.dt), hdt = T(0.5 * p.dt), dt6 = T(p.dt / 6.0);
    // ---- the solve tables of slot j into LDS, once (every beam of a shared-table plan uses the same ones)
    if (sub == 0) {
        auto put = [&](int k, T a0, T a1) { *reinterpret_cast<t2*>(tabs + (size_t(k) * NTB + tl) * 2) = t2{a0, a1}; };
        if (slot_ok) {
            const SlotConst<T>& sc = p.slot[j];
#pragma unroll
            for (int l = 0; l < LV; ++l) {
                const T* src = p.pcr_levels + (size_t(l) * size_t(S) + size_t(j)) * PCR_LEVEL_VALS;
#pragma unroll
                for (int k = 0; k < 5; ++k) put(5 * l + k, src[2 * k], src[2 * k + 1]);
            }
            const T* fin = p.pcr_final + size_t(j) * PCR_FINAL_VALS;
            put(5 * LV, fin[0], fin[1]);
            put(5 * LV + 1, fin[2], fin[3]);
            put(5 * LV + 2, fin[4], sc.drag);
            put(5 * LV + 3, sc.elem.c[0], sc.elem.c[1]);
            put(5 * LV + 4, sc.elem.c[2], sc.elem.c[3]);
            put(5 * LV + 5, sc.elem.c[4], sc.elem.c[5]);
            put(5 * LV + 6, sc.half_mass, j >= 1 ? p.slot[j - 1].half_mass : T(0));
            kinds[tl] = sc.elem.kind;
        } else {
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) put(k, T(0), T(0));
            kinds[tl] = KIND_NONE;
        }
    }
    int red[3] = {-1, -1, -1};
    if (HAS_REF && slot_ok) {
#pragma unroll
        for (int c = 0; c < 3; ++c) red[c] = P.red_map[3 * (j + p.off) + c];
    }
    const __amdgpu_buffer_rsrc_t ers =
        __builtin_amdgcn_make_buffer_rsrc(P.ebuf + size_t(group) * loop_ebuf_elems<NB>(), 0, unsigned(loop_ebuf_elems<NB>() * sizeof(T)), 0x00020000);
    const __amdgpu_buffer_rsrc_t urs =
        __builtin_amdgcn_make_buffer_rsrc(P.ubuf + size_t(group) * loop_ubuf_elems<NB>(), 0, unsigned(loop_ubuf_elems<NB>() * sizeof(T)), 0x00020000);
    // this thread's nodes in the private buffer: [pass][item 0..5][256 threads] 16-byte items {x0 q0 q1 | q2 v0 | v1 v2 | xs ...}
    // (a wave's access to one item is 1 KB contiguous), as a byte offset (< 4 GB) from the buffer's start -- an offset, not a
    // pointer, is what gets laundered below: a laundered pointer loses its address space
    const unsigned mine = unsigned((((size_t(group) * NB + y) * NPASS * 6) * 256 + t) * 2 * sizeof(T));
    auto own_at = [&](unsigned off, int pass, int item) {
        return reinterpret_cast<t2*>(reinterpret_cast<unsigned char*>(P.ownbuf) + off + unsigned(((pass * 6 + item) * 256) * 2 * sizeof(T)));
    };
    // byte offsets: this lane's first A fragment (row fi of a 16-beam tile, slot pair SPW wave, k quarter kq); this thread's
    // stage record / force of beam `bl` of the workgroup
    const unsigned a_off0 = unsigned((((SPW * wave) * 64 + fi) * 12 + 3 * kq) * sizeof(T));
    auto e_off = [&](int bl) { return unsigned((((j >> 1) * 64 + (y * BPW + bl)) * 12 + (j & 1) * 6) * sizeof(T)); };
    auto u_off = [&](int bl) { return unsigned(((((j >> 4) * 64) + (y * BPW + bl)) * 48 + 3 * (j & 15)) * sizeof(T)); };

    // inputs of one round of the stage phase (NBI beams per thread; beam i of round dp is pass dp NBI + i of the buffers):
    // the stage records (private buffer) and the feedback forces, requested one round ahead
    struct RoundIn { T sq[NBI][3], sv[NBI][3], uin[NBI][3]; };
    auto load_round = [&](RoundIn& in, unsigned own, unsigned ub, int dp, bool with_u) {
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            const t2 v3 = *own_at(own, dp * NBI + i, 3), v4 = *own_at(own, dp * NBI + i, 4), v5 = *own_at(own, dp * NBI + i, 5);
            in.sq[i][0] = v3[0]; in.sq[i][1] = v3[1]; in.sq[i][2] = v4[0]; in.sv[i][0] = v4[1]; in.sv[i][1] = v5[0]; in.sv[i][2] = v5[1];
            if (with_u) loop_ld3(urs, ub + unsigned((dp * NBI + i) * BPP * 48 * sizeof(T)), in.uin[i]);
        }
    };

    for (int rb = group; rb < P.n_rb; rb += P.n_groups) {
        // ---- prologue: the workgroup's beams into its private start-state buffer and into the fragment buffer
#pragma unroll 1
        for (int pass = 0; pass < NPASS; ++pass) {
            const int bl = pass * BPP + sub, beam = rb * 64 + y * BPW + bl;
            T xq[3] = {T(0), T(0), T(0)}, xv[3] = {T(0), T(0), T(0)};
            if (slot_ok && beam < p.B) {
                const SlotConst<T>& sc = p.slot[j];
                const size_t xoff = size_t(beam) * 2 * plane + size_t(j + p.off) * 4;
#pragma unroll
                for (int c = 0; c < 3; ++c) { xq[c] = p.x[xoff + c] * sc.mask[c]; xv[c] = p.x[xoff + plane + c] * sc.mask[c]; }
            }
            *own_at(mine, pass, 0) = t2{xq[0], xq[1]}; *own_at(mine, pass, 1) = t2{xq[2], xv[0]}; *own_at(mine, pass, 2) = t2{xv[1], xv[2]};
            *own_at(mine, pass, 3) = t2{xq[0], xq[1]}; *own_at(mine, pass, 4) = t2{xq[2], xv[0]}; *own_at(mine, pass, 5) = t2{xv[1], xv[2]};
            if (HAS_REF) {
                T eq[3] = {T(0), T(0), T(0)}, ev[3] = {T(0), T(0), T(0)};
                if (beam < p.B) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (red[c] >= 0) {
                            eq[c] = P.ref[size_t(beam) * 2 * P.n_red + red[c]] - xq[c];
                            ev[c] = P.ref[size_t(beam) * 2 * P.n_red + P.n_red + red[c]] - xv[c];
                        }
                }
                loop_st6(ers, e_off(bl), eq, ev);
            } else {
                loop_st6(ers, e_off(bl), xq, xv);
            }
        }
        if (!handoff()) return;   // (its barriers also publish the tables in LDS)
        const int my_kind = kinds[tl];
        T amps[NPASS];   // impulse amplitude of this thread's node in each of its beams (0 everywhere but at the forced node)
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            const int beam = rb * 64 + y * BPW + pass * BPP + sub;
            amps[pass] = T(0);
            if (p.amp && slot_ok && beam < p.B && j == (p.imp_node_b ? p.imp_node_b[beam] - p.off : p.imp_slot)) amps[pass] = p.amp[beam];
        }

        double tc = p.t0;
#pragma unroll 1
        for (int step = 0; step < p.n_steps; ++step) {
            const double t_half = __dadd_rn(tc, 0.5 * p.dt), t_full = __dadd_rn(tc, p.dt);
            const bool last_step = step + 1 == p.n_steps;
#pragma unroll 1
            for (int s = 0; s < 4; ++s) {
                // ======================================================== GEMM phase: U tile = E . Kslice^T
                CRB_LOOP_STAMP(5);
                acc4 acc[4][3];
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 3; ++n) acc[m][n] = acc4{T(0), T(0), T(0), T(0)};
                {
                    // ring of D register sets: the loads of slot pairs spl + 1 .. spl + D - 1 fly while spl is multiplied
                    // (pinned by scheduling barriers: left alone, the scheduler sinks every load to its first use)
                    T af[D][4][3], bs[D][3];
                    // (the streamed fragments by byte offset from the kernel argument: an offset can be laundered -- so that these
                    //  loads are not hoisted out of the step loop into registers -- without the pointer losing its address space)
                    unsigned b_off = unsigned((((size_t(y) * 4 + wave) * 3 + 2) * KSW * 64 + lane) * sizeof(T));
                    asm volatile("" : "+v"(b_off));
                    const T* bl = reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(P.kfrag) + b_off);
                    auto fetch = [&](int spl) {
#pragma unroll
                        for (int m = 0; m < 4; ++m) loop_ld3(ers, a_off0 + unsigned((spl * 64 + 16 * m) * 12 * sizeof(T)), af[spl % D][m]);
                        if (loop_b_streamed<NB>(2, 3 * spl)) {
#pragma unroll
                            for (int tt = 0; tt < 3; ++tt) bs[spl % D][tt] = bl[size_t(3 * spl + tt) * 64];
                        }
                    };
#pragma unroll
                    for (int d = 0; d < D - 1; ++d) fetch(d);
#pragma unroll
                    for (int spl = 0; spl < SPW; ++spl) {
                        if (spl + D - 1 < SPW) fetch(spl + D - 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int tt = 0; tt < 3; ++tt)
#pragma unroll
                            for (int m = 0; m < 4; ++m)
#pragma unroll
                                for (int n = 0; n < 3; ++n) {
                                    const bool streamed = loop_b_streamed<NB>(n, 3 * spl + tt);
                                    loop_mfma(streamed, acc[m][n], af[spl % D][m][tt], streamed ? bs[spl % D][tt] : Bf[n][3 * spl + tt]);
                                }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // (an MFMA's result is readable by other instructions only some cycles after issue; nothing pads that for asm)
                    asm volatile("s_nop 15\n\ts_nop 15"
                                 : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]),
                                   "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]));
                }
                CRB_LOOP_STAMP(0);
                // ---- split-K: wave w owns row tile m = w; partial tiles change hands through LDS, two owners per round; the
                // summed tile is then laid out in LDS as it lies in memory ([64 beams][48 columns], contiguous per workgroup)
                // and leaves in 1 KB pieces per store instruction: a write-through store costs a fabric write per lane unless
                // the lanes of an instruction fill whole lines.
                // (addresses are formed from ONE laundered base per use: hoisted out of the step loop, the 100-odd
                //  distinct addresses of this block would each take a register and end in scratch)
                unsigned red_base = unsigned(reinterpret_cast<unsigned char*>(scratch) - crb_smem) + unsigned(lane) * unsigned(sizeof(T));
                asm volatile("" : "+v"(red_base));
#pragma unroll
                for (int round = 0; round < 2; ++round) {
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        const int m = 2 * round + mm;
                        if (wave != m) {
                            const int rank = wave < m ? wave : wave - 1;
                            unsigned char* dst = crb_smem + red_base + unsigned((mm * 3 + rank) * 12 * 64 * sizeof(T));
#pragma unroll
                            for (int n = 0; n < 3; ++n)
#pragma unroll
                                for (int reg = 0; reg < 4; ++reg) *reinterpret_cast<T*>(dst + (n * 4 + reg) * 64 * sizeof(T)) = acc[m][n][reg];
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        const int m = 2 * round + mm;
                        if (wave == m) {
                            const unsigned char* src = crb_smem + red_base + unsigned(mm * 3 * 12 * 64 * sizeof(T));
#pragma unroll
                            for (int n = 0; n < 3; ++n)
#pragma unroll
                                for (int reg = 0; reg < 4; ++reg) {
                                    T sum = acc[m][n][reg];
#pragma unroll
                                    for (int k = 0; k < 3; ++k) sum += *reinterpret_cast<const T*>(src + (k * 12 + n * 4 + reg) * 64 * sizeof(T));
                                    acc[m][n][reg] = HAS_REF ? sum : -sum;
                                }
                        }
                    }
                    __syncthreads();
                }
                {
                    // row 16 wave + (lane >> 4) + 4 reg, column 16 n + fi of the workgroup's tile
                    unsigned char* tile = reinterpret_cast<unsigned char*>(scratch);
                    unsigned t_base = unsigned((((lane >> 4) + 16 * wave) * 48 + fi) * sizeof(T)), s_base = unsigned((wave * 64 + lane) * 16);
                    asm volatile("" : "+v"(t_base), "+v"(s_base));
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        if (wave == m) {
#pragma unroll
                            for (int n = 0; n < 3; ++n)
#pragma unroll
                                for (int reg = 0; reg < 4; ++reg) *reinterpret_cast<T*>(tile + t_base + (4 * reg * 48 + 16 * n) * sizeof(T)) = acc[m][n][reg];
                        }
                    __syncthreads();
                    const unsigned ug = unsigned(y * 64 * 48 * sizeof(T));
#pragma unroll
                    for (int c = 0; c < 6; ++c) {   // 24 KB = 24 pieces of 1 KB, six per wave
                        const loop_u4 v = *reinterpret_cast<const loop_u4*>(tile + s_base + c * 4096);
                        __builtin_amdgcn_raw_buffer_store_b128(v, urs, ug + s_base + c * 4096, 0, LOOP_SC1);
                    }
                }
                CRB_LOOP_STAMP(1);
                // ======================================================== stage phase: whole beams
                // A wave's memory counter retires in order and a write-through store takes over a microsecond to be
                // acknowledged, so no wait may sit between a pass's stores and the next pass's loads: the inputs of pass
                // p + 1 are requested at the top of pass p (ahead of p's stores; they are waited for at the top of p + 1,
                // with p's stores still in flight behind them), and nothing else in a pass touches the counter -- no
                // spill reloads (the register budget of this phase is why part of the gain slice is streamed), no
                // loads of per-beam constants.  The workgroup's own records of the first pass are requested while it
                // waits for the other workgroups' tiles.
                RoundIn cur, nxt;
                // (bases laundered per stage: hoisted out of the step loop, every address below would hold a register)
                unsigned ub = u_off(sub), acc_base = unsigned(reinterpret_cast<unsigned char*>(accs) - crb_smem) + unsigned(t) * unsigned(sizeof(T));
                unsigned own = mine;
                asm volatile("" : "+v"(ub), "+v"(acc_base), "+v"(own));
                arrive();
                // (the exchange columns share the scratch with the tile just stored -- every wave has read its pieces before the
                //  barrier inside arrive(): their "no neighbour" entries are zeroed again)
                if (LOGNW > 0 && tl == 0) {
#pragma unroll
                    for (int k = 0; k < 3 * NBI; ++k) lds_q[size_t(k) * (NTB + 1) + NTB] = T(0);
#pragma unroll
                    for (int k = 0; k < 8 * NBI; ++k) lds_a[size_t(k) * (NTB + 1) + NTB] = T(0);
                }
                load_round(cur, own, ub, 0, false);
                if (!wait()) return;
#pragma unroll
                for (int i = 0; i < NBI; ++i) loop_ld3(urs, ub + unsigned(i * BPP * 48 * sizeof(T)), cur.uin[i]);
                CRB_LOOP_STAMP(2);
                const double ts = (s == 0) ? tc : ((s == 3) ? t_full : t_half);
                const bool imp_on = ts < p.duration;
                const T w = (s == 0 || s == 3) ? T(1) : T(2);
                const T cs = (s == 2) ? dt : hdt;
#pragma unroll 1
                for (int dp = 0; dp < NDP; ++dp) {
                    nxt = cur;
                    if (dp + 1 < NDP) load_round(nxt, own, ub, dp + 1, true);
                    // the step-start records (needed after the right-hand sides), per-beam constants
                    T x0q[NBI][3], x0v[NBI][3], gx[NBI], gy[NBI], rq[HAS_REF ? NBI : 1][3], rv[HAS_REF ? NBI : 1][3];
#pragma unroll
                    for (int i = 0; i < NBI; ++i) {
                        const t2 v0 = *own_at(own, dp * NBI + i, 0), v1 = *own_at(own, dp * NBI + i, 1), v2 = *own_at(own, dp * NBI + i, 2);
                        x0q[i][0] = v0[0]; x0q[i][1] = v0[1]; x0q[i][2] = v1[0]; x0v[i][0] = v1[1]; x0v[i][1] = v2[0]; x0v[i][2] = v2[1];
                        gx[i] = p.gx; gy[i] = p.gy;
                        if ((GRAV && p.gvec) || HAS_REF) {
                            const int beam = rb * 64 + y * BPW + (dp * NBI + i) * BPP + sub;
                            const bool real = beam < p.B;
                            if (GRAV && p.gvec && real) { gx[i] = p.gvec[2 * size_t(beam)]; gy[i] = p.gvec[2 * size_t(beam) + 1]; }
                            if (HAS_REF) {
                                const T* rrow = P.ref + size_t(real ? beam : 0) * 2 * P.n_red;
#pragma unroll
                                for (int c = 0; c < 3; ++c) {
                                    rq[HAS_REF ? i : 0][c] = (red[c] >= 0 && real) ? rrow[red[c]] : T(0);
                                    rv[HAS_REF ? i : 0][c] = (red[c] >= 0 && real) ? rrow[P.n_red + red[c]] : T(0);
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    CRB_LOOP_STAMP(6);
                    T uadd[NBI][3], a[NBI][3];
#pragma unroll
                    for (int i = 0; i < NBI; ++i) {
                        T amp = amps[i];
#pragma unroll
                        for (int k = 1; k < NDP; ++k) amp = (dp == k) ? amps[k * NBI + i] : amp;   // (a select chain: no dynamic register index)
#pragma unroll
                        for (int c = 0; c < 3; ++c) uadd[i][c] = cur.uin[i][c] + ((imp_on && c == p.imp_dof) ? T(1) : T(0)) * amp;
                    }
                    // (beams past the ensemble's end run on zeros: finite, never stored into the state)
                    loop_rhs<T, LV, LOGNW, GRAV, EM, NBI>(tab, my_kind, gx, gy, corrected, drag_on, lds_q, lds_a, tl, lane, j, S, slot_ok, cur.sq, cur.sv, uadd, a);
                    CRB_LOOP_STAMP(7);
                    // ---- RK4 bookkeeping of this round
                    T eo[NBI][6];   // what the GEMM reads of each beam: the new stage state (with a reference: r - state)
#pragma unroll
                    for (int i = 0; i < NBI; ++i) {
                        const int pass = dp * NBI + i;
                        const int beam = rb * 64 + y * BPW + pass * BPP + sub;
                        const unsigned ab = acc_base + unsigned(pass * 256 * sizeof(T));
                        auto acc_at = [&](int c) -> T& { return *reinterpret_cast<T*>(crb_smem + ab + unsigned(c * NLI * sizeof(T))); };
                        T nq[3], nv[3], oq[3], ov[3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const T aq = s ? acc_at(c) : T(0), av = s ? acc_at(3 + c) : T(0);
                            nq[c] = aq + w * cur.sv[i][c];
                            nv[c] = av + w * a[i][c];
                            oq[c] = (s < 3) ? x0q[i][c] + cs * cur.sv[i][c] : x0q[i][c] + dt6 * nq[c];
                            ov[c] = (s < 3) ? x0v[i][c] + cs * a[i][c] : x0v[i][c] + dt6 * nv[c];
                        }
                        if (s < 3) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) { acc_at(c) = nq[c]; acc_at(3 + c) = nv[c]; }
                        } else if (!last_step) {
                            *own_at(own, pass, 0) = t2{oq[0], oq[1]}; *own_at(own, pass, 1) = t2{oq[2], ov[0]}; *own_at(own, pass, 2) = t2{ov[1], ov[2]};
                        } else if (slot_ok && beam < p.B) {   // the rollout's last stage: the state back into the device layout
                            typedef T rec4 __attribute__((ext_vector_type(4)));
                            T* dst = p.x + size_t(beam) * 2 * plane + size_t(j + p.off) * 4;
                            *reinterpret_cast<rec4*>(dst) = rec4{oq[0], oq[1], oq[2], T(0)};
                            *reinterpret_cast<rec4*>(dst + plane) = rec4{ov[0], ov[1], ov[2], T(0)};
                            mark_nonfinite<T>(p, beam, oq, ov);
                        }
                        if (HAS_REF) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                eo[i][c] = red[c] >= 0 ? rq[HAS_REF ? i : 0][c] - oq[c] : T(0);
                                eo[i][3 + c] = red[c] >= 0 ? rv[HAS_REF ? i : 0][c] - ov[c] : T(0);
                            }
                        } else {
#pragma unroll
                            for (int c = 0; c < 3; ++c) { eo[i][c] = oq[c]; eo[i][3 + c] = ov[c]; }
                        }
                        if (!(s == 3 && last_step)) {   // (nobody reads a stage state after the rollout's last stage)
                            *own_at(own, pass, 3) = t2{oq[0], oq[1]}; *own_at(own, pass, 4) = t2{oq[2], ov[0]}; *own_at(own, pass, 5) = t2{ov[1], ov[2]};
                        }
                    }
                    // ---- what the other workgroups read.  The fragment buffer is [slot pair][64 beams][12 values]: the R = BPP NBI
                    // beams of a round are neighbours in it, so per slot pair they form ONE contiguous segment of R x 96 bytes
                    // (384 at NB = 8: three whole lines).  The round's records are gathered in LDS in that order and leave in
                    // 1 KB pieces per store instruction -- whole lines: a write-through store costs a fabric write per lane
                    // unless the lanes of an instruction fill lines (48 bytes per thread straight from the registers took 7 us
                    // per stage, a quarter of it in the hand-off).
                    if (!(s == 3 && last_step)) {
                        constexpr int R = BPP * NBI, SEG = R * 12 * int(sizeof(T)), PIECES = (SPAD / 2) * SEG / 1024;
                        static_assert(((SPAD / 2) * SEG) % 4096 == 0, "whole 1 KB pieces, evenly over the waves");
                        unsigned char* stg = reinterpret_cast<unsigned char*>(scratch + size_t(BPP) * NBI * 3 * (NTB + 1));   // over the round-A columns
                        __syncthreads();   // (every wave has read its round-A columns)
#pragma unroll
                        for (int i = 0; i < NBI; ++i) {
                            t2* d = reinterpret_cast<t2*>(stg + (((j >> 1) * R + i * BPP + sub) * 12 + (j & 1) * 6) * sizeof(T));
                            d[0] = t2{eo[i][0], eo[i][1]}; d[1] = t2{eo[i][2], eo[i][3]}; d[2] = t2{eo[i][4], eo[i][5]};
                        }
                        __syncthreads();
                        const unsigned g0 = unsigned((y * BPW + dp * R) * 12 * sizeof(T));   // the round's first beam inside a slot pair's 64
                        unsigned l16 = unsigned(lane) * 16u;
                        asm volatile("" : "+v"(l16));   // (the piece addresses are formed here, not hoisted into registers that live across the launch)
#pragma unroll
                        for (int c = 0; c < PIECES / 4; ++c) {
                            const unsigned q = unsigned((c * 4 + wave) * 1024) + l16;     // byte inside the gathered image
                            const unsigned sp = q / unsigned(SEG), within = q - sp * unsigned(SEG);
                            const loop_u4 v = *reinterpret_cast<const loop_u4*>(stg + q);
                            __builtin_amdgcn_raw_buffer_store_b128(v, ers, sp * unsigned(64 * 12 * sizeof(T)) + g0 + within, 0, LOOP_SC1);
                        }
                    }
                    cur = nxt;
                    CRB_LOOP_STAMP(8);
                }
                CRB_LOOP_STAMP(3);
                // (the rollout's last stage hands nothing on: the next row block's first hand-off orders the rest)
                if (!(s == 3 && last_step)) {
                    if (!handoff()) return;
                }
                CRB_LOOP_STAMP(4);
            }
            tc = t_full;
        }
    }
#if CRB_LOOP_PROF
    if (t == 0) {
#pragma unroll
        for (int k = 0; k < 10; ++k) atomicAdd(reinterpret_cast<unsigned long long*>(P.sync + 2) + k, prof[k]);
    }
#endif
#undef CRB_LOOP_STAMP
}

}  // namespace crb
