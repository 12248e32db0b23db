// crb_feedback.h -- LQR feedback force of an ensemble as a fused fp64-MFMA GEMM, and state layout conversion.
#pragma once
#include <hip/hip_runtime.h>

#include "crb_generic.h"

namespace crb {

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
// the MFMA of each dtype: A / B fragments are one value per lane (A[i = lane & 15][k = lane >> 4]) for both,
// the C/D row of accumulator register `reg` differs (cdna_hip_programming.md, 'Fragment layout')
// (MfmaOps<T>: the matrix instruction of each dtype and its C/D row map -- crb_generic.h, shared with the fused small-beam loop)
template <typename T>
struct FeedbackParams {
    const T* xs;       // [B][2][n_node][4]
    const T* ref;      // [B][2n] reduced, or nullptr (= 0)
    const T* gain;     // [n][2n] row-major
    T* u;              // [B][n_node][4]; only free-DOF entries are written
    const int32_t* col_off; // [2n] offset of reduced state index j inside a beam's state record
    const int32_t* row_off; // [n]  offset of reduced position index i inside a beam's force record
    int B, n, n2;           // n2 = 2n
    size_t x_stride, u_stride;
    // grouped launches (per-group gains of heterogeneous ensembles, crb_feedback_force_grouped): row m of the product is beam
    // beam_idx[m] (nullptr: beam m), and the reference rows are the ensemble's padded reduced states [ref_ld] with the
    // velocities at ref_half (ungrouped: ref_ld = 2n, ref_half = n)
    const int32_t* beam_idx;
    int ref_ld, ref_half;
};
template <typename T>
__device__ __forceinline__ int feedback_beam(const FeedbackParams<T>& p, int row) { return p.beam_idx ? p.beam_idx[row] : row; }
template <typename T>
__device__ __forceinline__ int feedback_ref_index(const FeedbackParams<T>& p, int k) { return k < p.n ? k : p.ref_half + (k - p.n); }
// BM x BN outputs per 256-thread workgroup; the 4 waves form a WR x (4/WR) grid, each wave owning
// (BM/WR) x (BN/WC) outputs = TM x TN MFMA tiles of 16 x 16.  K advances in steps of BK through two LDS
// stages (one barrier per step).  The global loads of step s+1 are issued BEFORE the MFMAs of step s and
// only touched (negated, masked, stored to LDS) AFTER them, so they fly during the matrix work; every load is
// unconditional (rows / columns out of range read a clamped address and are zeroed by a select) -- a load
// under a per-lane branch would be followed by s_waitcnt vmcnt(0) inside the branch.  The reduced-index ->
// state-offset table sits in LDS.
template <typename T, int BM, int BN, int BK, int WR, bool HAS_REF>
__global__ void __launch_bounds__(256) crb_feedback_kernel(const FeedbackParams<T> p) {
    typedef typename MfmaOps<T>::acc_t crb_d4;
    constexpr int WC = 4 / WR, TM = BM / (16 * WR), TN = BN / (16 * WC), LD = BK + 2;
    constexpr int QA = BM * BK / 256, QB = BN * BK / 256, RSTEP = 256 / BK;
    static_assert(BM % (16 * WR) == 0 && BN % (16 * WC) == 0 && 256 % BK == 0 && BM % RSTEP == 0 && BN % RSTEP == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    T* const As = reinterpret_cast<T*>(crb_smem);          // [2][BM * LD]
    T* const Bs = As + 2 * BM * LD;                             // [2][BN * LD]
    int32_t* const coff_s = reinterpret_cast<int32_t*>(Bs + 2 * BN * LD);  // [n2]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int wm = (wave / WC) * (BM / WR), wn = (wave % WC) * (BN / WC);
    for (int k = t; k < p.n2; k += 256) coff_s[k] = p.col_off[k];
    crb_d4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = crb_d4{T(0), T(0), T(0), T(0)};

    // loader: this thread fetches column lk of rows lr + RSTEP*q of both tiles
    const int lk = t & (BK - 1), lr = t / BK;
    const T* xrow[QA];
    const T* rrow[QA];
    const T* grow[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int b = m0 + lr + RSTEP * q;
        const int bc = feedback_beam(p, b < p.B ? b : p.B - 1);
        xrow[q] = p.xs + size_t(bc) * p.x_stride;
        rrow[q] = HAS_REF ? p.ref + size_t(bc) * p.ref_ld : nullptr;
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
    struct Regs { T xa[QA], ra[QA], gb[QB]; T kmask; };
    Regs R0, R1;
    auto fetch = [&](Regs& R, int k0) {
        const int k = k0 + lk;
        const bool kok = k < p.n2;
        R.kmask = kok ? T(1) : T(0);
        const int kc = kok ? k : 0;
        const int coff = coff_s[kc];
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            R.xa[q] = xrow[q][coff];
            if (HAS_REF) R.ra[q] = rrow[q][feedback_ref_index(p, kc)];
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) R.gb[q] = grow[q][kc];
    };
    auto stash_piece = [&](const Regs& R, int st, int piece) {   // element `piece` of the QA + QB this thread stores
        T* A = As + st * BM * LD;
        T* Bt = Bs + st * BN * LD;
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
        const T* Aw = As + st * BM * LD + (wm + (lane & 15)) * LD + (lane >> 4);
        const T* Bw = Bs + st * BN * LD + (wn + (lane & 15)) * LD + (lane >> 4);
        T af[2][TM], bf[2][TN];   // fragments of sub-step kk+4 are read while the MFMAs of sub-step kk run
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
                    acc[a][b] = MfmaOps<T>::run(af[cur][a], bf[cur][b], acc[a][b]);
#pragma unroll
            for (int i = 0; i < PPS; ++i) stash_piece(Rcur, st ^ 1, (kk >> 2) * PPS + i);
        }
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
                const int brow = m0 + wm + 16 * a + MfmaOps<T>::row(lane, reg);
                if (brow < p.B) {
                    const int beam = feedback_beam(p, brow);
                    const T v = HAS_REF ? acc[a][b][reg] : -acc[a][b][reg];
                    T* dst = p.u + size_t(beam) * p.u_stride + roff;
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
constexpr int WS_NL = 512;       // loader threads of the wave-specialised kernel: two loader waves per SIMD behind the one matrix wave
                                 // (one: config 5 200.9 instead of 196.5 us per step)
constexpr int WS_PRIO_M = 2, WS_PRIO_L = 0;   // wave priorities of the two roles (swapped or equal: no difference measured)
template <typename T, int BN, int BK, bool HAS_REF>
__global__ void __launch_bounds__(256 + WS_NL) crb_feedback_ws_kernel(const FeedbackParams<T> p) {
    typedef typename MfmaOps<T>::acc_t crb_d4;
    constexpr int BM = 64, TN = BN / 16, LD = BK + 2, NL = WS_NL;   // NL loader threads behind the 256 matrix threads
    constexpr int QA = BM * BK / NL, QB = BN * BK / NL, RSTEP = NL / BK;
    static_assert(BN % 16 == 0 && NL % BK == 0 && BM % RSTEP == 0 && BN % RSTEP == 0, "tile shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char crb_smem[];
    T* const As = reinterpret_cast<T*>(crb_smem);          // [2][BM * LD]
    T* const Bs = As + 2 * BM * LD;                             // [2][BN * LD]
    int32_t* const coff_s = reinterpret_cast<int32_t*>(Bs + 2 * BN * LD);  // [n2]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    for (int k = t; k < p.n2; k += 256 + NL) coff_s[k] = p.col_off[k];
    const int nsteps = (p.n2 + BK - 1) / BK;
    __syncthreads();  // coff_s

    if (wave >= 4) {
        // ------------------------------------------------ loader role
        __builtin_amdgcn_s_setprio(WS_PRIO_L);
        const int lt = t - 256;
        const int lk = lt & (BK - 1), lr = lt / BK;
        const T* xrow[QA];
        const T* rrow[QA];
        const T* grow[QB];
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            const int b = m0 + lr + RSTEP * q;
            const int bc = feedback_beam(p, b < p.B ? b : p.B - 1);
            xrow[q] = p.xs + size_t(bc) * p.x_stride;
            rrow[q] = HAS_REF ? p.ref + size_t(bc) * p.ref_ld : nullptr;
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            const int i = n0 + lr + RSTEP * q;
            grow[q] = p.gain + size_t(i < p.n ? i : p.n - 1) * p.n2;
        }
        struct Regs { T xa[QA], ra[QA], gb[QB]; T kmask; };
        Regs R0, R1;
        auto fetch = [&](Regs& R, int k0) {   // unconditional loads (clamped), K tail zeroed on the gain side
            const int k = k0 + lk;
            const bool kok = k < p.n2;
            R.kmask = kok ? T(1) : T(0);
            const int kc = kok ? k : 0;
            const int coff = coff_s[kc];
#pragma unroll
            for (int q = 0; q < QA; ++q) {
                R.xa[q] = xrow[q][coff];
                if (HAS_REF) R.ra[q] = rrow[q][feedback_ref_index(p, kc)];
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) R.gb[q] = grow[q][kc];
        };
        auto stash = [&](const Regs& R, int st) {
            T* A = As + st * BM * LD;
            T* Bt = Bs + st * BN * LD;
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
    __builtin_amdgcn_s_setprio(WS_PRIO_M);
    crb_d4 acc[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[b] = crb_d4{T(0), T(0), T(0), T(0)};
    __syncthreads();                           // stage 0 ready
    for (int sidx = 0; sidx < nsteps; ++sidx) {
        const int st = sidx & 1;
        const T* Aw = As + st * BM * LD + (16 * wave + (lane & 15)) * LD + (lane >> 4);
        const T* Bw = Bs + st * BN * LD + (lane & 15) * LD + (lane >> 4);
        T af[2], bf[2][TN];               // fragments of sub-step kk+4 are read while the MFMAs of sub-step kk run
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
            for (int b = 0; b < TN; ++b) acc[b] = MfmaOps<T>::run(af[cur], bf[cur][b], acc[b]);
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
            const int brow = m0 + 16 * wave + MfmaOps<T>::row(lane, reg);
            if (brow < p.B) p.u[size_t(feedback_beam(p, brow)) * p.u_stride + roff] = HAS_REF ? acc[b][reg] : -acc[b][reg];
        }
    }
}
template <typename T, int BM, int BN, int BK>
__host__ __device__ constexpr size_t feedback_lds_bytes(int n2) {
    return size_t(2) * (BM + BN) * (BK + 2) * sizeof(T) + size_t(n2) * sizeof(int32_t);
}


// ------------------------------------------------------------------ layout conversion
// reduced [B][rows*n_free] <-> device [B][rows][n_node][4]; free_index[r] = 3*node + dof
template <typename T, bool PACK>
__global__ void crb_pack_kernel(const int32_t* free_index, size_t fi_stride, int n_free, int n_node, int rows, int B,
                                const T* src_red, T* dev, T* dst_red) {
    // fi_stride != 0: per-beam reduced -> full maps [B][n_free], -1 = padding of a beam with fewer free DOFs than the
    // ensemble's largest (mixed boundary conditions / lengths): ignored on pack, zero on unpack
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t per_beam = size_t(rows) * n_free;
    if (i >= size_t(B) * per_beam) return;
    const size_t b = i / per_beam, rem = i - b * per_beam;
    const int row = int(rem / n_free), r = int(rem - size_t(row) * n_free);
    const int fi = free_index[b * fi_stride + r];
    if (fi < 0) {
        if (!PACK) dst_red[i] = T(0);
        return;
    }
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
