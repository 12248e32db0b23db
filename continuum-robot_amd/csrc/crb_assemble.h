// crb_assemble.h -- plan-time device assembly and cyclic-reduction factorisation of the mass matrix.
#pragma once
#include <hip/hip_runtime.h>

#include "crb_generic.h"

namespace crb {

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
    const uint8_t* free_dof;    // [3*n_node]                 (+ beam * free_stride: per-beam boundary conditions)
    const double* wet; const double* cd;  // [n_elem] or null
    const GravTab* grav;        // [S] index table (host-built topology)   (+ beam * grav_stride)
    double fluid_density;
    uint32_t flags;
    int n_elem, n_node, off, S, levels_full;
    // per-beam topology / force parameters of a mixed ensemble (SURVEY f-3); null / 0 = the shared values above.
    // A beam shorter than the plan (n_elem_b[beam] < n_elem) ends in padding nodes: every DOF constrained
    // (free_dof = 0), no element, identity mass block -- they take no part in anything.
    const int32_t* n_elem_b;    // [nb]
    const double* fluid_density_b;  // [nb]
    const uint32_t* flags_b;    // [nb]
    size_t free_stride, grav_stride;
    // alpha != 0: factorise A = M + alpha K0 (K0 = linear element stiffness) instead of M -- the iteration matrix of
    // the implicit stepper (crb_stiff.h); slot_out is then null (the slot tables are the plan's)
    double alpha;
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
    const int node = j + p.off, ne = p.n_elem_b ? p.n_elem_b[beam] : p.n_elem, nn = p.n_node;
    const uint8_t* free_dof = p.free_dof + size_t(beam) * p.free_stride;
    const uint32_t flags = p.flags_b ? p.flags_b[beam] : p.flags;
    const double fluid_density = p.fluid_density_b ? p.fluid_density_b[beam] : p.fluid_density;
    auto fm = [&](int nd, int c) { return nd >= 0 && nd < nn && free_dof[3 * nd + c] != 0; };
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
        if (e >= 0 && e < ne) {
            kind = pNl[e] ? KIND_NONLINEAR : KIND_LINEAR;
            elem_coef_build<T>(sc.elem, kind, pL[e], pE[e], pI[e], pA[e]);
        } else {
            elem_coef_build<T>(sc.elem, KIND_NONE, 1.0, 1.0, 1.0, 1.0);
        }
        for (int c = 0; c < 3; ++c) sc.mask[c] = fm(node, c) ? T(1) : T(0);
        sc.pad0 = T(0);
        sc.drag = T(0);
        if ((flags & 1u) && fm(node, 1)) {
            const int row = node < ne ? node : ne - 1;
            sc.drag = T(0.5 * fluid_density * pCd[row] * pWet[row]);
        }
        sc.half_mass = ((flags & 2u) && j < ne) ? T(0.5 * (pRho[j] * pA[j] * pL[j])) : T(0);
        sc.grav = p.grav[size_t(beam) * p.grav_stride + j];
        if (p.slot_out) static_cast<SlotConst<T>*>(p.slot_out)[tab + j] = sc;

        // ---- mass matrix, node-block form, with the boundary-condition masks
        const int el = node - 1, er = node;
        if (el >= 0 && el < ne) mass_add_as_left_elem(cur, pL[el], pRho[el] * pA[el], j >= 1);
        if (er < ne) mass_add_as_right_elem(cur, pL[er], pRho[er] * pA[er]);
        if (p.alpha != 0.0) {
            if (el >= 0 && el < ne) stiff_add_as_left_elem(cur, pL[el], pE[el] * pA[el], pE[el] * pI[el], p.alpha, j >= 1);
            if (er < ne)   // (the shipped nonlinear f1 has no u2 coupling unless CRB_CORRECTED_AXIAL)
                stiff_add_as_right_elem(cur, pL[er], pE[er] * pA[er], pE[er] * pI[er], p.alpha, !pNl[er] || (flags & 4u) != 0);
        }
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

}  // namespace crb
