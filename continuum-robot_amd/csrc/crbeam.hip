// crbeam.hip -- C ABI of libcrbeam.so (include/crbeam.h): plan construction on the host,
// kernel launches on gfx950.  Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/crbeam.h"
#include "crb_kernels.h"
#include "crb_lean_launch.h"
#include "crb_loop_launch.h"
#include "crb_ctrl_launch.h"

using namespace crb;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) return fail(CRB_EHIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// ---- device-side assembly (crb_assemble_kernel) -------------------------------------------------
template <typename X>
struct DevBuf {
    X* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return hipMalloc(reinterpret_cast<void**>(&p), (n ? n : 1) * sizeof(X)) == hipSuccess ? 0 : -1; }
    int upload(const X* h, size_t n) {
        if (alloc(n)) return -1;
        return (!n || hipMemcpy(p, h, n * sizeof(X), hipMemcpyHostToDevice) == hipSuccess) ? 0 : -1;
    }
};
// The assembly kernel's inputs stay on the device for the life of the plan: the implicit stepper factorises
// A = M + alpha K0 from them for every new step size (crb_step_implicit).
struct AsmInputs {
    DevBuf<double> dL, dE, dI, dRho, dA, dWet, dCd, dFd, dNormScratch;
    DevBuf<uint8_t> dNl, dFree;
    DevBuf<crb::GravTab> dGrav;
    DevBuf<int32_t> dNe;
    DevBuf<uint32_t> dFl;
    crb::AsmParams a;   // the launch parameters of the plan's own assembly (pointers into the buffers above)
    int nd = 1, threads = 64;
    size_t smem = 0;
};

struct crb_plan {
    int device = -1, dtype = CRB_F64, B = 0;
    int n_elem = 0, n_node = 0, n_free = 0, off = 0, S = 0, G = 1, NT = 64;
    int levels = 0, levels_full = 0, lognw = 0;
    size_t slot_stride = 0, lv_stride = 0, fin_stride = 0;  // per-beam coefficient tables (0 = shared)
    bool canonical_gravity = false;  // gravity index table is the nearest-neighbour pattern of the plain cantilever
    uint32_t flags = 0;
    int elem_mode = 0;           // EM_* (crb_generic.h)
    double gx = 0, gy = 0;
    std::vector<int32_t> free_index;  // reduced -> full   (beam 0; every beam of a uniform-topology plan)
    std::vector<int32_t> full2red;    // full -> reduced or -1
    // mixed ensembles (SURVEY f-3): beams that differ in n_elem / boundary conditions / force parameters
    bool mixed_topology = false;      // the beams' free-DOF sets differ: reduced vectors are padded to n_free (the maximum)
    std::vector<int32_t> beam_n_elem, beam_n_free;        // [B] (empty for uniform plans)
    std::vector<std::vector<int32_t>> beam_free_index;    // [B][n_free_b] (empty for uniform plans)
    std::vector<uint8_t> any_free;    // [3 n_node]: DOF free in at least one beam
    size_t free_index_stride = 0;     // d_free_index is [B][n_free] (padded with -1) when != 0
    void* d_gvec = nullptr;           // [B][2] per-beam gravity vector in the plan dtype, or null (shared gx, gy)
    // implicit stepper: cyclic-reduction tables of A = M + alpha K0 for the step size last used (all levels: A is not
    // as diagonally dominant as M, nothing is truncated)
    AsmInputs* asm_in = nullptr;
    mutable void* d_alevels = nullptr;   // [nd][levels_full][S][10]   (the set in use: one of `stiff_sets`)
    mutable void* d_afinal = nullptr;    // [nd][S][6]
    mutable double stiff_alpha = 0.0;    // alpha the tables above were built for (0 = none yet)
    mutable int stiff_levels = 0;        // reduction levels of A that the implicit kernels run (the rest are below roundoff)
    // the table sets of the step sizes used last (a step-size controller alternates between h and h / 2: without this
    // every change would factorise again and synchronise the stream); the oldest set is overwritten
    struct StiffSet { void* lev = nullptr; void* fin = nullptr; double alpha = 0.0; int levels = 0; unsigned long long used = 0; };
    static constexpr int N_STIFF_SETS = 4;
    mutable StiffSet stiff_sets[N_STIFF_SETS];
    mutable unsigned long long stiff_clock = 0;
    // crb_solve_controlled: the table LADDERS of the implicit scheme (A = M + h^2/4 K0 for h = len / 2^r, r < rungs),
    // one per distinct piece length, the most recently used kept; the piece list of the last call
    struct Ladder { double len = 0.0; int rungs = 0; void* lev = nullptr; void* fin = nullptr; unsigned long long used = 0; };
    static constexpr int N_LADDERS = 6;
    mutable Ladder ladders[N_LADDERS];
    mutable void* d_pieces = nullptr;
    mutable size_t pieces_cap = 0;
    mutable void* d_rhs0 = nullptr;      // [B][2][n_node][4]: the RHS at the start of a crb_step_implicit_damped call (its a_0)
    // host-vector entry points (crb_rhs_host): full -> reduced map on the device, pinned staging, a stream of the plan's own
    mutable int32_t* d_red_map = nullptr;
    mutable double* h_stage = nullptr;   // pinned + mapped: [2n | n | 2n] doubles (x, u, out)
    mutable double* d_stage = nullptr;   // the device's view of h_stage
    mutable unsigned long long host_seq = 0;   // sequence number of the last flagged host-path launch
    mutable hipStream_t host_stream = nullptr;
    double host_spin_ms = 200.0;      // CRB_HOST_SPIN_MS at plan creation: how long a host-path call spins on the completion flag
    bool host_sync = false;           // CRB_HOST_SYNC at plan creation: wait for the stream instead of the flag
    int32_t* d_n_state = nullptr;     // [B] 2 * n_free_b, or null
    std::vector<double> h_levels, h_final, h_norms, h_mass, h_stiff;
    int first_nonlinear = -1;
    std::vector<crb::SlotConst<double>> h_slots;
    std::vector<int> h_kinds;
    // device
    void* d_slot = nullptr;
    void* d_levels = nullptr;
    void* d_final = nullptr;
    int32_t* d_free_index = nullptr;
    int32_t* d_col_off = nullptr;  // [2n] reduced state index -> offset in a beam's state record
    int32_t* d_row_off = nullptr;  // [n]  reduced position index -> offset in a beam's force record
    // crb_step_rk4_feedback replays one captured RK4 step (8 launches + clock) as a hipGraph on a stream of
    // its own (the caller's stream may be the legacy default stream, which cannot be captured)
    mutable hipStream_t aux_stream = nullptr;
    mutable hipEvent_t aux_in = nullptr, aux_out = nullptr;
    mutable hipGraphExec_t step_exec = nullptr;
    mutable std::vector<uint64_t> step_key;
    // crb_feedback_force_grouped: device tables of the last beam -> group assignment (beam lists, reduced -> layout offsets)
    struct GainGroup { int32_t* beam_idx = nullptr; int32_t* col = nullptr; int32_t* row = nullptr; int n = 0, count = 0; };
    mutable std::vector<GainGroup> gain_groups;
    mutable std::vector<int32_t> gain_group_key;
    mutable int32_t* d_status = nullptr;      // caller's per-beam status words (crb_plan_set_status), or null
    mutable long long status_steps = 0;       // steps the ensemble has taken since the status buffer was set
    mutable bool loop_used = false;   // the last crb_step_rk4_feedback ran the persistent stepper (its work buffer holds a status word)
};

extern "C" int crb_version(void) { return CRB_VERSION; }
extern "C" const char* crb_last_error(void) { return g_err.c_str(); }

namespace {

void mass_from_blocks(crb_plan* p, const std::vector<NodeBlocks>& blk) {
    const int n = int(p->free_index.size()), S = p->S;   // (beam 0's reduced size; n_free is the ensemble's maximum)
    p->h_mass.assign(size_t(n) * n, 0.0);
    auto put = [&](int fr, int fc, double v) {
        const int r = p->full2red[fr], c = p->full2red[fc];
        if (r >= 0 && c >= 0) p->h_mass[size_t(r) * n + c] += v;
    };
    for (int j = 0; j < S; ++j) {
        const int f0 = 3 * (j + p->off);
        const NodeBlocks& b = blk[j];
        put(f0, f0, b.b_ax);
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 2; ++c) put(f0 + 1 + r, f0 + 1 + c, b.B[2 * r + c]);
        if (j + 1 < S) {
            put(f0, f0 + 3, b.c_ax);
            put(f0 + 3, f0, b.c_ax);
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 2; ++c) {
                    put(f0 + 1 + r, f0 + 4 + c, b.C[2 * r + c]);
                    put(f0 + 4 + c, f0 + 1 + r, b.C[2 * r + c]);
                }
        }
    }
}

int pick_levels(crb_plan* p) {
    // a level whose multipliers are below the unit roundoff of the plan dtype cannot change a
    // result by more than half an ulp: the reduction stops there (exact to rounding)
    const double tol = (p->dtype == CRB_F64) ? std::ldexp(1.0, -53) : std::ldexp(1.0, -24);
    int used = p->levels_full;
    while (used > 0 && p->h_norms[used - 1] < tol) --used;
    p->levels = used;
    return used;
}

// Runs crb_assemble_kernel on the plan's device (one workgroup per described beam) and fills both
// the device tables the steppers load and, for beam 0, the host copies the crb_plan_get_* inspectors
// return.  nd == 1: coefficients shared by all beams of the plan; nd == n_beams: per-beam coefficients.
// Integer topology of one beam inside a plan of n_node nodes (built on the host): free-DOF flags (padding nodes
// beyond the beam's own last node are fully constrained), the reduced <-> full maps and the gravity index table.
struct BeamTopo {
    int n_elem = 0, n_free = 0;
    bool canonical_gravity = false;
    std::vector<uint8_t> free_dof;      // [3 n_node]
    std::vector<int32_t> full2red, free_index;
    std::vector<GravTab> grav;          // [S]
};

template <typename T>
int device_assemble(crb_plan* p, const crb_beam_desc* descs, int nd, const std::vector<SlotConst<double>>& slots,
                    const std::vector<const BeamTopo*>& topo /* [nd] */, bool per_beam_topo) {
    const int S = p->S, ne = p->n_elem, lf = p->levels_full, nn = p->n_node;
    const crb_beam_desc* d = descs;
    p->asm_in = new AsmInputs();
    AsmInputs& in = *p->asm_in;
    DevBuf<double>&dL = in.dL, &dE = in.dE, &dI = in.dI, &dRho = in.dRho, &dA = in.dA, &dWet = in.dWet, &dCd = in.dCd, &dFd = in.dFd;
    DevBuf<uint8_t>&dNl = in.dNl, &dFree = in.dFree;
    DevBuf<GravTab>& dGrav = in.dGrav;
    DevBuf<int32_t>& dNe = in.dNe;
    DevBuf<uint32_t>& dFl = in.dFl;
    DevBuf<double> dFinAll, dNorms, dBlocks, dLv64;
    const int nt = per_beam_topo ? nd : 1;   // beams whose integer topology is uploaded
    std::vector<GravTab> grav(size_t(nt) * S);
    std::vector<uint8_t> free_dof(size_t(nt) * nn * 3);
    for (int b = 0; b < nt; ++b) {
        std::memcpy(&grav[size_t(b) * S], topo[b]->grav.data(), size_t(S) * sizeof(GravTab));
        std::memcpy(&free_dof[size_t(b) * nn * 3], topo[b]->free_dof.data(), size_t(nn) * 3);
    }
    (void)slots;
    bool drag = false;
    for (int b = 0; b < nd; ++b) drag = drag || (descs[b].flags & CRB_FORCE_DRAG);
    // element columns, [nd][n_elem]; a beam shorter than the plan is padded (1.0: never read by an element)
    auto gather = [&](const double* crb_beam_desc::*col) {
        std::vector<double> v(size_t(nd) * ne, 1.0);
        for (int b = 0; b < nd; ++b)
            if (descs[b].*col) std::memcpy(&v[size_t(b) * ne], descs[b].*col, size_t(descs[b].n_elem) * sizeof(double));
        return v;
    };
    std::vector<uint8_t> nl(size_t(nd) * ne, 0);
    for (int b = 0; b < nd; ++b) std::memcpy(&nl[size_t(b) * ne], descs[b].nonlinear, size_t(descs[b].n_elem));
    std::vector<int32_t> ne_b(nd);
    std::vector<uint32_t> fl_b(nd);
    std::vector<double> fd_b(nd);
    for (int b = 0; b < nd; ++b) { ne_b[b] = descs[b].n_elem; fl_b[b] = descs[b].flags; fd_b[b] = descs[b].fluid_density; }
    const auto vL = gather(&crb_beam_desc::length), vE = gather(&crb_beam_desc::elastic_modulus),
               vI = gather(&crb_beam_desc::moment_inertia), vR = gather(&crb_beam_desc::density),
               vA = gather(&crb_beam_desc::cross_area);
    std::vector<double> vW, vC;
    if (drag) { vW = gather(&crb_beam_desc::wetted_area); vC = gather(&crb_beam_desc::drag_coef); }
    if (dL.upload(vL.data(), vL.size()) || dE.upload(vE.data(), vE.size()) || dI.upload(vI.data(), vI.size()) ||
        dRho.upload(vR.data(), vR.size()) || dA.upload(vA.data(), vA.size()) || dNl.upload(nl.data(), nl.size()) ||
        dFree.upload(free_dof.data(), free_dof.size()) || dGrav.upload(grav.data(), grav.size()) ||
        dNe.upload(ne_b.data(), nd) || dFl.upload(fl_b.data(), nd) || dFd.upload(fd_b.data(), nd) ||
        (drag && (dWet.upload(vW.data(), vW.size()) || dCd.upload(vC.data(), vC.size()))) ||
        dFinAll.alloc(size_t(lf + 1) * S * PCR_FINAL_VALS) || dNorms.alloc(lf) || dBlocks.alloc(size_t(S) * 15) ||
        dLv64.alloc(size_t(lf) * S * PCR_LEVEL_VALS))
        return fail(CRB_EHIP, "crb_plan_create: device allocation/upload failed");
    HIP_TRY(hipMalloc(&p->d_slot, size_t(nd) * S * sizeof(SlotConst<T>)));
    HIP_TRY(hipMalloc(&p->d_levels, size_t(nd) * size_t(lf > 0 ? lf : 1) * S * PCR_LEVEL_VALS * sizeof(T)));
    HIP_TRY(hipMalloc(&p->d_final, size_t(nd) * S * PCR_FINAL_VALS * sizeof(T)));
    if (p->mixed_topology) {   // per-beam reduced -> full maps, padded with -1 to the largest beam
        std::vector<int32_t> fi(size_t(nd) * p->n_free, -1), nst(nd);
        for (int b = 0; b < nd; ++b) {
            std::memcpy(&fi[size_t(b) * p->n_free], topo[b]->free_index.data(), topo[b]->free_index.size() * sizeof(int32_t));
            nst[b] = 2 * topo[b]->n_free;
        }
        p->free_index_stride = size_t(p->n_free);
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_free_index), fi.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(p->d_free_index, fi.data(), fi.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_n_state), nst.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(p->d_n_state, nst.data(), nst.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    } else {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_free_index), p->free_index.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(p->d_free_index, p->free_index.data(), p->free_index.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    {   // per-beam gravity vectors (only when they differ)
        bool differ = false;
        for (int b = 1; b < nd; ++b) differ = differ || descs[b].gravity[0] != d->gravity[0] || descs[b].gravity[1] != d->gravity[1];
        if (differ) {
            std::vector<T> gv(size_t(nd) * 2);
            for (int b = 0; b < nd; ++b) { gv[2 * b] = T(descs[b].gravity[0]); gv[2 * b + 1] = T(descs[b].gravity[1]); }
            HIP_TRY(hipMalloc(&p->d_gvec, gv.size() * sizeof(T)));
            HIP_TRY(hipMemcpy(p->d_gvec, gv.data(), gv.size() * sizeof(T), hipMemcpyHostToDevice));
        }
    }
    {   // offsets of the reduced ordering inside the device layouts (crb_feedback_force)
        const int n = p->n_free;
        std::vector<int32_t> col(size_t(2) * n), row(n);
        for (int r = 0; r < n; ++r) {
            const int f = r < int(p->free_index.size()) ? p->free_index[r] : 0;   // (mixed plans: beam 0 may be shorter)
            row[r] = (f / 3) * 4 + (f % 3);
            col[r] = row[r];
            col[n + r] = p->n_node * 4 + row[r];
        }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_col_off), col.size() * sizeof(int32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_row_off), row.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(p->d_col_off, col.data(), col.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_row_off, row.data(), row.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    if (nd > 1) {
        p->slot_stride = size_t(S);
        p->lv_stride = size_t(lf) * S * PCR_LEVEL_VALS;
        p->fin_stride = size_t(S) * PCR_FINAL_VALS;
    }

    AsmParams a;
    std::memset(&a, 0, sizeof(a));
    a.L = dL.p; a.E = dE.p; a.I = dI.p; a.rho = dRho.p; a.A = dA.p;
    a.nonlinear = dNl.p; a.free_dof = dFree.p; a.wet = dWet.p; a.cd = dCd.p; a.grav = dGrav.p;
    a.fluid_density = d->fluid_density; a.flags = d->flags;
    a.n_elem = ne; a.n_node = p->n_node; a.off = p->off; a.S = S; a.levels_full = lf;
    if (nd > 1) { a.n_elem_b = dNe.p; a.fluid_density_b = dFd.p; a.flags_b = dFl.p; }
    a.free_stride = per_beam_topo ? size_t(nn) * 3 : 0;
    a.grav_stride = per_beam_topo ? size_t(S) : 0;
    a.slot_out = p->d_slot; a.lv64 = dLv64.p; a.lvT = p->d_levels; a.fin64_all = dFinAll.p; a.norms = dNorms.p;
    a.blocks0 = dBlocks.p; a.finT = nullptr; a.fin_level = -1;
    a.elem_stride = nd > 1 ? size_t(ne) : 0;
    const int nta = (S + 63) / 64 * 64;
    const size_t smem = size_t(S) * sizeof(NodeBlocks);
    if (smem > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&crb_assemble_kernel<T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, int(smem)));
    // pass 1: constants, multipliers of every level, their norms (max over beams)
    HIP_TRY(hipMemset(dNorms.p, 0, size_t(lf ? lf : 1) * sizeof(double)));
    hipLaunchKernelGGL((crb_assemble_kernel<T>), dim3(nd), dim3(nta), smem, nullptr, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    p->h_levels.assign(size_t(lf > 0 ? lf : 1) * S * PCR_LEVEL_VALS, 0.0);
    p->h_norms.assign(size_t(lf > 0 ? lf : 1), 0.0);
    if (lf > 0) {
        HIP_TRY(hipMemcpy(p->h_levels.data(), dLv64.p, size_t(lf) * S * PCR_LEVEL_VALS * sizeof(double), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(p->h_norms.data(), dNorms.p, size_t(lf) * sizeof(double), hipMemcpyDeviceToHost));
    }
    const int used = pick_levels(p);
    if (used > MAX_LV)
        return fail(CRB_EUNSUPPORTED, "mass matrix needs more cyclic-reduction levels than the kernels carry in registers");
    in.a = a; in.nd = nd; in.threads = nta; in.smem = smem;   // (kept for the implicit stepper's factorisations)
    // pass 2: the final inverses after `used` levels, straight into the steppers' table
    a.finT = p->d_final; a.fin_level = used; a.lv64 = nullptr; a.blocks0 = nullptr; a.fin64_all = nullptr;
    hipLaunchKernelGGL((crb_assemble_kernel<T>), dim3(nd), dim3(nta), smem, nullptr, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());

    // ---- inspection copies (beam 0)
    p->h_final.assign(size_t(S) * PCR_FINAL_VALS, 0.0);
    HIP_TRY(hipMemcpy(p->h_final.data(), dFinAll.p + size_t(used) * S * PCR_FINAL_VALS,
                      size_t(S) * PCR_FINAL_VALS * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<double> b0(size_t(S) * 15);
    HIP_TRY(hipMemcpy(b0.data(), dBlocks.p, b0.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<NodeBlocks> blk(S);
    for (int j = 0; j < S; ++j) {
        const double* b = &b0[size_t(j) * 15];
        blk[j].a_ax = b[0]; blk[j].b_ax = b[1]; blk[j].c_ax = b[2];
        for (int k = 0; k < 4; ++k) { blk[j].A[k] = b[3 + k]; blk[j].B[k] = b[7 + k]; blk[j].C[k] = b[11 + k]; }
    }
    mass_from_blocks(p, blk);
    std::vector<SlotConst<T>> hs(S);
    HIP_TRY(hipMemcpy(hs.data(), p->d_slot, size_t(S) * sizeof(SlotConst<T>), hipMemcpyDeviceToHost));
    p->h_slots.assign(S, SlotConst<double>());
    p->h_kinds.assign(S, KIND_NONE);
    for (int j = 0; j < S; ++j) {
        SlotConst<double>& o = p->h_slots[j];
        std::memset(&o, 0, sizeof(o));
        o.drag = double(hs[j].drag);
        o.half_mass = double(hs[j].half_mass);
        for (int c = 0; c < 3; ++c) o.mask[c] = double(hs[j].mask[c]);
        o.grav = hs[j].grav;
        p->h_kinds[j] = hs[j].elem.kind;
    }
    return CRB_OK;
}

}  // namespace

static int plan_create_impl(crb_plan** out, int device, int dtype, int n_beams, const crb_beam_desc* d, int nd);

extern "C" int crb_plan_create(crb_plan** out, int device, int dtype, int n_beams, const crb_beam_desc* d) {
    return plan_create_impl(out, device, dtype, n_beams, d, 1);
}

extern "C" int crb_plan_create_ensemble(crb_plan** out, int device, int dtype, int n_beams, const crb_beam_desc* descs) {
    if (!out || !descs) return fail(CRB_EINVAL, "crb_plan_create_ensemble: null argument");
    if (n_beams < 1) return fail(CRB_EINVAL, "n_beams must be >= 1");
    if (device < 0) return fail(CRB_EUNSUPPORTED, "per-beam coefficient plans are built on the device (no host-only form)");
    const crb_beam_desc& a = descs[0];
    for (int b = 1; b < n_beams; ++b)   // the one thing an ensemble must share: the element variant compiled into the kernels
        if ((descs[b].flags & CRB_CORRECTED_AXIAL) != (a.flags & CRB_CORRECTED_AXIAL))
            return fail(CRB_EINVAL, "ensemble beams must share the CRB_CORRECTED_AXIAL option");
    return plan_create_impl(out, device, dtype, n_beams, descs, n_beams);
}

namespace {
// CSV-row validation of one beam: Properties.__post_init__ / EulerBernoulliBeam._validate_parameters
// (abstractions.py:39-53, euler_bernoulli_beam.py:100-103), ForceParams / FluidDragForce (force_params.py:41-43,
// fluid_forces.py:59-72).  Returns nullptr or the reference's message.
const char* validate_desc(const crb_beam_desc& o) {
    if (o.n_elem < 1) return "n_elem must be >= 1";
    if (!o.length || !o.elastic_modulus || !o.moment_inertia || !o.density || !o.cross_area || !o.nonlinear || !o.node_bc)
        return "beam description has null columns";
    for (int e = 0; e < o.n_elem; ++e) {
        if (!(o.length[e] > 0) || !(o.elastic_modulus[e] > 0) || !(o.moment_inertia[e] > 0) || !(o.density[e] > 0) ||
            !(o.cross_area[e] > 0))
            return "All numeric parameters must be positive";
        if (o.nonlinear[e] > 1) return "Invalid element type";
    }
    for (int i = 0; i <= o.n_elem; ++i)
        if (o.node_bc[i] != CRB_BC_NONE && o.node_bc[i] != CRB_BC_FIXED && o.node_bc[i] != CRB_BC_PINNED)
            return "Unsupported boundary condition type";
    if (o.flags & CRB_FORCE_DRAG) {
        if (!o.wetted_area || !o.drag_coef) return "drag enabled but wetted_area/drag_coef missing";
        if (!(o.fluid_density > 0)) return "fluid_density must be positive when fluid effects are enabled";
        for (int e = 0; e < o.n_elem; ++e) {
            if (o.drag_coef[e] < 0) return "Drag coefficients cannot be negative";
            if (o.wetted_area[e] < 0) return "Wetted areas cannot be negative";
        }
    }
    return nullptr;
}

// Integer topology of a beam of d.n_elem elements inside a plan of nn nodes whose first `off` nodes carry no slot:
// boundary conditions -> free masks and the reduced ordering (euler_bernoulli_beam.py:240-259); the gravity index
// table with the reference's reduced-index addressing (gravity_forces.py:97-146).
BeamTopo build_topology(const crb_beam_desc& d, int nn, int off) {
    BeamTopo t;
    const int ne = d.n_elem, S = nn - off;
    t.n_elem = ne;
    t.free_dof.assign(size_t(nn) * 3, 0);    // nodes past the beam's own last node (index ne) stay fully constrained
    for (int i = 0; i <= ne; ++i) {
        const int bc = d.node_bc[i];
        const bool fx = bc == CRB_BC_FIXED, pin = bc == CRB_BC_PINNED;
        t.free_dof[3 * i] = t.free_dof[3 * i + 1] = (fx || pin) ? 0 : 1;
        t.free_dof[3 * i + 2] = fx ? 0 : 1;
    }
    t.full2red.assign(size_t(nn) * 3, -1);
    for (int f = 0; f < 3 * nn; ++f)
        if (t.free_dof[f]) { t.full2red[f] = int32_t(t.free_index.size()); t.free_index.push_back(f); }
    t.n_free = int(t.free_index.size());
    const int n = t.n_free;
    t.grav.resize(S);
    const bool grav = d.flags & CRB_FORCE_GRAVITY;
    for (int j = 0; j < S; ++j) {
        GravTab& gt = t.grav[j];
        gt.phiA = gt.phiB = -1;
        for (int c = 0; c < 3; ++c) { gt.segA[c] = gt.segB[c] = -1; gt.comp[c] = 0; }
        gt.pad = 0;
        if (!grav) continue;
        const int node = j + off;
        auto enc = [&](int red) -> int16_t {
            const int f = t.free_index[red];
            return int16_t(((f / 3) - off) * 4 + (f % 3));
        };
        if (j < ne) {  // this thread evaluates segment j (gravity_forces.py:97-128)
            const int sp = 3 * j + 2, ep = 3 * (j + 1) + 2;  // indices into the REDUCED vector
            if (sp < n) gt.phiA = enc(sp);
            if (ep < n) gt.phiB = enc(ep);
        }
        for (int c = 0; c < 3; ++c) {  // which segments add to this DOF (gravity_forces.py:130-146)
            const int r = t.full2red[3 * node + c];
            if (r < 0 || r % 3 == 2) continue;
            gt.comp[c] = int8_t(r % 3);
            if (r / 3 < ne) gt.segA[c] = int16_t(r / 3);
            if (r / 3 - 1 >= 0) gt.segB[c] = int16_t(r / 3 - 1);
        }
    }
    // does the gravity table reduce to "segment j <-> slots j, j+1" (the plain cantilever filling the whole plan)?
    t.canonical_gravity = false;
    if (grav && off == 1 && ne == S) {
        bool canon = true;
        for (int j = 0; j < S && canon; ++j) {
            const GravTab& gt = t.grav[j];
            canon = gt.phiA == int16_t(j * 4 + 2) && gt.phiB == (j + 1 < S ? int16_t((j + 1) * 4 + 2) : int16_t(-1)) &&
                    gt.segA[2] < 0 && gt.segB[2] < 0;
            for (int c = 0; c < 2 && canon; ++c)
                canon = gt.comp[c] == c && gt.segA[c] == int16_t(j) && gt.segB[c] == int16_t(j - 1 >= 0 ? j - 1 : -1);
        }
        t.canonical_gravity = canon;
    }
    return t;
}
}  // namespace

static int plan_create_impl(crb_plan** out, int device, int dtype, int n_beams, const crb_beam_desc* d, int nd) {
    if (!out || !d) return fail(CRB_EINVAL, "crb_plan_create: null argument");
    *out = nullptr;
    if (dtype != CRB_F64 && dtype != CRB_F32) return fail(CRB_EINVAL, "dtype must be CRB_F64 or CRB_F32");
    if (n_beams < 1) return fail(CRB_EINVAL, "n_beams must be >= 1");
    for (int b = 0; b < nd; ++b)
        if (const char* msg = validate_desc(d[b])) return fail(CRB_EINVAL, msg);
    // the plan's node count is the longest beam's; shorter beams end in padding nodes (fully constrained, no element)
    int ne = 0;
    bool all_fixed_root = true, any_drag = false, any_grav = false;
    for (int b = 0; b < nd; ++b) {
        ne = d[b].n_elem > ne ? d[b].n_elem : ne;
        all_fixed_root = all_fixed_root && d[b].node_bc[0] == CRB_BC_FIXED;
        any_drag = any_drag || (d[b].flags & CRB_FORCE_DRAG);
        any_grav = any_grav || (d[b].flags & CRB_FORCE_GRAVITY);
    }
    const bool grav = any_grav;

    crb_plan* p = new crb_plan();
    p->device = device;
    if (const char* env = std::getenv("CRB_HOST_SPIN_MS")) p->host_spin_ms = std::atof(env);
    p->host_sync = std::getenv("CRB_HOST_SYNC") != nullptr;
    p->dtype = dtype;
    p->B = n_beams;
    p->n_elem = ne;
    p->n_node = ne + 1;
    // force terms compiled into a launch = the union over the beams; a beam without drag / gravity has zero drag
    // factors / zero segment masses in its slot table
    p->flags = (d->flags & CRB_CORRECTED_AXIAL) | (any_drag ? CRB_FORCE_DRAG : 0u) | (any_grav ? CRB_FORCE_GRAVITY : 0u);
    p->gx = d->gravity[0];
    p->gy = d->gravity[1];
    const int nn = p->n_node;
    // node 0 carries no thread slot when it is FIXED in every beam (the cantilever of the examples)
    p->off = all_fixed_root ? 1 : 0;
    p->S = nn - p->off;
    const int S = p->S;
    if (S > 1024) { delete p; return fail(CRB_EUNSUPPORTED, "more than 1024 thread-carried nodes per beam"); }

    // ---- integer topology per DISTINCT (n_elem, boundary conditions, gravity on/off) among the beams
    std::vector<BeamTopo> topos;
    std::vector<int> topo_of(nd, 0);
    for (int b = 0; b < nd; ++b) {
        int hit = -1;
        for (int c = 0; c < b && hit < 0; ++c)
            if (d[c].n_elem == d[b].n_elem && std::memcmp(d[c].node_bc, d[b].node_bc, size_t(d[b].n_elem) + 1) == 0 &&
                ((d[c].flags ^ d[b].flags) & CRB_FORCE_GRAVITY) == 0)
                hit = topo_of[c];
        if (hit < 0) { topos.push_back(build_topology(d[b], nn, p->off)); hit = int(topos.size()) - 1; }
        topo_of[b] = hit;
    }
    std::vector<const BeamTopo*> topo(nd);
    for (int b = 0; b < nd; ++b) topo[b] = &topos[topo_of[b]];
    const BeamTopo& t0 = *topo[0];
    p->free_index = t0.free_index;
    p->full2red = t0.full2red;
    p->n_free = t0.n_free;
    p->any_free = t0.free_dof;
    bool per_beam_topo = topos.size() > 1;
    for (int b = 1; b < nd; ++b) {
        if (topo[b]->free_dof != t0.free_dof) p->mixed_topology = true;
        if (topo[b]->n_free > p->n_free) p->n_free = topo[b]->n_free;
        for (size_t f = 0; f < p->any_free.size(); ++f) p->any_free[f] = p->any_free[f] | topo[b]->free_dof[f];
    }
    // per-beam sizes whenever they differ -- also when the free-DOF sets agree (a longer beam whose extra nodes are all FIXED
    // has the shorter beam's masks: its element count, and with it the tip node, is still its own)
    bool sizes_differ = p->mixed_topology;
    for (int b = 1; b < nd; ++b) sizes_differ = sizes_differ || topo[b]->n_elem != t0.n_elem;
    if (sizes_differ) {
        p->beam_n_elem.resize(nd); p->beam_n_free.resize(nd); p->beam_free_index.resize(nd);
        for (int b = 0; b < nd; ++b) {
            p->beam_n_elem[b] = topo[b]->n_elem; p->beam_n_free[b] = topo[b]->n_free; p->beam_free_index[b] = topo[b]->free_index;
        }
    }
    for (int b = 0; b < nd; ++b)
        if (topo[b]->n_free == 0) { delete p; return fail(CRB_EINVAL, "Cannot constrain all degrees of freedom"); }

    // S >= 64: one beam per workgroup of NW = 2^lognw wavefronts (slots interleaved over the waves);
    // S < 64: G = 64/S whole beams per single-wave workgroup
    p->lognw = 0;
    if (S >= 64) {
        p->G = 1;
        while ((64 << p->lognw) < S) ++p->lognw;
        p->NT = 64 << p->lognw;
    } else {
        p->G = 64 / S;
        p->NT = 64;
    }
    int lf = 0;
    while ((1 << lf) < S) ++lf;
    p->levels_full = lf;

    // ---- per-slot constants of beam 0 (host copies for the inspectors; the device tables come from crb_assemble_kernel)
    std::vector<SlotConst<double>> slots(S);
    std::vector<int> kinds(S, KIND_NONE);
    const bool drag = d->flags & CRB_FORCE_DRAG;
    for (int j = 0; j < S; ++j) {
        SlotConst<double>& s = slots[j];
        std::memset(&s, 0, sizeof(s));
        const int node = j + p->off;
        const int e = node - 1;
        if (e >= 0 && e < d->n_elem) kinds[j] = d->nonlinear[e] ? KIND_NONLINEAR : KIND_LINEAR;
        for (int c = 0; c < 3; ++c) s.mask[c] = t0.free_dof[3 * node + c] ? 1.0 : 0.0;
        if (drag && t0.free_dof[3 * node + 1]) {
            // fluid_forces.py:59-61, 87-90: the node's own segment row, last row repeated for the tip
            const int row = node < d->n_elem ? node : d->n_elem - 1;
            s.drag = 0.5 * d->fluid_density * d->drag_coef[row] * d->wetted_area[row];
        }
        s.grav = t0.grav[j];
        if ((d->flags & CRB_FORCE_GRAVITY) && j < d->n_elem) s.half_mass = 0.5 * (d->density[j] * d->cross_area[j] * d->length[j]);
    }
    p->h_slots = slots;
    p->h_kinds = kinds;
    {   // one element kind over the whole ensemble?
        bool all_nl = true, all_lin = true;
        for (int b = 0; b < nd; ++b)
            for (int e = 0; e < d[b].n_elem; ++e) { all_nl = all_nl && d[b].nonlinear[e]; all_lin = all_lin && !d[b].nonlinear[e]; }
        p->elem_mode = all_lin ? EM_LINEAR : (all_nl && !(d->flags & CRB_CORRECTED_AXIAL)) ? EM_NONLINEAR : EM_MIXED;
        if (std::getenv("CRB_DISABLE_ELEM_MODE")) p->elem_mode = EM_MIXED;
    }
    if (grav) {   // nearest-neighbour gravity (lean kernels) only if EVERY beam that has gravity is the canonical cantilever
        bool canon = true;
        for (int b = 0; b < nd; ++b)
            if (d[b].flags & CRB_FORCE_GRAVITY) canon = canon && topo[b]->canonical_gravity;
            else canon = canon && d[b].n_elem == S && p->off == 1;   // (a gravity-free beam only needs the same slot layout)
        p->canonical_gravity = canon;
    }

    const int n = t0.n_free;
    {   // dense reduced stiffness of beam 0's linear elements (get_stiffness_matrix); columns = K_e * unit vectors
        p->h_stiff.assign(size_t(n) * n, 0.0);
        for (int e = 0; e < d->n_elem; ++e) {
            if (d->nonlinear[e]) { if (p->first_nonlinear < 0) p->first_nonlinear = e; continue; }
            ElemCoef<double> ec;
            elem_coef_build<double>(ec, KIND_LINEAR, d->length[e], d->elastic_modulus[e], d->moment_inertia[e],
                                    d->cross_area[e]);
            for (int col = 0; col < 6; ++col) {
                double xe[6] = {0, 0, 0, 0, 0, 0}, fl[3], fr[3];
                xe[col] = 1.0;
                elem_force_linear<double>(ec.c, xe, xe + 3, fl, fr);
                const int rc = p->full2red[3 * e + col];
                if (rc < 0) continue;
                for (int row = 0; row < 6; ++row) {
                    const int rr = p->full2red[3 * e + row];
                    if (rr >= 0) p->h_stiff[size_t(rr) * n + rc] += row < 3 ? fl[row] : fr[row - 3];
                }
            }
        }
    }
    if (device < 0) {
    // ======== host-only plan (inspection, one shared beam): same arithmetic (crb_math.h) in plain C++ ========
    const std::vector<uint8_t>& free_dof = t0.free_dof;
    // ---- mass matrix in node-block form + cyclic-reduction factorisation (all fp64)
    std::vector<NodeBlocks> cur(S), nxt(S);
    auto fm = [&](int node, int c) { return node >= 0 && node < nn && free_dof[3 * node + c] != 0; };
    for (int j = 0; j < S; ++j) {
        NodeBlocks nb;
        std::memset(&nb, 0, sizeof(nb));
        const int node = j + p->off;
        const int el = node - 1, er = node;
        if (el >= 0) mass_add_as_left_elem(nb, d->length[el], d->density[el] * d->cross_area[el], j >= 1);
        if (er < ne) mass_add_as_right_elem(nb, d->length[er], d->density[er] * d->cross_area[er]);
        const bool hl = j >= 1, hr = j + 1 < S;
        mass_apply_masks(nb, fm(node, 0), fm(node, 1), fm(node, 2), hl && fm(node - 1, 0), hl && fm(node - 1, 1),
                         hl && fm(node - 1, 2), hr && fm(node + 1, 0), hr && fm(node + 1, 1), hr && fm(node + 1, 2));
        cur[j] = nb;
    }
    mass_from_blocks(p, cur);
    std::vector<std::vector<NodeBlocks>> states;
    states.push_back(cur);
    p->h_levels.assign(size_t(lf > 0 ? lf : 1) * S * PCR_LEVEL_VALS, 0.0);
    p->h_norms.assign(size_t(lf > 0 ? lf : 1), 0.0);
    for (int l = 0; l < lf; ++l) {
        const int s = 1 << l;
        double nm = 0.0;
        for (int j = 0; j < S; ++j) {
            PcrLevel lv;
            const bool hl = j - s >= 0, hh = j + s < S;
            pcr_factor_level(cur[j], cur[hl ? j - s : j], hl, cur[hh ? j + s : j], hh, lv, nxt[j]);
            double* o = &p->h_levels[(size_t(l) * S + j) * PCR_LEVEL_VALS];
            o[0] = lv.al_ax;
            o[1] = lv.ga_ax;
            for (int k = 0; k < 4; ++k) { o[2 + k] = lv.al[k]; o[6 + k] = lv.ga[k]; }
            const int node = j + p->off;
            const int el = node - 1 >= 0 ? node - 1 : 0;
            const double nj = pcr_level_norm(lv, d->length[el < ne ? el : ne - 1]);
            nm = nj > nm ? nj : nm;
        }
        p->h_norms[l] = nm;
        cur.swap(nxt);
        states.push_back(cur);
    }
    const int used = pick_levels(p);
    if (used > MAX_LV) {
        delete p;
        return fail(CRB_EUNSUPPORTED, "mass matrix needs more cyclic-reduction levels than the kernels carry in registers");
    }
    p->h_final.assign(size_t(S) * PCR_FINAL_VALS, 0.0);
    for (int j = 0; j < S; ++j) {
        const NodeBlocks& b = states[used][j];
        double Bi[4];
        inv2(b.B, Bi);
        double* o = &p->h_final[size_t(j) * PCR_FINAL_VALS];
        // constrained DOFs: zero row/column in the final inverse, so the kernels need no mask multiply
        const double mu = slots[j].mask[0], mw = slots[j].mask[1], mp = slots[j].mask[2];
        o[0] = mu / b.b_ax;
        o[1] = mw * mw * Bi[0];
        o[2] = mw * mp * Bi[1];
        o[3] = mp * mw * Bi[2];
        o[4] = mp * mp * Bi[3];
        o[5] = 0.0;
    }

    }
    if (device >= 0) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || device >= ndev) {
            delete p;
            return fail(CRB_ENODEV, "crb_plan_create: HIP device not available (no fallback path exists)");
        }
        // plan creation works on the plan's device and hands the caller's current device back (launch entry points
        // select the plan's device and leave it selected, as a HIP stream of that device must be current anyway)
        int prev = -1;
        (void)hipGetDevice(&prev);
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) { delete p; return fail(CRB_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e)); }
        // ======== device plan: crb_assemble_kernel builds every floating-point table ========
        const int rc = (dtype == CRB_F64) ? device_assemble<double>(p, d, nd, slots, topo, per_beam_topo)
                                          : device_assemble<float>(p, d, nd, slots, topo, per_beam_topo);
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
        if (rc != CRB_OK) { crb_plan_destroy(p); return rc; }
    }
    *out = p;
    return CRB_OK;
}

namespace {
void free_gain_groups(const crb_plan* p);
}
extern "C" void crb_plan_destroy(crb_plan* p) {
    if (!p) return;
    if (p->device >= 0) {
        if (p->aux_stream) (void)hipStreamSynchronize(p->aux_stream);
        if (p->step_exec) (void)hipGraphExecDestroy(p->step_exec);
        if (p->aux_in) (void)hipEventDestroy(p->aux_in);
        if (p->aux_out) (void)hipEventDestroy(p->aux_out);
        if (p->aux_stream) (void)hipStreamDestroy(p->aux_stream);
        (void)hipFree(p->d_slot);
        (void)hipFree(p->d_levels);
        (void)hipFree(p->d_final);
        (void)hipFree(p->d_free_index);
        free_gain_groups(p);
        (void)hipFree(p->d_col_off);
        (void)hipFree(p->d_row_off);
        (void)hipFree(p->d_gvec);
        (void)hipFree(p->d_n_state);
        for (auto& set : p->stiff_sets) { (void)hipFree(set.lev); (void)hipFree(set.fin); }
        for (auto& lad : p->ladders) { (void)hipFree(lad.lev); (void)hipFree(lad.fin); }
        (void)hipFree(p->d_pieces);
        (void)hipFree(p->d_rhs0);
        (void)hipFree(p->d_red_map);
        if (p->h_stage) (void)hipHostFree(p->h_stage);
        if (p->host_stream) (void)hipStreamDestroy(p->host_stream);
        delete p->asm_in;
    }
    delete p;
}

extern "C" int crb_plan_set_status(const crb_plan* p, void* status, long long steps_done) {
    if (!p) return fail(CRB_EINVAL, "crb_plan_set_status: null plan");
    if (steps_done < 0) return fail(CRB_EINVAL, "crb_plan_set_status: steps_done must be >= 0");
    p->d_status = static_cast<int32_t*>(status);
    p->status_steps = steps_done;
    return CRB_OK;
}

extern "C" int crb_plan_get_layout(const crb_plan* p, crb_layout* o) {
    if (!p || !o) return fail(CRB_EINVAL, "null argument");
    o->dtype = p->dtype;
    o->n_beams = p->B;
    o->n_elem = p->n_elem;
    o->n_node = p->n_node;
    o->n_free = p->n_free;
    o->node_offset = p->off;
    o->n_slots = p->S;
    o->beams_per_group = p->G;
    o->threads = p->NT;
    o->pcr_levels = p->levels;
    o->pcr_levels_full = p->levels_full;
    o->mixed_topology = p->mixed_topology ? 1 : 0;
    return CRB_OK;
}

extern "C" int crb_plan_get_free_index(const crb_plan* p, int32_t* out) {
    if (!p || !out) return fail(CRB_EINVAL, "null argument");
    std::memcpy(out, p->free_index.data(), p->free_index.size() * sizeof(int32_t));
    return CRB_OK;
}

extern "C" int crb_plan_get_beam_info(const crb_plan* p, int beam, int32_t* n_elem, int32_t* n_free) {
    if (!p) return fail(CRB_EINVAL, "null argument");
    if (beam < 0 || beam >= p->B) return fail(CRB_EINVAL, "crb_plan_get_beam_info: beam out of range");
    const bool per_beam = !p->beam_n_elem.empty();
    if (n_elem) *n_elem = per_beam ? p->beam_n_elem[beam] : p->n_elem;
    if (n_free) *n_free = per_beam ? p->beam_n_free[beam] : int32_t(p->free_index.size());
    return CRB_OK;
}

extern "C" int crb_plan_get_beam_free_index(const crb_plan* p, int beam, int32_t* out) {
    if (!p || !out) return fail(CRB_EINVAL, "null argument");
    if (beam < 0 || beam >= p->B) return fail(CRB_EINVAL, "crb_plan_get_beam_free_index: beam out of range");
    const std::vector<int32_t>& fi = !p->beam_free_index.empty() ? p->beam_free_index[beam] : p->free_index;
    std::memcpy(out, fi.data(), fi.size() * sizeof(int32_t));
    return CRB_OK;
}

extern "C" int crb_plan_get_pcr_tables(const crb_plan* p, double* levels, double* final_, double* norms) {
    if (!p) return fail(CRB_EINVAL, "null argument");
    const size_t S = size_t(p->S);
    if (levels) std::memcpy(levels, p->h_levels.data(), size_t(p->levels_full) * S * PCR_LEVEL_VALS * sizeof(double));
    if (final_) std::memcpy(final_, p->h_final.data(), S * PCR_FINAL_VALS * sizeof(double));
    if (norms) std::memcpy(norms, p->h_norms.data(), size_t(p->levels_full) * sizeof(double));
    return CRB_OK;
}

extern "C" int crb_plan_get_slot_tables(const crb_plan* p, double* drag, double* half_mass, double* mask, int16_t* grav,
                                        int32_t* elem_kind) {
    if (!p) return fail(CRB_EINVAL, "null argument");
    for (int j = 0; j < p->S; ++j) {
        const SlotConst<double>& s = p->h_slots[j];
        if (drag) drag[j] = s.drag;
        if (half_mass) half_mass[j] = s.half_mass;
        if (mask) for (int c = 0; c < 3; ++c) mask[3 * j + c] = s.mask[c];
        if (grav) {
            int16_t* g = grav + 12 * j;
            g[0] = s.grav.phiA;
            g[1] = s.grav.phiB;
            for (int c = 0; c < 3; ++c) { g[2 + c] = s.grav.segA[c]; g[5 + c] = s.grav.segB[c]; g[8 + c] = s.grav.comp[c]; }
            g[11] = 0;
        }
        if (elem_kind) elem_kind[j] = p->h_kinds[j];
    }
    return CRB_OK;
}

extern "C" int crb_plan_get_mass(const crb_plan* p, double* M) {
    if (!p || !M) return fail(CRB_EINVAL, "null argument");
    std::memcpy(M, p->h_mass.data(), p->h_mass.size() * sizeof(double));
    return CRB_OK;
}

extern "C" int crb_plan_get_stiffness(const crb_plan* p, double* K) {
    if (!p || !K) return fail(CRB_EINVAL, "null argument");
    if (p->first_nonlinear >= 0)
        return fail(CRB_EINVAL, "Cannot extract stiffness matrix from beam with nonlinear segments. Segment " +
                                    std::to_string(p->first_nonlinear) +
                                    " is nonlinear. Stiffness matrix is only valid for purely linear beams.");
    std::memcpy(K, p->h_stiff.data(), p->h_stiff.size() * sizeof(double));
    return CRB_OK;
}

// ------------------------------------------------------------------ launches
namespace {

int need_device(const crb_plan* p, const char* who) {
    if (!p) return fail(CRB_EINVAL, std::string(who) + ": null plan");
    if (p->device < 0)
        return fail(CRB_ENODEV, std::string(who) + ": host-only plan; the stepper has no CPU path, a HIP device is required");
    hipError_t e = hipSetDevice(p->device);
    if (e != hipSuccess) return fail(CRB_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return CRB_OK;
}

template <typename T>
KParams<T> base_params(const crb_plan* p) {
    KParams<T> k;
    std::memset(&k, 0, sizeof(k));
    k.slot = static_cast<const SlotConst<T>*>(p->d_slot);
    k.pcr_levels = static_cast<const T*>(p->d_levels);
    k.pcr_final = static_cast<const T*>(p->d_final);
    k.B = p->B;
    k.S = p->S;
    k.G = p->G;
    k.n_node = p->n_node;
    k.off = p->off;
    k.levels = p->levels;
    k.lognw = p->lognw;
    k.slot_stride = p->slot_stride; k.lv_stride = p->lv_stride; k.fin_stride = p->fin_stride;
    k.flags = p->flags;
    k.imp_slot = -1;
    k.imp_dof = 0;
    k.imp_node_b = nullptr;
    k.duration = 0.0;
    k.gx = T(p->gx);
    k.gy = T(p->gy);
    k.gvec = static_cast<const T*>(p->d_gvec);
    return k;
}

// per-beam status of a stepper launch of `n_steps` steps (crb_plan_set_status): what a beam found non-finite at the end of
// this launch is marked with -- the ensemble's step count by then
template <typename T>
void arm_status(const crb_plan* p, KParams<T>& k, int n_steps) {
    k.status = p->d_status;
    if (!p->d_status) return;
    p->status_steps += n_steps;
    k.status_value = int32_t(p->status_steps < 2147483647LL ? p->status_steps : 2147483647LL);
}

// dynamic LDS above 64 KiB must be opted into per kernel (the CU has 160 KiB)
template <typename K>
int allow_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
    return CRB_OK;
}

template <typename T, int MODE, int LV, bool LEAN>
int launch_beam_lean(const crb_plan* p, const KParams<T>& k, hipStream_t st) {
    const dim3 grid((p->B + p->G - 1) / p->G), block(p->NT);
    const size_t smem = lds_bytes<T>(p->NT);
    // register budget: <=256-thread groups run 2 groups per CU (2 waves/SIMD, 256 VGPRs) so that the
    // solve multipliers stay in registers; 1024-thread groups get what their size allows
    if (p->NT <= 256) {
        hipLaunchKernelGGL((crb_beam_kernel<T, MODE, LV, 256, 2, LEAN>), grid, block, smem, st, k);
    } else if constexpr (LV >= 4) {
        // (beams of more than 256 slots always run a truncated reduction -- their full one has 9 or 10 levels -- and the
        //  truncation never lands below 4 levels: the multipliers of level 3 are ~1e-3; fewer levels are not built)
        if (p->NT <= 512) {
            if (int rc = allow_lds(crb_beam_kernel<T, MODE, LV, 512, 2, LEAN>, smem)) return rc;
            hipLaunchKernelGGL((crb_beam_kernel<T, MODE, LV, 512, 2, LEAN>), grid, block, smem, st, k);
        } else {
            if (int rc = allow_lds(crb_beam_kernel<T, MODE, LV, 1024, 4, LEAN>, smem)) return rc;
            hipLaunchKernelGGL((crb_beam_kernel<T, MODE, LV, 1024, 4, LEAN>), grid, block, smem, st, k);
        }
    } else {
        return fail(CRB_EUNSUPPORTED, "a beam of more than 256 thread-carried nodes with fewer than 4 cyclic-reduction levels");
    }
    HIP_TRY(hipGetLastError());
    return CRB_OK;
}

template <typename T, int MODE, int LV>
int launch_beam_lv(const crb_plan* p, const KParams<T>& k, hipStream_t st) {
    // LEAN: the stepper without gravity tables and without a held input (BASELINE config 3/4)
    if (MODE == MODE_STEP && !(p->flags & CRB_FORCE_GRAVITY) && !k.u_held)
        return launch_beam_lean<T, MODE, LV, MODE == MODE_STEP>(p, k, st);
    return launch_beam_lean<T, MODE, LV, false>(p, k, st);
}

template <typename T, int MODE>
int launch_beam(const crb_plan* p, const KParams<T>& k, hipStream_t st) {
#ifdef CRB_FAST_BUILD  // kernel-tuning build (make fast): only the config-3 lean stepper is instantiated
    return fail(CRB_EUNSUPPORTED, "CRB_FAST_BUILD: generic kernels not built");
#else
    switch (p->levels) {
        case 0: return launch_beam_lv<T, MODE, 0>(p, k, st);
        case 1: return launch_beam_lv<T, MODE, 1>(p, k, st);
        case 2: return launch_beam_lv<T, MODE, 2>(p, k, st);
        case 3: return launch_beam_lv<T, MODE, 3>(p, k, st);
        case 4: return launch_beam_lv<T, MODE, 4>(p, k, st);
        case 5: return launch_beam_lv<T, MODE, 5>(p, k, st);
        case 6: return launch_beam_lv<T, MODE, 6>(p, k, st);
        case 7: return launch_beam_lv<T, MODE, 7>(p, k, st);
        case 8: return launch_beam_lv<T, MODE, 8>(p, k, st);
        default: return fail(CRB_EUNSUPPORTED, "unsupported number of cyclic-reduction levels");
    }
#endif
}

// Fast path of crb_step_rk4: no held input, one beam per workgroup, gravity absent or of the plain
// cantilever's nearest-neighbour form.  The kernels live in crb_lean.hip (one translation unit per dtype).
inline bool lean_eligible(const crb_plan* p, const void* held) {
    const bool grav = (p->flags & CRB_FORCE_GRAVITY) != 0;
    (void)held;   // (a held input has its own instantiation of the lean stepper)
    // (G > 1: beams of fewer than 64 slots packed into one wave, the PACK instantiation of the one-wave stepper)
    const bool packed = p->G > 1 && p->lognw == 0 && p->NT == 64 && std::getenv("CRB_DISABLE_LEAN_PACK") == nullptr;
    // (beams of more than 64 slots: only the level counts the truncated reduction lands on are built, crb_lean.hip:by_nw)
    const int lv_long = p->dtype == CRB_F64 ? 5 : 4;
    const bool levels_ok = p->lognw == 0 ? (p->levels >= 3 && p->levels <= 6) : (p->levels == lv_long || p->levels == lv_long + 1);
    return (!grav || p->canonical_gravity) && (p->G == 1 || packed) && p->NT == (64 << p->lognw) && p->lognw <= 3 &&
           levels_ok && std::getenv("CRB_DISABLE_LEAN") == nullptr;
}
template <typename T>
int launch_lean(const crb_plan* p, const KParams<T>& k, hipStream_t st) {
#ifdef CRB_FAST_BUILD   // (make fast: the fp64 config-3 instance only; make fast32: the fp32 config-4 one)
#ifdef CRB_FAST_F32
    if constexpr (sizeof(T) == 8) return fail(CRB_EUNSUPPORTED, "CRB_FAST_BUILD (fp32): fp64 lean stepper not built");
#else
    if constexpr (sizeof(T) == 4) return fail(CRB_EUNSUPPORTED, "CRB_FAST_BUILD: fp32 lean stepper not built");
#endif
    else
#endif
    HIP_TRY(crb::launch_lean(k, p->B, p->levels, p->lognw, (p->flags & CRB_FORCE_GRAVITY) != 0, p->elem_mode, st));
    return CRB_OK;
}

// One RK4 stage through the lean machinery (crb_stage_lean_kernel): same eligibility as the lean stepper,
// except that a per-node input force is part of the stage contract.  Shared-table plans run a bounded
// number of workgroups, each walking over several beams with the solve tables in registers.
inline bool stage_lean_eligible(const crb_plan* p) {
    return lean_eligible(p, nullptr) && p->G == 1 && std::getenv("CRB_DISABLE_LEAN_STAGE") == nullptr;
}
template <typename T>
int launch_stage_lean(const crb_plan* p, const KParams<T>& k, hipStream_t st) {
    const bool shared = p->slot_stride == 0 && p->lv_stride == 0 && p->fin_stride == 0;
    int groups = p->B;
    if (shared) {
        // one wave per SIMD (256 CUs x 4) / waves per group: measured best at 2048 x 128 (208 us per step
        // against 220 with two waves per SIMD and 236 with one group per beam) -- the table reload per
        // group costs more than the extra latency hiding gains
        const int resident = 256 * 4 / (1 << p->lognw);
        const char* env = std::getenv("CRB_STAGE_GROUPS");
        const int cap = env ? std::atoi(env) : resident;
        if (cap > 0 && groups > cap) groups = cap;
    }
#ifdef CRB_FAST_BUILD
#ifdef CRB_FAST_F32
    if constexpr (sizeof(T) == 8) return fail(CRB_EUNSUPPORTED, "CRB_FAST_BUILD (fp32): fp64 lean stage kernel not built");
#else
    if constexpr (sizeof(T) == 4) return fail(CRB_EUNSUPPORTED, "CRB_FAST_BUILD: fp32 lean stage kernel not built");
#endif
    else
#endif
    HIP_TRY(crb::launch_stage_lean(k, groups, p->levels, p->lognw, (p->flags & CRB_FORCE_GRAVITY) != 0, p->elem_mode, st));
    return CRB_OK;
}

template <typename T>
int pack_impl(const crb_plan* p, bool pack, int rows, const void* red_in, void* dev, void* red_out, hipStream_t st) {
    const size_t total = size_t(p->B) * rows * p->n_free;
    const int bs = 256;
    const unsigned grid = unsigned((total + bs - 1) / bs);
    if (pack) {
        HIP_TRY(hipMemsetAsync(dev, 0, size_t(p->B) * rows * p->n_node * 4 * sizeof(T), st));
        hipLaunchKernelGGL((crb_pack_kernel<T, true>), dim3(grid), dim3(bs), 0, st, p->d_free_index, p->free_index_stride, p->n_free, p->n_node,
                           rows, p->B, static_cast<const T*>(red_in), static_cast<T*>(dev), static_cast<T*>(nullptr));
    } else {
        hipLaunchKernelGGL((crb_pack_kernel<T, false>), dim3(grid), dim3(bs), 0, st, p->d_free_index, p->free_index_stride, p->n_free, p->n_node,
                           rows, p->B, static_cast<const T*>(nullptr), static_cast<T*>(dev), static_cast<T*>(red_out));
    }
    HIP_TRY(hipGetLastError());
    return CRB_OK;
}

int pack_dispatch(const crb_plan* p, bool pack, int rows, const void* red_in, void* dev, void* red_out, void* stream,
                  const char* who) {
    if (int rc = need_device(p, who)) return rc;
    if (!dev || (pack && !red_in) || (!pack && !red_out)) return fail(CRB_EINVAL, std::string(who) + ": null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    return p->dtype == CRB_F64 ? pack_impl<double>(p, pack, rows, red_in, dev, red_out, st)
                               : pack_impl<float>(p, pack, rows, red_in, dev, red_out, st);
}

}  // namespace

extern "C" int crb_pack_state(const crb_plan* p, const void* x_red, void* x, void* stream) {
    return pack_dispatch(p, true, 2, x_red, x, nullptr, stream, "crb_pack_state");
}
extern "C" int crb_unpack_state(const crb_plan* p, const void* x, void* x_red, void* stream) {
    return pack_dispatch(p, false, 2, nullptr, const_cast<void*>(x), x_red, stream, "crb_unpack_state");
}
extern "C" int crb_pack_vec(const crb_plan* p, const void* v_red, void* v, void* stream) {
    return pack_dispatch(p, true, 1, v_red, v, nullptr, stream, "crb_pack_vec");
}
extern "C" int crb_unpack_vec(const crb_plan* p, const void* v, void* v_red, void* stream) {
    return pack_dispatch(p, false, 1, nullptr, const_cast<void*>(v), v_red, stream, "crb_unpack_vec");
}

extern "C" int crb_internal_force(const crb_plan* p, const void* x, void* kout, void* stream) {
    if (int rc = need_device(p, "crb_internal_force")) return rc;
    if (!x || !kout) return fail(CRB_EINVAL, "crb_internal_force: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->dtype == CRB_F64) {
        KParams<double> k = base_params<double>(p);
        k.x = static_cast<double*>(const_cast<void*>(x));
        k.out = static_cast<double*>(kout);
        return launch_beam<double, MODE_KQ>(p, k, st);
    }
    KParams<float> k = base_params<float>(p);
    k.x = static_cast<float*>(const_cast<void*>(x));
    k.out = static_cast<float*>(kout);
    return launch_beam<float, MODE_KQ>(p, k, st);
}

extern "C" int crb_rhs(const crb_plan* p, const void* x, const void* u, void* xdot, void* stream) {
    if (int rc = need_device(p, "crb_rhs")) return rc;
    if (!x || !xdot) return fail(CRB_EINVAL, "crb_rhs: null pointer");
    if (x == xdot) return fail(CRB_EINVAL, "crb_rhs: xdot must not alias x");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->dtype == CRB_F64) {
        KParams<double> k = base_params<double>(p);
        k.x = static_cast<double*>(const_cast<void*>(x));
        k.u_held = static_cast<const double*>(u);
        k.out = static_cast<double*>(xdot);
        return launch_beam<double, MODE_RHS>(p, k, st);
    }
    KParams<float> k = base_params<float>(p);
    k.x = static_cast<float*>(const_cast<void*>(x));
    k.u_held = static_cast<const float*>(u);
    k.out = static_cast<float*>(xdot);
    return launch_beam<float, MODE_RHS>(p, k, st);
}

extern "C" int crb_step_rk4(const crb_plan* p, void* x, double t0, double dt, int n_steps, const crb_input_desc* in,
                            double* t_end, void* stream) {
    return crb_step_rk4_rec(p, x, t0, dt, n_steps, in, nullptr, t_end, stream);
}

extern "C" int crb_step_rk4_rec(const crb_plan* p, void* x, double t0, double dt, int n_steps, const crb_input_desc* in,
                                const crb_record_desc* rec, double* t_end, void* stream) {
    if (int rc = need_device(p, "crb_step_rk4")) return rc;
    int rec_slot = -1, rec_comp = 0, rec_every = 1, rec_n = 0;
    void* rec_out = nullptr;
    if (rec && rec->node == CRB_RECORD_ALL) {   // whole-state snapshots
        if (rec->every < 1 || !rec->out) return fail(CRB_EINVAL, "crb_step_rk4_rec: bad record description");
        rec_n = n_steps / rec->every;
        if (rec_n > 0) { rec_slot = REC_ALL_SLOTS; rec_every = rec->every; rec_out = rec->out; }
    } else if (rec) {
        if (rec->plane < 0 || rec->plane > 1 || rec->node < 0 || rec->node >= p->n_node || rec->dof < 0 || rec->dof > 2 ||
            rec->every < 1 || !rec->out)
            return fail(CRB_EINVAL, "crb_step_rk4_rec: bad record description");
        rec_n = n_steps / rec->every;
        if (rec->node - p->off >= 0 && rec_n > 0) {  // (the dropped FIXED node 0 records nothing: it is 0 forever)
            rec_slot = rec->node - p->off; rec_comp = rec->plane * 3 + rec->dof; rec_every = rec->every; rec_out = rec->out;
        }
    }
    if (!x) return fail(CRB_EINVAL, "crb_step_rk4: null state");
    if (n_steps < 0) return fail(CRB_EINVAL, "crb_step_rk4: n_steps must be >= 0");
    if (!(dt > 0)) return fail(CRB_EINVAL, "crb_step_rk4: dt must be positive");
    int imp_slot = -1, imp_dof = 0;
    double duration = 0.0;
    const void* amp = nullptr;
    const void* held = nullptr;
    if (in) {
        held = in->f_held;
        if (in->kind == CRB_INPUT_IMPULSE) {
            if (in->node < 0 || in->node >= p->n_node || in->dof < 0 || in->dof > 2)
                return fail(CRB_EINVAL, "crb_step_rk4: impulse node/dof out of range");
            if (!in->amp) return fail(CRB_EINVAL, "crb_step_rk4: impulse amplitude array is null");
            if (!p->any_free[3 * in->node + in->dof])   // (mixed ensembles: a beam in which the DOF is constrained ignores it)
                return fail(CRB_EINVAL, "crb_step_rk4: impulse targets a constrained DOF");
            imp_slot = in->node - p->off;
            imp_dof = in->dof;
            duration = in->duration;
            amp = in->amp;
        } else if (in->kind != CRB_INPUT_NONE) {
            return fail(CRB_EINVAL, "crb_step_rk4: unknown input kind");
        }
    }
    // the clock the kernel will hold after n_steps additions (same fp64 additions on the host)
    if (t_end) {
        double t = t0;
        for (int i = 0; i < n_steps; ++i) t = t + dt;
        *t_end = t;
    }
    if (n_steps == 0) return CRB_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->dtype == CRB_F64) {
        KParams<double> k = base_params<double>(p);
        k.x = static_cast<double*>(x);
        k.u_held = static_cast<const double*>(held);
        k.amp = static_cast<const double*>(amp);
        k.imp_slot = imp_slot; k.imp_dof = imp_dof; k.duration = duration;
        k.imp_node_b = (in && in->kind == CRB_INPUT_IMPULSE) ? in->node_b : nullptr;
        k.t0 = t0; k.dt = dt; k.n_steps = n_steps;
    arm_status(p, k, n_steps);
        k.rec_out = static_cast<double*>(rec_out); k.rec_slot = rec_slot; k.rec_comp = rec_comp; k.rec_every = rec_every; k.rec_n = rec_n;
        if (lean_eligible(p, held)) return launch_lean<double>(p, k, st);
        return launch_beam<double, MODE_STEP>(p, k, st);
    }
    KParams<float> k = base_params<float>(p);
    k.x = static_cast<float*>(x);
    k.u_held = static_cast<const float*>(held);
    k.amp = static_cast<const float*>(amp);
    k.imp_slot = imp_slot; k.imp_dof = imp_dof; k.duration = duration;
    k.imp_node_b = (in && in->kind == CRB_INPUT_IMPULSE) ? in->node_b : nullptr;
    k.t0 = t0; k.dt = dt; k.n_steps = n_steps;
    arm_status(p, k, n_steps);
    k.rec_out = static_cast<float*>(rec_out); k.rec_slot = rec_slot; k.rec_comp = rec_comp; k.rec_every = rec_every; k.rec_n = rec_n;
    if (lean_eligible(p, held)) return launch_lean<float>(p, k, st);
    return launch_beam<float, MODE_STEP>(p, k, st);
}

// ------------------------------------------------------------------ host-vector entry points (single-beam closures)
namespace {
// device copy of the full -> reduced index map (plans with one free-DOF set)
int ensure_red_map(const crb_plan* p) {
    if (p->d_red_map) return CRB_OK;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p->d_red_map), p->full2red.size() * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(p->d_red_map, p->full2red.data(), p->full2red.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return CRB_OK;
}
int host_path_setup(const crb_plan* p, const char* who) {
    if (int rc = need_device(p, who)) return rc;
    if (p->dtype != CRB_F64) return fail(CRB_EUNSUPPORTED, std::string(who) + ": host-vector calls need an fp64 plan");
    if (p->mixed_topology) return fail(CRB_EUNSUPPORTED, std::string(who) + ": host-vector calls need one free-DOF set for the plan");
    if (p->h_stage) return CRB_OK;
    if (int rc = ensure_red_map(p)) return rc;
    const size_t n = p->free_index.size();
    // (+ 64 bytes: the completion flag of one-workgroup launches)
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->h_stage), size_t(p->B) * 5 * n * sizeof(double) + 64, hipHostMallocMapped));
    std::memset(p->h_stage + size_t(p->B) * 5 * n, 0, 64);
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->d_stage), p->h_stage, 0));
    HIP_TRY(hipStreamCreateWithFlags(&p->host_stream, hipStreamNonBlocking));
    return CRB_OK;
}
}  // namespace

namespace {
// Waits for a host-path launch.  One workgroup: the kernel raises a flag in host-mapped memory after its output stores
// and the host spins on it (a stream synchronise costs ~8 us of wake-up latency on top of a 5 us kernel).  The spin is
// bounded by the wall clock (CRB_HOST_SPIN_MS, default 200 ms) and looks at the stream every 4096 polls: a launch that failed
// ends the wait with its error at once, a flag that never comes falls back to synchronising the stream.
int host_path_wait(const crb_plan* p, bool flagged, unsigned long long seq) {
    if (flagged) {
        const size_t n = p->free_index.size();
        volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(p->h_stage + size_t(p->B) * 5 * n);
        timespec t0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (unsigned long spin = 1;; ++spin) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return CRB_OK;
            __builtin_ia32_pause();
            if ((spin & 4095ul) == 0) {
                const hipError_t q = hipStreamQuery(p->host_stream);
                if (q != hipSuccess && q != hipErrorNotReady) return fail(CRB_EHIP, std::string("host-path launch: ") + hipGetErrorString(q));
                if (q == hipSuccess) break;   // the stream is idle: the kernel has run (its flag store is visible after the synchronise below)
                timespec t1;
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1000.0 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > p->host_spin_ms) break;
            }
        }
    }
    HIP_TRY(hipStreamSynchronize(p->host_stream));
    return CRB_OK;
}
}  // namespace

extern "C" int crb_rhs_host(const crb_plan* p, const double* x_red, const double* u_red, double* xdot_red) {
    if (int rc = host_path_setup(p, "crb_rhs_host")) return rc;
    if (!x_red || !xdot_red) return fail(CRB_EINVAL, "crb_rhs_host: null pointer");
    const size_t n = p->free_index.size(), B = size_t(p->B);
    double *hx = p->h_stage, *hu = hx + B * 2 * n, *ho = hu + B * n;
    std::memcpy(hx, x_red, B * 2 * n * sizeof(double));
    if (u_red) std::memcpy(hu, u_red, B * n * sizeof(double));
    KParams<double> k = base_params<double>(p);
    k.x = p->d_stage;
    k.u_held = u_red ? p->d_stage + B * 2 * n : nullptr;
    k.out = p->d_stage + B * 3 * n;
    k.red_map = p->d_red_map;
    k.n_red = int(n);
    const bool flagged = (p->B + p->G - 1) / p->G == 1 && !p->host_sync;
    if (flagged) {
        k.done_flag = reinterpret_cast<unsigned long long*>(p->d_stage + B * 5 * n);
        k.done_seq = ++p->host_seq;
    }
    if (int rc = launch_beam<double, MODE_RHS>(p, k, p->host_stream)) return rc;
    if (int rc = host_path_wait(p, flagged, k.done_seq)) return rc;
    std::memcpy(xdot_red, ho, B * 2 * n * sizeof(double));
    return CRB_OK;
}

extern "C" int crb_internal_force_host(const crb_plan* p, const double* q_red, double* k_red) {
    if (int rc = host_path_setup(p, "crb_internal_force_host")) return rc;
    if (!q_red || !k_red) return fail(CRB_EINVAL, "crb_internal_force_host: null pointer");
    const size_t n = p->free_index.size(), B = size_t(p->B);
    double *hx = p->h_stage, *ho = hx + B * 3 * n;
    for (size_t b = 0; b < B; ++b) std::memcpy(hx + b * 2 * n, q_red + b * n, n * sizeof(double));   // (state rows of 2n: q | unused)
    KParams<double> k = base_params<double>(p);
    k.x = p->d_stage;
    k.out = p->d_stage + B * 3 * n;
    k.red_map = p->d_red_map;
    k.n_red = int(n);
    const bool flagged = (p->B + p->G - 1) / p->G == 1 && !p->host_sync;
    if (flagged) {
        k.done_flag = reinterpret_cast<unsigned long long*>(p->d_stage + B * 5 * n);
        k.done_seq = ++p->host_seq;
    }
    if (int rc = launch_beam<double, MODE_KQ>(p, k, p->host_stream)) return rc;
    if (int rc = host_path_wait(p, flagged, k.done_seq)) return rc;
    std::memcpy(k_red, ho, B * n * sizeof(double));
    return CRB_OK;
}

// ------------------------------------------------------------------ implicit stepper (crb_stiff.h)
namespace {
// (re)builds the cyclic-reduction tables of A = M + alpha K0 on `st` when the step size changed
template <typename T>
int stiff_tables(const crb_plan* p, double alpha, hipStream_t st) {
    if (p->stiff_alpha == alpha && p->d_alevels) return CRB_OK;
    AsmInputs& in = *p->asm_in;
    const int S = p->S, lf = p->levels_full, nd = in.nd;
    // a set built earlier for this step size, else the least recently used one (work queued on `st` that still reads
    // the overwritten set is ordered before the assembly launched below on the same stream)
    crb_plan::StiffSet* set = nullptr;
    for (auto& c : p->stiff_sets)
        if (c.lev && c.alpha == alpha) set = &c;
    const bool hit = set != nullptr;
    if (!set) {
        set = &p->stiff_sets[0];
        for (auto& c : p->stiff_sets)
            if (c.used < set->used) set = &c;
    }
    if (!set->lev || !set->fin) {
        // both tables or neither: a set with one of them would pass for built at the next call
        void *lev = nullptr, *fin = nullptr;
        const bool ok = hipMalloc(&lev, size_t(nd) * size_t(lf > 0 ? lf : 1) * S * PCR_LEVEL_VALS * sizeof(T)) == hipSuccess &&
                        hipMalloc(&fin, size_t(nd) * S * PCR_FINAL_VALS * sizeof(T)) == hipSuccess &&
                        (in.dNormScratch.p || in.dNormScratch.alloc(size_t(lf > 0 ? lf : 1)) == 0);
        if (!ok) {
            if (lev) (void)hipFree(lev);
            if (fin) (void)hipFree(fin);
            (void)hipGetLastError();
            return fail(CRB_EHIP, "crb_step_implicit: device allocation failed");
        }
        set->lev = lev;
        set->fin = fin;
    }
    set->used = ++p->stiff_clock;
    p->d_alevels = set->lev;
    p->d_afinal = set->fin;
    if (hit) {
        p->stiff_levels = set->levels;
        p->stiff_alpha = alpha;
        return CRB_OK;
    }
    set->alpha = 0.0;   // (not valid until the build below has been queued; neither is "the set in use" if a launch below fails)
    p->stiff_alpha = 0.0;
    AsmParams a = in.a;
    a.alpha = alpha;
    a.slot_out = nullptr; a.lv64 = nullptr; a.fin64_all = nullptr; a.blocks0 = nullptr;
    a.lvT = p->d_alevels; a.finT = p->d_afinal; a.fin_level = lf; a.norms = in.dNormScratch.p;
    HIP_TRY(hipMemsetAsync(in.dNormScratch.p, 0, size_t(lf > 0 ? lf : 1) * sizeof(double), st));
    hipLaunchKernelGGL((crb_assemble_kernel<T>), dim3(nd), dim3(in.threads), in.smem, st, a);
    HIP_TRY(hipGetLastError());
    // The reduction of A stops where its multipliers fall below the unit roundoff, like the mass matrix's (pick_levels):
    // for the step sizes the examples use (h = 1e-4: the stride-32 multipliers of a 256-node Nitinol rod are ~1e-19) that is
    // 5 levels instead of 8; large steps (h = 1e-3) keep them all.  One stream synchronisation per new step size.
    int used = lf;
    if (lf > 0 && std::getenv("CRB_STIFF_ALL_LEVELS") == nullptr) {
        std::vector<double> norms(static_cast<size_t>(lf));
        HIP_TRY(hipMemcpyAsync(norms.data(), in.dNormScratch.p, size_t(lf) * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        while (used > 0 && norms[size_t(used) - 1] < std::ldexp(1.0, -53)) --used;
        // (the lean kernels exist for 5 ... full levels)
        const bool lean_shape = p->G == 1 && p->lognw <= 2 && p->NT == (64 << p->lognw) && lf == 6 + p->lognw;
        if (lean_shape && used < 5 && std::getenv("CRB_DISABLE_LEAN") == nullptr && std::getenv("CRB_DISABLE_LEAN_IMPLICIT") == nullptr) used = 5;
        // (several beams per wave: the packed lean kernels exist for 3 ... 5 levels)
        const bool pack_shape = p->G > 1 && p->lognw == 0 && p->NT == 64 && lf >= 3 && lf <= 5;
        if (pack_shape && used < 3) used = 3;
        if (used < lf) {   // the final block inverses after `used` levels
            a.fin_level = used;
            hipLaunchKernelGGL((crb_assemble_kernel<T>), dim3(nd), dim3(in.threads), in.smem, st, a);
            HIP_TRY(hipGetLastError());
        }
    }
    p->stiff_levels = used;
    p->stiff_alpha = alpha;
    set->levels = used;
    set->alpha = alpha;
    return CRB_OK;
}

template <typename T, int LV>
int launch_implicit_lv(const crb_plan* p, const KParams<T>& k, const StiffParams<T>& q, hipStream_t st) {
    const dim3 grid((p->B + p->G - 1) / p->G), block(p->NT);
    const size_t smem = lds_bytes<T>(p->NT);
    // one wave per SIMD: the thread's rows of A's tables (10 per level, up to 8 levels) stay in registers
    if (p->NT > 256) return fail(CRB_EUNSUPPORTED, "crb_step_implicit: beams of more than 256 thread-carried nodes are not supported");
    // Up to 5 levels two waves per SIMD fit the register file (65536 x 10: 32 instead of 42 us per step); a launch that
    // does not fill the GPU at one wave per SIMD keeps the whole file per wave (ONE 10-element beam: 3.6 instead of 4.3 us
    // per step).  The full tables of long beams take the file whole.
    const long waves = long(grid.x) * ((p->NT + 63) / 64);
    if (LV <= 5 && waves > 1024) hipLaunchKernelGGL((crb_implicit_kernel<T, LV, 256, (LV <= 5 ? 2 : 1)>), grid, block, smem, st, k, q);
    else hipLaunchKernelGGL((crb_implicit_kernel<T, LV, 256, 1>), grid, block, smem, st, k, q);
    HIP_TRY(hipGetLastError());
    return CRB_OK;
}
template <typename T>
int launch_implicit(const crb_plan* p, const KParams<T>& k, const StiffParams<T>& q, hipStream_t st) {
#ifdef CRB_FAST_BUILD
    return fail(CRB_EUNSUPPORTED, "CRB_FAST_BUILD: implicit stepper not built");
#else
    // the lean form: one beam per workgroup of 1 / 2 / 4 waves with ALL its reduction levels, gravity absent or canonical
    const bool grav = (p->flags & CRB_FORCE_GRAVITY) != 0;
    const bool one_per_group = p->G == 1 && p->lognw <= 2 && p->NT == (64 << p->lognw) && p->levels_full == 6 + p->lognw;
    const bool packed = p->G > 1 && p->lognw == 0 && p->NT == 64 && p->levels_full >= 3 && p->levels_full <= 5 && p->stiff_levels >= 3 &&
                        std::getenv("CRB_DISABLE_LEAN_PACK") == nullptr;
    if ((one_per_group || packed) && (!grav || p->canonical_gravity) &&
        std::getenv("CRB_DISABLE_LEAN") == nullptr && std::getenv("CRB_DISABLE_LEAN_IMPLICIT") == nullptr) {
        // one wave per SIMD (the tables of A fill the register file): 256 CUs x 4 / waves per beam workgroups are resident
        int groups = (p->B + p->G - 1) / p->G;
        const bool shared = p->slot_stride == 0 && q.alv_stride == 0 && q.afin_stride == 0;
        int resident = 256 * 4 / (1 << p->lognw) * implicit_lean_minw(p->stiff_levels, grav, p->lognw);   // (waves per SIMD: crb_stiff.h)
        if (const char* env = std::getenv("CRB_LEAN_MAX_GROUPS")) resident = std::atoi(env) > 0 ? std::atoi(env) : resident;   // (tests)
        if (shared && groups > resident) {
            const int rounds = (groups + resident - 1) / resident;
            groups = (groups + rounds - 1) / rounds;
        }
        HIP_TRY(crb::launch_implicit_lean(k, q, groups, p->stiff_levels, p->lognw, grav, p->elem_mode, st));
        return CRB_OK;
    }
    switch (p->stiff_levels) {   // (the levels of A whose multipliers matter, stiff_tables)
        case 0: return launch_implicit_lv<T, 0>(p, k, q, st);
        case 1: return launch_implicit_lv<T, 1>(p, k, q, st);
        case 2: return launch_implicit_lv<T, 2>(p, k, q, st);
        case 3: return launch_implicit_lv<T, 3>(p, k, q, st);
        case 4: return launch_implicit_lv<T, 4>(p, k, q, st);
        case 5: return launch_implicit_lv<T, 5>(p, k, q, st);
        case 6: return launch_implicit_lv<T, 6>(p, k, q, st);
        case 7: return launch_implicit_lv<T, 7>(p, k, q, st);
        case 8: return launch_implicit_lv<T, 8>(p, k, q, st);
        default: return fail(CRB_EUNSUPPORTED, "crb_step_implicit: beams of more than 256 thread-carried nodes are not supported");
    }
#endif
}

// the general implicit kernel in its damped form (generalised-alpha): any level count, one wave per SIMD
template <typename T, int LV>
int launch_implicit_damped_lv(const crb_plan* p, const KParams<T>& k, const StiffParams<T>& q, hipStream_t st) {
    const dim3 grid((p->B + p->G - 1) / p->G), block(p->NT);
    hipLaunchKernelGGL((crb_implicit_kernel<T, LV, 256, 1, true>), grid, block, lds_bytes<T>(p->NT), st, k, q);
    HIP_TRY(hipGetLastError());
    return CRB_OK;
}
template <typename T>
int launch_implicit_damped(const crb_plan* p, const KParams<T>& k, const StiffParams<T>& q, hipStream_t st) {
#ifdef CRB_FAST_BUILD
    return fail(CRB_EUNSUPPORTED, "CRB_FAST_BUILD: implicit stepper not built");
#else
    if (p->NT > 256) return fail(CRB_EUNSUPPORTED, "crb_step_implicit: beams of more than 256 thread-carried nodes are not supported");
    switch (p->stiff_levels) {
        case 0: return launch_implicit_damped_lv<T, 0>(p, k, q, st);
        case 1: return launch_implicit_damped_lv<T, 1>(p, k, q, st);
        case 2: return launch_implicit_damped_lv<T, 2>(p, k, q, st);
        case 3: return launch_implicit_damped_lv<T, 3>(p, k, q, st);
        case 4: return launch_implicit_damped_lv<T, 4>(p, k, q, st);
        case 5: return launch_implicit_damped_lv<T, 5>(p, k, q, st);
        case 6: return launch_implicit_damped_lv<T, 6>(p, k, q, st);
        case 7: return launch_implicit_damped_lv<T, 7>(p, k, q, st);
        case 8: return launch_implicit_damped_lv<T, 8>(p, k, q, st);
        default: return fail(CRB_EUNSUPPORTED, "crb_step_implicit: beams of more than 256 thread-carried nodes are not supported");
    }
#endif
}

template <typename T>
int step_implicit_impl(const crb_plan* p, void* x, double t0, double h, int n_steps, int n_iter, double rho, const crb_input_desc* in,
                       int imp_slot, int imp_dof, double duration, const void* amp, const void* held, void* rec_out,
                       int rec_slot, int rec_comp, int rec_every, int rec_n, hipStream_t st) {
    const bool damped = rho < 1.0;
    // generalised-alpha coefficients (crb_stiff.h: StiffParams); rho = 1 is the midpoint rule with kappa = h^2 / 4
    const double am_ = (2.0 * rho - 1.0) / (rho + 1.0), af_ = rho / (rho + 1.0);
    const double gam = 0.5 - am_ + af_, bet = 0.25 * (1.0 - am_ + af_) * (1.0 - am_ + af_);
    const double kappa = damped ? (1.0 - af_) * bet * h * h / (1.0 - am_) : 0.25 * h * h;
    if (int rc = stiff_tables<T>(p, kappa, st)) return rc;
    KParams<T> k = base_params<T>(p);
    k.x = static_cast<T*>(x);
    k.u_held = static_cast<const T*>(held);
    k.amp = static_cast<const T*>(amp);
    k.imp_slot = imp_slot; k.imp_dof = imp_dof; k.duration = duration;
    k.imp_node_b = (in && in->kind == CRB_INPUT_IMPULSE) ? in->node_b : nullptr;
    k.t0 = t0; k.dt = h; k.n_steps = n_steps;
    arm_status(p, k, n_steps);
    k.rec_out = static_cast<T*>(rec_out); k.rec_slot = rec_slot; k.rec_comp = rec_comp; k.rec_every = rec_every; k.rec_n = rec_n;
    StiffParams<T> q;
    q.a_levels = static_cast<const T*>(p->d_alevels);
    q.a_final = static_cast<const T*>(p->d_afinal);
    q.alv_stride = p->asm_in->nd > 1 ? size_t(p->levels_full) * p->S * PCR_LEVEL_VALS : 0;
    q.afin_stride = p->asm_in->nd > 1 ? size_t(p->S) * PCR_FINAL_VALS : 0;
    q.h = h;
    q.n_iter = n_iter;
    q.a0 = nullptr;
    if (!damped) return launch_implicit<T>(p, k, q, st);
    // a_0: the RHS of the state at t0 with the input of t0 (one launch of the RHS kernel into the plan's scratch)
    if (!p->d_rhs0) HIP_TRY(hipMalloc(&p->d_rhs0, size_t(p->B) * 2 * p->n_node * 4 * sizeof(T)));
    {
        KParams<T> r = base_params<T>(p);
        r.x = static_cast<T*>(x);
        r.u_held = static_cast<const T*>(held);
        r.out = static_cast<T*>(p->d_rhs0);
        r.amp = static_cast<const T*>(amp);
        r.imp_slot = imp_slot; r.imp_dof = imp_dof; r.duration = duration; r.imp_node_b = k.imp_node_b;
        r.t0 = t0;
        if (int rc = launch_beam<T, MODE_RHS>(p, r, st)) return rc;
    }
    q.a0 = static_cast<const T*>(p->d_rhs0);
    const double cv = (1.0 - af_) * gam * h / (1.0 - am_);
    q.kappa = kappa; q.cv = cv;
    q.c_qv = (1.0 - af_) * h; q.c_qa = (1.0 - af_) * h * h * (0.5 - bet) - kappa * am_;
    q.c_va = (1.0 - af_) * h * (1.0 - gam) - cv * am_;
    q.tf_frac = 1.0 - af_; q.alpha_m = am_; q.inv1m = 1.0 / (1.0 - am_);
    q.c_q0 = h * h * (0.5 - bet); q.c_q1 = h * h * bet; q.c_v0 = h * (1.0 - gam); q.c_v1 = h * gam;
    return launch_implicit_damped<T>(p, k, q, st);
}
}  // namespace

extern "C" int crb_step_implicit(const crb_plan* p, void* x, double t0, double h, int n_steps, int n_iter,
                                 const crb_input_desc* in, const crb_record_desc* rec, double* t_end, void* stream) {
    return crb_step_implicit_damped(p, x, t0, h, n_steps, n_iter, 1.0, in, rec, t_end, stream);
}

extern "C" int crb_step_implicit_damped(const crb_plan* p, void* x, double t0, double h, int n_steps, int n_iter, double rho_inf,
                                        const crb_input_desc* in, const crb_record_desc* rec, double* t_end, void* stream) {
    if (int rc = need_device(p, "crb_step_implicit")) return rc;
    if (!(rho_inf >= 0.0 && rho_inf <= 1.0)) return fail(CRB_EINVAL, "crb_step_implicit_damped: rho_inf must be in [0, 1]");
    if (!x) return fail(CRB_EINVAL, "crb_step_implicit: null pointer");
    if (n_steps < 0) return fail(CRB_EINVAL, "crb_step_implicit: n_steps must be >= 0");
    if (!(h > 0)) return fail(CRB_EINVAL, "crb_step_implicit: h must be positive");
    if (n_iter < 1) return fail(CRB_EINVAL, "crb_step_implicit: n_iter must be >= 1");
    if (p->dtype != CRB_F64)   // cond(M + h^2/4 K0) is 1e6 ... 1e9 at the step sizes this stepper is for
        return fail(CRB_EUNSUPPORTED, "crb_step_implicit: the implicit stepper needs an fp64 plan (the iteration matrix is too ill-conditioned for fp32)");
    int rec_slot = -1, rec_comp = 0, rec_every = 1, rec_n = 0;
    void* rec_out = nullptr;
    if (rec && rec->node == CRB_RECORD_ALL) {
        if (rec->every < 1 || !rec->out) return fail(CRB_EINVAL, "crb_step_implicit: bad record description");
        rec_n = n_steps / rec->every;
        if (rec_n > 0) { rec_slot = REC_ALL_SLOTS; rec_every = rec->every; rec_out = rec->out; }
    } else if (rec) {
        if (rec->plane < 0 || rec->plane > 1 || rec->node < 0 || rec->node >= p->n_node || rec->dof < 0 || rec->dof > 2 ||
            rec->every < 1 || !rec->out)
            return fail(CRB_EINVAL, "crb_step_implicit: bad record description");
        rec_n = n_steps / rec->every;
        if (rec->node - p->off >= 0 && rec_n > 0) {
            rec_slot = rec->node - p->off; rec_comp = rec->plane * 3 + rec->dof; rec_every = rec->every; rec_out = rec->out;
        }
    }
    int imp_slot = -1, imp_dof = 0;
    double duration = 0.0;
    const void* amp = nullptr;
    const void* held = nullptr;
    if (in) {
        held = in->f_held;
        if (in->kind == CRB_INPUT_IMPULSE) {
            if (in->node < 0 || in->node >= p->n_node || in->dof < 0 || in->dof > 2 || !in->amp ||
                !p->any_free[3 * in->node + in->dof])
                return fail(CRB_EINVAL, "crb_step_implicit: bad impulse description");
            imp_slot = in->node - p->off; imp_dof = in->dof; duration = in->duration; amp = in->amp;
        } else if (in->kind != CRB_INPUT_NONE) {
            return fail(CRB_EINVAL, "crb_step_implicit: unknown input kind");
        }
    }
    if (t_end) {
        double t = t0;
        for (int i = 0; i < n_steps; ++i) t = t + h;
        *t_end = t;
    }
    if (n_steps == 0) return CRB_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return step_implicit_impl<double>(p, x, t0, h, n_steps, n_iter, rho_inf, in, imp_slot, imp_dof, duration, amp, held, rec_out, rec_slot,
                                      rec_comp, rec_every, rec_n, st);
}

// ------------------------------------------------------------------ step-size control in the kernel (crb_ctrl.h)
namespace {
// the ladder of A's tables for pieces of length `len`: rung r holds the factorisation for h = len / 2^r (all levels)
int ctrl_ladder(const crb_plan* p, double len, int rungs, hipStream_t st, const crb_plan::Ladder** out) {
    AsmInputs& in = *p->asm_in;
    const int S = p->S, lf = p->levels_full, nd = in.nd;
    const size_t lv_rung = size_t(nd) * size_t(lf > 0 ? lf : 1) * S * PCR_LEVEL_VALS, fin_rung = size_t(nd) * S * PCR_FINAL_VALS;
    crb_plan::Ladder* lad = nullptr;
    for (auto& c : p->ladders)
        if (c.lev && c.len == len && c.rungs >= rungs) lad = &c;
    if (lad) {
        lad->used = ++p->stiff_clock;
        *out = lad;
        return CRB_OK;
    }
    lad = &p->ladders[0];
    for (auto& c : p->ladders)
        if (c.used < lad->used) lad = &c;
    if (lv_rung * size_t(rungs) * sizeof(double) > (size_t(8) << 30))
        return fail(CRB_EUNSUPPORTED, "crb_solve_controlled: the table ladder of this ensemble (per-beam tables) exceeds 8 GiB; give the step count instead");
    // (work queued on `st` that still reads a replaced ladder is ordered before the frees below: hipFree waits for the device)
    if (lad->lev) (void)hipFree(lad->lev);
    if (lad->fin) (void)hipFree(lad->fin);
    lad->lev = lad->fin = nullptr;
    lad->len = 0.0; lad->rungs = 0;
    void *lev = nullptr, *fin = nullptr;
    const bool ok = hipMalloc(&lev, lv_rung * size_t(rungs) * sizeof(double)) == hipSuccess &&
                    hipMalloc(&fin, fin_rung * size_t(rungs) * sizeof(double)) == hipSuccess &&
                    (in.dNormScratch.p || in.dNormScratch.alloc(size_t(lf > 0 ? lf : 1)) == 0);
    if (!ok) {
        if (lev) (void)hipFree(lev);
        if (fin) (void)hipFree(fin);
        (void)hipGetLastError();
        return fail(CRB_EHIP, "crb_solve_controlled: device allocation failed");
    }
    lad->lev = lev;
    lad->fin = fin;
    HIP_TRY(hipMemsetAsync(in.dNormScratch.p, 0, size_t(lf > 0 ? lf : 1) * sizeof(double), st));
    for (int r = 0; r < rungs; ++r) {
        const double h = len / double(1 << r);
        AsmParams a = in.a;
        a.alpha = 0.25 * h * h;
        a.slot_out = nullptr; a.lv64 = nullptr; a.fin64_all = nullptr; a.blocks0 = nullptr;
        a.lvT = static_cast<double*>(lev) + size_t(r) * lv_rung;
        a.finT = static_cast<double*>(fin) + size_t(r) * fin_rung;
        a.fin_level = lf;
        a.norms = in.dNormScratch.p;
        hipLaunchKernelGGL((crb_assemble_kernel<double>), dim3(nd), dim3(in.threads), in.smem, st, a);
        HIP_TRY(hipGetLastError());
    }
    lad->len = len;
    lad->rungs = rungs;
    lad->used = ++p->stiff_clock;
    *out = lad;
    return CRB_OK;
}
}  // namespace

extern "C" int crb_solve_controlled(const crb_plan* p, void* x, double t0, double dt_eval, int n_intervals,
                                    const crb_control_desc* ctl, const crb_input_desc* in, const void* gain, const void* ref,
                                    void* y_out, void* stats, void* used, void* stream) {
    if (int rc = need_device(p, "crb_solve_controlled")) return rc;
    if (!x || !ctl || !stats) return fail(CRB_EINVAL, "crb_solve_controlled: null pointer");
    if (n_intervals < 0 || !(dt_eval > 0)) return fail(CRB_EINVAL, "crb_solve_controlled: n_intervals >= 0 and dt_eval > 0 required");
    if (!(ctl->rtol > 0) || !(ctl->atol >= 0)) return fail(CRB_EINVAL, "crb_solve_controlled: tolerances must be positive");
    if (p->dtype != CRB_F64) return fail(CRB_EUNSUPPORTED, "crb_solve_controlled: the controlled steppers need an fp64 plan");
    if (p->NT > 256) return fail(CRB_EUNSUPPORTED, "crb_solve_controlled: beams of more than 256 thread-carried nodes are not supported");
    const bool fb = gain != nullptr;
    // (one modified-Newton iteration per step: at the step sizes the controller takes the previous step's iterate is converged
    //  after one -- same errors and step counts +-15 % as with two on linear, nonlinear and mixed rods at rtol 1e-1 .. 1e-3,
    //  half the time: profiles/exp_niter.py; what is left of the iteration error scales with h and is part of the estimate)
    const int n_iter = ctl->n_iter > 0 ? ctl->n_iter : 1;
    const int rungs = ctl->max_rungs > 0 ? ctl->max_rungs : 15;   // up to 2^14 fine steps per piece
    if (rungs < 2 || rungs > 24) return fail(CRB_EINVAL, "crb_solve_controlled: max_rungs must be in 2 .. 24");
    if (fb) {
        if (p->mixed_topology) return fail(CRB_EUNSUPPORTED, "crb_solve_controlled: one gain matrix for the ensemble needs one free-DOF set");
        if (in && in->f_held) return fail(CRB_EUNSUPPORTED, "crb_solve_controlled: the closed loop runs without a held force");
        if (p->levels > 6) return fail(CRB_EUNSUPPORTED, "crb_solve_controlled: unsupported number of cyclic-reduction levels");
        if (ctrl_lds_bytes<double>(p->NT, true, p->n_free) > size_t(160) * 1024)
            return fail(CRB_EUNSUPPORTED, "crb_solve_controlled: the gain matrix does not fit the LDS (beams of up to ~30 elements); use crb_step_rk4_feedback");
    } else if (ref) {
        return fail(CRB_EINVAL, "crb_solve_controlled: a reference without a gain");
    }
    int imp_slot = -1, imp_dof = 0;
    double t_switch = 0.0;
    bool impulse = false;
    const void* amp = nullptr;
    const void* held = nullptr;
    if (in) {
        held = in->f_held;
        if (in->kind == CRB_INPUT_IMPULSE) {
            if (in->node < 0 || in->node >= p->n_node || in->dof < 0 || in->dof > 2 || !in->amp || !p->any_free[3 * in->node + in->dof])
                return fail(CRB_EINVAL, "crb_solve_controlled: bad impulse description");
            imp_slot = in->node - p->off; imp_dof = in->dof; t_switch = in->duration; amp = in->amp;
            impulse = true;
        } else if (in->kind != CRB_INPUT_NONE) {
            return fail(CRB_EINVAL, "crb_solve_controlled: unknown input kind");
        }
    }
    if (n_intervals == 0) return CRB_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);

    // the pieces: whole t_eval intervals, the one the impulse ends in cut at that instant (inside a piece the input is constant)
    std::vector<crb::CtrlPiece> pieces;
    pieces.reserve(size_t(n_intervals) + 1);
    double lens[crb::CTRL_MAX_LADDERS] = {dt_eval, 0.0, 0.0};
    int n_ladders = 1;
    for (int k = 0; k < n_intervals; ++k) {
        const double t_k = t0 + k * dt_eval, t_n = t0 + (k + 1) * dt_eval;
        const bool cut = impulse && t_k + 1e-9 * dt_eval < t_switch && t_switch < t_n - 1e-9 * dt_eval;
        const int parts = cut ? 2 : 1;
        for (int part = 0; part < parts; ++part) {
            crb::CtrlPiece c;
            c.t_a = part == 0 ? t_k : t_switch;
            const double t_b = (cut && part == 0) ? t_switch : t_n;
            c.len = cut ? t_b - c.t_a : dt_eval;
            c.ladder = 0;
            if (cut) { c.ladder = n_ladders; lens[n_ladders++] = c.len; }
            c.on = (impulse && 0.5 * (c.t_a + t_b) < t_switch) ? 1 : 0;
            c.rec = (part == parts - 1 && y_out) ? k : -1;
            c.interval = k;
            pieces.push_back(c);
        }
    }
    crb::CtrlParams<double> q;
    std::memset(&q, 0, sizeof(q));
    if (!fb) {
        if (p->levels_full > 8) return fail(CRB_EUNSUPPORTED, "crb_solve_controlled: beams of more than 256 thread-carried nodes are not supported");
        const int lf = p->levels_full, nd = p->asm_in->nd;
        for (int l = 0; l < n_ladders; ++l) {
            const crb_plan::Ladder* lad = nullptr;
            if (int rc = ctrl_ladder(p, lens[l], rungs, st, &lad)) return rc;
            q.a_levels[l] = static_cast<const double*>(lad->lev);
            q.a_final[l] = static_cast<const double*>(lad->fin);
        }
        for (int l = n_ladders; l < crb::CTRL_MAX_LADDERS; ++l) { q.a_levels[l] = q.a_levels[0]; q.a_final[l] = q.a_final[0]; }
        q.lv_rung = size_t(nd) * size_t(lf > 0 ? lf : 1) * p->S * PCR_LEVEL_VALS;
        q.fin_rung = size_t(nd) * p->S * PCR_FINAL_VALS;
        q.alv_stride = nd > 1 ? size_t(lf > 0 ? lf : 1) * p->S * PCR_LEVEL_VALS : 0;
        q.afin_stride = nd > 1 ? size_t(p->S) * PCR_FINAL_VALS : 0;
    }
    // (the piece list of an earlier call may still be read by its launch: wait for the stream before replacing it)
    const size_t piece_bytes = pieces.size() * sizeof(crb::CtrlPiece);
    HIP_TRY(hipStreamSynchronize(st));
    if (p->pieces_cap < piece_bytes) {
        if (p->d_pieces) (void)hipFree(p->d_pieces);
        p->d_pieces = nullptr; p->pieces_cap = 0;
        HIP_TRY(hipMalloc(&p->d_pieces, piece_bytes));
        p->pieces_cap = piece_bytes;
    }
    HIP_TRY(hipMemcpy(p->d_pieces, pieces.data(), piece_bytes, hipMemcpyHostToDevice));
    q.pieces = static_cast<const crb::CtrlPiece*>(p->d_pieces);
    q.n_pieces = int(pieces.size());
    q.n_rungs = rungs; q.n_iter = n_iter; q.n_intervals = n_intervals;
    q.rtol = ctl->rtol; q.atol = ctl->atol;
    // the first coarse solution: h ~ 1e-4 s for the implicit scheme, 5e-6 s for RK4 (near the closed loop's stability limit
    // for the Nitinol examples, 8.6e-6 s)
    q.rate0 = ctl->first_rate > 0 ? ctl->first_rate : (fb ? 2e5 : 1e4);
    q.n_state = 2 * p->n_free; q.n_state_b = p->d_n_state; q.positions_only = ctl->positions_only ? 1 : 0;
    q.stats = static_cast<int32_t*>(stats);
    q.used = static_cast<int32_t*>(used);
    q.y_out = static_cast<double*>(y_out);
    q.series_out = nullptr; q.series_slot = -1; q.series_comp = 0;
    if (ctl->series_out) {
        if (ctl->series_plane < 0 || ctl->series_plane > 1 || ctl->series_node < 0 || ctl->series_node >= p->n_node || ctl->series_dof < 0 ||
            ctl->series_dof > 2)
            return fail(CRB_EINVAL, "crb_solve_controlled: bad series description");
        if (ctl->series_node - p->off >= 0) {   // (a node without a thread slot -- the fixed root -- stays at the caller's zeros)
            q.series_out = static_cast<double*>(ctl->series_out);
            q.series_slot = ctl->series_node - p->off;
            q.series_comp = 3 * ctl->series_plane + ctl->series_dof;
        }
    }

    KParams<double> k = base_params<double>(p);
    k.G = 1;
    k.x = static_cast<double*>(x);
    k.u_held = static_cast<const double*>(held);
    k.amp = static_cast<const double*>(amp);
    k.imp_slot = imp_slot; k.imp_dof = imp_dof; k.duration = t_switch;
    k.imp_node_b = impulse ? in->node_b : nullptr;
    k.t0 = t0;
    arm_status(p, k, 1);   // (per-beam status, crb_plan_set_status: a controlled launch counts as one step of the ensemble)
    if (fb) {
        if (int rc = ensure_red_map(p)) return rc;
        k.red_map = p->d_red_map;
        k.n_red = p->n_free;
        k.fb_gain = static_cast<const double*>(gain);
        k.fb_ref = static_cast<const double*>(ref);
    }
    const int threads = p->NT;
    // the lean iteration (crb_stiff.h) where the plan allows it: gravity absent or of the plain cantilever's form
    const bool grav = (p->flags & CRB_FORCE_GRAVITY) != 0;
    const bool lean_shape = p->lognw <= 2 && threads == (64 << p->lognw) && (!grav || p->canonical_gravity) && std::getenv("CRB_DISABLE_LEAN") == nullptr;
    const bool lean = fb ? (lean_shape && p->lognw == 0 && p->levels >= 1 && p->levels <= 6 && std::getenv("CRB_DISABLE_LEAN_FEEDBACK") == nullptr)
                         : (lean_shape && p->levels_full >= 1 && std::getenv("CRB_DISABLE_LEAN_IMPLICIT") == nullptr);
    const int lean_lognw = lean ? p->lognw : -1;
    // per_wave: short beams packed G to a wave, one step sequence per wave (the worst of its beams decides)
    const bool pack = ctl->per_wave != 0 && !fb && lean && p->G > 1 && p->lognw == 0 && p->levels_full <= 5;
    if (ctl->per_wave != 0 && !pack)
        return fail(CRB_EUNSUPPORTED, "crb_solve_controlled: per_wave packs beams of 2 .. 32 thread-carried nodes of the implicit scheme (gravity absent or canonical)");
    if (pack) k.G = p->G;
    HIP_TRY(crb::launch_controlled(k, q, fb ? p->levels : p->levels_full, fb, lean_lognw, grav, pack, threads,
                                   ctrl_lds_bytes<double>(threads, fb, p->n_free, fb ? -1 : lean_lognw), st));
    return CRB_OK;
}

static int rk4_stage_impl(const crb_plan* p, void* x, const void* xs, void* acc, void* xs_next, const void* u_stage, int stage,
                         double t_stage, const double* t_dev, double dt, const crb_input_desc* in, void* stream);

namespace {
// The persistent closed-loop stepper (crb_loop.h): fp64 plans with one table set and one free-DOF set, beams of 33 .. 128
// thread-carried nodes, gravity absent or of the plain cantilever's form, no held input.  CRB_LOOP=0 / 1 in the
// environment: never / whenever eligible; unset: ensembles of at least LOOP_MIN_BEAMS beams (a group of workgroups owns
// 64 beams: small ensembles leave most of the chip idle and are faster through the stage-split launches).
constexpr int LOOP_MIN_BEAMS = 512;
int loop_nb(const crb_plan* p) { return p->lognw == 1 ? 8 : 4; }
bool loop_shape_ok(const crb_plan* p) {
    const bool grav = (p->flags & CRB_FORCE_GRAVITY) != 0;
    return p->dtype == CRB_F64 && !p->mixed_topology && p->slot_stride == 0 && p->lv_stride == 0 && p->fin_stride == 0 && p->G == 1 &&
           (p->lognw == 0 || p->lognw == 1) && p->NT == (64 << p->lognw) && p->S > 32 && (p->levels == 5 || p->levels == 6) &&
           (!grav || p->canonical_gravity);
}
bool loop_eligible(const crb_plan* p, const crb_input_desc* in) {
    if (!loop_shape_ok(p) || (in && in->f_held)) return false;
    const char* env = std::getenv("CRB_LOOP");
    if (env) return std::atoi(env) != 0;
    return p->B >= LOOP_MIN_BEAMS;
}
}  // namespace

extern "C" size_t crb_feedback_work_bytes(const crb_plan* p) {
    if (!p) return 0;
    const size_t state = size_t(p->B) * 2 * p->n_node * 4 * (p->dtype == CRB_F64 ? sizeof(double) : sizeof(float)), force = state / 2;
    size_t need = 3 * state + force + 256;   // + the device clock of the replayed step
    if (loop_shape_ok(p)) {
        const int n_rb = (p->B + 63) / 64;
        const size_t loop = crb::loop_work_layout(loop_nb(p), n_rb < crb::LOOP_MAX_GROUPS ? n_rb : crb::LOOP_MAX_GROUPS).total;
        if (loop > need) need = loop;
    }
    return need;
}

static bool fused_feedback_eligible(const crb_plan* p, const crb_input_desc* in);
extern "C" int crb_feedback_path(const crb_plan* p) {
    if (!p || p->device < 0) return 0;
    if (loop_eligible(p, nullptr)) return 2;
    return fused_feedback_eligible(p, nullptr) ? 1 : 0;
}

extern "C" int crb_feedback_status(const crb_plan* p, const void* work, int32_t* status, void* stream) {
    if (int rc = need_device(p, "crb_feedback_status")) return rc;
    if (!work || !status) return fail(CRB_EINVAL, "crb_feedback_status: null pointer");
    *status = 0;
    if (!p->loop_used) return CRB_OK;   // (only the persistent stepper keeps a status word)
    unsigned words[2] = {0, 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    HIP_TRY(hipMemcpyAsync(words, work, sizeof(words), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *status = int32_t(words[1]);
    return CRB_OK;
}

namespace {
// the eight launches of one closed-loop RK4 step; t_dev != nullptr: stage times come from the device clock
int feedback_step_launches(const crb_plan* p, void* x, void* acc, void* const bufs[2], void* u, const void* gain, const void* ref,
                           const crb_input_desc* in, double t, const double* t_dev, double dt, void* stream) {
    const double th = t + 0.5 * dt, t1 = t + dt;   // same clock convention as crb_step_rk4
    const double ts[4] = {t, th, th, t1};
    const void* cur = x;
    for (int stage = 0; stage < 4; ++stage) {
        if (int rc = crb_feedback_force(p, cur, gain, ref, u, stream)) return rc;
        void* nxt = bufs[stage & 1];
        if (int rc = rk4_stage_impl(p, x, cur, acc, nxt, u, stage, ts[stage], t_dev, dt, in, stream)) return rc;
        cur = nxt;
    }
    return CRB_OK;
}
}  // namespace

// Beams that live in one wave and whose gain fits LDS: the whole closed-loop rollout as ONE launch of the general
// stepper's feedback instantiation (crb_generic.h, FB) instead of eight launches per step.
static bool fused_feedback_eligible(const crb_plan* p, const crb_input_desc* in) {
    if (p->lognw != 0 || p->NT != 64 || p->mixed_topology || (in && in->f_held)) return false;
    const char* env = std::getenv("CRB_FUSED_FEEDBACK");     // 0 = never, 1 = whenever the gain fits LDS, unset = choose
    if (env && std::atoi(env) == 0) return false;
    const size_t need = p->dtype == CRB_F64 ? fb_lds_bytes<double>(p->NT, p->G, p->n_free) : fb_lds_bytes<float>(p->NT, p->G, p->n_free);
    if (need > size_t(144) * 1024) return false;
    if (env) return true;
    // a workgroup is ONE wave here: with a gain of more than ~50 KB few of them share a CU, and an ensemble that needs
    // several rounds of workgroups is faster through the stage-split path (2048 x 27 elements: 125 against 67 us/step;
    // 64 x 27: 31 against 48; up to 16 elements the fused form wins at every size: 21 against 40 - 48 us/step)
    const size_t per_cu = (size_t(160) * 1024) / need;
    const size_t groups = size_t((p->B + p->G - 1) / p->G);
    return need <= size_t(52) * 1024 || groups <= per_cu * 256;
}
namespace {
template <typename T, int LV>
int launch_fused_feedback_lv(const crb_plan* p, const KParams<T>& k, hipStream_t st) {
    const dim3 grid((p->B + p->G - 1) / p->G), block(p->NT);
    const size_t smem = fb_lds_bytes<T>(p->NT, p->G, p->n_free);
    if (int rc = allow_lds(crb_beam_kernel<T, MODE_STEP, LV, 64, 1, false, true>, smem)) return rc;
    hipLaunchKernelGGL((crb_beam_kernel<T, MODE_STEP, LV, 64, 1, false, true>), grid, block, smem, st, k);
    HIP_TRY(hipGetLastError());
    return CRB_OK;
}
template <typename T>
int fused_feedback_impl(const crb_plan* p, void* x, double t0, double dt, int n_steps, const void* gain, const void* ref,
                        const crb_input_desc* in, hipStream_t st) {
#ifdef CRB_FAST_BUILD
    return fail(CRB_EUNSUPPORTED, "CRB_FAST_BUILD: generic kernels not built");
#else
    if (int rc = ensure_red_map(p)) return rc;
    KParams<T> k = base_params<T>(p);
    k.x = static_cast<T*>(x);
    k.t0 = t0; k.dt = dt; k.n_steps = n_steps;
    arm_status(p, k, n_steps);
    k.rec_slot = -1; k.rec_every = 1;
    if (in && in->kind == CRB_INPUT_IMPULSE) {
        k.amp = static_cast<const T*>(in->amp);
        k.imp_slot = in->node - p->off; k.imp_dof = in->dof; k.duration = in->duration;
        k.imp_node_b = in->node_b;
    }
    k.red_map = p->d_red_map;
    k.n_red = p->n_free;
    k.fb_gain = static_cast<const T*>(gain);
    k.fb_ref = static_cast<const T*>(ref);
    // several beams per wave with 3 .. 5 reduction levels, gravity absent or canonical: the packed LEAN stepper with the feedback
    // inside its stages (its right-hand side costs half of the general kernel's)
    const bool grav = (p->flags & CRB_FORCE_GRAVITY) != 0;
    if (p->G > 1 && lean_eligible(p, nullptr) && p->levels >= 3 && p->levels <= 5 &&
        fb_lean_lds_bytes<T>(p->G, p->n_free) <= size_t(144) * 1024 && std::getenv("CRB_DISABLE_LEAN_FEEDBACK") == nullptr) {
        HIP_TRY(crb::launch_lean_feedback(k, p->B, p->levels, grav, st));
        return CRB_OK;
    }
    switch (p->levels) {   // (a beam inside one wave: at most 6 levels)
        case 0: return launch_fused_feedback_lv<T, 0>(p, k, st);
        case 1: return launch_fused_feedback_lv<T, 1>(p, k, st);
        case 2: return launch_fused_feedback_lv<T, 2>(p, k, st);
        case 3: return launch_fused_feedback_lv<T, 3>(p, k, st);
        case 4: return launch_fused_feedback_lv<T, 4>(p, k, st);
        case 5: return launch_fused_feedback_lv<T, 5>(p, k, st);
        case 6: return launch_fused_feedback_lv<T, 6>(p, k, st);
        default: return fail(CRB_EUNSUPPORTED, "crb_step_rk4_feedback: unsupported number of cyclic-reduction levels");
    }
#endif
}
}  // namespace

extern "C" int crb_step_rk4_feedback(const crb_plan* p, void* x, double t0, double dt, int n_steps, const void* gain,
                                     const void* ref, const crb_input_desc* in, void* work, double* t_end, void* stream) {
    if (int rc = need_device(p, "crb_step_rk4_feedback")) return rc;
    if (!x || !gain || !work) return fail(CRB_EINVAL, "crb_step_rk4_feedback: null pointer");
    if (n_steps < 0 || !(dt > 0)) return fail(CRB_EINVAL, "crb_step_rk4_feedback: n_steps >= 0 and dt > 0 required");
    p->loop_used = false;
    if (loop_eligible(p, in)) {
        if (in && in->kind == CRB_INPUT_IMPULSE &&
            (in->node < 0 || in->node >= p->n_node || in->dof < 0 || in->dof > 2 || !in->amp || !p->any_free[3 * in->node + in->dof]))
            return fail(CRB_EINVAL, "crb_step_rk4_feedback: bad impulse description");
        if (in && in->kind != CRB_INPUT_IMPULSE && in->kind != CRB_INPUT_NONE)
            return fail(CRB_EINVAL, "crb_step_rk4_feedback: unknown input kind");
        if (t_end) {
            double t = t0;
            for (int i = 0; i < n_steps; ++i) t = t + dt;
            *t_end = t;
        }
        if (n_steps == 0) return CRB_OK;
        if (int rc = ensure_red_map(p)) return rc;
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int nb = loop_nb(p), n_rb = (p->B + 63) / 64;
        const crb::LoopWork lay = crb::loop_work_layout(nb, n_rb < crb::LOOP_MAX_GROUPS ? n_rb : crb::LOOP_MAX_GROUPS);
        char* w = static_cast<char*>(work);
        crb::LoopParams<double> P;
        std::memset(&P, 0, sizeof(P));
        P.k = base_params<double>(p);
        P.k.x = static_cast<double*>(x);
        P.k.t0 = t0; P.k.dt = dt; P.k.n_steps = n_steps;
        arm_status(p, P.k, n_steps);
        if (in && in->kind == CRB_INPUT_IMPULSE) {
            P.k.amp = static_cast<const double*>(in->amp);
            P.k.imp_slot = in->node - p->off; P.k.imp_dof = in->dof; P.k.duration = in->duration;
            P.k.imp_node_b = in->node_b;
        }
        P.ref = static_cast<const double*>(ref);
        P.red_map = p->d_red_map;
        P.n_red = p->n_free;
        P.sync = reinterpret_cast<unsigned*>(w);
        P.kfrag = reinterpret_cast<const double*>(w + lay.kfrag);
        P.ebuf = reinterpret_cast<double*>(w + lay.ebuf);
        P.ubuf = reinterpret_cast<double*>(w + lay.ubuf);
        P.ownbuf = reinterpret_cast<double*>(w + lay.ownbuf);
        P.n_rb = n_rb;
        const char* fenv = std::getenv("CRB_LOOP_FENCES");
        P.fences = fenv ? std::atoi(fenv) : 0;
        const char* tenv = std::getenv("CRB_LOOP_TIMEOUT_MS");
        P.timeout = (unsigned long long)(tenv ? std::atol(tenv) : 2000) * 100000ull;   // 100 MHz ticks
        // every polled word starts at zero, in every call
        HIP_TRY(hipMemsetAsync(w, 0, size_t(crb::LOOP_SYNC_WORDS) * sizeof(unsigned), st));
        const bool grav = (p->flags & CRB_FORCE_GRAVITY) != 0;
        const hipError_t e = p->lognw == 1 ? crb::launch_loop_long(P, static_cast<const double*>(gain), p->levels, grav, p->elem_mode, st)
                                           : crb::launch_loop_short(P, static_cast<const double*>(gain), p->levels, grav, p->elem_mode, st);
        if (e != hipSuccess) return fail(CRB_EHIP, std::string("crb_step_rk4_feedback (persistent stepper): ") + hipGetErrorString(e));
        p->loop_used = true;
        return CRB_OK;
    }
    if (fused_feedback_eligible(p, in)) {
        if (in && in->kind == CRB_INPUT_IMPULSE &&
            (in->node < 0 || in->node >= p->n_node || in->dof < 0 || in->dof > 2 || !in->amp || !p->any_free[3 * in->node + in->dof]))
            return fail(CRB_EINVAL, "crb_step_rk4_feedback: bad impulse description");
        if (in && in->kind != CRB_INPUT_IMPULSE && in->kind != CRB_INPUT_NONE)
            return fail(CRB_EINVAL, "crb_step_rk4_feedback: unknown input kind");
        if (t_end) {
            double t = t0;
            for (int i = 0; i < n_steps; ++i) t = t + dt;
            *t_end = t;
        }
        if (n_steps == 0) return CRB_OK;
        hipStream_t st = static_cast<hipStream_t>(stream);
        return p->dtype == CRB_F64 ? fused_feedback_impl<double>(p, x, t0, dt, n_steps, gain, ref, in, st)
                                   : fused_feedback_impl<float>(p, x, t0, dt, n_steps, gain, ref, in, st);
    }
    const size_t state = size_t(p->B) * 2 * p->n_node * 4 * (p->dtype == CRB_F64 ? sizeof(double) : sizeof(float));
    char* w = static_cast<char*>(work);
    void* acc = w;
    void* bufs[2] = {w + state, w + 2 * state};
    void* u = w + 3 * state;
    double* t_dev = reinterpret_cast<double*>(w + 3 * state + state / 2);
    hipStream_t user = static_cast<hipStream_t>(stream);
    double t = t0;
    // CRB_USE_GRAPH=1: one step (8 launches + clock) is captured into a hipGraph and replayed; the stage kernels
    // then read the stage time from a device clock that the graph's last node advances, so that no launch
    // argument changes from step to step.  Opt-in: on ROCm 7.2 the replay measured SLOWER than the plain
    // launches (70 vs 65 us per step for 1..256 beams of 64 elements, 218 vs 205 us at 2048 x 128) -- the
    // small-ensemble loop is bound by the ~7 us of dependent-kernel latency per launch, which a graph of
    // kernel nodes does not remove.
    const char* genv = std::getenv("CRB_USE_GRAPH");
    const bool use_graph = n_steps >= 8 && genv && std::atoi(genv) != 0;
    if (!use_graph) {
        // entries of u outside the free DOFs are never written by the GEMM and must read as zero
        HIP_TRY(hipMemsetAsync(u, 0, state / 2, user));
        for (int s = 0; s < n_steps; ++s) {
            if (int rc = feedback_step_launches(p, x, acc, bufs, u, gain, ref, in, t, nullptr, dt, stream)) return rc;
            t = t + dt;
        }
        if (t_end) *t_end = t;
        return CRB_OK;
    }
    if (!p->aux_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&p->aux_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&p->aux_in, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&p->aux_out, hipEventDisableTiming));
    }
    hipStream_t aux = p->aux_stream;
    HIP_TRY(hipEventRecord(p->aux_in, user));                 // everything the caller queued so far ...
    HIP_TRY(hipStreamWaitEvent(aux, p->aux_in, 0));           // ... happens before the replayed steps
    HIP_TRY(hipMemsetAsync(u, 0, state / 2, aux));
    hipLaunchKernelGGL(crb_clock_kernel<0>, dim3(1), dim3(64), 0, aux, t_dev, dt, t0, 1);
    HIP_TRY(hipGetLastError());
    // the captured step depends on every pointer and scalar below: rebuild when one of them changes
    std::vector<uint64_t> key = {uint64_t(reinterpret_cast<uintptr_t>(x)), uint64_t(reinterpret_cast<uintptr_t>(gain)),
                                 uint64_t(reinterpret_cast<uintptr_t>(ref)), uint64_t(reinterpret_cast<uintptr_t>(work))};
    uint64_t dtb;
    std::memcpy(&dtb, &dt, sizeof(dtb));
    key.push_back(dtb);
    if (in) {
        uint64_t dur;
        std::memcpy(&dur, &in->duration, sizeof(dur));
        key.insert(key.end(), {uint64_t(in->kind), uint64_t(in->node), uint64_t(in->dof), dur,
                               uint64_t(reinterpret_cast<uintptr_t>(in->amp)), uint64_t(reinterpret_cast<uintptr_t>(in->f_held)),
                               uint64_t(reinterpret_cast<uintptr_t>(in->node_b))});
    }
    const char* tile = std::getenv("CRB_FEEDBACK_TILE");
    key.push_back(tile ? uint64_t(std::atoi(tile)) : 0);
    if (!p->step_exec || key != p->step_key) {
        if (p->step_exec) {
            HIP_TRY(hipStreamSynchronize(aux));                // the old graph may still be running
            HIP_TRY(hipGraphExecDestroy(p->step_exec));
            p->step_exec = nullptr;
        }
        hipGraph_t graph = nullptr;
        HIP_TRY(hipStreamBeginCapture(aux, hipStreamCaptureModeThreadLocal));
        int rc = feedback_step_launches(p, x, acc, bufs, u, gain, ref, in, 0.0, t_dev, dt, aux);
        if (rc == CRB_OK) {
            hipLaunchKernelGGL(crb_clock_kernel<0>, dim3(1), dim3(64), 0, aux, t_dev, dt, 0.0, 0);
            if (hipGetLastError() != hipSuccess) rc = fail(CRB_EHIP, "crb_step_rk4_feedback: clock kernel launch failed");
        }
        const hipError_t ec = hipStreamEndCapture(aux, &graph);   // (always: leaves the stream usable)
        if (rc != CRB_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        HIP_TRY(ec);
        const hipError_t ei = hipGraphInstantiate(&p->step_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIP_TRY(ei);
        p->step_key = key;
    }
    for (int s = 0; s < n_steps; ++s) {
        HIP_TRY(hipGraphLaunch(p->step_exec, aux));
        t = t + dt;
    }
    HIP_TRY(hipEventRecord(p->aux_out, aux));
    HIP_TRY(hipStreamWaitEvent(user, p->aux_out, 0));         // the caller's stream continues after the rollout
    if (t_end) *t_end = t;
    return CRB_OK;
}

namespace {
template <typename T, int LV>
int launch_rk45_lv(const crb_plan* p, const KParams<T>& k, const Rk45Params& q, hipStream_t st) {
    const dim3 grid(p->B), block(p->NT);
    const size_t smem = rk45_lds_bytes<T>(p->NT);
    if (p->NT <= 256) {
        if (int rc = allow_lds(crb_rk45_kernel<T, LV, 256, 1>, smem)) return rc;
        hipLaunchKernelGGL((crb_rk45_kernel<T, LV, 256, 1>), grid, block, smem, st, k, q);
    } else if constexpr (LV >= 4) {
        if (int rc = allow_lds(crb_rk45_kernel<T, LV, 1024, 1>, smem)) return rc;
        hipLaunchKernelGGL((crb_rk45_kernel<T, LV, 1024, 1>), grid, block, smem, st, k, q);
    } else {
        return fail(CRB_EUNSUPPORTED, "a beam of more than 256 thread-carried nodes with fewer than 4 cyclic-reduction levels");
    }
    HIP_TRY(hipGetLastError());
    return CRB_OK;
}
template <typename T>
int launch_rk45(const crb_plan* p, const KParams<T>& k, const Rk45Params& q, hipStream_t st) {
#ifdef CRB_FAST_BUILD
    return fail(CRB_EUNSUPPORTED, "CRB_FAST_BUILD: rk45 not built");
#else
    // plans without gravity, one beam per workgroup of <= 4 waves: the lean RHS (crb_lean.hip)
    const int lv_long = p->dtype == CRB_F64 ? 5 : 4;
    const bool levels_ok = p->lognw == 0 ? (p->levels >= 3 && p->levels <= 6) : (p->levels == lv_long || p->levels == lv_long + 1);
    if (!(p->flags & CRB_FORCE_GRAVITY) && p->NT == (64 << p->lognw) && p->lognw <= 2 && levels_ok &&
        std::getenv("CRB_DISABLE_LEAN") == nullptr) {
        HIP_TRY(crb::launch_rk45_lean(k, q, p->B, p->levels, p->lognw, p->elem_mode, st));
        return CRB_OK;
    }
    switch (p->levels) {
        case 0: return launch_rk45_lv<T, 0>(p, k, q, st);
        case 1: return launch_rk45_lv<T, 1>(p, k, q, st);
        case 2: return launch_rk45_lv<T, 2>(p, k, q, st);
        case 3: return launch_rk45_lv<T, 3>(p, k, q, st);
        case 4: return launch_rk45_lv<T, 4>(p, k, q, st);
        case 5: return launch_rk45_lv<T, 5>(p, k, q, st);
        case 6: return launch_rk45_lv<T, 6>(p, k, q, st);
        default: return fail(CRB_EUNSUPPORTED, "crb_solve_rk45: unsupported number of cyclic-reduction levels");
    }
#endif
}
}  // namespace

extern "C" int crb_solve_rk45(const crb_plan* p, void* x, double t0, double t_end, double rtol, double atol,
                              const crb_input_desc* in, void* h, void* stats, int max_steps, void* stream) {
    return crb_solve_rk45_eval(p, x, t0, t_end, rtol, atol, in, h, stats, max_steps, nullptr, 0.0, 0.0, 0, stream);
}

extern "C" int crb_solve_rk45_eval(const crb_plan* p, void* x, double t0, double t_end, double rtol, double atol,
                                   const crb_input_desc* in, void* h, void* stats, int max_steps,
                                   const crb_record_desc* rec, double eval_t0, double eval_dt, int n_eval, void* stream) {
    if (int rc = need_device(p, "crb_solve_rk45")) return rc;
    const bool eval_all = rec && rec->node == CRB_RECORD_ALL;
    if (rec && n_eval > 0) {
        if ((!eval_all && (rec->plane < 0 || rec->plane > 1 || rec->node < 0 || rec->node >= p->n_node || rec->dof < 0 ||
                           rec->dof > 2)) ||
            !rec->out || !(eval_dt > 0) || eval_t0 < t0)
            return fail(CRB_EINVAL, "crb_solve_rk45_eval: bad t_eval description");
    }
    if (!x) return fail(CRB_EINVAL, "crb_solve_rk45: null state");
    if (!(t_end > t0)) return fail(CRB_EINVAL, "crb_solve_rk45: t_end must be greater than t0");
    if (!(rtol > 0) || !(atol >= 0)) return fail(CRB_EINVAL, "crb_solve_rk45: tolerances must be positive");
    // (beams of fewer than 64 slots run one per wave here, not packed: every beam has its own step sequence)
    if (rk45_lds_bytes<double>(p->NT) > 160 * 1024)
        return fail(CRB_EUNSUPPORTED, "crb_solve_rk45: beam too long for the LDS-resident stage storage");
    int imp_slot = -1, imp_dof = 0;
    double duration = 0.0;
    const void* amp = nullptr;
    const void* held = nullptr;
    if (in) {
        held = in->f_held;
        if (in->kind == CRB_INPUT_IMPULSE) {
            if (in->node < 0 || in->node >= p->n_node || in->dof < 0 || in->dof > 2 || !in->amp ||
                !p->any_free[3 * in->node + in->dof])
                return fail(CRB_EINVAL, "crb_solve_rk45: bad impulse description");
            imp_slot = in->node - p->off; imp_dof = in->dof; duration = in->duration; amp = in->amp;
        }
    }
    Rk45Params q;
    q.t0 = t0; q.t_end = t_end; q.rtol = rtol; q.atol = atol;
    q.h_io = static_cast<double*>(h); q.stats = static_cast<int32_t*>(stats);
    q.n_state = 2 * p->n_free; q.n_state_b = p->d_n_state; q.max_steps = max_steps > 0 ? max_steps : 100000000;
    q.eval_out = nullptr; q.eval_t0 = eval_t0; q.eval_dt = eval_dt; q.n_eval = 0; q.eval_slot = -1; q.eval_comp = 0;
    if (rec && n_eval > 0 && eval_all) {
        q.eval_out = rec->out; q.n_eval = n_eval; q.eval_slot = REC_ALL_SLOTS;
    } else if (rec && n_eval > 0 && rec->node - p->off >= 0) {
        q.eval_out = rec->out; q.n_eval = n_eval; q.eval_slot = rec->node - p->off; q.eval_comp = rec->plane * 3 + rec->dof;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->dtype == CRB_F64) {
        KParams<double> k = base_params<double>(p);
        k.x = static_cast<double*>(x); k.u_held = static_cast<const double*>(held); k.amp = static_cast<const double*>(amp);
        k.imp_slot = imp_slot; k.imp_dof = imp_dof; k.duration = duration;
        k.imp_node_b = (in && in->kind == CRB_INPUT_IMPULSE) ? in->node_b : nullptr;
        return launch_rk45<double>(p, k, q, st);
    }
    KParams<float> k = base_params<float>(p);
    k.x = static_cast<float*>(x); k.u_held = static_cast<const float*>(held); k.amp = static_cast<const float*>(amp);
    k.imp_slot = imp_slot; k.imp_dof = imp_dof; k.duration = duration;
    k.imp_node_b = (in && in->kind == CRB_INPUT_IMPULSE) ? in->node_b : nullptr;
    return launch_rk45<float>(p, k, q, st);
}

namespace {
template <typename T, int BM, int BN, int BK, int WR>
int launch_feedback_tile(const FeedbackParams<T>& f, int ksplit, hipStream_t st) {
    const dim3 grid((f.B + BM - 1) / BM, (f.n + BN - 1) / BN, ksplit);
    const size_t smem = feedback_lds_bytes<T, BM, BN, BK>(f.n2);
    if (ksplit > 1) HIP_TRY(hipMemsetAsync(f.u, 0, size_t(f.B) * f.u_stride * sizeof(T), st));
    if (f.ref) {
        if (int rc = allow_lds(crb_feedback_kernel<T, BM, BN, BK, WR, true>, smem)) return rc;
        hipLaunchKernelGGL((crb_feedback_kernel<T, BM, BN, BK, WR, true>), grid, dim3(256), smem, st, f);
    } else {
        if (int rc = allow_lds(crb_feedback_kernel<T, BM, BN, BK, WR, false>, smem)) return rc;
        hipLaunchKernelGGL((crb_feedback_kernel<T, BM, BN, BK, WR, false>), grid, dim3(256), smem, st, f);
    }
    return CRB_OK;
}
template <typename T, int BN, int BK>
int launch_feedback_ws(const FeedbackParams<T>& f, hipStream_t st) {
    const dim3 grid((f.B + 63) / 64, (f.n + BN - 1) / BN);
    const size_t smem = feedback_lds_bytes<T, 64, BN, BK>(f.n2);
    if (f.ref) {
        if (int rc = allow_lds(crb_feedback_ws_kernel<T, BN, BK, true>, smem)) return rc;
        hipLaunchKernelGGL((crb_feedback_ws_kernel<T, BN, BK, true>), grid, dim3(256 + WS_NL), smem, st, f);
    } else {
        if (int rc = allow_lds(crb_feedback_ws_kernel<T, BN, BK, false>, smem)) return rc;
        hipLaunchKernelGGL((crb_feedback_ws_kernel<T, BN, BK, false>), grid, dim3(256 + WS_NL), smem, st, f);
    }
    return CRB_OK;
}
// tile: 0 = choose.  Large ensembles: 64 x 48 outputs per workgroup, wave-specialised (one workgroup per CU
// and the fewest L2 reads at the config-5 shape: 32.7 us at 2048 x 768 x 384 against 37 us for 32 x 32);
// small ensembles: 32 x 32 for the larger grid.  CRB_FEEDBACK_TILE=48|32 forces one (tests cover both).
template <typename T>
int launch_feedback(int tile, const FeedbackParams<T>& f, hipStream_t st) {
    const long wide_groups = long((f.B + 63) / 64) * ((f.n + 47) / 48);
    const bool wide_fits = feedback_lds_bytes<T, 64, 48, 64>(f.n2) <= size_t(160) * 1024;
    if (tile == 0) tile = (wide_groups >= 192 && wide_fits) ? 48 : 32;
    if (tile == 48 && wide_fits) return launch_feedback_ws<T, 48, 64>(f, st);
    return launch_feedback_tile<T, 32, 32, 32, 2>(f, 1, st);
}
template <typename T>
int feedback_force_impl(const crb_plan* p, const void* xs, const void* gain, const void* ref, void* u, void* stream) {
    FeedbackParams<T> f;
    f.xs = static_cast<const T*>(xs);
    f.ref = static_cast<const T*>(ref);
    f.gain = static_cast<const T*>(gain);
    f.u = static_cast<T*>(u);
    f.col_off = p->d_col_off;
    f.row_off = p->d_row_off;
    f.B = p->B; f.n = p->n_free; f.n2 = 2 * p->n_free;
    f.x_stride = size_t(2) * p->n_node * 4;
    f.u_stride = size_t(p->n_node) * 4;
    f.beam_idx = nullptr; f.ref_ld = 2 * p->n_free; f.ref_half = p->n_free;
    const char* force = std::getenv("CRB_FEEDBACK_TILE");
    if (int rc = launch_feedback<T>(force ? std::atoi(force) : 0, f, static_cast<hipStream_t>(stream))) return rc;
    HIP_TRY(hipGetLastError());
    return CRB_OK;
}
}  // namespace

extern "C" int crb_feedback_force(const crb_plan* p, const void* xs, const void* gain, const void* ref, void* u, void* stream) {
    if (int rc = need_device(p, "crb_feedback_force")) return rc;
    if (!xs || !gain || !u) return fail(CRB_EINVAL, "crb_feedback_force: null pointer");
    if (p->mixed_topology)
        return fail(CRB_EUNSUPPORTED, "crb_feedback_force: one gain matrix for the ensemble needs one free-DOF set (uniform boundary conditions / lengths)");
    return p->dtype == CRB_F64 ? feedback_force_impl<double>(p, xs, gain, ref, u, stream)
                               : feedback_force_impl<float>(p, xs, gain, ref, u, stream);
}

namespace {
void free_gain_groups(const crb_plan* p) {
    for (auto& g : p->gain_groups) {
        if (g.beam_idx) (void)hipFree(g.beam_idx);
        if (g.col) (void)hipFree(g.col);
        if (g.row) (void)hipFree(g.row);
    }
    p->gain_groups.clear();
    p->gain_group_key.clear();
}
// device tables for a beam -> group assignment: per group the list of its beams and the offsets of ITS reduced ordering
// inside the device layouts (the beams of a group must share the free-DOF set); cached for the assignment last used
int ensure_gain_groups(const crb_plan* p, int n_groups, const int32_t* beam_group) {
    std::vector<int32_t> key(beam_group, beam_group + p->B);
    key.push_back(n_groups);
    if (key == p->gain_group_key) return CRB_OK;
    free_gain_groups(p);
    std::vector<std::vector<int32_t>> members(static_cast<size_t>(n_groups));
    for (int b = 0; b < p->B; ++b) {
        const int g = beam_group[b];
        if (g >= n_groups) return fail(CRB_EINVAL, "crb_feedback_force_grouped: group index out of range");
        if (g >= 0) members[size_t(g)].push_back(b);
    }
    auto free_index_of = [&](int b) -> const std::vector<int32_t>& { return p->beam_free_index.empty() ? p->free_index : p->beam_free_index[size_t(b)]; };
    p->gain_groups.resize(static_cast<size_t>(n_groups));
    for (int g = 0; g < n_groups; ++g) {
        const auto& m = members[size_t(g)];
        crb_plan::GainGroup& G = p->gain_groups[size_t(g)];
        G.count = int(m.size());
        if (m.empty()) continue;
        const std::vector<int32_t>& fi = free_index_of(m[0]);
        for (int b : m)
            if (free_index_of(b) != fi) {
                free_gain_groups(p);
                return fail(CRB_EINVAL, "crb_feedback_force_grouped: the beams of a gain group must share one free-DOF set (boundary conditions and length)");
            }
        G.n = int(fi.size());
        std::vector<int32_t> col(size_t(2) * G.n), row(size_t(G.n));
        for (int r = 0; r < G.n; ++r) {
            row[r] = (fi[r] / 3) * 4 + (fi[r] % 3);
            col[r] = row[r];
            col[G.n + r] = p->n_node * 4 + row[r];
        }
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&G.beam_idx), m.size() * sizeof(int32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&G.col), col.size() * sizeof(int32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&G.row), row.size() * sizeof(int32_t)));
        HIP_TRY(hipMemcpy(G.beam_idx, m.data(), m.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(G.col, col.data(), col.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(G.row, row.data(), row.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    p->gain_group_key = key;
    return CRB_OK;
}
template <typename T>
int feedback_force_grouped_impl(const crb_plan* p, const void* xs, int n_groups, const void* const* gains, const void* ref, void* u,
                                void* stream) {
    for (int g = 0; g < n_groups; ++g) {
        const crb_plan::GainGroup& G = p->gain_groups[size_t(g)];
        if (G.count == 0) continue;
        if (!gains[g]) return fail(CRB_EINVAL, "crb_feedback_force_grouped: null gain for a group that has beams");
        FeedbackParams<T> f;
        f.xs = static_cast<const T*>(xs);
        f.ref = static_cast<const T*>(ref);
        f.gain = static_cast<const T*>(gains[g]);
        f.u = static_cast<T*>(u);
        f.col_off = G.col;
        f.row_off = G.row;
        f.B = G.count; f.n = G.n; f.n2 = 2 * G.n;
        f.x_stride = size_t(2) * p->n_node * 4;
        f.u_stride = size_t(p->n_node) * 4;
        f.beam_idx = G.beam_idx; f.ref_ld = 2 * p->n_free; f.ref_half = p->n_free;
        const char* force = std::getenv("CRB_FEEDBACK_TILE");
        if (int rc = launch_feedback<T>(force ? std::atoi(force) : 0, f, static_cast<hipStream_t>(stream))) return rc;
        HIP_TRY(hipGetLastError());
    }
    return CRB_OK;
}
}  // namespace

extern "C" int crb_feedback_force_grouped(const crb_plan* p, const void* xs, int n_groups, const int32_t* beam_group,
                                          const void* const* gains, const void* ref, void* u, void* stream) {
    if (int rc = need_device(p, "crb_feedback_force_grouped")) return rc;
    if (!xs || !u || !beam_group || !gains || n_groups < 1) return fail(CRB_EINVAL, "crb_feedback_force_grouped: null pointer or no group");
    if (int rc = ensure_gain_groups(p, n_groups, beam_group)) return rc;
    return p->dtype == CRB_F64 ? feedback_force_grouped_impl<double>(p, xs, n_groups, gains, ref, u, stream)
                               : feedback_force_grouped_impl<float>(p, xs, n_groups, gains, ref, u, stream);
}

static int rk4_stage_impl(const crb_plan* p, void* x, const void* xs, void* acc, void* xs_next, const void* u_stage, int stage,
                         double t_stage, const double* t_dev, double dt, const crb_input_desc* in, void* stream);

extern "C" int crb_step_rk4_feedback_grouped(const crb_plan* p, void* x, double t0, double dt, int n_steps, int n_groups,
                                             const int32_t* beam_group, const void* const* gains, const void* ref,
                                             const crb_input_desc* in, void* work, double* t_end, void* stream) {
    if (int rc = need_device(p, "crb_step_rk4_feedback_grouped")) return rc;
    if (!x || !work || !beam_group || !gains || n_groups < 1) return fail(CRB_EINVAL, "crb_step_rk4_feedback_grouped: null pointer or no group");
    if (n_steps < 0 || !(dt > 0)) return fail(CRB_EINVAL, "crb_step_rk4_feedback_grouped: n_steps >= 0 and dt > 0 required");
    if (int rc = ensure_gain_groups(p, n_groups, beam_group)) return rc;
    p->loop_used = false;
    const size_t state = size_t(p->B) * 2 * p->n_node * 4 * (p->dtype == CRB_F64 ? sizeof(double) : sizeof(float));
    char* w = static_cast<char*>(work);
    void* acc = w;
    void* bufs[2] = {w + state, w + 2 * state};
    void* u = w + 3 * state;
    // entries of u that no group writes (constrained DOFs, beams without a gain) must read as zero
    HIP_TRY(hipMemsetAsync(u, 0, state / 2, static_cast<hipStream_t>(stream)));
    double t = t0;
    for (int s = 0; s < n_steps; ++s) {
        const double th = t + 0.5 * dt, t1 = t + dt;   // same clock convention as crb_step_rk4
        const double ts[4] = {t, th, th, t1};
        const void* cur = x;
        for (int stage = 0; stage < 4; ++stage) {
            if (int rc = crb_feedback_force_grouped(p, cur, n_groups, beam_group, gains, ref, u, stream)) return rc;
            void* nxt = bufs[stage & 1];
            if (int rc = rk4_stage_impl(p, x, cur, acc, nxt, u, stage, ts[stage], nullptr, dt, in, stream)) return rc;
            cur = nxt;
        }
        t = t + dt;
    }
    if (t_end) *t_end = t;
    return CRB_OK;
}

extern "C" int crb_rk4_stage(const crb_plan* p, void* x, const void* xs, void* acc, void* xs_next, const void* u_stage,
                             int stage, double t_stage, double dt, const crb_input_desc* in, void* stream) {
    return rk4_stage_impl(p, x, xs, acc, xs_next, u_stage, stage, t_stage, nullptr, dt, in, stream);
}
static int rk4_stage_impl(const crb_plan* p, void* x, const void* xs, void* acc, void* xs_next, const void* u_stage, int stage,
                         double t_stage, const double* t_dev, double dt, const crb_input_desc* in, void* stream) {
    if (int rc = need_device(p, "crb_rk4_stage")) return rc;
    if (!x || !xs || !acc) return fail(CRB_EINVAL, "crb_rk4_stage: null pointer");
    if (stage < 0 || stage > 3) return fail(CRB_EINVAL, "crb_rk4_stage: stage must be 0..3");
    if (stage < 3 && (!xs_next || xs_next == xs)) return fail(CRB_EINVAL, "crb_rk4_stage: xs_next must be a distinct buffer");
    if (!(dt > 0)) return fail(CRB_EINVAL, "crb_rk4_stage: dt must be positive");
    int imp_slot = -1, imp_dof = 0;
    double duration = 0.0;
    const void* amp = nullptr;
    if (in && in->kind == CRB_INPUT_IMPULSE) {
        if (in->node < 0 || in->node >= p->n_node || in->dof < 0 || in->dof > 2 || !in->amp ||
            !p->any_free[3 * in->node + in->dof])
            return fail(CRB_EINVAL, "crb_rk4_stage: bad impulse description");
        imp_slot = in->node - p->off; imp_dof = in->dof; duration = in->duration; amp = in->amp;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (p->dtype == CRB_F64) {
        KParams<double> k = base_params<double>(p);
        k.x = static_cast<double*>(x); k.xs = static_cast<const double*>(xs); k.acc = static_cast<double*>(acc);
        k.out = static_cast<double*>(xs_next); k.u_held = static_cast<const double*>(u_stage);
        k.amp = static_cast<const double*>(amp);
        k.imp_slot = imp_slot; k.imp_dof = imp_dof; k.duration = duration;
        k.imp_node_b = (in && in->kind == CRB_INPUT_IMPULSE) ? in->node_b : nullptr;
        k.stage = stage; k.t0 = t_stage; k.dt = dt; k.t_dev = t_dev;
    if (stage == 3) arm_status(p, k, 1);
        if (stage_lean_eligible(p)) return launch_stage_lean<double>(p, k, st);
        return launch_beam<double, MODE_STAGE>(p, k, st);
    }
    KParams<float> k = base_params<float>(p);
    k.x = static_cast<float*>(x); k.xs = static_cast<const float*>(xs); k.acc = static_cast<float*>(acc);
    k.out = static_cast<float*>(xs_next); k.u_held = static_cast<const float*>(u_stage);
    k.amp = static_cast<const float*>(amp);
    k.imp_slot = imp_slot; k.imp_dof = imp_dof; k.duration = duration;
    k.imp_node_b = (in && in->kind == CRB_INPUT_IMPULSE) ? in->node_b : nullptr;
    k.stage = stage; k.t0 = t_stage; k.dt = dt; k.t_dev = t_dev;
    if (stage == 3) arm_status(p, k, 1);
    if (stage_lean_eligible(p)) return launch_stage_lean<float>(p, k, st);
    return launch_beam<float, MODE_STAGE>(p, k, st);
}

extern "C" int crb_gather_dof(const crb_plan* p, const void* x, int plane, int node, int dof, void* out, void* stream) {
    if (int rc = need_device(p, "crb_gather_dof")) return rc;
    if (!x || !out) return fail(CRB_EINVAL, "crb_gather_dof: null pointer");
    if (plane < 0 || plane > 1 || node < 0 || node >= p->n_node || dof < 0 || dof > 2)
        return fail(CRB_EINVAL, "crb_gather_dof: index out of range");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t stride = size_t(2) * p->n_node * 4, off = (size_t(plane) * p->n_node + node) * 4 + dof;
    const int bs = 256, grid = (p->B + bs - 1) / bs;
    if (p->dtype == CRB_F64)
        hipLaunchKernelGGL((crb_gather_kernel<double>), dim3(grid), dim3(bs), 0, st, static_cast<const double*>(x), stride,
                           off, p->B, static_cast<double*>(out));
    else
        hipLaunchKernelGGL((crb_gather_kernel<float>), dim3(grid), dim3(bs), 0, st, static_cast<const float*>(x), stride, off,
                           p->B, static_cast<float*>(out));
    HIP_TRY(hipGetLastError());
    return CRB_OK;
}
