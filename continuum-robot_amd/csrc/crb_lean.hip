// crb_lean.hip -- instantiations and launch of crb_step_lean_kernel for ONE dtype (-DCRB_LEAN_T=double|float).
#include <cstdlib>

#include "crb_lean_launch.h"

#ifndef CRB_LEAN_T
#error "compile with -DCRB_LEAN_T=double or -DCRB_LEAN_T=float"
#endif

namespace crb {
namespace {
typedef CRB_LEAN_T T;

// Workgroups a launch of `kernel` keeps resident on the device (CUs x workgroups per CU by the occupancy query).
template <typename K>
int resident_groups(K kernel, int threads, size_t smem) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, smem) != hipSuccess || cus < 1 || per_cu < 1)
        return 0;
    return cus * per_cu;
}

template <int LV, int LOGNW, bool GRAV, int EM, bool HELD, bool PACK = false>
hipError_t one_held(const KParams<T>& k, int n_beams, hipStream_t st) {
    const int groups = PACK ? (n_beams + k.G - 1) / k.G : n_beams;
    const size_t smem = lean_lds_bytes<T>(64 << LOGNW, LOGNW);
    auto kernel = crb_step_lean_kernel<T, LV, LOGNW, GRAV, EM, HELD, PACK>;
    if (smem > 64 * 1024) {  // dynamic LDS above 64 KiB is opt-in per kernel (the CU has 160 KiB)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(smem));
        if (e != hipSuccess) return e;
    }
    // Shared-table plans: a workgroup loads its rows of the solve tables once and walks over several beams, so the
    // grid is what the device keeps resident, split evenly (4096 beams = 8 beams for each of 512 workgroups).
    // Per-beam tables are reloaded per beam anyway: one workgroup per beam, dispatched by the hardware.
    int grid = groups;
    const bool shared = k.slot_stride == 0 && k.lv_stride == 0 && k.fin_stride == 0;
    static int resident = -1;   // (per instantiation; every device of a node is the same part)
    if (shared && std::getenv("CRB_LEAN_NO_WALK") == nullptr) {
        if (resident < 0) resident = resident_groups(kernel, 64 << LOGNW, smem);
        int cap = resident;
        if (const char* env = std::getenv("CRB_LEAN_MAX_GROUPS")) cap = std::atoi(env);   // (tests: walk with a handful of beams)
        if (cap > 0 && groups > cap) {
            const int rounds = (groups + cap - 1) / cap;
            grid = (groups + rounds - 1) / rounds;
        }
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 << LOGNW), smem, st, k);
    return hipGetLastError();
}
template <int LV, int LOGNW, bool GRAV, int EM>
hipError_t one(const KParams<T>& k, int n_beams, hipStream_t st) {
#ifdef CRB_FAST_BUILD
    return one_held<LV, LOGNW, GRAV, EM, false>(k, n_beams, st);
#else
    if (LOGNW == 0 && k.G > 1)   // several beams per wave
        return k.u_held ? one_held<LV, 0, GRAV, EM, true, true>(k, n_beams, st) : one_held<LV, 0, GRAV, EM, false, true>(k, n_beams, st);
    return k.u_held ? one_held<LV, LOGNW, GRAV, EM, true>(k, n_beams, st) : one_held<LV, LOGNW, GRAV, EM, false>(k, n_beams, st);
#endif
}
template <int LV, int LOGNW, bool GRAV, int EM>
hipError_t one_stage(const KParams<T>& k, int n_groups, hipStream_t st) {
    const dim3 grid(n_groups), block(64 << LOGNW);
    const size_t smem = stage_lean_lds_bytes<T>(64 << LOGNW, LOGNW);
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(crb_stage_lean_kernel<T, LV, LOGNW, GRAV, EM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(smem));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((crb_stage_lean_kernel<T, LV, LOGNW, GRAV, EM>), grid, block, smem, st, k);
    return hipGetLastError();
}
// STAGE selects the kernel family: the fused multi-step stepper or the one-stage kernel
template <int LV, int LOGNW, bool GRAV, bool STAGE>
hipError_t by_em(const KParams<T>& k, int n, int em, hipStream_t st) {
    if (STAGE) {
        switch (em) {
            case EM_LINEAR: return one_stage<LV, LOGNW, GRAV, EM_LINEAR>(k, n, st);
            case EM_NONLINEAR: return one_stage<LV, LOGNW, GRAV, EM_NONLINEAR>(k, n, st);
            default: return one_stage<LV, LOGNW, GRAV, EM_MIXED>(k, n, st);
        }
    }
    switch (em) {
        case EM_LINEAR: return one<LV, LOGNW, GRAV, EM_LINEAR>(k, n, st);
        case EM_NONLINEAR: return one<LV, LOGNW, GRAV, EM_NONLINEAR>(k, n, st);
        default: return one<LV, LOGNW, GRAV, EM_MIXED>(k, n, st);
    }
}
template <int LV, bool GRAV, bool STAGE>
hipError_t by_nw(const KParams<T>& k, int n, int lognw, int em, hipStream_t st) {
    // Beams of more than 64 slots run the TRUNCATED reduction (their full one has >= 7 levels): the level count is where
    // the multipliers fall below the unit roundoff -- 5 (6 for slowly decaying mass matrices) in fp64, 4 (5) in fp32 --
    // so only those are instantiated for LOGNW >= 1 (crbeam.hip:lean_eligible sends anything else to the general kernel)
    constexpr bool long_ok = sizeof(T) == 8 ? (LV == 5 || LV == 6) : (LV == 4 || LV == 5);
    if (lognw == 0) return by_em<LV, 0, GRAV, STAGE>(k, n, em, st);
    if constexpr (long_ok) {
        switch (lognw) {
            case 1: return by_em<LV, 1, GRAV, STAGE>(k, n, em, st);
            case 2: return by_em<LV, 2, GRAV, STAGE>(k, n, em, st);
            case 3: return by_em<LV, 3, GRAV, STAGE>(k, n, em, st);
            default: return hipErrorInvalidValue;
        }
    }
    return hipErrorInvalidValue;
}
template <bool GRAV, bool STAGE>
hipError_t by_lv(const KParams<T>& k, int n, int levels, int lognw, int em, hipStream_t st) {
    switch (levels) {
        case 3: return by_nw<3, GRAV, STAGE>(k, n, lognw, em, st);
        case 4: return by_nw<4, GRAV, STAGE>(k, n, lognw, em, st);
        case 5: return by_nw<5, GRAV, STAGE>(k, n, lognw, em, st);
        case 6: return by_nw<6, GRAV, STAGE>(k, n, lognw, em, st);
        default: return hipErrorInvalidValue;
    }
}
template <int LV, int LNW, int EM>
hipError_t one_rk45(const KParams<T>& k, const Rk45Params& q, int n_beams, hipStream_t st) {
    constexpr int NT = 64 << LNW;
    // waves per SIMD the register allocation aims at: two, with 60 .. 160 spilled VGPRs -- measured against a spill-free
    // build at one wave per SIMD on single-wave beams: 4096 x 64 integrates in 1.26 ms against 1.76 ms (1024 x 64: 0.60 against 0.54)
    constexpr int MINW = 2;
    const size_t smem = rk45_lds_bytes<T>(NT, true);
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(crb_rk45_kernel<T, LV, 256, MINW, LNW, EM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, int(smem));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((crb_rk45_kernel<T, LV, 256, MINW, LNW, EM>), dim3(n_beams), dim3(NT), smem, st, k, q);
    return hipGetLastError();
}
template <int LV, int LNW>
hipError_t rk45_by_em(const KParams<T>& k, const Rk45Params& q, int n, int em, hipStream_t st) {
    switch (em) {
        case EM_LINEAR: return one_rk45<LV, LNW, EM_LINEAR>(k, q, n, st);
        case EM_NONLINEAR: return one_rk45<LV, LNW, EM_NONLINEAR>(k, q, n, st);
        default: return one_rk45<LV, LNW, EM_MIXED>(k, q, n, st);
    }
}
template <int LV>
hipError_t rk45_by_nw(const KParams<T>& k, const Rk45Params& q, int n, int lognw, int em, hipStream_t st) {
    constexpr bool long_ok = sizeof(T) == 8 ? (LV == 5 || LV == 6) : (LV == 4 || LV == 5);   // (as by_nw above)
    if (lognw == 0) return rk45_by_em<LV, 0>(k, q, n, em, st);
    if constexpr (long_ok) {
        if (lognw == 1) return rk45_by_em<LV, 1>(k, q, n, em, st);
        if (lognw == 2) return rk45_by_em<LV, 2>(k, q, n, em, st);
    }
    return hipErrorInvalidValue;
}
}  // namespace

// The instantiations are spread over translation units that build in parallel (Makefile: -DCRB_LEAN_PART=1|2|3 per
// dtype; undefined = everything in one unit, the `make fast` tuning build): 1 = stepper without gravity (+ the
// dispatcher), 2 = stepper with nearest-neighbour gravity, 3 = one-stage kernel and RK45 with the lean RHS.
#ifndef CRB_LEAN_PART
#define CRB_LEAN_PART 0
#endif
hipError_t launch_lean_grav(const KParams<T>& k, int n_beams, int levels, int lognw, int elem_mode, hipStream_t st);

#if CRB_LEAN_PART == 0 || CRB_LEAN_PART == 3
hipError_t launch_rk45_lean(const KParams<T>& k, const Rk45Params& q, int n_beams, int levels, int lognw, int elem_mode, hipStream_t st) {
#ifdef CRB_FAST_BUILD
    return hipErrorInvalidValue;
#else
    switch (levels) {
        case 3: return rk45_by_nw<3>(k, q, n_beams, lognw, elem_mode, st);
        case 4: return rk45_by_nw<4>(k, q, n_beams, lognw, elem_mode, st);
        case 5: return rk45_by_nw<5>(k, q, n_beams, lognw, elem_mode, st);
        case 6: return rk45_by_nw<6>(k, q, n_beams, lognw, elem_mode, st);
        default: return hipErrorInvalidValue;
    }
#endif
}

#endif

#if CRB_LEAN_PART == 0 || CRB_LEAN_PART == 2
hipError_t launch_lean_grav(const KParams<T>& k, int n_beams, int levels, int lognw, int elem_mode, hipStream_t st) {
#ifdef CRB_FAST_BUILD
    return hipErrorInvalidValue;
#else
    return by_lv<true, false>(k, n_beams, levels, lognw, elem_mode, st);
#endif
}
#endif

#if (CRB_LEAN_PART == 0 || CRB_LEAN_PART == 1) && !defined(CRB_FAST_BUILD)
// the packed one-wave stepper with the state feedback inside its stages (crb_step_lean_kernel<..., FB>): k.G >= 2 beams per wave,
// gain / reference / reduced map in k; mixed element kinds are evaluated per lane
namespace {
template <int LV, bool GRAV>
hipError_t one_fb(const KParams<T>& k, int n_beams, hipStream_t st) {
    auto kernel = crb_step_lean_kernel<T, LV, 0, GRAV, EM_MIXED, false, true, true>;
    const size_t smem = fb_lean_lds_bytes<T>(k.G, k.n_red);
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(smem));
        if (e != hipSuccess) return e;
    }
    const int groups = (n_beams + k.G - 1) / k.G;
    int grid = groups;
    const bool shared = k.slot_stride == 0 && k.lv_stride == 0 && k.fin_stride == 0;
    static int resident = -1;   // (per instantiation)
    if (shared) {   // a workgroup loads its tables and the gain once and walks over several groups of beams
        if (resident < 0) resident = resident_groups(kernel, 64, smem);
        int cap = resident;
        if (const char* env = std::getenv("CRB_LEAN_MAX_GROUPS")) cap = std::atoi(env);
        if (cap > 0 && groups > cap) {
            const int rounds = (groups + cap - 1) / cap;
            grid = (groups + rounds - 1) / rounds;
        }
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), smem, st, k);
    return hipGetLastError();
}
}  // namespace
hipError_t launch_lean_feedback(const KParams<T>& k, int n_beams, int levels, bool grav, hipStream_t st) {
    if (k.G < 2 || !k.fb_gain || !k.red_map) return hipErrorInvalidValue;
    switch (levels) {
        case 3: return grav ? one_fb<3, true>(k, n_beams, st) : one_fb<3, false>(k, n_beams, st);
        case 4: return grav ? one_fb<4, true>(k, n_beams, st) : one_fb<4, false>(k, n_beams, st);
        case 5: return grav ? one_fb<5, true>(k, n_beams, st) : one_fb<5, false>(k, n_beams, st);
        default: return hipErrorInvalidValue;
    }
}
#endif

#if CRB_LEAN_PART == 0 || CRB_LEAN_PART == 1
hipError_t launch_lean(const KParams<T>& k, int n_beams, int levels, int lognw, bool grav, int elem_mode, hipStream_t st) {
#ifdef CRB_FAST_BUILD  // kernel-tuning build (make fast): only the config-3 instance
    if (sizeof(T) == 8 && levels == 5 && lognw == 2 && !grav && elem_mode == EM_NONLINEAR)
        return one<5, 2, false, EM_NONLINEAR>(k, n_beams, st);
    if (sizeof(T) == 4 && levels == 4 && lognw == 2 && !grav && elem_mode == EM_NONLINEAR)   // (make fast32)
        return one<4, 2, false, EM_NONLINEAR>(k, n_beams, st);
    return hipErrorInvalidValue;
#else
    return grav ? launch_lean_grav(k, n_beams, levels, lognw, elem_mode, st)
                : by_lv<false, false>(k, n_beams, levels, lognw, elem_mode, st);
#endif
}
#endif

#if (CRB_LEAN_PART == 0 || CRB_LEAN_PART == 3) && !defined(CRB_FAST_BUILD)
namespace {
template <int LV, int LOGNW, bool GRAV, bool PACK = false>
hipError_t implicit_by_em(const KParams<T>& k, const StiffParams<T>& q, int groups, int em, hipStream_t st) {
    const dim3 grid(groups), block(64 << LOGNW);
    const size_t smem = implicit_lean_lds_bytes<T>(64 << LOGNW, LOGNW);
    switch (em) {
        case EM_LINEAR: hipLaunchKernelGGL((crb_implicit_lean_kernel<T, LV, LOGNW, GRAV, EM_LINEAR, PACK>), grid, block, smem, st, k, q); break;
        case EM_NONLINEAR: hipLaunchKernelGGL((crb_implicit_lean_kernel<T, LV, LOGNW, GRAV, EM_NONLINEAR, PACK>), grid, block, smem, st, k, q); break;
        default: hipLaunchKernelGGL((crb_implicit_lean_kernel<T, LV, LOGNW, GRAV, EM_MIXED, PACK>), grid, block, smem, st, k, q); break;
    }
    return hipGetLastError();
}
}  // namespace
// levels: 5 ... the full count ceil(log2 S) (6 / 7 / 8 for one / two / four waves per beam): where the reduction of
// A = M + h^2/4 K0 stops for the step size at hand (crbeam.hip: stiff_tables)
hipError_t launch_implicit_lean(const KParams<T>& k, const StiffParams<T>& q, int groups, int levels, int lognw, bool grav,
                                int elem_mode, hipStream_t st) {
    if constexpr (sizeof(T) == 8) {   // (fp64 plans only: crb_step_implicit refuses fp32)
        if (k.G > 1) {   // several beams per wave (fewer than 33 slots each: 3 ... 5 levels)
            if (lognw != 0) return hipErrorInvalidValue;
#define CRB_IMPL_PACK(LVV) \
            if (levels == LVV) \
                return grav ? implicit_by_em<LVV, 0, true, true>(k, q, groups, elem_mode, st) : implicit_by_em<LVV, 0, false, true>(k, q, groups, elem_mode, st);
            CRB_IMPL_PACK(3) CRB_IMPL_PACK(4) CRB_IMPL_PACK(5)
#undef CRB_IMPL_PACK
            return hipErrorInvalidValue;
        }
#define CRB_IMPL_CASE(LVV, NWW) \
        if (levels == LVV && lognw == NWW) \
            return grav ? implicit_by_em<LVV, NWW, true>(k, q, groups, elem_mode, st) : implicit_by_em<LVV, NWW, false>(k, q, groups, elem_mode, st);
        CRB_IMPL_CASE(5, 0) CRB_IMPL_CASE(6, 0)
        CRB_IMPL_CASE(5, 1) CRB_IMPL_CASE(6, 1) CRB_IMPL_CASE(7, 1)
        CRB_IMPL_CASE(5, 2) CRB_IMPL_CASE(6, 2) CRB_IMPL_CASE(7, 2) CRB_IMPL_CASE(8, 2)
#undef CRB_IMPL_CASE
    }
    return hipErrorInvalidValue;
}
#endif

#if CRB_LEAN_PART == 0 || CRB_LEAN_PART == 3
hipError_t launch_stage_lean(const KParams<T>& k, int n_groups, int levels, int lognw, bool grav, int elem_mode, hipStream_t st) {
#ifdef CRB_FAST_BUILD  // kernel-tuning build: the config-5 instance (128 linear elements + gravity, fp64)
    if (sizeof(T) == 8 && levels == 5 && lognw == 1 && grav && elem_mode == EM_LINEAR)
        return one_stage<5, 1, true, EM_LINEAR>(k, n_groups, st);
    return hipErrorInvalidValue;
#else
    return grav ? by_lv<true, true>(k, n_groups, levels, lognw, elem_mode, st)
                : by_lv<false, true>(k, n_groups, levels, lognw, elem_mode, st);
#endif
}
#endif
}  // namespace crb
