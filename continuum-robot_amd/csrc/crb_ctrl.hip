// crb_ctrl.hip -- the step-size-controlled steppers (crb_ctrl.h), one translation unit of their own.
#include "crb_ctrl_launch.h"

namespace crb {
namespace {
template <int LV, bool FB, int LNW = -1, bool GRAV = false, bool PACK = false>
hipError_t one_controlled(const KParams<double>& k, const CtrlParams<double>& q, int threads, size_t lds, hipStream_t st) {
    auto kern = crb_controlled_kernel<double, LV, FB, LNW, GRAV, PACK>;
    if (lds > size_t(48) * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(PACK ? (k.B + k.G - 1) / k.G : k.B), dim3(threads), lds, st, k, q);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_controlled(const KParams<double>& k, const CtrlParams<double>& q, int levels, bool feedback, int lean_lognw, bool grav,
                             bool pack, int threads, size_t lds, hipStream_t st) {
    if (threads < 64 || threads > 256 || (threads & 63)) return hipErrorInvalidValue;
#ifdef CRB_FAST_BUILD
    return hipErrorInvalidValue;
#else
    if (pack) {   // several short beams per wave (k.G of them, fewer than 33 slots each), one step sequence per wave
        if (feedback || lean_lognw != 0 || threads != 64 || k.G < 2) return hipErrorInvalidValue;
#define CRB_CTRL_PACK(LVV) \
        if (levels == LVV) \
            return grav ? one_controlled<LVV, false, 0, true, true>(k, q, threads, lds, st) : one_controlled<LVV, false, 0, false, true>(k, q, threads, lds, st);
        CRB_CTRL_PACK(1) CRB_CTRL_PACK(2) CRB_CTRL_PACK(3) CRB_CTRL_PACK(4) CRB_CTRL_PACK(5)
#undef CRB_CTRL_PACK
        return hipErrorInvalidValue;
    }
    if (lean_lognw >= 0 && feedback) {   // closed-loop RK4 with the lean right-hand side of one wave (the mass matrix's `levels` levels)
        if (lean_lognw != 0 || threads != 64) return hipErrorInvalidValue;
#define CRB_CTRL_FBL(LVV) \
        if (levels == LVV) \
            return grav ? one_controlled<LVV, true, 0, true>(k, q, threads, lds, st) : one_controlled<LVV, true, 0, false>(k, q, threads, lds, st);
        CRB_CTRL_FBL(1) CRB_CTRL_FBL(2) CRB_CTRL_FBL(3) CRB_CTRL_FBL(4) CRB_CTRL_FBL(5) CRB_CTRL_FBL(6)
#undef CRB_CTRL_FBL
        return hipErrorInvalidValue;
    }
    if (lean_lognw >= 0) {   // the lean iteration: all ceil(log2 S) levels of a beam of 2 .. 64 / 65 .. 128 / 129 .. 256 slots
        if (threads != (64 << lean_lognw)) return hipErrorInvalidValue;
#define CRB_CTRL_LEAN(LVV, NWW) \
        if (levels == LVV && lean_lognw == NWW) \
            return grav ? one_controlled<LVV, false, NWW, true>(k, q, threads, lds, st) : one_controlled<LVV, false, NWW, false>(k, q, threads, lds, st);
        CRB_CTRL_LEAN(1, 0) CRB_CTRL_LEAN(2, 0) CRB_CTRL_LEAN(3, 0) CRB_CTRL_LEAN(4, 0) CRB_CTRL_LEAN(5, 0) CRB_CTRL_LEAN(6, 0)
        CRB_CTRL_LEAN(7, 1) CRB_CTRL_LEAN(8, 2)
#undef CRB_CTRL_LEAN
        return hipErrorInvalidValue;
    }
    if (feedback) {
        switch (levels) {
            case 0: return one_controlled<0, true>(k, q, threads, lds, st);
            case 1: return one_controlled<1, true>(k, q, threads, lds, st);
            case 2: return one_controlled<2, true>(k, q, threads, lds, st);
            case 3: return one_controlled<3, true>(k, q, threads, lds, st);
            case 4: return one_controlled<4, true>(k, q, threads, lds, st);
            case 5: return one_controlled<5, true>(k, q, threads, lds, st);
            case 6: return one_controlled<6, true>(k, q, threads, lds, st);
            default: return hipErrorInvalidValue;
        }
    }
    switch (levels) {
        case 0: return one_controlled<0, false>(k, q, threads, lds, st);
        case 1: return one_controlled<1, false>(k, q, threads, lds, st);
        case 2: return one_controlled<2, false>(k, q, threads, lds, st);
        case 3: return one_controlled<3, false>(k, q, threads, lds, st);
        case 4: return one_controlled<4, false>(k, q, threads, lds, st);
        case 5: return one_controlled<5, false>(k, q, threads, lds, st);
        case 6: return one_controlled<6, false>(k, q, threads, lds, st);
        case 7: return one_controlled<7, false>(k, q, threads, lds, st);
        case 8: return one_controlled<8, false>(k, q, threads, lds, st);
        default: return hipErrorInvalidValue;
    }
#endif
}
}  // namespace crb
