// crb_loop.hip -- instantiations and launch of crb_loop_kernel (the persistent closed-loop stepper, crb_loop.h).
#include <cstdlib>

#include "crb_loop_launch.h"

namespace crb {
namespace {
typedef double T;

template <int LV, int LOGNW, int NB, bool GRAV, int EM, bool HAS_REF>
hipError_t one_loop(LoopParams<T> P, const T* gain, hipStream_t st) {
    auto kernel = crb_loop_kernel<T, LV, LOGNW, NB, GRAV, EM, HAS_REF>;
    const size_t smem = loop_lds_bytes<T, LV, LOGNW, NB>();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(smem));
    if (e != hipSuccess) return e;
    // the grid is what the device keeps resident, in whole groups: a workgroup that is not running cannot arrive
    static int resident = -1;   // (per instantiation; every device of a node is the same part)
    if (resident < 0) {
        int dev = 0, cus = 0, per_cu = 0;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, smem)) != hipSuccess) return e;
        resident = cus * per_cu;
    }
    int groups = resident / NB;
    if (const char* env = std::getenv("CRB_LOOP_MAX_GROUPS")) groups = std::atoi(env);   // (tests: several row blocks per group)
    if (groups > LOOP_MAX_GROUPS) groups = LOOP_MAX_GROUPS;
    if (groups > P.n_rb) groups = P.n_rb;
    if (groups < 1) return hipErrorInvalidConfiguration;
    P.n_groups = groups;
    constexpr int frag_vals = NB * 4 * 3 * 6 * NB * 64;
    hipLaunchKernelGGL((crb_loop_gain_kernel<T, NB>), dim3((frag_vals + 255) / 256), dim3(256), 0, st, gain, P.red_map, P.n_red, P.k.S, P.k.off,
                       const_cast<T*>(P.kfrag));
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(groups * NB), dim3(256), smem, st, P);
    return hipGetLastError();
}
template <int LV, int LOGNW, int NB, bool GRAV, int EM>
hipError_t by_ref(const LoopParams<T>& P, const T* gain, hipStream_t st) {
    return P.ref ? one_loop<LV, LOGNW, NB, GRAV, EM, true>(P, gain, st) : one_loop<LV, LOGNW, NB, GRAV, EM, false>(P, gain, st);
}
template <int LV, int LOGNW, int NB, bool GRAV>
hipError_t by_em(const LoopParams<T>& P, const T* gain, int em, hipStream_t st) {
    // (a gain is designed on the linear model: all-linear topologies get the straight-line force, anything else the per-lane branch)
    return em == EM_LINEAR ? by_ref<LV, LOGNW, NB, GRAV, EM_LINEAR>(P, gain, st) : by_ref<LV, LOGNW, NB, GRAV, EM_MIXED>(P, gain, st);
}
template <int LV, int LOGNW, int NB>
hipError_t by_grav(const LoopParams<T>& P, const T* gain, bool grav, int em, hipStream_t st) {
    return grav ? by_em<LV, LOGNW, NB, true>(P, gain, em, st) : by_em<LV, LOGNW, NB, false>(P, gain, em, st);
}
}  // namespace

#ifndef CRB_LOOP_PART   // 0 = everything in one unit; 1 / 2 = beams of 65 .. 128 / 33 .. 64 slots (Makefile: built in parallel)
#define CRB_LOOP_PART 0
#endif
#if CRB_LOOP_PART == 0 || CRB_LOOP_PART == 1
hipError_t launch_loop_long(const LoopParams<double>& P, const double* gain, int levels, bool grav, int elem_mode, hipStream_t st) {
#ifdef CRB_FAST_BUILD   // kernel-tuning build: the config-5 instance (128 linear elements + gravity, regulation to 0)
    if (levels == 5 && grav && elem_mode == EM_LINEAR && !P.ref) return one_loop<5, 1, 8, true, EM_LINEAR, false>(P, gain, st);
    return hipErrorInvalidValue;
#else
    if (levels == 5) return by_grav<5, 1, 8>(P, gain, grav, elem_mode, st);
    if (levels == 6) return by_grav<6, 1, 8>(P, gain, grav, elem_mode, st);
    return hipErrorInvalidValue;
#endif
}
#endif
#if (CRB_LOOP_PART == 0 || CRB_LOOP_PART == 2) && defined(CRB_FAST_BUILD)
hipError_t launch_loop_short(const LoopParams<double>&, const double*, int, bool, int, hipStream_t) { return hipErrorInvalidValue; }
#endif
#if (CRB_LOOP_PART == 0 || CRB_LOOP_PART == 2) && !defined(CRB_FAST_BUILD)
hipError_t launch_loop_short(const LoopParams<double>& P, const double* gain, int levels, bool grav, int elem_mode, hipStream_t st) {
    if (levels == 5) return by_grav<5, 0, 4>(P, gain, grav, elem_mode, st);
    if (levels == 6) return by_grav<6, 0, 4>(P, gain, grav, elem_mode, st);
    return hipErrorInvalidValue;
}
#endif
}  // namespace crb
