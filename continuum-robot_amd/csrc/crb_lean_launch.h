// crb_lean_launch.h -- host entry of the lean stepper's translation units (crb_lean.hip is compiled once
// per dtype so that the kernel instantiations build in parallel with the rest of the library).
#pragma once
#include <hip/hip_runtime.h>

#include "crb_kernels.h"

namespace crb {
// launches crb_step_lean_kernel<T, levels, lognw, grav, elem_mode> on `n_beams` workgroups.
// levels in 3..6, lognw in 0..3 (callers check eligibility); hipErrorInvalidValue otherwise.
hipError_t launch_lean(const KParams<double>& k, int n_beams, int levels, int lognw, bool grav, int elem_mode, hipStream_t st);
hipError_t launch_lean(const KParams<float>& k, int n_beams, int levels, int lognw, bool grav, int elem_mode, hipStream_t st);
// launches the packed one-wave stepper with the LQR feedback inside its stages (crb_step_lean_kernel<..., FB>): k.G >= 2 beams
// per wave, levels in 3..5, gain / reference / reduced map in k; hipErrorInvalidValue otherwise
hipError_t launch_lean_feedback(const KParams<double>& k, int n_beams, int levels, bool grav, hipStream_t st);
hipError_t launch_lean_feedback(const KParams<float>& k, int n_beams, int levels, bool grav, hipStream_t st);
// launches crb_stage_lean_kernel<T, levels, lognw, grav, elem_mode> on `n_groups` workgroups (each walks
// over beams blockIdx.x, blockIdx.x + n_groups, ...)
hipError_t launch_stage_lean(const KParams<double>& k, int n_groups, int levels, int lognw, bool grav, int elem_mode, hipStream_t st);
hipError_t launch_stage_lean(const KParams<float>& k, int n_groups, int levels, int lognw, bool grav, int elem_mode, hipStream_t st);
// launches crb_rk45_kernel with the lean RHS (plans without gravity, one beam per workgroup of 2^lognw <= 4
// waves, levels 3..6); hipErrorInvalidValue otherwise
hipError_t launch_rk45_lean(const KParams<double>& k, const Rk45Params& q, int n_beams, int levels, int lognw, int elem_mode, hipStream_t st);
hipError_t launch_rk45_lean(const KParams<float>& k, const Rk45Params& q, int n_beams, int levels, int lognw, int elem_mode, hipStream_t st);
// launches crb_implicit_lean_kernel (crb_stiff.h) on `groups` workgroups, each walking over beams; (lognw, levels_full) in
// {(0, 6), (1, 7), (2, 8)}: 33..64 / 65..128 / 129..256 slots per beam; hipErrorInvalidValue otherwise
hipError_t launch_implicit_lean(const KParams<double>& k, const StiffParams<double>& q, int groups, int levels_full, int lognw,
                                bool grav, int elem_mode, hipStream_t st);
hipError_t launch_implicit_lean(const KParams<float>& k, const StiffParams<float>& q, int groups, int levels_full, int lognw,
                                bool grav, int elem_mode, hipStream_t st);
}  // namespace crb
