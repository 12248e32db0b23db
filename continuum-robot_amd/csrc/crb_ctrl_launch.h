// crb_ctrl_launch.h -- host entry of the controlled steppers' translation unit (crb_ctrl.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "crb_ctrl.h"

namespace crb {
// launches crb_controlled_kernel<double, levels, feedback> on a grid of k.B workgroups of `threads` threads (one beam per
// workgroup).  levels: the reduction levels of the tables the kernel solves with -- ALL levels of the beam for the
// implicit scheme (A = M + h^2/4 K0 is not as diagonally dominant as M, and one instance must serve every rung), the
// plan's truncated count for the closed-loop RK4 (the mass matrix's tables).  hipErrorInvalidValue when no instance
// covers the plan (more than 8 / 6 levels).
// lean_lognw >= 0 (implicit scheme): the lean iteration with 2^lean_lognw waves per beam (threads = 64 << lean_lognw, levels =
// ceil(log2 S) >= 1), gravity absent or canonical (`grav`); -1: the general RHS.
// pack (lean form with lean_lognw == 0, k.G >= 2 beams of fewer than 33 slots per wave): a grid of ceil(k.B / k.G) one-wave
// workgroups, every wave with ONE step sequence for its beams.
hipError_t launch_controlled(const KParams<double>& k, const CtrlParams<double>& q, int levels, bool feedback, int lean_lognw, bool grav,
                             bool pack, int threads, size_t lds_bytes, hipStream_t st);
}  // namespace crb
