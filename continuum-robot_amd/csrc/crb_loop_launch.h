// crb_loop_launch.h -- host entry of the persistent closed-loop stepper's translation units (crb_loop.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "crb_loop.h"

namespace crb {
// launches crb_loop_kernel<double, levels, ...> for beams of 65 .. 128 thread-carried nodes (two waves per beam, groups
// of 8 workgroups) / 33 .. 64 (one wave per beam, groups of 4); levels in {5, 6}; P.n_groups is chosen here (what the
// device keeps resident).  hipErrorInvalidValue when no instance covers the plan.
// `gain` [n][2n] is re-laid into P.kfrag by a small kernel of the same launch sequence (crb_loop_gain_kernel).
hipError_t launch_loop_long(const LoopParams<double>& P, const double* gain, int levels, bool grav, int elem_mode, hipStream_t st);
hipError_t launch_loop_short(const LoopParams<double>& P, const double* gain, int levels, bool grav, int elem_mode, hipStream_t st);

// Layout of the work buffer (bytes from its start): [sync words][gain fragments][E][U][own], every part sized for
// `groups` groups of NB workgroups (NB = 8 for beams of more than 64 thread-carried nodes, else 4)
struct LoopWork {
    size_t kfrag, ebuf, ubuf, ownbuf, total;
};
inline LoopWork loop_work_layout(int nb, int groups) {
    LoopWork w;
    const size_t e = size_t(8 * nb) * 64 * 12 * sizeof(double), u = size_t(64) * 48 * nb * sizeof(double), own = size_t(64) * 16 * nb * 12 * sizeof(double);
    w.kfrag = size_t(LOOP_SYNC_WORDS) * sizeof(unsigned);
    w.ebuf = w.kfrag + size_t(nb) * 4 * 3 * 6 * nb * 64 * sizeof(double);
    w.ubuf = w.ebuf + e * groups;
    w.ownbuf = w.ubuf + u * groups;
    w.total = w.ownbuf + own * groups;
    return w;
}
}  // namespace crb
