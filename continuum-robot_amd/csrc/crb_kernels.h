// crb_kernels.h -- all gfx950 kernels of the beam stepper (umbrella header).
//   crb_generic.h   parameter blocks, slot topology, general kernels (crb_beam_kernel)
//   crb_rk45.h      adaptive RK45 (crb_rk45_kernel; uses the lean RHS where a plan allows it)
//   crb_lean.h      lean stepper / lean RHS / lean stage kernel (the headline path)
//   crb_assemble.h  plan-time assembly + mass-matrix factorisation
//   crb_feedback.h  feedback GEMM (fp64 MFMA) and layout conversion
//   crb_stiff.h     implicit (trapezoidal, modified Newton) stepper for stiff integrations
#pragma once
#include "crb_generic.h"
#include "crb_rk45.h"
#include "crb_lean.h"
#include "crb_assemble.h"
#include "crb_feedback.h"
#include "crb_stiff.h"
