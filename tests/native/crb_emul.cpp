// crb_emul.cpp -- TEST HARNESS ONLY: compiles continuum-robot_amd/csrc/crb_math.h for the host
// with g++ so that the arithmetic the gfx950 kernels inline (element forces, gravity, drag, the
// cyclic-reduction solve step) can be checked against the oracle in the GPU-less build container.
// Nothing in the product loads this file.
#include "../../continuum-robot_amd/csrc/crb_math.h"

using namespace crb;

template <typename T>
static void elem_force_t(int kind, double L, double E, double I, double A, const double* ql, const double* qr,
                         int corrected, double* fl, double* fr) {
    ElemCoef<T> e;
    elem_coef_build<T>(e, kind, L, E, I, A);
    T l[3], r[3], a[3], b[3];
    for (int c = 0; c < 3; ++c) { l[c] = T(ql[c]); r[c] = T(qr[c]); }
    elem_force<T>(e, l, r, corrected != 0, a, b);
    for (int c = 0; c < 3; ++c) { fl[c] = double(a[c]); fr[c] = double(b[c]); }
}

extern "C" {
void emul_elem_force_f64(int kind, double L, double E, double I, double A, const double* ql, const double* qr,
                         int corrected, double* fl, double* fr) {
    elem_force_t<double>(kind, L, E, I, A, ql, qr, corrected, fl, fr);
}
void emul_elem_force_f32(int kind, double L, double E, double I, double A, const double* ql, const double* qr,
                         int corrected, double* fl, double* fr) {
    elem_force_t<float>(kind, L, E, I, A, ql, qr, corrected, fl, fr);
}
// the nonlinear element in its two forms: which = 0 literal (the reference's coefficients),
// 1 symmetric variables with rational coefficients (the form the kernels use)
void emul_elem_nonlinear_form(int which, double L, double E, double I, double A, const double* ql, const double* qr,
                              int corrected, double* fl, double* fr) {
    if (which == 0) {
        const double lit[5] = {L, E * A, E * I, 1.0 / (L * L), 0.1 / (L * L * L)};
        elem_force_nonlinear_literal<double>(lit, ql, qr, corrected != 0, fl, fr);
    } else {
        ElemCoef<double> e;
        elem_coef_build<double>(e, KIND_NONLINEAR, L, E, I, A);
        elem_force_nonlinear_sym<double>(e.c, ql, qr, corrected != 0, fl, fr);
    }
}
void emul_gravity_segment(double phi, double gx, double gy, double half_mass, double* out) {
    gravity_segment<double>(phi, gx, gy, half_mass, out);
}
double emul_drag(double coef, double v) { return drag_force<double>(coef, v); }
// r <- one cyclic-reduction level applied at one node
void emul_pcr_level(const double* cf, const double* rlo, const double* rhi, double* r) {
    pcr_apply_level<double>(cf, rlo, rhi, r);
}
void emul_pcr_final(const double* cf, const double* r, double* x) { pcr_apply_final<double>(cf, r, x); }
}
