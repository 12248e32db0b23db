/*
 * crbeam.h -- C ABI of libcrbeam.so, the MI355X (gfx950) beam-dynamics stepper.
 *
 * The reference (cram9030/continuum-robot, pure Python) has no FFI/plugin interface: its
 * boundary for this path is the Python class API of
 * src/continuum_robot/models/dynamic_beam_model.py (DynamicEulerBernoulliBeam) and the
 * scipy.solve_ivp call sites that step its RHS.  This header is the native boundary a
 * maintainer binds with ctypes; every entry point names the reference
 * interface it stands in for.  Plain pointers and sizes only; no torch / C++ types.
 *
 * Conventions
 *   - every function returns 0 (CRB_OK) or a negative CRB_E* code; crb_last_error() returns
 *     a thread-local message for the last failure on the calling thread.
 *   - "device pointer" = HIP device memory of the plan's device and dtype; the caller owns
 *     all state/force tensors, the library owns only what hangs off the opaque plan.
 *   - launches are asynchronous on the passed stream (a hipStream_t passed as void*; NULL =
 *     the default stream).  A plan is not thread-safe; distinct plans are independent.
 *   - device state layout  x[B][2][n_node][4] : plane 0 = positions, plane 1 = velocities,
 *     record = {u, w, phi, 0}.  This is the reference's interleaved full DOF order
 *     (euler_bernoulli_beam.py:111-133) padded to 32-byte (fp64) / 16-byte (fp32) records;
 *     constrained DOFs keep their slot and hold 0.  Generalised-force tensors use
 *     f[B][n_node][4] the same way.  crb_pack_* / crb_unpack_* convert to and from the
 *     reference's REDUCED ordering (euler_bernoulli_beam.py:258-259, state = [q_red ; v_red],
 *     dynamic_beam_model.py:135-145).
 */
#ifndef CRBEAM_H
#define CRBEAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRB_VERSION 106

enum { CRB_OK = 0, CRB_EINVAL = -1, CRB_EHIP = -2, CRB_ENODEV = -3, CRB_EUNSUPPORTED = -4 };
enum { CRB_F64 = 0, CRB_F32 = 1 };
enum { CRB_BC_NONE = 0, CRB_BC_FIXED = 1, CRB_BC_PINNED = 2 }; /* abstractions.py:16-20 */
enum {
    CRB_FORCE_DRAG = 1u,      /* FluidDragForce auto-registration, dynamic_beam_model.py:223-231 */
    CRB_FORCE_GRAVITY = 2u,   /* GravityForce auto-registration,   dynamic_beam_model.py:234-241 */
    CRB_CORRECTED_AXIAL = 4u  /* opt-in f1 = -f2 instead of the shipped f1 (segments.py:178-208) */
};
enum { CRB_INPUT_NONE = 0, CRB_INPUT_IMPULSE = 1 };

typedef struct crb_plan crb_plan;

/* One beam topology shared by all beams of the ensemble: the CSV schema of
 * dynamic_beam_model.py:78-90 plus ForceParams (force_params.py:16-22).  Host pointers. */
typedef struct crb_beam_desc {
    int32_t n_elem;
    const double* length;          /* [n_elem] */
    const double* elastic_modulus; /* [n_elem] */
    const double* moment_inertia;  /* [n_elem] */
    const double* density;         /* [n_elem] */
    const double* cross_area;      /* [n_elem] */
    const uint8_t* nonlinear;      /* [n_elem] CSV 'type': 0 linear, 1 nonlinear */
    const uint8_t* node_bc;        /* [n_elem+1] CRB_BC_* per NODE (CSV row i -> node i, last node free) */
    const double* wetted_area;     /* [n_elem] or NULL when CRB_FORCE_DRAG is off */
    const double* drag_coef;       /* [n_elem] or NULL */
    double fluid_density;          /* ForceParams.fluid_density */
    double gravity[3];             /* ForceParams.gravity_vector (gz unused, gravity_forces.py:18) */
    uint32_t flags;                /* CRB_FORCE_* | CRB_CORRECTED_AXIAL */
} crb_beam_desc;

typedef struct crb_layout {
    int32_t dtype;        /* CRB_F64 / CRB_F32 */
    int32_t n_beams;
    int32_t n_elem;
    int32_t n_node;       /* n_elem + 1 */
    int32_t n_free;       /* size n of the reference's reduced position vector */
    int32_t node_offset;  /* 1 when node 0 is FIXED and therefore has no thread slot, else 0 */
    int32_t n_slots;      /* nodes carried by threads = n_node - node_offset */
    int32_t beams_per_group; /* beams handled by one workgroup */
    int32_t threads;      /* workgroup size */
    int32_t pcr_levels;   /* cyclic-reduction levels the mass solve applies */
    int32_t pcr_levels_full; /* ceil(log2(n_slots)): levels of the untruncated reduction */
    int32_t mixed_topology; /* 1: the beams' free-DOF sets differ (per-beam n_elem / boundary conditions): n_elem and
                             * n_free are the ensemble's maxima, reduced vectors are padded to n_free per beam
                             * (crb_plan_get_beam_info / crb_plan_get_beam_free_index give each beam's own) */
} crb_layout;

/* External generalised force u(t) added to the right-hand side, the `u` of
 * get_dynamic_system()(t, x, u) (dynamic_beam_model.py:343-362).
 *   held    : f_held[B][n_node][4] (device, may be NULL) is constant over the call.
 *   impulse : the examples' forcing (example_utilities.py:144-148, lqr_control.py:33-41):
 *             amp[b] on DOF (node, dof) while t < duration, else 0; evaluated at every RK4
 *             stage time. */
typedef struct crb_input_desc {
    int32_t kind;        /* CRB_INPUT_NONE / CRB_INPUT_IMPULSE */
    int32_t node;        /* impulse: node index (0..n_elem) */
    int32_t dof;         /* impulse: 0 = u, 1 = w, 2 = phi */
    int32_t reserved;
    double duration;     /* impulse: seconds */
    const void* amp;     /* impulse: device [B], plan dtype */
    const void* f_held;  /* device [B][n_node][4] or NULL */
    const int32_t* node_b; /* impulse: device [B] per-beam node (overrides `node`), or NULL -- beams of different length
                            * each forced at their own tip (`u[-2]` of example_utilities.py:147 per beam) */
} crb_input_desc;

/* Strided on-device recording of one DOF during crb_step_rk4_rec: the `t_eval` output of the
 * reference's solve_ivp calls (example_utilities.py:153-159) reduced to what the examples read
 * (tip displacement, lqr_control.py:168; example_utilities.py:173-205).  After every `every`-th step
 * out[b][k] = x[b][plane][node][dof], k = (step+1)/every - 1; out holds floor(n_steps/every)
 * values per beam (device, plan dtype). */
#define CRB_RECORD_ALL (-1) /* node = CRB_RECORD_ALL: whole-state snapshots (every DOF of sol.y on the t_eval grid,
                             * what examples/example_utilities.py:173-205 reads beam shapes from): out is device
                             * [floor(n_steps/every)][B][2][n_node][4], zero-initialised by the caller */
typedef struct crb_record_desc {
    int32_t plane;   /* 0 position, 1 velocity */
    int32_t node;
    int32_t dof;
    int32_t every;   /* >= 1 */
    void* out;       /* device [B][floor(n_steps/every)] */
} crb_record_desc;

int crb_version(void);
const char* crb_last_error(void);

/* Builds everything DynamicEulerBernoulliBeam.__init__ precomputes (dynamic_beam_model.py:25-74):
 * element coefficient packs (segments.py:32-78), the boundary-condition masks
 * (euler_bernoulli_beam.py:221-298), the mass matrix in node-block form and its cyclic-reduction
 * factorisation (replaces M_inv = inv(M), dynamic_beam_model.py:60), drag factors
 * (fluid_forces.py:50-101) and the gravity index table (gravity_forces.py:97-146).
 * device >= 0: HIP device ordinal, tables are uploaded.  device == -1: host-only plan for
 * inspection (crb_plan_get_* work, every launch returns CRB_ENODEV). */
int crb_plan_create(crb_plan** out, int device, int dtype, int n_beams, const crb_beam_desc* desc);
/* Heterogeneous ensembles (SURVEY f-3): descs[n_beams], one CSV-schema record + ForceParams per beam -- what the
 * reference's parallel examples run as independent simulations (examples/beam_comparison_fluid.py:49-83: six beams,
 * three without and three with fluid; beam_comparison_gravity.py:53-66).  Per beam: every element column and the
 * linear/nonlinear type; n_elem (the plan is laid out for the longest beam, a shorter one ends in padding nodes that
 * are fully constrained and carry no element); node_bc (dynamic_beam_model.py:205-218, euler_bernoulli_beam.py:221-298);
 * fluid_density, the gravity vector and the DRAG / GRAVITY flags (dynamic_beam_model.py:220-241).  Only
 * CRB_CORRECTED_AXIAL must be common.  Assembly and factorisation run batched on the device (one workgroup per beam).
 * When the beams' free-DOF sets differ (crb_layout.mixed_topology) the reduced vectors of crb_pack_* / crb_unpack_*
 * are [B][rows][n_free] with n_free the ensemble's maximum: beam b uses the first n_free_b entries of each row, the
 * rest is ignored on pack and zero on unpack; crb_feedback_force (one gain for the whole ensemble) is not available --
 * crb_feedback_force_grouped takes one gain per group of like beams.
 * The host inspectors (crb_plan_get_free_index/_mass/_stiffness/_pcr_tables/_slot_tables) describe beam 0. */
int crb_plan_create_ensemble(crb_plan** out, int device, int dtype, int n_beams, const crb_beam_desc* descs);
/* n_elem and n_free (size of the reference's reduced position vector) of one beam of the plan; either may be NULL */
int crb_plan_get_beam_info(const crb_plan* plan, int beam, int32_t* n_elem, int32_t* n_free);
/* reduced index -> full index 3*node+dof of one beam, [n_free of that beam] host */
int crb_plan_get_beam_free_index(const crb_plan* plan, int beam, int32_t* full_index);
void crb_plan_destroy(crb_plan* plan);
int crb_plan_get_layout(const crb_plan* plan, crb_layout* out);
/* reduced index -> full index 3*node+dof, ascending (euler_bernoulli_beam.py:258-259); [n_free] host */
int crb_plan_get_free_index(const crb_plan* plan, int32_t* full_index);
/* Host copies of the solve tables in fp64, for inspection/tests:
 *   levels [pcr_levels_full][n_slots][10], final_ [n_slots][6] (computed after `pcr_levels`
 *   levels), norms [pcr_levels_full].  Any pointer may be NULL. */
int crb_plan_get_pcr_tables(const crb_plan* plan, double* levels, double* final_, double* norms);
/* Host copies of the per-slot constants the kernels keep in registers, for inspection/tests:
 *   drag [n_slots], half_mass [n_slots], mask [n_slots][3],
 *   grav [n_slots][12] = {phiA, phiB, segA[3], segB[3], comp[3], 0} (see crb_generic.h GravTab),
 *   elem_kind [n_slots] (0 none, 1 linear, 2 nonlinear; the element LEFT of the slot's node).
 * Any pointer may be NULL. */
int crb_plan_get_slot_tables(const crb_plan* plan, double* drag, double* half_mass, double* mask, int16_t* grav,
                             int32_t* elem_kind);
/* Dense reduced mass matrix as assembled by the plan, [n_free][n_free] host fp64
 * (EulerBernoulliBeam.get_mass_matrix, euler_bernoulli_beam.py:358-362). */
int crb_plan_get_mass(const crb_plan* plan, double* M);
/* Dense reduced stiffness matrix of an all-linear beam, [n_free][n_free] host fp64
 * (EulerBernoulliBeam.get_stiffness_matrix, euler_bernoulli_beam.py:422-511); CRB_EINVAL with the
 * reference's message when a segment is nonlinear. */
int crb_plan_get_stiffness(const crb_plan* plan, double* K);

/* reduced [B][2n] <-> device layout [B][2][n_node][4]  (stiffness_with_boundary's
 * scatter/gather, euler_bernoulli_beam.py:280-289, done once at the API edge) */
int crb_pack_state(const crb_plan* plan, const void* x_red, void* x, void* stream);
int crb_unpack_state(const crb_plan* plan, const void* x, void* x_red, void* stream);
/* reduced [B][n] <-> device layout [B][n_node][4] for force / position-like vectors */
int crb_pack_vec(const crb_plan* plan, const void* v_red, void* v, void* stream);
int crb_unpack_vec(const crb_plan* plan, const void* v, void* v_red, void* stream);

/* k(q): EulerBernoulliBeam.get_stiffness_function() (euler_bernoulli_beam.py:163-219, 270-289).
 * x: device state (only the position plane is read); k: device [B][n_node][4]. */
int crb_internal_force(const crb_plan* plan, const void* x, void* k, void* stream);

/* xdot = [v ; Minv(-k(q) + f_drag + f_gravity + u)]: get_dynamic_system()(t, x, u)
 * (dynamic_beam_model.py:256-272, 294-328, 343-362).  u: device [B][n_node][4] or NULL. */
int crb_rhs(const crb_plan* plan, const void* x, const void* u, void* xdot, void* stream);

/* The same two functions for HOST vectors in the reference's reduced ordering, synchronously: what the single-beam
 * closures handed to scipy.solve_ivp call (get_dynamic_system()(t, x, u), dynamic_beam_model.py:343-362;
 * get_stiffness_function(), euler_bernoulli_beam.py:364-368) -- 6e5 calls per simulated second under LSODA.  The
 * (un)packing is fused into the kernel's own loads and stores, the vectors travel through pinned, device-mapped
 * staging of the plan: ONE launch + one stream synchronisation per call (the device-pointer forms above need
 * pack + kernel + unpack + copies).  x_red / xdot_red: [n_beams][2 n_free], u_red: [n_beams][n_free] or NULL,
 * q_red / k_red: [n_beams][n_free].  fp64 plans with one free-DOF set; not thread-safe per plan. */
int crb_rhs_host(const crb_plan* plan, const double* x_red, const double* u_red, double* xdot_red);
int crb_internal_force_host(const crb_plan* plan, const double* q_red, double* k_red);

/* n_steps classical RK4 steps of size dt, in place, in ONE launch (replaces the
 * scipy.solve_ivp call sites example_utilities.py:153-159, lqr_control.py:117-125).  The clock
 * starts at t0 and accumulates by addition (t <- t + dt); stage times t, t+dt/2, t+dt.
 * *t_end (host, may be NULL) receives the final clock value, to be passed as the next t0. */
int crb_step_rk4(const crb_plan* plan, void* x, double t0, double dt, int n_steps, const crb_input_desc* input,
                 double* t_end, void* stream);

/* Adaptive integration t0 -> t_end of every beam with embedded Dormand-Prince 5(4) and per-beam step-size
 * control, one launch: the algorithm of scipy.integrate.solve_ivp(method="RK45"), which the reference's
 * tests hand their RHS to (tests/test_dynamic_beam.py:218-220, test_functional_composition.py:539-546).
 *   h    device [B] fp64: in  first step per beam (<= 0: scipy's select_initial_step), out next step
 *   stats device [B][4] int32: accepted steps, rejected steps, RHS evaluations, status (0 ok / 1 step
 *        too small or max_steps exceeded).  h and stats may be NULL.  One beam per workgroup (small beams are
 *        not packed here: every beam has its own step sequence); beams of up to 256 slots (LDS-resident stages). */
int crb_solve_rk45(const crb_plan* plan, void* x, double t0, double t_end, double rtol, double atol,
                   const crb_input_desc* input, void* h, void* stats, int max_steps, void* stream);

/* crb_solve_rk45 plus solve_ivp's `t_eval` for one DOF: values on the uniform grid
 * t_eval[k] = eval_t0 + k*eval_dt (k < n_eval, eval_t0 >= t0) by scipy's dense output (RkDenseOutput, the
 * 4th-order interpolant of RK45) -- what the examples read from `sol.y` (example_utilities.py:153-159,
 * 173-205).  rec->every is unused here; rec->out is device [B][n_eval], plan dtype; with rec->node =
 * CRB_RECORD_ALL it is [n_eval][B][2][n_node][4] (zero-initialised by the caller): every DOF of sol.y. */
int crb_solve_rk45_eval(const crb_plan* plan, void* x, double t0, double t_end, double rtol, double atol,
                        const crb_input_desc* input, void* h, void* stats, int max_steps,
                        const crb_record_desc* rec, double eval_t0, double eval_dt, int n_eval, void* stream);

/* Implicit fixed-step integration for the stiff end of the reference's call sites: the examples hand the beam RHS to
 * solve_ivp(method="LSODA") for 1 s (examples/example_utilities.py:153-159, examples/lqr_control.py:117-125) because
 * explicit steppers are stability-limited to dt <= ~7e-5 s.  n_steps steps of size h of the implicit midpoint rule
 * (for linear systems the trapezoidal rule / Newmark average acceleration: A-stable, second order, no numerical
 * damping) on M a = -k(q) + f_drag(v) + f_gravity(q) + u(t), each step solved by n_iter modified-Newton iterations
 * with the constant matrix A = M + (h^2/4) K0 (K0 = element tangent stiffness at q = 0; cyclic reduction with
 * multipliers factorised on the device whenever h changes); one launch, in place.
 *   n_iter 1 is exact for linear elements without drag / gravity; 2 reproduces the converged step otherwise (the
 *          state dependence of drag, gravity and the geometric stiffness is not stiff); every step starts from the
 *          previous step's iterate, the first step of a call from 0.
 *   input  sampled at the step MIDPOINT (impulse: on while t + h/2 < duration), held force as in crb_step_rk4.
 *   rec    as crb_step_rk4_rec (may be NULL).  Beams of up to 256 thread-carried nodes; fp64 plans only
 *          (cond(A) = 1e6 ... 1e9 at these step sizes).
 * Modes with |lambda| h >> 1 are not resolved (their amplitude is kept, their phase is not): displacements converge
 * at second order in h from h ~ 1e-3 s down, velocities only once h resolves the modes they contain. */
int crb_step_implicit(const crb_plan* plan, void* x, double t0, double h, int n_steps, int n_iter,
                      const crb_input_desc* input, const crb_record_desc* rec, double* t_end, void* stream);
/* The numerically DAMPED member of the same family: generalised-alpha (Chung & Hulbert) with spectral radius rho_inf in
 * [0, 1] at infinite frequency -- second order, unconditionally stable, modes with |lambda| h >> 1 lose the factor rho_inf
 * per step instead of ringing on with the wrong phase (rho_inf = 1: crb_step_implicit exactly; 0: they are gone after one
 * step).  What LSODA's BDF formulas do to the unresolved modes of the examples' runs (example_utilities.py:153-159) and the
 * midpoint rule does not.  Same iteration (matrix M + kappa K0, kappa = (1 - alpha_f) beta h^2 / (1 - alpha_m)); the
 * acceleration history starts from the RHS at t0 (one extra launch per call), the input is sampled at
 * t + (1 - alpha_f) h.  Runs the general kernel (one wave per SIMD) for every beam shape. */
int crb_step_implicit_damped(const crb_plan* plan, void* x, double t0, double h, int n_steps, int n_iter, double rho_inf,
                             const crb_input_desc* input, const crb_record_desc* rec, double* t_end, void* stream);

/* The examples' integration call with its TOLERANCES, one launch for the whole span and the whole ensemble:
 * solve_ivp(f, t_span, x0, method="LSODA", t_eval=np.arange(t0, t1, DT)) at scipy's default rtol 1e-3 / atol 1e-6
 * (examples/example_utilities.py:153-159) and the closed loop of examples/lqr_control.py:117-125 at rtol 1e-8 / atol 1e-10.
 * The step size is controlled INSIDE the kernel, per beam (one workgroup per beam; csrc/crb_ctrl.h):
 *   gain == NULL  the implicit midpoint rule of crb_step_implicit (order 2);
 *   gain != NULL  RK4 with u = K (ref - x) in every stage (order 4; device [n][2n] / [B][2n] like crb_step_rk4_feedback;
 *                 beams whose gain fits the LDS, up to ~30 elements; one free-DOF set, no held force).
 * by step doubling per piece -- a t_eval interval [t0 + k dt_eval, t0 + (k+1) dt_eval], cut at input->duration when the
 * impulse ends inside it (the impulse is then simply on or off within a piece): m = 2^r and 2m steps from the same state,
 * (fine - coarse) / (2^order - 1) measured in scipy's norm (RMS over the beam's reduced state -- its position half with
 * positions_only -- of err / (atol + rtol max(|fine|, |start|))); above 1 or not finite: twice the steps; the fine
 * solution is the accepted one; far below 1: half the rate for the next piece.  The cyclic-reduction tables of
 * A = M + (h^2/4) K0 for the whole ladder h = len / 2^r, r < max_rungs, are factorised on the device before the launch (and
 * kept with the plan).
 *   y_out  device [n_intervals][B][2][n_node][4] (the state at t0 + (k+1) dt_eval in slot k) or NULL
 *   stats  device int32 [B][4]: fine steps accepted, doublings, status (0 = reached the end, 2 = a piece asked for more than
 *          2^(max_rungs-1) steps: the beam stops at the start of that piece), rung of the last piece
 *   used   device int32 [B][n_intervals] fine steps accepted per interval, or NULL
 * x ends at t0 + n_intervals dt_eval.  fp64 plans, beams of up to 256 thread-carried nodes. */
typedef struct crb_control_desc {
    double rtol, atol;
    double first_rate;       /* steps per second of the first coarse solution; <= 0: 1e4 (implicit) / 2e5 (closed-loop RK4) */
    int32_t positions_only;  /* measure the position half of the state only */
    int32_t n_iter;          /* implicit scheme: modified-Newton iterations per step; <= 0: 1 (enough at the controller's step sizes) */
    int32_t max_rungs;       /* <= 0: 15 */
    int32_t per_wave;        /* 0: every beam its own step sequence (one workgroup per beam).  1 (implicit scheme, beams of 2 .. 32
                              * thread-carried nodes): G = 64 / n_slots beams share a wave AND its step sequence -- the worst of them
                              * decides -- so thousands of short beams fill the chip with a fifth of the waves */
    /* one DOF on the t_eval grid instead of (or next to) whole-state snapshots -- what the examples read (tip displacement,
     * lqr_control.py:168): series_out != NULL: device [B][n_intervals], plan dtype, series_out[b][k] = x[b][plane][node][dof] at
     * t0 + (k + 1) dt_eval (4096 beams x 1000 outputs: 33 MB instead of the 2.9 GB of y_out) */
    int32_t series_plane, series_node, series_dof, series_pad;
    void* series_out;
} crb_control_desc;
int crb_solve_controlled(const crb_plan* plan, void* x, double t0, double dt_eval, int n_intervals,
                         const crb_control_desc* control, const crb_input_desc* input, const void* gain, const void* ref,
                         void* y_out, void* stats, void* used, void* stream);

/* Per-beam status of the fixed-step steppers (crb_step_rk4[_rec], crb_step_implicit, crb_step_rk4_feedback, crb_rk4_stage at
 * stage 3).  status: device int32 [B], zeroed by the caller, or NULL to switch the reporting off; steps_done: the number of
 * steps the ensemble has taken so far.  A launch that leaves a non-finite value in beam b's state writes status[b] = the
 * ensemble's step count at the END of that launch (once: later launches leave a marked beam alone), so 0 = finite and k > 0 =
 * "went non-finite within the launch that ended at step k".  The reference has no such report: the shipped nonlinear element
 * (segments.py:178-208) is linearly unstable for long chains and solve_ivp simply returns NaN rows; this turns that into a
 * condition the planning layer can read per rollout.  The beams are independent: a diverged beam never affects another. */
int crb_plan_set_status(const crb_plan* plan, void* status, long long steps_done);

/* Feedback force of a whole ensemble, u = K (r - x) (FullStateLinear.compute_input,
 * control/full_state_linear.py:81; loop of examples/lqr_control.py:95-111), as one MFMA GEMM in the plan's dtype
 * (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32) with the gather from the state layout and the scatter into
 * the force layout fused in.
 *   xs   device [B][2][n_node][4]           gain  device [n][2n] row-major, reduced ordering
 *   ref  device [B][2n] reduced or NULL=0   u     device [B][n_node][4]: free-DOF entries are
 *                                                  overwritten, the rest must already be zero */
int crb_feedback_force(const crb_plan* plan, const void* xs, const void* gain, const void* ref, void* u, void* stream);

/* The same for heterogeneous ensembles (crb_plan_create_ensemble; SURVEY f-3): one gain per GROUP of beams, as the reference
 * designs one gain per model (examples/lqr_control.py:46-84, control/linear_quadratic_regulator.py:84-191).
 *   beam_group  host int32 [B]: the group of every beam, or -1 (no feedback: its u stays 0)
 *   gains       host array of n_groups device pointers, gains[g] = [n_g][2 n_g] row-major in the reduced ordering of group g's
 *               beams, which must share one free-DOF set (n_g = their n_free)
 *   ref         device [B][2 n_free] padded reduced states as crb_pack_state takes them, or NULL
 * One MFMA launch per group (rows gathered through the group's beam list); u must be zero where no group writes. */
int crb_feedback_force_grouped(const crb_plan* plan, const void* xs, int n_groups, const int32_t* beam_group,
                               const void* const* gains, const void* ref, void* u, void* stream);
/* crb_step_rk4_feedback with per-group gains: the stage-split loop (grouped force, then crb_rk4_stage, per stage). */
int crb_step_rk4_feedback_grouped(const crb_plan* plan, void* x, double t0, double dt, int n_steps, int n_groups,
                                  const int32_t* beam_group, const void* const* gains, const void* ref,
                                  const crb_input_desc* input, void* work, double* t_end, void* stream);

/* ONE stage of the stage-split RK4 stepper, for inputs that change from stage to stage -- state
 * feedback u = K(r - x) evaluated inside the RHS as examples/lqr_control.py:95-111 does
 * (FullStateLinear.compute_input, control/full_state_linear.py:81).  The caller computes this
 * stage's generalised force u_stage[B][n_node][4] from xs (e.g. one GEMM over the whole ensemble) and
 * calls:   k = f(t_stage, xs, u_stage + impulse);  acc = (stage ? acc : 0) + w_stage * k;
 *          stage < 3: xs_next = x + c_stage * k;      stage == 3: x += dt/6 * acc.
 * Stage 0 passes xs == x.  x, xs, acc, xs_next: device [B][2][n_node][4]; xs_next may not alias xs.
 * All state / force buffers of this header must be aligned to 32 bytes (a node record is read as one
 * vector); hipMalloc and torch allocations are. */
int crb_rk4_stage(const crb_plan* plan, void* x, const void* xs, void* acc, void* xs_next, const void* u_stage,
                  int stage, double t_stage, double dt, const crb_input_desc* input, void* stream);

/* The closed loop of examples/lqr_control.py:95-125 as ONE call: n_steps RK4 steps with u = K (r - x)
 * re-evaluated at every stage (crb_feedback_force, then crb_rk4_stage), all launches issued from here
 * (either dtype).  work: device scratch of crb_feedback_work_bytes(plan) bytes (three state-sized buffers and
 * one force-sized buffer and the device clock; contents need not be initialised).  Returns the accumulated clock
 * in *t_end.  CRB_USE_GRAPH=1 in the environment replays one captured step as a hipGraph on a stream of the
 * plan's own (ordered after / before the caller's stream by events).
 * Beams that live in one wave (fewer than 64 thread-carried nodes: the reference's own LQR example has 6 elements)
 * and whose gain fits LDS take ONE launch for the whole rollout instead: the packed lean stepper (several beams per wave,
 * 3 .. 5 reduction levels, gravity absent or of the plain cantilever's form; the general stepper otherwise) with the gain
 * resident in LDS and K (r - x) formed per stage -- on the matrix cores for gains of 21 .. 32 rows, by the node threads
 * otherwise (10 instead of 36 us per step at 6 and at 10 elements, 21 instead of 40 - 48 at 16 .. 20).
 * CRB_FUSED_FEEDBACK=0 / 1 in the environment forces the stage-split / the fused form.
 * Large ensembles of beams with 33 .. 128 thread-carried nodes (fp64, one table set) take ONE persistent launch for the
 * whole rollout (csrc/crb_loop.h): groups of workgroups own 64 beams each, keep their slice of the gain in registers,
 * and alternate between the fp64-MFMA product and the stage arithmetic, handing tiles to each other through L2.
 * CRB_LOOP=0 / 1 forces the stage-split / the persistent form.  A hand-off that does not complete within
 * CRB_LOOP_TIMEOUT_MS (default 2000) makes every workgroup leave; crb_feedback_status reports it. */
size_t crb_feedback_work_bytes(const crb_plan* plan);
/* Which form crb_step_rk4_feedback takes for this plan (without a held input): 0 stage-split, 1 fused (gain in LDS), 2 persistent. */
int crb_feedback_path(const crb_plan* plan);
/* *status = 0, or (group + 1) of the first hand-off of the last crb_step_rk4_feedback call on `work` that timed out (the
 * state is then unusable).  Synchronises `stream`. */
int crb_feedback_status(const crb_plan* plan, const void* work, int32_t* status, void* stream);
int crb_step_rk4_feedback(const crb_plan* plan, void* x, double t0, double dt, int n_steps, const void* gain,
                          const void* ref, const crb_input_desc* input, void* work, double* t_end, void* stream);

/* crb_step_rk4 plus strided recording of one DOF (rec may be NULL). */
int crb_step_rk4_rec(const crb_plan* plan, void* x, double t0, double dt, int n_steps, const crb_input_desc* input,
                     const crb_record_desc* rec, double* t_end, void* stream);

/* out[b] = x[b][plane][node][dof]  (e.g. tip displacement = plane 0, node n_elem, dof 1;
 * lqr_control.py:168) */
int crb_gather_dof(const crb_plan* plan, const void* x, int plane, int node, int dof, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CRBEAM_H */
