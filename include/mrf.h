/* mrf.h -- C ABI of the MI355X-native multi-robot fabrics hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI of its own: the
 * path is a Python object API that ends in CasADi function evaluations
 *   planner.compute_action(**kwargs)                  examples/example_pandas_Jointspace.py:441,444
 *   ForwardFabricsPlanner.get_velocity_rollouts(...)  multi_robot_fabrics/fabrics_planner/forward_planner_Jointspace.py:298-336
 *   ForwardFabricsPlanner.rollouts_numerical(...)     forward_planner_Jointspace.py:338-423
 *   FabricsRollouts.get_velocity_rollouts / rollouts_numerical
 *                                                     multi_robot_fabrics/fabrics_planner/forward_planner_Cartesian.py:538-563
 *   fk / jac / jac_dot helper functions               multi_robot_fabrics/utils/utils.py:16-54,87-119
 * Every entry point below states which of those it replaces.  Plain pointers and sizes only;
 * no torch types, no exceptions, no global mutable state.
 *
 * Conventions
 *   - All device arrays are caller-owned, never retained past the call, and laid out
 *     component-major over the batch ("SoA"):  a[c][row]  ==  a[c * rows + row].
 *   - A "row" is one (scenario, robot) pair:  row = scenario * n_robots + robot.
 *     The robot index (row % n_robots) selects the mount transform.
 *   - The scalar type of every array is cfg.scalar (f64 or f32) for the whole handle.
 *   - Calls are asynchronous on the given hipStream_t (passed as void*; NULL = default stream).
 *   - Return 0 on success, a negative mrf_status otherwise; mrf_last_error() gives text.
 *   - A handle is NOT thread-safe; distinct handles are independent.
 *
 * Numerical contract
 *   - float64 results agree with the float64 CPU restatement (oracle/) to 1e-9 relative on every row whose barrier
 *     coordinates are positive: x = d/(r_o + r_b) - 1 of every collision leaf, the plane leaves' clearance, every joint's
 *     distance to its limits (tests/test_gpu_parity.py; error against the smallest x: tests/test_gpu_error_curve.py).
 *   - The solve kernels are compiled with -ffast-math (finite-math-only, re-association; the control-step unit --
 *     deadlock logic, state machine -- is not).  The reference's barrier strings are even powers of 1/x with no clamp
 *     (EXJ:88-89), so a row with a coordinate x <= 0 -- overlapping spheres, a point below the plane, a joint past its limit
 *     -- has no meaningful result in the reference either; here such a row returns UNSPECIFIED values, finite or not (a NaN
 *     is not guaranteed to propagate).  mrf_deadlock_step counts non-finite rollout averages (MRF_DL_NONFINITE) and treats
 *     them as "no deadlock", as the reference's comparison would.
 *   - Such a row cannot reach the rows of OTHER scenarios: the kernels share per-wave exchange tiles, but a scenario's
 *     rows read only their own scenario's entries.  Every other row's output equals a run without the bad row -- bit for
 *     bit for a single evaluation (compute_action), to round-off (<= 1e-12) for rollouts, whose system_step picks the
 *     incremental or the full sincos by a wave-wide vote that a diverging row can change
 *     (tests/test_gpu_fastmath_contract.py).  Within its scenario every robot is affected from the next rollout step on,
 *     as in the reference's coupled recurrence (FPJ:211-233).
 */
#ifndef MRF_H_
#define MRF_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRF_ABI_VERSION 6
#define MRF_MAX_ROBOTS 16
#define MRF_MAX_SPHERES 32 /* exchanged spheres per robot */
#define MRF_DOF_MAX 7
#define MRF_N_EGO 6 /* ego collision points of a Panda: origins of panda_link3..8 */

/* per-row parameter vector (29 scalars), the runtime kwargs of compute_action
 * (example_pandas_Jointspace.py:421-439) minus the obstacle lists */
#define MRF_P_X_GOAL_0 0      /* [3]  (planar3: [2])                         */
#define MRF_P_WEIGHT_GOAL_0 3 /*                                              */
#define MRF_P_ANGLE_GOAL_1 4  /* [9]  row-major 3x3                           */
#define MRF_P_X_GOAL_1 13     /* [3]                                          */
#define MRF_P_WEIGHT_GOAL_1 16
#define MRF_P_X_GOAL_2 17
#define MRF_P_WEIGHT_GOAL_2 18
#define MRF_P_CONSTRAINT_0 19 /* [4]  plane a.x + d                           */
#define MRF_P_RADIUS_BODY 23  /* [6]  links 3..8 (planar3: [1] base_link)     */
#define MRF_NPARAM 29

typedef enum { MRF_OK = 0, MRF_E_ARG = -1, MRF_E_CONFIG = -2, MRF_E_DEVICE = -3, MRF_E_LAUNCH = -4 } mrf_status;
typedef enum { MRF_MODEL_PANDA7 = 0, MRF_MODEL_PLANAR3 = 1 } mrf_model;
typedef enum { MRF_MODE_ACC = 0, MRF_MODE_VEL = 1 } mrf_mode;
typedef enum { MRF_F64 = 0, MRF_F32 = 1 } mrf_scalar;
typedef enum { MRF_FAMILY_POW = 0, MRF_FAMILY_LOGISTIC = 1 } mrf_family;
typedef enum { MRF_GATE_NONE = 0, MRF_GATE_NEG = 1 } mrf_gate;

/* One leaf string of the reference, reduced to its family.  The strings the reference
 * passes (example_pandas_Jointspace.py:87-89, example_pointmasses_static.py:106-107) are
 *   POW      :  k / x**p * gate(xdot) * xdot**2
 *   LOGISTIC :  k * (1/(1 + c*exp(-s*x)) - 1) * gate(xdot) * xdot**2
 * gate NEG is (1 - heaviside(xdot)) == -0.5*(sign(xdot)-1): 1 for xdot<0, 0 for xdot>0, 0.5 at 0.
 * As a geometry the value is h(x,xdot); as a Finsler energy L the metric is d2L/dxdot2. */
typedef struct {
  int32_t family;
  int32_t gate;
  int32_t p;
  int32_t reserved;
  double k, c, s;
} mrf_leaf_fn;

typedef struct {
  int32_t abi_version; /* MRF_ABI_VERSION */
  int32_t model;       /* mrf_model */
  int32_t scalar;      /* mrf_scalar */
  int32_t mode;        /* mrf_mode: action = qddot | qdot + dt*qddot   (EXJ:133) */
  int32_t n_robots;    /* N: robots per scenario */
  int32_t n_spheres;   /* S: exchanged spheres per robot in rollouts (reference: 8 link origins) */
  int32_t horizon;     /* H */
  int32_t dynamic;     /* STATIC_OR_DYN_FABRICS: 0 zeroes the exchanged v,a (FPJ:215-217) */
  int32_t n_ego;       /* 6 = collision+plane leaves on the links of ego_link_mask; 0 = "grasp" planner (EXJ:160-166); planar3: 1 or 0 */
  int32_t n_planes;    /* 0 or 1 plane constraint per ego point */
  int32_t use_limits;  /* joint-limit leaves on/off */
  int32_t n_goals;     /* attractors in use: panda 0..3, planar3 0..1 */
  int32_t plane_abs;   /* |a.x+d| in the plane task map */
  int32_t zero_small_action; /* action := 0 when |action| < eps */
  int32_t obst_dim;    /* dynamic_obstacle_dimension: 3 (pandas) or 2 (point robots, dynamic example) */
  int32_t goal_estimate_mask; /* bit i: robot i's x_goal_0 is replaced by x_ee + goal_estimate_T * v_ee at rollout start (EXC:355-357) */
  double dt;           /* planner time step (PM:8) */
  double eps;          /* regulariser in (M + eps I)^-1 and in the energization denominators */
  double jdot_sign;    /* sign in c = jdot_sign * d(J qdot)/dq qdot   (utils.py:28) */
  double goal_estimate_T; /* 20*0.01 (EXJ:347) */
  double base_mass;    /* base energy 0.5*m*qdot.qdot */
  double attr_k, attr_alpha;        /* psi = w*k*(|x| + log(1+exp(-2 alpha |x|))/alpha) */
  double attr_mu, attr_ml, attr_a;  /* A = (mu-ml)*exp(-(a|x|)^2) + ml ;  L = xdot^T A xdot */
  double beta_a, beta_r, beta_b, beta_s; /* 0.5*(tanh(-a(|x|-r))+1)*b + s + max(0, a_ex - a_le) */
  double eta_a, eta_s;              /* 0.5*(tanh(-a*qdot.qdot - s)+1) ; a = 0.9*(1-1/2) */
  double mount[MRF_MAX_ROBOTS][12]; /* 3x4 row-major [R|t] per robot (EXJ:107-118) */
  double limits[MRF_DOF_MAX][2];    /* EXJ:97-105 */
  int32_t sphere_link[MRF_MAX_SPHERES];     /* parent link number 1..8 */
  double sphere_offset[MRF_MAX_SPHERES][3]; /* link-local xyz (SIM:188-245) */
  double sphere_radius[MRF_MAX_SPHERES];
  mrf_leaf_fn collision_geometry, collision_finsler;
  mrf_leaf_fn plane_geometry, plane_finsler;
  mrf_leaf_fn limit_geometry, limit_finsler;
  int32_t kernel_select; /* coupled kernels: 0 = auto by batch size, 1 = row-per-lane (throughput), 2 = one wave per scenario
                          * (latency), 3 = a pair of waves per row, two resident waves per SIMD (joint-space rollout with
                          * float64, the reference's leaf strings and link-origin spheres; everything else -- and every
                          * configuration in a library built without -DMRF_WITH_WP, mrf_build_has_wp() -- as 1) */
  int32_t ego_link_mask; /* panda7, n_ego == 6: bit (l-3) set = panda_link l (l = 3..8) carries collision and plane leaves --
                            the collision_links list of set_components (EXJ:91-96,123-125; the Cartesian rollout class
                            defaults to link 7 alone, FPC:20-21).  0x3F = all six (the examples' setting). */
  int32_t exchange;      /* robot-sharded rollouts (mrf_rollout_sharded): what a robot puts on the wire per scenario and
                          * step -- mrf_exchange_kind; 0 = MRF_EXCHANGE_JOINTS */
  int32_t reserved_tail; /* 0 */
} mrf_config;

/* Payload of the exchange step of a robot-sharded rollout (FPJ:211-225 across GPUs), per robot, scenario and step:
 *   MRF_EXCHANGE_JOINTS   cos q, sin q, qdot of the 7 joints = 21 scalars (168 B in float64) whatever the sphere table;
 *                         the receiver re-walks the sender's chain (its mount is part of every rank's configuration) and
 *                         streams the spheres x, v, a into its leaf sums -- the spheres themselves never exist in memory.
 *   MRF_EXCHANGE_SPHERES  the SX = mrf_exchange_spheres() predicted spheres as 9 scalars each (x, v, a): 432 B for the
 *                         reference's link-origin table, 1 440 B for 20 spheres -- the literal "all-gather of sphere
 *                         centres".
 * Either way robots that live on the SAME rank exchange on chip (per-wave LDS tile, as mrf_rollout does); only robots of
 * other ranks go through the exchange buffers.  Results agree to round-off (different summation order). */
typedef enum { MRF_EXCHANGE_JOINTS = 0, MRF_EXCHANGE_SPHERES = 1 } mrf_exchange_kind;
#define MRF_JOINT_STATE_SCALARS 21 /* [7][3]: row 3*j + 0 = cos q_j, + 1 = sin q_j, + 2 = qdot_j */

typedef struct mrf_handle mrf_handle;

/* Fill cfg with the reference's Panda defaults (leaf strings EXJ:87-89, limits EXJ:97-105,
 * dt/mode PM:8,12, 8 link-origin spheres r=0.08 PM:23-26) for n_robots in {2,3} mounts
 * (PM:83-105) or a ring of mounts otherwise.  Host-only, no device needed. */
void mrf_default_config_panda(mrf_config* cfg, int32_t n_robots, int32_t horizon);
/* Defaults of the 4-point-robot examples (example_pointmasses_static.py:102-129). */
void mrf_default_config_planar3(mrf_config* cfg, int32_t n_robots);

/* Validates cfg, selects the device (device_id >= 0) and uploads the immutable constants.
 * There is no CPU path: without a usable HIP device this returns MRF_E_DEVICE. */
int mrf_create(const mrf_config* cfg, int32_t device_id, mrf_handle** out);
void mrf_destroy(mrf_handle* h);
const char* mrf_last_error(const mrf_handle* h);
int mrf_abi_version(void);
/* 1 when the library carries the float32 instantiations of the kernels (built with -DMRF_WITH_F32), else 0: mrf_create
 * then refuses cfg.scalar = MRF_F32 with MRF_E_CONFIG.  The default build is float64 only -- the reference's arithmetic
 * (CasADi SX / DM); float32 is a throughput option for well-separated robots (accuracy: DESIGN.md section 3). */
int mrf_build_has_f32(void);
/* 1 when the library carries k_rollout_panda_wp (built with -DMRF_WITH_WP): kernel_select = 3 then runs the wave-pair
 * rollout where it applies.  The default build does not (round 6: measured 7-9 % slower than the row-per-lane kernel,
 * never auto-selected); kernel_select = 3 then runs the row-per-lane kernel. */
int mrf_build_has_wp(void);
int64_t mrf_config_sizeof(void); /* sizeof(mrf_config) as compiled, for FFI layout checks */

/* Replaces ParameterizedFabricPlanner.compute_action (EXJ:441,444; EXC:447,449; FPC:150-190).
 *   q, qdot      [dof][rows]
 *   params       [MRF_NPARAM][rows]
 *   obst_x/v/a   [n_obst][3][rows]   obst_r [n_obst][rows]
 *                the first n_obst_static obstacles are the planner's static spheres (x_obsts / radius_obsts:
 *                full 3-D distance, v and a not read); the rest are its dynamic spheres (x_obsts_dynamic, ...,
 *                distance in cfg.obst_dim dimensions).  obst_v / obst_a may be NULL (= zeros).
 *   qddot_out    [dof][rows] (may be NULL)     action_out [dof][rows]
 */
int mrf_compute_action(mrf_handle* h, int64_t rows, const void* q, const void* qdot, const void* params,
                       int32_t n_obst, int32_t n_obst_static, const void* obst_x, const void* obst_v, const void* obst_a,
                       const void* obst_r, void* qddot_out, void* action_out, void* stream);

/* compute_action for all robots of n_scenarios scenarios with the host-side obstacle assembly of the control loop
 * (example_pandas_Jointspace.py:394-412) done on the device: the dynamic obstacles of robot i are the configured
 * spheres (cfg.sphere_*) of every other robot of its scenario -- x from forward kinematics, v = J qdot
 * (zero when cfg.dynamic == 0, EXJ:336-337), a = 0 as the reference passes it (EXJ:411) or, with use_accel != 0,
 * jac_dot*qdot as in the rollouts (FPJ:97-99).  Arrays as mrf_compute_action, rows = n_scenarios * n_robots. */
int mrf_compute_action_coupled(mrf_handle* h, int64_t n_scenarios, const void* q, const void* qdot,
                               const void* params, int32_t use_accel, void* qddot_out, void* action_out, void* stream);

/* Replaces ForwardFabricsPlanner.get_velocity_rollouts / rollouts_numerical (FPJ:190-249,298-423):
 * the coupled N-robot, H-step joint-space rollout.  rows = n_scenarios * n_robots.
 *   q0, qdot0 [dof][rows]   params [MRF_NPARAM][rows]
 *   avg_vel_out [rows]                       mean squared joint velocity (FPJ:102-116)
 *   traj_q, traj_qdot [H][dof][rows]         optional (NULL to skip), == q_N_fun / q_dot_N_fun
 */
int mrf_rollout(mrf_handle* h, int64_t n_scenarios, const void* q0, const void* qdot0, const void* params,
                void* avg_vel_out, void* traj_q, void* traj_qdot, void* stream);

/* Replaces ForwardFabricsPlanner.rollouts_numerical_obstacles (FPJ:425-512; the per-step obstacle outputs of the
 * rollout graph, FPJ:211-225,283-289): the sphere states every robot PUBLISHES at each horizon step of a rollout whose
 * trajectories traj_q / traj_qd [H][dof][rows] mrf_rollout has written -- positions after the step's position update,
 * v = J qdot and a = jdot_sign * Jdot qdot with the velocities that ENTER the step (qdot0 for step 0, traj_qd[k-1]
 * after).  x_out, v_out, a_out [H][S][3][rows] (v_out / a_out may be NULL); one mrf_fk_spheres launch per step. */
/* The clock the LAST row-per-lane mrf_rollout of this handle ran at, measured inside the kernel: its first and its last
 * workgroup stamp the shader-cycle counter (s_memtime) and the constant-rate wall clock (s_memrealtime) on entry and on
 * exit.  Synchronises the device.  out[i], i < n <= MRF_ROLLOUT_CLOCK_N:
 *   0, 1  shader clock [GHz] seen by the first / last workgroup (0: no rollout yet, or the LAST mrf_rollout took a
 *         cooperative kernel, which does not stamp -- the stamps carry the serial number of the call they belong to; a grid
 *         of one workgroup fills both)
 *   2, 3  lifetime of that workgroup [ms]      4  wall-clock rate [GHz]
 * Two boxes (or two runs) whose kernel times differ can be told apart by it: same cycles at a lower clock, or more cycles. */
#define MRF_ROLLOUT_CLOCK_N 5
int mrf_rollout_clock(mrf_handle* h, double* out, int32_t n);
int mrf_rollout_sphere_traj(mrf_handle* h, int64_t n_scenarios, const void* qdot0, const void* traj_q, const void* traj_qd,
                            void* x_out, void* v_out, void* a_out, void* stream);

/* Replaces FabricsRollouts.get_velocity_rollouts / rollouts_numerical (FPC:347-489,538-563):
 * per-row independent rollout, action-then-step, obstacles at constant Cartesian velocity. */
int mrf_rollout_cartesian(mrf_handle* h, int64_t rows, const void* q0, const void* qdot0, const void* params,
                          int32_t n_obst, int32_t n_obst_static, const void* obst_x0, const void* obst_v, const void* obst_a,
                          const void* obst_r, void* avg_vel_out, void* traj_q, void* traj_qdot, void* stream);

/* Replaces the fk/jac/jac_dot helper evaluations (utils.py:16-54,87-119; FPJ:82-100; UFK:3-33):
 * for every row the S configured spheres  x, v = J qdot, a = jdot_sign * Jdot qdot.
 *   x_out, v_out, a_out [S][3][rows]  (v_out/a_out may be NULL) */
int mrf_fk_spheres(mrf_handle* h, int64_t rows, const void* q, const void* qdot, void* x_out, void* v_out,
                   void* a_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Host-buffer entry points: the same four calls for callers that hold HOST arrays of one (or a few) scenarios, as the
 * reference's call sites do (numpy in, numpy out: example_pandas_Jointspace.py:374,441; example_pandas_cartesian.py:404).
 * All arrays are float64 host memory in the layout of the device versions; the library packs them into a pinned staging
 * buffer of the handle, moves them with one copy each way on its own stream and returns after that stream has been
 * synchronised (so these calls are synchronous; results are valid on return).  Optional arrays as in the device
 * versions (NULL = absent). */
int mrf_compute_action_host(mrf_handle* h, int64_t rows, const double* q, const double* qdot, const double* params,
                            int32_t n_obst, int32_t n_obst_static, const double* obst_x, const double* obst_v,
                            const double* obst_a, const double* obst_r, double* qddot_out, double* action_out);
int mrf_rollout_host(mrf_handle* h, int64_t n_scenarios, const double* q0, const double* qdot0, const double* params,
                     double* avg_vel_out, double* traj_q, double* traj_qdot);
int mrf_rollout_cartesian_host(mrf_handle* h, int64_t rows, const double* q0, const double* qdot0, const double* params,
                               int32_t n_obst, int32_t n_obst_static, const double* obst_x0, const double* obst_v,
                               const double* obst_a, const double* obst_r, double* avg_vel_out, double* traj_q,
                               double* traj_qdot);
int mrf_fk_spheres_host(mrf_handle* h, int64_t rows, const double* q, const double* qdot, double* x_out, double* v_out,
                        double* a_out);

/* Robot-sharded rollout (one or a few robots per GPU).  One rollout step is
 *   mrf_step_predict : q += dt*qdot for the owned robots; writes their spheres (x,v,a)
 *   <all-gather of the sphere block across ranks -- done by the host over RCCL>
 *   mrf_step_action  : fabric solve of the owned robots against all other robots' spheres;
 *                      qdot := action; sumsq += |qdot|^2
 * Owned rows are [n_scenarios][robot_count] with robot index robot_first + (row % robot_count).
 *   sph_own  [robot_count][SX][9][n_scenarios]   (x,v,a interleaved as 9 components)
 *   sph_all  [n_robots  ][SX][9][n_scenarios]
 * SX = mrf_exchange_spheres(h): cfg.n_spheres, minus one for each pair of coincident spheres of the link-origin table
 * (the origins of links 1/2 and of links 5/6 are the same point; with equal radii only one of them travels and the
 * receiver counts it twice) -- 6 instead of 8 for the reference's rollout table.
 */
int32_t mrf_exchange_spheres(const mrf_handle* h);
/* Once per rollout, before the first step: params_out := params with x_goal_0 replaced by x_ee + goal_estimate_T * v_ee
 * for the owned robots in cfg.goal_estimate_mask (RF-CV, EXC:355-357) -- what mrf_rollout does in its prologue. */
int mrf_step_prepare(mrf_handle* h, int64_t n_scenarios, int32_t robot_first, int32_t robot_count, const void* q,
                     const void* qdot, const void* params, void* params_out, void* stream);
int mrf_step_predict(mrf_handle* h, int64_t n_scenarios, int32_t robot_first, int32_t robot_count, void* q_io,
                     const void* qdot, void* sph_own, void* stream);
int mrf_step_action(mrf_handle* h, int64_t n_scenarios, int32_t robot_first, int32_t robot_count, const void* q,
                    void* qdot_io, const void* params, const void* sph_all, void* sumsq_io, void* stream);
/* The same step with the MRF_EXCHANGE_JOINTS payload:
 *   mrf_step_predict_joints : q += dt*qdot for the owned robots; writes cos q, sin q, qdot  -> jst_own
 *   <all-gather of the joint-state block across ranks>
 *   mrf_step_action_joints  : fabric solve of the owned robots; the spheres of the robots of OTHER ranks are re-derived
 *                             from jst_all by a chain walk per (lane, robot), the owned robots exchange theirs on chip
 *                             (their entries of jst_all are read as the own state: cos q / sin q are not recomputed)
 *   jst_own  [robot_count][21][n_scenarios]      jst_all  [n_robots][21][n_scenarios]   (MRF_JOINT_STATE_SCALARS rows) */
int mrf_step_predict_joints(mrf_handle* h, int64_t n_scenarios, int32_t robot_first, int32_t robot_count, void* q_io,
                            const void* qdot, void* jst_own, void* stream);
int mrf_step_action_joints(mrf_handle* h, int64_t n_scenarios, int32_t robot_first, int32_t robot_count, const void* q,
                           void* qdot_io, const void* params, const void* jst_all, void* sumsq_io, void* stream);
/* mrf_step_action_joints of step k AND mrf_step_predict_joints of step k + 1 in one launch: after the solve, q_io += dt *
 * action and the joint state of the following step (cos q, sin q rotated by the increment as mrf_rollout does, the action
 * as qdot) goes to jst_next_own [robot_count][21][n_scenarios] -- the block the next all-gather sends.  jst_next_own may be
 * the rank's own block INSIDE jst_all when nobody else reads it during the launch (a group of one).  What
 * mrf_rollout_sharded's RCCL transport runs for steps 1 .. H-1. */
int mrf_step_action_predict_joints(mrf_handle* h, int64_t n_scenarios, int32_t robot_first, int32_t robot_count, void* q_io,
                                   void* qdot_io, const void* params, const void* jst_all, void* sumsq_io,
                                   void* jst_next_own, void* stream);
/* scalars one robot puts on the wire per scenario and step under cfg.exchange: 21, or 9 * mrf_exchange_spheres() */
int32_t mrf_exchange_scalars(const mrf_handle* h);

/* ------------------------------------------------------------------------------------------------------------
 * Robot-sharded rollout INSIDE the library (SURVEY 8b "mrf_comm_init / mrf_rollout_sharded", 8e).
 * A group of `world` processes, one per GPU, shares the robots of every scenario: rank r owns the contiguous robot
 * block that mrf_comm_partition() reports (block sizes differ by at most one).  mrf_rollout_sharded runs the whole
 * H-step recurrence of forward_planner_Jointspace.py:190-249 for the owned robots; the exchange step of that graph
 * (FPJ:211-225: at step k every robot reads every other robot's predicted spheres x, v, a) is done by the library:
 *
 *   MRF_TRANSPORT_RCCL   per step: mrf_step_predict -> ncclAllGather (RCCL, over xGMI between GPUs) of the ranks' sphere
 *                        blocks -> mrf_step_action, all enqueued on the caller's stream from C++ (no host round trip).
 *                        librccl is dlopen'ed on first use; the library has no link-time dependency on it.
 *   MRF_TRANSPORT_PEER   device-initiated exchange: every rank maps every other rank's exchange buffer
 *                        (hipIpcGetMemHandle / hipIpcOpenMemHandle; the caller carries the 64-byte handles between the
 *                        processes by whatever means it has); ONE persistent kernel per rollout walks the H steps and,
 *                        per step, stores its robots' spheres straight into all peers' buffers, raises a per-workgroup
 *                        flag there and polls the peers' flags for the same scenarios (bounded spin: a peer that never
 *                        arrives ends the kernel and is reported by mrf_comm_status, it cannot hang the GPU).
 *
 * world <= n_robots; further GPUs run replicas of the group on other scenario batches (independent, no exchange).
 * Every rank of the group must make the same sequence of mrf_rollout_sharded calls with the same n_scen.  A timed-out
 * peer exchange is the GROUP's and sticky: the rank whose wait ran out raises the error word of every rank and stops
 * publishing, so every rank that folded (or would fold) its missing spheres ends in error too.  The PEER kernel writes
 * to staging arrays; a commit pass after it reads the error word ONCE and either advances ALL rows of q_io / qdot_io or
 * none of them, in which case avg_vel is NaN for every row (a result that cannot be mistaken for a rollout).  Every later
 * wait of that communicator ends at once, until mrf_comm_reset or mrf_comm_destroy.  (A rank that had already completed
 * its rollout with valid peers' data when another rank gave up keeps its -- correct -- result and learns of the broken
 * group from mrf_comm_status.)  The PEER kernel's grid is capped at the resident workgroup count, so no wait depends on a
 * workgroup that has not been dispatched.
 *   q_io, qdot_io [dof][n_scen*count]  the OWNED rows, row = scenario*count + (robot - first); advanced in place
 *   params        [MRF_NPARAM][n_scen*count]      avg_vel_out [n_scen*count]
 */
typedef enum { MRF_TRANSPORT_NONE = 0, MRF_TRANSPORT_RCCL = 1, MRF_TRANSPORT_PEER = 2 } mrf_transport;
#define MRF_COMM_ID_BYTES 128   /* sizeof(ncclUniqueId) */
#define MRF_IPC_HANDLE_BYTES 64 /* sizeof(hipIpcMemHandle_t) */
/* rank 0 of the group: a fresh communicator id (ncclGetUniqueId) to hand to every rank's mrf_comm_init */
int mrf_comm_unique_id(void* id_out);
/* RCCL transport.  world == 1 with unique_id == NULL creates a group of one without touching RCCL. */
int mrf_comm_init(mrf_handle* h, int32_t rank, int32_t world, const void* unique_id);
/* PEER transport, two phases around the caller's exchange of the handles: open allocates this rank's exchange buffer
 * for up to max_scenarios scenarios and writes its IPC handle (MRF_IPC_HANDLE_BYTES); connect maps the peers' buffers
 * from ipc_handles_all [world][MRF_IPC_HANDLE_BYTES] (rank order; the own entry is not opened). */
int mrf_comm_peer_open(mrf_handle* h, int32_t rank, int32_t world, int64_t max_scenarios, void* ipc_handle_out);
int mrf_comm_peer_connect(mrf_handle* h, const void* ipc_handles_all);
/* The same transport for a group whose ranks live in ONE process (one handle and one stream per rank; the ranks' devices may
 * differ): instead of IPC handles the ranks exchange the plain device pointers of their exchange buffers --
 * mrf_comm_peer_local_base() of every rank, then mrf_comm_peer_connect_local(h, bases_all[world]) on every rank (peer access
 * between different devices is enabled here).  The rollouts of the ranks must be issued on DIFFERENT streams (they wait for
 * each other on the device) before any of them is synchronised.  Ranks that share a device share its workgroup slots.
 * Used by the single-process tests of the multi-rank kernels at production grid sizes and by tools/shard_local.py. */
int mrf_comm_peer_local_base(const mrf_handle* h, void** base_out);
/* *concurrent_out = 1 when a kernel on stream_b runs while a kernel on stream_a is still running (HIP maps streams onto a
 * few hardware queues; two streams that share one run their kernels one after the other, and persistent kernels that wait
 * for each other would then wait for ever -- until the bounded wait gives up).  Takes up to 20 ms when they do not. */
int mrf_streams_concurrent(int32_t device, void* stream_a, void* stream_b, int32_t* concurrent_out);
int mrf_comm_peer_connect_local(mrf_handle* h, void* const* bases_all);
int mrf_comm_partition(const mrf_handle* h, int32_t* robot_first, int32_t* robot_count);
/* What the communicator of this handle is, for logs that must prove what ran (bench.py's robot_sharded block): out[i],
 * i < n <= MRF_COMM_INFO_N:
 *   0 transport  1 rank  2 world  3 robot_first  4 robot_count
 *   5 ncclCommCount of the RCCL communicator (0: none)   6 ncclCommUserRank (-1: none)   7 ncclCommCuDevice (-1: none)
 *   8 HIP device of the handle   9 peer exchange buffers mapped from other ranks (PEER transport after connect)
 *   10 cfg.exchange (mrf_exchange_kind)   11 scalars per robot, scenario and step on the wire (mrf_exchange_scalars)
 *   12 peers whose mapped buffer lies on a device this one reaches in ONE hop (hipExtGetLinkTypeAndHopCount; -1: not
 *      a connected PEER communicator)
 *   13 workgroups of the persistent PEER kernel's footprint found CO-RESIDENT on this device by the roll call of
 *      mrf_comm_peer_connect -- the cap of that kernel's grid (0: not measured: a group of one, or ranks sharing a device)
 *   14 the last mrf_rollout_sharded of the PEER transport walked its blocks in PAIRS (k_rollout_peer_paired: joint payload,
 *      more blocks than workgroup slots; the exchange of one block runs under the step of the other)
 *   15 the PEER communicator exchanges TAGGED payload words (opt-in MRF_PEER_TAGGED=1 at mrf_comm_peer_open: every 32 bits
 *      of joint state in one 8-byte store with the tag of its step, polled by the reader; no flags; twice the link bytes) */
#define MRF_COMM_INFO_N 16
int mrf_comm_info(const mrf_handle* h, int32_t* out, int32_t n);
int32_t mrf_comm_transport(const mrf_handle* h);
/* Where the exchange buffers of a connected PEER communicator really are, per rank g of the group: out[g*MRF_PEER_INFO_N + i]
 *   0 device ordinal that owns the mapped allocation (hipPointerGetAttributes on the IPC mapping; the own rank: the
 *     handle's device; -1: unknown)      1 hipDeviceCanAccessPeer(own device -> that device) (own rank: 1)
 *   2 link type and 3 hop count of hipExtGetLinkTypeAndHopCount(own, that) (own rank / same device: 0, 0; -1: the call
 *     failed)   link type: the HSA_AMD_LINK_INFO_TYPE_* value (2 = PCIe, 4 = xGMI)
 * n = number of int32 available in out (>= world * MRF_PEER_INFO_N).  A first multi-GPU bench line can thereby show
 * whether the peer stores crossed xGMI or fell back to PCIe. */
#define MRF_PEER_INFO_N 4
int mrf_comm_peer_info(const mrf_handle* h, int32_t* out, int32_t n);
/* The node as HIP shows it to this process, no handle needed: *n_devices = hipGetDeviceCount; for i, j < min(n_devices,
 * cap): can_access[i*cap + j] = hipDeviceCanAccessPeer(i, j) (diagonal 1), link_type / hops[i*cap + j] from
 * hipExtGetLinkTypeAndHopCount (diagonal 0; -1 where the call fails).  Does not create contexts on other devices. */
int mrf_device_topology(int32_t* n_devices, int32_t* can_access, int32_t* link_type, int32_t* hops, int32_t cap);
int mrf_rollout_sharded(mrf_handle* h, int64_t n_scen, void* q_io, void* qdot_io, const void* params, void* avg_vel_out,
                        void* stream);
/* Waits for the stream of the last mrf_rollout_sharded and reports a timed-out exchange (MRF_E_LAUNCH) or MRF_OK. */
int mrf_comm_status(mrf_handle* h);
/* Makes a communicator usable again after a timed-out PEER exchange: synchronises this rank's stream, clears the error
 * word and the flags in this rank's exchange buffer and restarts the sequence numbers in a new EPOCH.  The reset count
 * is the high part of every sequence number AND the tag of the error word: a flag of the old sequence that a slower peer
 * stores after the clearing can satisfy no wait of the new one, and a peer kernel of the old epoch that times out late
 * writes the OLD tag, which the kernels, the commit pass and mrf_comm_status of the new epoch do not take for an error.
 * Call it on EVERY rank of the group the same number of times, BETWEEN two barriers of the caller's (no rank may start a
 * mrf_rollout_sharded while another one still resets); afterwards the ranks continue with the same sequence of calls
 * again.  A no-op for the RCCL transport. */
int mrf_comm_reset(mrf_handle* h);
#define MRF_PEER_TIMEOUT_DEFAULT_MS 10000 /* bounded flag wait of the PEER kernel; override: env MRF_PEER_TIMEOUT_MS */
/* env MRF_PEER_DEVICE_SHARE = k: k ranks of a group run on ONE device (single-GPU test setups); the PEER kernel then caps
 * its grid at 1/k of the device's resident workgroups so that all k kernels fit at once. */
void mrf_comm_destroy(mrf_handle* h); /* also done by mrf_destroy */

/* ------------------------------------------------------------------------------------------------------------
 * Device-resident control step: everything between two simulator steps of
 * examples/example_pandas_Jointspace.py:280-458 stays on the GPU,
 *     mrf_control_prepare   end-effector FK + RF-CV goal estimate                 (EXJ:325-329, 346-348)
 *     mrf_rollout           Rollout Fabrics -> mean squared velocity per robot    (EXJ:354-375)
 *     mrf_deadlock_step     deadlock detection / resolution                      (EXJ:377-383, DP:50-118)
 *     mrf_compute_action_coupled                                                 (EXJ:394-448)
 *     mrf_apply_action      clip + exact velocity integration                     (EXJ:452-453, urdfenvs 'vel' mode)
 * mrf_episode_run chains them for n_steps control steps without a host round trip (optionally as a replayed HIP graph).
 */
typedef struct mrf_deadlock_config { /* thresholds of deadlock_prevention.py:12-27 and the literals of :50-118 */
  double avg_vel_constant;     /* DP:20 (0.16; point masses DP:13 0.03)  deadlock if avg velocity signal below */
  double dist_constant;        /* DP:21 (0; 1)        ... and the pair's summed goal distance above            */
  double goal_weight_follower; /* DP:22 (2; 10) */
  double goal_weight_leader;   /* DP:23 (3; 1)  */
  double nr_goal_scale;        /* DP:25 (2; 100) */
  double ee_distance;          /* DP:63  0.35   ... and the end effectors closer than this                     */
  double follower_offset;      /* DP:95  0.3    follower goal = x_follower - offset * unit(x_leader - x_follower) */
  double min_goal_norm;        /* DP:94  0.05   */
  double z_floor;              /* DP:98-99 0.1  replaces a negative follower-goal height                       */
  int32_t time_wait;           /* DP:24 (300; 50) steps the resolution is held after the deadlock disappears   */
  int32_t min_time_step;       /* DP:66,81  10 */
  int32_t grasp_state;         /* DP:108  2   state-machine state that cancels the hold ...                    */
  int32_t grasp_timeout;       /* DP:109  400 ... by setting time_deadlock_out to this                         */
} mrf_deadlock_config;
void mrf_default_deadlock_config(mrf_deadlock_config* c, int32_t point_mass);
int64_t mrf_deadlock_config_sizeof(void); /* sizeof(mrf_deadlock_config) as compiled, for FFI layout checks */

/* per-scenario deadlock state, int32 dl_state[MRF_DL_NSTATE][n_scenarios] + scalar dl_goal[3][n_scenarios]
 * (= the attributes of the reference's deadlockprevention object that its logic reads back, DP:9-34) */
#define MRF_DL_LEADER 0
#define MRF_DL_FOLLOWER 1
#define MRF_DL_DEAD0 2
#define MRF_DL_DEAD1 3
#define MRF_DL_TIME_IN_DEADLOCK 4
#define MRF_DL_TIME_DEADLOCK_OUT 5 /* the driver's loop variable, initial value 1000 (EXJ:273) */
#define MRF_DL_TIME_STEP 6         /* control-step counter w (EXJ:279), advanced by mrf_deadlock_step */
#define MRF_DL_NONFINITE 7         /* control steps whose rollout average was not finite (the test DP:66 is then false) */
#define MRF_DL_NSTATE 8
int mrf_deadlock_init(mrf_handle* h, int64_t n_scenarios, int32_t* dl_state, void* dl_goal, void* stream);

/* rows = n_scenarios * n_robots.  Copies params_nominal to params_work (they may alias), writes the hand position
 * x_ee [3][rows] and, when apply_estimate != 0, overwrites x_goal_0 in params_work with x_ee + goal_estimate_T * v_ee
 * for the robots in cfg.goal_estimate_mask -- as the reference does, where the estimate then also reaches
 * deadlock_checking and compute_action (EXJ:346-348 -> :377, :425).  With apply_estimate == 0 the estimate stays
 * inside mrf_rollout (which applies the same mask itself) and the other stages see the true goals. */
int mrf_control_prepare(mrf_handle* h, int64_t n_scenarios, const void* q, const void* qdot, const void* params_nominal,
                        void* params_work, int32_t apply_estimate, void* x_ee_out, void* stream);

/* Replaces deadlockprevention.deadlock_checking (DP:50-118) for every scenario: avg_sum = sum_i avg_vel[i] / N
 * (EXJ:375), x_robots = x_ee, goals / weights = x_goal_0 / weight_goal_0 of params_work (rewritten in place for the
 * leader and the follower), sm_state [rows] the robots' state-machine states (NULL = all 0, "approaching").
 * time_step < 0 uses (and advances) the device counter MRF_DL_TIME_STEP; otherwise the given step is used. */
int mrf_deadlock_step(mrf_handle* h, int64_t n_scenarios, const mrf_deadlock_config* dl, int32_t time_step,
                      const void* x_ee, const void* avg_vel, const int32_t* sm_state, void* params_work,
                      int32_t* dl_state, void* dl_goal, void* stream);

/* action := clip(action, +-vel_limit) (EXJ:452; vel_limit [dof] host array), q += dt * action, qdot := action
 * (mode 'vel').  stop_margin >= 0 additionally clamps q to [lo + margin, hi - margin] of cfg.limits, the hard joint
 * stops a simulator imposes; < 0 disables it. */
int mrf_apply_action(mrf_handle* h, int64_t rows, void* q_io, void* qdot_io, void* action_io, const double* vel_limit,
                     double stop_margin, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Pick-and-place sequencing on the device: StateMachine.get_state_machine_panda of
 * multi_robot_fabrics/others_planner/state_machine.py:133-214 and get_gripper_action_panda (:66-86) for every row,
 * as the driver runs them per robot before the fabric calls (example_pandas_Jointspace.py:304-316, 448).
 * Per-row state:  int32 sm_state[MRF_SM_NSTATE][rows],  scalar sm_goal[MRF_SM_NGOAL][rows].
 * States (SM:150-199): 0 go home / open, 1 above the block, 2 down to it, 3 grip (0.3 s), 12 lift, 4 carry home,
 * 5 release, 10 all blocks done.  The Kinova variant of the reference (a second robot type) is not built. */
#define MRF_SM_STATE 0    /* the value get_state_machine_panda returns; this row is the sm_state input of mrf_deadlock_step */
#define MRF_SM_PICKED 1   /* nr_blocks_panda_success */
#define MRF_SM_FAILED 2   /* nr_blocks_panda_failed  */
#define MRF_SM_T_GRIP 3   /* time_gripping_panda     */
#define MRF_SM_GRIPPER 4  /* 0 "open", 1 "close"     */
#define MRF_SM_STOP 5     /* stop_time_panda         */
#define MRF_SM_NSTATE 6
#define MRF_SM_GOAL 0       /* [3] self.goal            */
#define MRF_SM_GOAL_ABOVE 3 /* [3] self.goal_above_block */
#define MRF_SM_WEIGHT 6     /* self.weight_goal         */
#define MRF_SM_NGOAL 7
typedef struct mrf_state_machine_config { /* literals of state_machine.py:133-214 and :66-86 */
  double reach_home, reach_pregrasp, reach_block, reach_lift, reach_drop; /* 0.05 0.013 0.013 0.04 0.15 */
  double pregrasp_height, lift_height;    /* 0.1, 0.15 */
  double grip_steps;                      /* 0.3/0.01: the grip is held while time_gripping <= this */
  double open_tol, dropped_below_z;       /* 0.005, 0.6 */
  double weight_high, weight_low;         /* 2, 0 */
  double gripper_open[2];                 /* [0.04, 0.04] */
  double v_close, v_open;                 /* -0.05, 0.4 */
  int32_t nr_blocks;                      /* blocks per robot (n_cubes / nr_robots, EXJ:255) */
  int32_t model;                          /* 0: block and gripper observations are inputs (as from a simulator);
                                             1: minimal built-in model for simulator-free episodes: the block of a row is
                                             blocks[min(picked, nr_blocks-1)], carried with the hand while the gripper is
                                             closed in states 12/4, and the finger joints integrate their own velocity
                                             command (q += dt * action, clipped to [0, gripper_open]) */
} mrf_state_machine_config;
void mrf_default_state_machine_config(mrf_state_machine_config* c, int32_t nr_blocks);
int64_t mrf_state_machine_config_sizeof(void);
/* initial object state (SM:9-38): state 1, goal = start_goal [3][rows], weight 2, gripper open */
int mrf_state_machine_init(mrf_handle* h, int64_t rows, const void* start_goal, int32_t* sm_state, void* sm_goal, void* stream);
/* One update of every row.  x_ee [3][rows] (mrf_control_prepare), start_goal [3][rows],
 * blocks [n_block_arrays][3][rows]: model 0 reads blocks[0] as this step's goal_block (already lifted by 0.1 as the
 * driver does, EXJ:303); model 1 indexes it by the picked count.  q_gripper_io [2][rows] finger joints (model 1 advances
 * them).  Writes weight_goal_0 of params_work for every row and x_goal_0 for the rows whose robot bit is NOT in
 * skip_robot_mask (the RF-CV estimate of mrf_control_prepare overrides the state machine's GOAL there, EXJ:346-348; the
 * weight of an estimated robot is still the state machine's, EXJ:313-316 -> :363), and the gripper velocity command
 * gripper_action_out [2][rows] (may be NULL). */
int mrf_state_machine_step(mrf_handle* h, int64_t rows, const mrf_state_machine_config* sm, const void* x_ee,
                           const void* start_goal, const void* blocks, int32_t n_block_arrays, void* q_gripper_io,
                           int32_t* sm_state, void* sm_goal, void* params_work, int32_t skip_robot_mask,
                           void* gripper_action_out, void* stream);
/* Attaches pick-and-place buffers to an action handle: every control step of mrf_episode_run on that handle then runs
 * mrf_state_machine_step between mrf_control_prepare and the rollout, feeds sm_state row MRF_SM_STATE to the
 * deadlock logic (the `sm_state` argument of mrf_episode_run is ignored), and selects the action per row as the
 * driver does (EXJ:414-445): states 3 and 5 stand still (zero action); state 2 takes the action of h_grasp -- the
 * "grasp" planner without collision links (EXJ:160-166; a handle created with n_ego = 0 and the same mounts), written
 * to action_grasp_work [dof][rows]; NULL h_grasp keeps the full planner in state 2.  All arrays are caller-owned and
 * must outlive the episodes; sm == NULL detaches. */
int mrf_episode_set_pick_place(mrf_handle* h_action, const mrf_state_machine_config* sm, const void* start_goal,
                               const void* blocks, int32_t n_block_arrays, void* q_gripper_io, int32_t* sm_state,
                               void* sm_goal, void* gripper_action_out, mrf_handle* h_grasp, void* action_grasp_work);

/* The Cartesian Rollout Fabrics of the reference's second driver with the obstacle assembly on the device:
 * FabricsRollouts.get_velocity_rollouts for EVERY robot of every scenario (EXC:366-399; FPC:421-458,561-563) against
 * the obstacles compute_x_obsts_dyn_0 hands it (utils_fabrics_kinematics.py:3-33, EXC:330-352): the configured spheres
 * (cfg.sphere_link / _offset / _radius) of all other robots of the scenario at their current positions, moving with
 * their current velocities J qdot (zero when cfg.dynamic == 0) for the whole horizon, zero accelerations (FPC:33).
 * rows = n_scenarios * n_robots as in mrf_rollout.  Three forms, chosen by table and batch size: small batches run one
 * wave per scenario; tables of up to 8 spheres per robot (or the link-origin table) in mode 'vel' keep the other robots'
 * start states in an LDS tile for the whole horizon and touch no obstacle array at all; anything else assembles obstacle
 * arrays in a work buffer owned by the handle (allocated on the first call of a batch size; not inside a stream capture)
 * and runs mrf_rollout_cartesian on them.  avg_vel_out [rows]; traj_q / traj_qd [H][7][rows] or NULL. */
int mrf_rollout_cartesian_coupled(mrf_handle* h, int64_t n_scenarios, const void* q0, const void* qdot0, const void* params,
                                  void* avg_vel_out, void* traj_q, void* traj_qd, void* stream);

/* Which rollout the control steps of mrf_episode_run take from this ROLLOUT handle: the coupled joint-space rollout
 * (mrf_rollout, example_pandas_Jointspace.py) or the per-robot Cartesian one (mrf_rollout_cartesian_coupled,
 * example_pandas_cartesian.py).  Default MRF_ROLLOUT_JOINTSPACE. */
#define MRF_ROLLOUT_JOINTSPACE 0
#define MRF_ROLLOUT_CARTESIAN 1
int mrf_episode_set_rollout(mrf_handle* h_rollout, int32_t kind);

/* Attaches a recorder to an action handle: every control step of mrf_episode_run on it then also stores, at index
 * i = *step_counter (nothing once i >= capacity; the counter is advanced by the step itself), what a host loop would
 * otherwise read back after every step -- so that any number of steps can be queued back to back:
 *   q_hist  [capacity][dof][rows]    joint positions after the step                                   (or NULL)
 *   sm_hist [capacity][rows] int32   state-machine states of the step (pick-and-place attached, else the sm_state argument) (or NULL)
 *   t_begin, t_end [capacity] int64  the device's constant-rate wall clock (s_memrealtime ticks; rate: mrf_rollout_clock
 *                                    out[4]) when the first kernel of the step started / when its last kernel ran   (or NULL)
 *   done_at [rows] int32             first step index at which the row's state equalled done_state; the caller
 *                                    initialises it to -1                                                (or NULL)
 * All arrays are caller-owned device memory that must outlive the episodes; step_counter is int32[1].  capacity <= 0 or a
 * NULL counter detaches. */
int mrf_episode_set_recorder(mrf_handle* h_action, void* q_hist, int32_t* sm_hist, int64_t* t_begin, int64_t* t_end,
                             int32_t* done_at, int32_t* step_counter, int32_t capacity, int32_t done_state);

/* n_steps control steps on the device.  h_rollout may be NULL (no Rollout Fabrics, no deadlock logic: plain MRDF);
 * dl may be NULL (rollouts monitored, no deadlock logic).  Work buffers are caller-owned:
 *   params_work [MRF_NPARAM][rows]  x_ee_work [3][rows]  avg_work [rows]  action_out [dof][rows] (last step's action)
 * use_graph != 0 captures one control step into a HIP graph (cached in h_action) and replays it. */
int mrf_episode_run(mrf_handle* h_rollout, mrf_handle* h_action, int64_t n_scenarios, int32_t n_steps,
                    const mrf_deadlock_config* dl, int32_t apply_estimate, const double* vel_limit, double stop_margin,
                    void* q_io, void* qdot_io, const void* params_nominal, void* params_work, const int32_t* sm_state,
                    int32_t* dl_state, void* dl_goal, void* x_ee_work, void* avg_work, void* action_out,
                    int32_t use_graph, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MRF_H_ */
