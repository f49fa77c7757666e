/*
 * myriad_hip.h -- C-ABI of libmyriad_hip.so, the MI355X (gfx950) batched trajectory-optimisation engine.
 *
 * This is the drop-in boundary for the reference's hot path (nikihowe/myriad, citations relative to
 * /root/reference/).  The reference is pure Python; the interface it would bind through ctypes is:
 *
 *   reference interface                                              entry point here
 *   --------------------------------------------------------------  ---------------------------
 *   get_optimizer(hp,cfg,system)  trajectory_optimizers/__init__.py:12-28     myr_create / myr_destroy
 *   jit(objective), jit(grad(objective)), jit(constraints),
 *   jit(jacrev(constraints))      nlp_solvers/__init__.py:32-40               myr_eval
 *   solve(hp,cfg,opt_dict) -> minimize_ipopt(...)
 *                                 nlp_solvers/__init__.py:18-98 (:57-58)      myr_solve
 *   get_state_trajectory_and_cost / get_defect   utils.py:258-324            myr_rollout
 *   Python exceptions / solution['success']      nlp_solvers/__init__.py:59-64  return codes, status[], myr_last_error
 *   jax.grad(lagrangian, argnums=0)(x, lmbda)    nlp_solvers/extra_gradient.py:21-33,
 *                                                experiments/e2e_sysid.py:113-125           myr_vjp (add_gradf=1), myr_jvp
 *   step(x, lmbda) (extragradient iteration)     nlp_solvers/extra_gradient.py:25-33        myr_exgd
 *   FBSM(hp,cfg,system).solve()                  trajectory_optimizers/forward_backward_sweep.py:88-116   myr_fbsm
 *
 * Conventions
 *   - all floating point is IEEE fp64 (the reference sets jax_enable_x64, run.py:15);
 *   - every array is caller-allocated, C-contiguous; the library never retains or frees caller memory;
 *   - `mem` says whether ALL array arguments of that call are host pointers (MYR_MEM_HOST: staged through
 *     the handle's device scratch) or device pointers on the handle's device (MYR_MEM_DEVICE: zero-copy);
 *   - per-instance arrays are instance-major: z[B][n] uses exactly the reference's ravel_pytree layout
 *     per instance (states row-major first, then controls);
 *   - calls block until the result is complete; a handle is not thread-safe;
 *   - functions return 0 on success, <0 on error (myr_last_error() gives the thread-local message);
 *     non-convergence is NOT an error (as in the reference): it is reported per instance in status[].
 */
#ifndef MYRIAD_HIP_H
#define MYRIAD_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* system ids: members of myriad.systems.SystemType on the hot path (systems/__init__.py:29-50) */
enum { MYR_SYS_CARTPOLE = 0, MYR_SYS_VANDERPOL = 1, MYR_SYS_CANCERTREATMENT = 2, MYR_SYS_SIMPLECASE = 3,
       /* NodeSystem over CARTPOLE with a (64,64) sigmoid MLP (systems/neural_ode/node_system.py:14-42,
          neural_ode/create_node.py:110-117): params = the 4804 weights, see csrc/node_system.h for the order */
       MYR_SYS_NODE_CARTPOLE = 4,
       /* further autonomous systems without terminal cost (systems/lenhart/<name>.py, miscellaneous/seir.py) */
       MYR_SYS_BIOREACTOR = 5, MYR_SYS_GLUCOSE = 6, MYR_SYS_MOULDFUNGICIDE = 7, MYR_SYS_SIMPLECASEWITHBOUNDS = 8,
       MYR_SYS_HIVTREATMENT = 9, MYR_SYS_EPIDEMICSEIRN = 10, MYR_SYS_SEIR = 11, MYR_SYS_BEARPOPULATIONS = 12,
       /* classical_control/{pendulum,mountain_car}.py (gym-style clips kept), miscellaneous/rocket_landing.py */
       MYR_SYS_PENDULUM = 13, MYR_SYS_MOUNTAINCAR = 14, MYR_SYS_ROCKETLANDING = 15,
       /* systems with a (linear) terminal cost: lenhart/bacteria.py, miscellaneous/tumour.py.  The terminal
          term is applied where the reference applies it for collocation -- the TRAPEZOIDAL objective (trapezoidal.py:126-127)
          the rollout (utils.py:295-296) and, on the integrated end state of the last interval, the SHOOTING objective
          (shooting.py:206-208); not the Hermite-Simpson objective */
       MYR_SYS_BACTERIA = 16, MYR_SYS_TUMOUR = 17,
       /* running cost g(x,u,t) with explicit time (lenhart/harvest.py:61-62, timber_harvest.py:84-85): every transcription
          evaluates it at the reference's point / step times */
       MYR_SYS_HARVEST = 18, MYR_SYS_TIMBERHARVEST = 19,
       /* lenhart/predator_prey.py: terminal cost and ONE pinned terminal state (x_T = [None, None, B]) */
       MYR_SYS_PREDATORPREY = 20,
       /* ELASTIC twins (100 + id; no counterpart in the reference -- the feasibility-restoration device of the solver, DESIGN.md
          "Elastic mode"): x' = f(x,u) + s with NS slack controls s appended to u and the running cost g + rho/2 |s|^2, rho the
          LAST model parameter.  All entry points work on them like on any system (NU = nu + ns controls), except myr_solve under
          MYR_TR_SHOOTING (every twin): MYR_E_UNSUPPORTED, "... built for the collocation transcriptions".  (Up to round 4 ROCKETLANDING's
          twin had no trapezoidal solver either.)  Every system with pinned terminal states has one (round 4). */
       MYR_SYS_CARTPOLE_ELASTIC = 100, MYR_SYS_VANDERPOL_ELASTIC = 101, MYR_SYS_PENDULUM_ELASTIC = 113, MYR_SYS_MOUNTAINCAR_ELASTIC = 114,
       MYR_SYS_ROCKETLANDING_ELASTIC = 115,
       /* lenhart/invasive_plant.py: DISCRETE-time (five foci, five controls).  Only myr_fbsm (its discrete recurrences)
          accepts it; the direct-transcription entry points return MYR_E_UNSUPPORTED, as the reference's direct optimisers
          raise NotImplementedError for it (trajectory_optimizers/base.py:66-67) */
       MYR_SYS_INVASIVEPLANT = 21 };
/* transcription: OptimizerType x QuadratureRule (config.py:12-57) */
enum { MYR_TR_HERMITE_SIMPSON = 0, MYR_TR_TRAPEZOIDAL = 1, MYR_TR_SHOOTING = 2 };
/* IntegrationMethod (config.py:46-50) */
enum { MYR_INT_EULER = 0, MYR_INT_HEUN = 1, MYR_INT_MIDPOINT = 2, MYR_INT_RK4 = 3 };
enum { MYR_MEM_HOST = 0, MYR_MEM_DEVICE = 1 };
/* per-instance solve status */
enum { MYR_STATUS_CONVERGED = 0, MYR_STATUS_MAXITER = 1, MYR_STATUS_NAN = 2, MYR_STATUS_STALLED = 3,
       /* never written by myr_solve itself: the verdict of the host's elastic phase ("Elastic mode") on an instance
          whose elastic twin converges with a slack that does not vanish as its penalty grows */
       MYR_STATUS_INFEASIBLE = 4 };
/* kernel ids for myr_kernel_time */
enum { MYR_K_EVAL = 0, MYR_K_SOLVE = 1, MYR_K_ROLLOUT = 2, MYR_K_RESID = 3, MYR_K_PROD = 4, MYR_K_FBSM = 5, MYR_K_COUNT = 6 };
/* error codes */
enum { MYR_OK = 0, MYR_E_ARG = -1, MYR_E_UNSUPPORTED = -2, MYR_E_HIP = -3, MYR_E_CAPACITY = -4 };

typedef struct myr_handle_s* myr_handle;

typedef struct {
  int32_t system_id;             /* MYR_SYS_*                                         */
  int32_t transcription;         /* MYR_TR_*                                          */
  int32_t integration_method;    /* MYR_INT_* (shooting + rollout)                    */
  int32_t intervals;             /* hp.intervals                                      */
  int32_t controls_per_interval; /* hp.controls_per_interval (1 for collocation)      */
  int32_t device;                /* HIP device ordinal                                */
  int32_t max_batch;             /* HINT: batch the handle pre-sizes its scratch for; larger batches are accepted
                                    (scratch grows on demand, MYR_E_HIP if the device cannot hold it)          */
  int32_t reserved;
  double  T;                     /* horizon system.T                                  */
} myr_problem_desc;

typedef struct {
  int32_t n;        /* decision variables per instance                                  */
  int32_t m;        /* equality constraints per instance                                */
  int32_t ns, nu, np;
  int32_t x_rows, u_rows;   /* rows of the unravelled (xs, us)                          */
  int32_t jblk;     /* doubles of stage-block Jacobian per instance (see myr_eval)      */
  int32_t ngrad;    /* doubles of objective-gradient output per instance (see myr_eval) */
  int32_t reserved;
} myr_dims;

typedef struct {
  int32_t max_iter;     /* hp.max_iter (config.py:70): bounds the iterations of ONE attempt   */
  int32_t restarts;     /* attempts after a first one that ends without a KKT point: -1 = the library's default (2 for the
                           shooting wavefront kernel, 0 for every other kernel), 0 = a single attempt of at most max_iter
                           iterations -- `iters` <= max_iter then holds --, k > 0 = up to k more (at most 4).  With restarts
                           the first attempt is cut at max(100, max_iter / 8) iterations, a restart takes the caller's point
                           again with another initial barrier parameter (mu_init x 3, / 3, x 9, / 9: a different one per attempt) and the full max_iter, and
                           `iters` reports the sum over the attempts.  Only the shooting wavefront kernel restarts; the lane
                           kernels (MYRIAD_SOLVE_MODE=lane, iterates too large for LDS) ignore the field. */
  double  tol_feas;     /* converged: max|c| <= tol_feas                  (default 1e-8)  */
  double  tol_stat;     /* converged: scaled ||grad f + J^T lam - zL + zU||_inf <= tol_stat (1e-6) */
  double  tol_compl;    /* converged: complementarity <= tol_compl        (default 1e-7)  */
  double  mu_init;      /* initial barrier parameter                      (default 0.1)   */
  int32_t restoration;  /* what stands in for IPOPT's feasibility-restoration phase (the reference gets it from inside the one
                           minimize_ipopt call, nlp_solvers/__init__.py:57-58), applied INSIDE myr_solve / myr_solve_x0 to the
                           instances the first attempt leaves without a KKT point -- a bit mask:
                             1  elastic phase: the instance is solved on the system's ELASTIC TWIN (x' = f(x,u) + s, cost
                                g + rho/2 |s|^2; systems listed above as *_ELASTIC, collocation transcriptions) for rho = 1,
                                1e2, 1e4, each from the previous solution, and the twin's trajectory starts the problem itself;
                                a twin that converges with a slack that neither vanishes (> 1e-3) nor shrinks (> 1/10 of its
                                value at the previous rho) marks the instance MYR_STATUS_INFEASIBLE;
                             2  second starts: excitation guesses -- controls u(t) = centre + 0.95 amp sin(2 pi c t / T) over
                                the control bounds, states by a rollout of the true dynamics -- for c = 2, 3, 5 cycles, first
                                success wins;
                           0 = one attempt from the caller's point and nothing else (the reference's call, minus IPOPT's own
                           restoration); -1 = the library's default (3; the environment variables MYRIAD_ELASTIC=0 /
                           MYRIAD_SECOND_STARTS=0 clear the bits, MYRIAD_SECOND_STARTS="2,7" sets other cycle counts).
                           `iters` then sums the attempts of an instance; myr_solve_info reports which start produced it. */
  int32_t park_iter;    /* two-phase launch of the one-wavefront collocation kernel (a scheduling matter: the iterates of an instance
                           are the same, bit for bit).  A batch much larger than the number of resident wavefronts runs as several
                           whole solves per wavefront and ends with the wavefronts that drew the long solves last; instead every
                           instance first gets `park_iter` iterations, the unfinished ones are parked (40 KB each) and resumed
                           longest-first by their residuals at that point.  0 = the library decides (Hermite-Simpson, closed-form systems: 8 / 10 /
                           12 iterations from 1.5 / 2 / 3 solves per resident wavefront on; whole solves otherwise; MYRIAD_PARK_ITER overrides), -1 = whole
                           solves, k > 0 = k iterations (an explicit k > 0 also selects the one-wavefront form for small batches, which the
                           library would otherwise give two wavefronts per trajectory: the two-wavefront form has no parking).  Needs the
                           per-instance `status` and `kkt` outputs; ignored by the other kernels. */
} myr_solve_opts;

int myr_create(const myr_problem_desc* desc, myr_handle* out);
int myr_destroy(myr_handle h);
int myr_get_dims(myr_handle h, myr_dims* out);
void myr_default_solve_opts(myr_solve_opts* o);

/*
 * Evaluate the transcription at B decision vectors: replaces the four jitted callbacks of
 * nlp_solvers/__init__.py:32-40 for B instances at once.
 *   z      [B][n]      in
 *   params [B][np] if params_stride==np, or [np] shared if params_stride==0; NULL = system defaults
 *   f      [B]         out  objective
 *   gradf  [B][ngrad]  out  d f / d z.  If the system's running cost does not depend on x (CARTPOLE)
 *                           only the control part is stored (ngrad = u_rows*nu, the non-zeros);
 *                           otherwise ngrad = n in z layout.
 *   c      [B][m]      out  constraints, reference row order
 *   jblk   [B][jblk]   out  constraint Jacobian as stage blocks.  HERMITE_SIMPSON: per interval k,
 *                           row-major blocks  Dxs,Dxm,Dxe (ns x ns), Dus,Dum,Due (ns x nu)  [defect rows]
 *                           then              Ixs,Ixe (ns x ns), Ius,Iue (ns x nu)           [interp rows];
 *                           the identity d interp / d x_m is implied and not stored
 *                           (jblk = N*(5 ns^2 + 5 ns nu)).
 *                           TRAPEZOIDAL: per interval  Cxs = h/2 A_s + I, Cxe = h/2 A_e - I (ns x ns),
 *                           Cus = h/2 B_s, Cue = h/2 B_e (ns x nu)   (jblk = N*(2 ns^2 + 2 ns nu)).
 *                           SHOOTING: per interval  Jx = d c_k/d x_k (ns x ns), then Ju = d c_k / d (the interval's control
 *                           rows) (ns x (mc cpi + 1) nu, mc = 2 for RK4 else 1); d c_k/d x_{k+1} = -I implied;
 *                           gradf is the full gradient (ngrad = n).
 * Any output pointer may be NULL to skip it.
 */
int myr_eval(myr_handle h, int32_t B, const double* z, const double* params, int32_t params_stride,
             double* f, double* gradf, double* c, double* jblk, int32_t mem);

/*
 * Solve B independent NLP instances (replaces the minimize_ipopt call, nlp_solvers/__init__.py:57-58).
 *   z      [B][n]  in: initial guess (reference rule: linspace(x0,xT) / zeros)   out: solution
 *   lb,ub  [B][n]  bounds per instance (lb==ub pins a variable, e.g. x[0]=x0, x[-1]=x_T;
 *                  +-inf allowed)
 *   lam    [B][m]  out: equality multipliers, sign convention of scipy/ipopt `mult_g`
 *                  (stationarity: grad f + J^T lam - zL + zU = 0)
 *   cost   [B]     out: objective at the solution
 *   status [B], iters [B]  out (int32): MYR_STATUS_* per instance (non-convergence is a status, never an error return);
 *                  iterations spent.  SHOOTING on the wavefront kernel gives the first start max(100, max_iter / 8) iterations
 *                  and restarts a solve that ended without a KKT point from the caller's point with another initial barrier
 *                  parameter (x3, then /3, each with the full max_iter): `iters` sums the attempts and may exceed max_iter.
 *   kkt    [B][3]  out (may be NULL): final {max|c|, stationarity, complementarity}
 * Restoration (myr_solve_opts.restoration, on by default): instances the first attempt leaves without a KKT point go through the
 * elastic phase and the second starts described at the field, inside this call; a converged instance is never touched.
 */
int myr_solve(myr_handle h, int32_t B, double* z, const double* lb, const double* ub,
              const double* params, int32_t params_stride, const myr_solve_opts* opts,
              double* lam, double* cost, int32_t* status, int32_t* iters, double* kkt, int32_t mem);

/*
 * Per-instance account of the LAST myr_solve / myr_solve_x0 call on this handle (HOST arrays of B int32 each, any may be NULL):
 *   start    0 = the caller's point produced the returned result, c > 0 = the excitation guess with c cycles
 *   attempts device solves the instance went through (1 = the first attempt only; the elastic phase counts 4: three twin solves
 *            and the solve of the problem itself)
 *   restored 1 = the returned result comes out of the elastic phase
 * B must be the batch size of that call.
 */
int myr_solve_info(myr_handle h, int32_t B, int32_t* start, int32_t* attempts, int32_t* restored);

/*
 * How the library ran the FIRST attempt of the last myr_solve / myr_solve_x0 call on this handle (a scheduling matter: no entry changes a result).
 * One source of truth for what a measurement says about its own launch (bench.py used to restate the library's rules).  plan[8]:
 *   [0] kernel form: 0 one trajectory per lane, 1 fused-phase wavefront kernel, 2 round-2 wavefront kernel, 3 shooting wavefront kernel
 *   [1] wavefronts per trajectory            [2] iterations of phase 1 of the two-phase launch (0 = whole solves)
 *   [3] solver-kernel launches per solve     [4] resident trajectory slots of the launch
 *   [5] helper workgroups per trajectory at most (network kernel; 0 = none)     [6], [7] reserved (0)
 */
int myr_solve_plan(myr_handle h, int32_t* plan);

/*
 * myr_solve for B instances that differ in their START STATE only (EXTENSION; the reference builds guess and bounds of an
 * instance from system.x_0 in the optimiser's constructor: collocation/hermite_simpson.py:37-48 (guess), :55-81 (bounds);
 * collocation/trapezoidal.py:36-50, 55-77; shooting.py:56-74, 247-275 -- this entry point applies that constructor to B start
 * states ON THE DEVICE, so a caller sends B x ns doubles instead of three [B][n] arrays).
 *   x0s    [B][ns]  start states
 *   g0,g1  [n]      guess rule: z0[b][i] = g0[i] + g1[i] * x0s[b][i mod ns] for the state entries (i < x_rows*ns: the product and
 *                   the sum are rounded separately, i.e. bit for bit numpy's x0*(1-lin) + x_T*lin with g1 = 1-lin, g0 = x_T*lin),
 *                   z0[b][i] = g0[i] for the controls.  A guess that does not depend on x0 (ones*0.1) has g1 = 0.
 *   lb,ub  [n]      bounds shared by the instances; the first point's state entries (i < ns) are replaced by x0s[b] (x[0] = x0)
 *   z      [B][n]   out: solutions (NOT read on entry)
 *   params, opts, lam, cost, status, iters, kkt, mem: as myr_solve.  MYR_MEM_DEVICE: every pointer is a device pointer.
 * Results are those of myr_solve on the expanded arrays (tests/test_gpu_solve.py::test_solve_x0_matches_solve).
 */
int myr_solve_x0(myr_handle h, int32_t B, const double* x0s, const double* g0, const double* g1, const double* lb,
                 const double* ub, const double* params, int32_t params_stride, const myr_solve_opts* opts,
                 double* z, double* lam, double* cost, int32_t* status, int32_t* iters, double* kkt, int32_t mem);

/*
 * Variable scaling of the SOLVE path.  scale [ns+nu] (states, then controls; NULL = all 1): myr_solve then works on
 * z/s, lb/s, ub/s with the dynamics f(s x)/s -- the same optimisation problem in better-conditioned variables (IPOPT's
 * user scaling, which the reference reaches through `nlp_scaling_method`); inputs and outputs of myr_solve stay in the
 * ORIGINAL variables (z, lam) except kkt[][0], the constraint violation, which is measured in scaled states.
 * Not available for NODE systems.  Power-of-two scales make the change of variables exact in floating point.
 */
int myr_set_var_scale(myr_handle h, const double* scale);

/*
 * True-dynamics rollout under given controls (utils.py:258-298) + terminal state.
 *   x0 [B][ns], us [B][u_rows_rollout][nu]  ->  xs [B][num_steps+1][ns] (may be NULL), cost [B]
 *   u_rows_rollout = (RK4 ? 2 : 1)*num_steps + 1 consumed entries (reference indexing utils.py:57-65).
 */
int myr_rollout(myr_handle h, int32_t B, int32_t num_steps, int32_t u_rows, const double* x0, const double* us,
                const double* params, int32_t params_stride, double* xs, double* cost, int32_t mem);

/*
 * Matrix-free products with the constraint Jacobian -- what jax.grad(lagrangian) computes through the dense transcription in
 * nlp_solvers/extra_gradient.py:21-33 and experiments/e2e_sysid.py:113-141.
 *   myr_vjp:  out[B][n] = J(z)^T lam   (+ grad f(z) when add_gradf != 0: the gradient of the Lagrangian in z)
 *   myr_jvp:  out[B][m] = J(z) v
 * z [B][n], lam [B][m], v [B][n]; layouts and row order as in myr_eval.  Collocation: pointwise / intervalwise kernels;
 * SHOOTING (EULER, HEUN, MIDPOINT, RK4 -- whatever hp.integration_method selects, utils.py:91-96, shooting.py:230-241): reverse
 * sweep seeded with lam / forward tangents through the steps (RK4: two control rows per step, u[2i], u[2i+1], u[2i+2]).
 */
int myr_vjp(myr_handle h, int32_t B, const double* z, const double* lam, const double* params, int32_t params_stride,
            double* out, int32_t add_gradf, int32_t mem);
int myr_jvp(myr_handle h, int32_t B, const double* z, const double* v, const double* params, int32_t params_stride,
            double* out, int32_t mem);

/*
 * `nsteps` extragradient iterations on B instances, iterate resident on the device (extra_gradient.py:25-33):
 *   x_bar = clip(x - eta_x dL/dx(x, lam), lb, ub);  x_new = clip(x - eta_x dL/dx(x_bar, lam), lb, ub);
 *   lam_new = lam + eta_v c(x_new)
 * z [B][n] and lam [B][m] are updated in place; lb, ub [B][n].
 */
int myr_exgd(myr_handle h, int32_t B, double* z, double* lam, const double* lb, const double* ub, const double* params,
             int32_t params_stride, double eta_x, double eta_v, int32_t nsteps, int32_t mem);

/*
 * Batched Forward-Backward Sweep, the reference's indirect solver (trajectory_optimizers/forward_backward_sweep.py:20-116;
 * RK4 sweeps utils.py:138-197; stopping rule trajectory_optimizers/base.py:128-141) for B instances of the handle's
 * system.  Built for all fourteen IndirectFHCS systems: SIMPLECASE, CANCERTREATMENT, BACTERIA, BEARPOPULATIONS, BIOREACTOR,
 * EPIDEMICSEIRN, GLUCOSE, HARVEST, HIVTREATMENT, MOULDFUNGICIDE, SIMPLECASEWITHBOUNDS, TIMBERHARVEST, PREDATORPREY, and the
 * discrete-time INVASIVEPLANT (others return MYR_E_UNSUPPORTED).  The secant `sequencesolver` for a terminal state
 * condition (PREDATORPREY) is a host loop over this call with different adj_T.  The handle's transcription is not used.
 * Discrete systems (forward_backward_sweep.py:33-41, utils.py:184-188): N = int(T) unit steps, direct recurrences instead
 * of RK4, `us` is [B][N][nu] (one control row per step) and `params` is required.
 *   N = hp.fbsm_intervals; x0 [B][ns]; adj_T [ns] or NULL (= 0); params as in myr_eval;
 *   clip_lo / clip_hi [nu]: the bounds each control's optim_characterization is clipped with (+-inf = not clipped);
 *   bang: max|bounds[-1]| for the bang-bang characterisations; delta: stopping tolerance (0.001)
 *   xs, adjs [B][N+1][ns], us [B][N+1][nu], sweeps [B] (may be NULL).  Host arrays only.
 */
int myr_fbsm(myr_handle h, int32_t B, int32_t N, const double* x0, const double* adj_T, const double* params,
             int32_t params_stride, const double* clip_lo, const double* clip_hi, double bang, double delta,
             int32_t max_sweeps, double* xs, double* us, double* adjs, int32_t* sweeps, int32_t mem);

/* Average device time (HIP events on the handle's stream) of the launches of one kernel since the last reset. */
int myr_kernel_time(myr_handle h, int32_t kernel_id, double* avg_ms, int32_t* launches);
int myr_kernel_time_reset(myr_handle h);

/* Devices the library can create handles on (hipGetDeviceCount; 0 when there is none or the runtime fails).  The reference
 * has no multi-device code; the host mirror uses it to fan a batch out over the GPUs of a node beneath the
 * unchanged TrajectoryOptimizer API (myriad/trajectory_optimizers/base.py:69-93): one handle and one host thread per device. */
int myr_device_count(void);

const char* myr_last_error(void);
const char* myr_version(void);

/* ABI guard (round 5): the bytes this build of the library reads and writes through a `myr_solve_opts*`, a `myr_problem_desc*` and a
 * `myr_dims*`.  The structs carry no size field and `myr_solve_opts` has grown before (round 4: restoration, park_iter: 40 -> 48 bytes);
 * a binding compares these numbers with the sizes of its OWN declarations once, at load time, and refuses a library that disagrees
 * (myriad_amd/_lib.py: load()) instead of letting the library read past a shorter struct.  `which`: 0 = myr_solve_opts,
 * 1 = myr_problem_desc, 2 = myr_dims; anything else returns -1. */
int32_t myr_abi_sizeof(int32_t which);

#ifdef __cplusplus
}
#endif
#endif
