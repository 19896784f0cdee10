/*
 * ccal.h -- C ABI of the MI355X-native reprojection residual + Jacobian engine
 * and Gauss-Newton / Levenberg-Marquardt normal-equation builder.
 *
 * This is the drop-in boundary for the hot path of
 * powei-lin/camera-intrinsic-calibration-rs (paths below are relative to the
 * reference tree).  A Rust caller binds these symbols with `extern "C"`
 * (INTEGRATION.md shows the stub); everything is plain pointers + sizes, no
 * C++ types, no exceptions, no torch types.  All floating point is f64 except
 * the detected-corner inputs, which are f32 exactly as the reference holds
 * them (src/detected_points.rs:6-9, widened at src/optimization/factors.rs:141-143).
 *
 * What each entry point replaces:
 *   ccal_problem_create      the `problem.add_residual_block(2, [...], ReprojectionFactor|
 *                            OtherCamReprojectionFactor, HuberLoss(1.0))` loops
 *                            src/util.rs:401-414 and src/util.rs:595-631
 *   ccal_set_bounds          tiny_solver::Problem::set_variable_bounds   src/util.rs:36-47
 *   ccal_fix_param           tiny_solver::Problem::fix_variable          src/util.rs:61,461,666
 *   ccal_apply_reference_bounds / ccal_disable_distortions
 *                            set_problem_parameter_bound / _disabled     src/util.rs:29-71
 *   ccal_eval                Factor::residual_func evaluated with dual numbers for every
 *                            block (src/optimization/factors.rs:152-173, 204-228)
 *   ccal_build_normal        tiny-solver's J^T J / J^T r assembly (call sites src/util.rs:455,670),
 *                            with the per-frame pose blocks eliminated exactly (Schur)
 *   ccal_solve / _dev        GaussNewtonOptimizer::optimize(&problem, &initial_values, None)
 *                            src/util.rs:443-463 and 668-670 (+ an LM mode)
 *   ccal_init_poses          the unproject + sqpnp pose initialisation inside calib_camera  src/util.rs:418-436
 *   ccal_reprojection_errors / ccal_validation
 *                            validation()                                 src/util.rs:721-795
 *
 * Parameter layout.
 *   intr   [n_cams][CCAL_PMAX]  FULL model parameters [fx,fy,cx,cy,dist...] per camera, like
 *                               GenericModel::params().  With xy_same_focal the solver variable is
 *                               f = intr[c][0] and fy is ignored on input / set to f on output
 *                               (src/util.rs:391-395, 467-470).
 *   poses  [n_slots][6]         rvec(3), tvec(3) of T_cam0_board per frame slot
 *                               ("rvec{i}","tvec{i}" / "rvec_0_b_{f}","tvec_0_b_{f}").
 *   extr   [n_cams][6]          rvec, tvec of T_cam_i_cam0 ("rvec_{c}_0","tvec_{c}_0"); row 0 unused.
 * Solver-visible intrinsic index space ("eff" indices): full index with fy removed when
 * xy_same_focal (the `shift` of src/util.rs:35).  Bounds / fixed flags use eff indices, exactly
 * like the reference's set_variable_bounds / fix_variable calls.
 *
 * Jacobian column order of one block (ccal_eval): [theta_c (P_eff) | rvec_0_b | tvec_0_b] for
 * camera 0 blocks and [theta_c | rvec_0_b | tvec_0_b | rvec_c_0 | tvec_c_0] for camera c>0
 * (src/util.rs:411, 621-627).
 * Reduced ("camera") system column order (ccal_build_normal): for c = 0..n_cams-1:
 * theta_c (P_eff(c)), then for c>0 rvec_c_0, tvec_c_0.  K = sum P_eff + 6 (n_cams-1).
 *
 * Environment.  The behaviour contract of this ABI does not depend on the environment.  libccal_hip.so reads exactly TWO
 * variables: CCAL_RCCL_LIB (the RCCL library to dlopen when none is loaded in the process yet) and CCAL_MULTI_TRANSPORT
 * (inproc | rccl: the transport ccal_multi_create picks when the caller leaves the choice to it).  Neither changes a result.
 * The A/B developer switches and the test hooks live in a second library that only the parity tests load
 * (libccal_hip_legacy.so); a binding sets nothing.
 */
#ifndef CCAL_H
#define CCAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCAL_PMAX 10          /* row stride of intr / bounds arrays (max model has 9 params) */
#define CCAL_MAX_CAMS 8
#define CCAL_KMAX 128         /* the reduced camera system has fewer columns than this: eight cameras of any model fit (8 x 9 + 7 x 6 = 114) */

typedef enum {
    CCAL_OK = 0,
    CCAL_ERR_INVALID_ARG = 1,
    CCAL_ERR_HIP = 2,
    CCAL_ERR_NONFINITE = 3,       /* NaN/Inf cost: tiny-solver returns None (src/util.rs:455-457) */
    CCAL_ERR_NOT_PD = 4,          /* Cholesky of the normal equations failed: None */
    CCAL_ERR_NO_CONVERGENCE = 5,  /* informational: max_iterations reached (reference still returns Some) */
    CCAL_ERR_UNSUPPORTED = 6,
    CCAL_ERR_NO_MEMORY = 7        /* host allocation failed inside the library (never thrown across the ABI) */
} ccal_status;

typedef enum {                    /* camera-intrinsic-model GenericModel variants on the hot path */
    CCAL_MODEL_UCM = 0,           /* [fx,fy,cx,cy,alpha]              tests/optimization_test.rs:41 */
    CCAL_MODEL_EUCM = 1,          /* [fx,fy,cx,cy,alpha,beta]         data/eucm.json              */
    CCAL_MODEL_KB4 = 2,           /* [fx,fy,cx,cy,k1,k2,k3,k4]                                     */
    CCAL_MODEL_OPENCV5 = 3,       /* [fx,fy,cx,cy,k1,k2,p1,p2,k3]                                  */
    CCAL_MODEL_EUCMT = 4          /* [fx,fy,cx,cy,alpha,beta,t1,t2]: PARAMETER CONTAINER ONLY - the target of the closed-form
                                     UCM -> EUCMT conversion (src/util.rs:236-243); its projection exists only in the absent
                                     crate, so problems with it are refused with CCAL_ERR_UNSUPPORTED */
} ccal_model;

/* What this build assumes about the crate camera-intrinsic-model 0.8 (Cargo.toml:25; its source is not part of the
 * reference tree) and can be changed at run time, per context, without rebuilding a kernel.  Defaults:
 * camera_intrinsic_calibration_rs_amd/csrc/ccal_models.hpp. */
typedef struct {
    double kb4_small_radius;       /* KB4 project_one: sqrt(x^2+y^2) <= this -> pinhole limit.  Default 1e-8 */
    double dist_lo[4][5];          /* [ccal_model][i]: lower / upper bound of distortion parameter i in the order of the     */
    double dist_hi[4][5];          /* model rows of ccal_model above (OPENCV5: k1, k2, p1, p2, k3 whatever ocv5_order says), */
                                   /* i.e. what distortion_params_bound() returns (applied at src/util.rs:40-48)             */
    double unproject_small_radius; /* unprojection (pose initialisation, convert_model's rays): image-plane radius below which
                                      the ray is the optical axis.  Default 1e-8 */
    int32_t ocv5_order[5];         /* WHERE k1, k2, p1, p2, k3 sit among the five distortion parameters of an OPENCV5 params()
                                      vector: parameter 4 + ocv5_order[i] is k1 (i = 0), k2, p1, p2, k3 (i = 4).  Default
                                      {0, 1, 2, 3, 4} = OpenCV's own order.  Every array of this ABI that follows the parameter
                                      order (intrinsics in / out, eff indices of bounds and fixed parameters, Jacobian and
                                      normal-equation columns, "the last k distortion parameters") follows THIS order */
    int32_t reserved_;
} ccal_model_conventions;

typedef struct ccal_ctx ccal_ctx;          /* one GPU + one HIP stream; single caller */
typedef struct ccal_problem ccal_problem;  /* inputs resident in HBM + workspaces */

/* Problem description = the reference's calib-frame inputs flattened.
 * An "observation frame" is one (camera, frame slot) pair with its detected corners stored
 * contiguously: corners [obs_offsets[i], obs_offsets[i+1]) of the SoA arrays.  Single camera:
 * n_obs == n_slots, obs_cam[i] == 0, obs_slot[i] == i. */
typedef struct {
    int32_t n_cams;
    const int32_t* model;          /* [n_cams] ccal_model */
    const double* width;           /* [n_cams] image width  (bounds, src/util.rs:38) */
    const double* height;          /* [n_cams] image height (bounds, src/util.rs:39) */
    int32_t xy_same_focal;         /* ReprojectionFactor::xy_same_focal */
    int32_t n_slots;               /* board-pose slots (frames) */
    int32_t n_obs;                 /* observation frames */
    const int32_t* obs_cam;        /* [n_obs] */
    const int32_t* obs_slot;       /* [n_obs] */
    const int64_t* obs_offsets;    /* [n_obs+1] */
    const float* p3d_x;            /* [n_corners] board point (FeaturePoint::p3d), f32 */
    const float* p3d_y;
    const float* p3d_z;
    const float* p2d_u;            /* [n_corners] detected pixel (FeaturePoint::p2d), f32 */
    const float* p2d_v;
    double huber_delta;            /* HuberLoss::new(1.0) in the reference; <= 0 disables the loss */
} ccal_problem_desc;

typedef enum { CCAL_METHOD_GN = 0, CCAL_METHOD_LM = 1 } ccal_method;

typedef struct {
    int32_t method;                /* CCAL_METHOD_GN is the reference's optimizer (parity mode) */
    int32_t max_iterations;        /* tiny-solver default 100 */
    double min_abs_error_decrease; /* 1e-5 */
    double min_rel_error_decrease; /* 1e-5 */
    double min_error;              /* 1e-10 */
    double lm_initial_radius;      /* 1e4   (LM only; Ceres-style trust region, lambda = 1/radius) */
    double lm_min_diagonal;        /* 1e-6 */
    double lm_max_diagonal;        /* 1e32 */
    int32_t verbose;
    int32_t timeout_s;             /* host-side watchdog of the device-resident loop, seconds without a step completing.
                                      0 = default: 30 s on a single GPU, 600 s in a sharded solve (a communicator or an
                                      all-reduce callback is set): a step that waits for a peer - still uploading, still
                                      setting up RCCL channels - is waited for, a peer that died does not hang the
                                      survivors for ever.  Give every rank the same value: a rank that gives up
                                      (CCAL_ERR_HIP, "timed out") leaves the shared sequence of collectives while its peers
                                      are inside ncclAllReduce - see "Errors" below: exit or abort the communicator, never
                                      re-exec a process that has touched the GPU */
    int32_t error_metric;          /* what the stop rules above call "error" - a guess about the absent crate made switchable
                                      (tiny-solver 0.18, Optimizer::compute_error: squared_norm_l2() or norm_l2() of the
                                      loss-corrected residuals; call sites src/util.rs:443,455).  CCAL_ERROR_SQUARED_NORM (0,
                                      the default): sum rho'(s) s, what ccal_report::initial_cost / final_cost hold.
                                      CCAL_ERROR_NORM (1): its square root - min_error and the absolute / relative decreases are
                                      then compared on sqrt(cost); the reports still hold the squared norm */
    int32_t reserved_;             /* 0 */
} ccal_solver_opts;
typedef enum { CCAL_ERROR_SQUARED_NORM = 0, CCAL_ERROR_NORM = 1 } ccal_error_metric;

typedef struct {
    int32_t status;                /* ccal_status of the solve */
    int32_t iterations;            /* linear solves performed */
    int32_t lm_accepted;
    int32_t lm_rejected;
    double initial_cost;           /* sum over blocks of rho'(s) * s at the start (tiny-solver's "error") */
    double final_cost;
    double solve_ms;               /* wall time of the iteration loop */
    int32_t lm_spec_hits;          /* LM: accepted steps whose speculative elimination was the next system (one group) */
    int32_t lm_spec_misses;        /*     ... and those that needed a re-elimination group                              */
} ccal_report;

/* Frame-sharded multi-GPU solves (one process per GPU, every rank holds a contiguous range of frame slots and the
 * same camera parameters): ONE in-place sum of a small device buffer ([reduced system | cost | model decrease | failed
 * blocks], 100..400 doubles) over all ranks per optimizer step, Gauss-Newton and Levenberg-Marquardt alike.
 *   ccal_set_rccl_comm   the library calls ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, comm, stream) itself,
 *                        stream-ordered, and keeps enqueueing steps ahead of the host.  The production path.
 *   ccal_set_allreduce   a callback does the sum (ordered on `hip_stream`) - for transports other than RCCL
 *                        (the tests use gloo); the loop then waits for every step before it enqueues the next.
 * Neither set = single GPU.
 * Errors: every rank issues the same sequence of collectives by construction (one per step, decisions taken from
 * all-reduced sums).  If a sharded ccal_solve nevertheless returns an error other than the solver's own verdicts
 * (NONFINITE / NOT_PD / NO_CONVERGENCE, which all ranks reach together), collectives of this rank may still be queued
 * with no partner: treat the communicator as INVALID - abort it (ncclCommAbort) or exit the process; do not reuse it and
 * do not wait for the stream. */
typedef int (*ccal_allreduce_fn)(void* user, double* device_buf, size_t count, void* hip_stream);
#define CCAL_RCCL_UNIQUE_ID_BYTES 128

/* ---- context ----------------------------------------------------------------------------
 * A problem holds its context: ccal_ctx_destroy with problems still alive is deferred to the last ccal_problem_destroy
 * (any destruction order is safe); a caller-provided stream must stay alive while work is enqueued on it. */
int ccal_ctx_create(int device_id, void* hip_stream /* hipStream_t or NULL = own stream */, ccal_ctx** out);
void ccal_ctx_destroy(ccal_ctx* ctx);
const char* ccal_last_error(const ccal_ctx* ctx);
const char* ccal_version(void);
int ccal_model_num_params(int model);            /* 5 / 6 / 8 / 9 (/ 8 for the EUCMT container), -1 if unknown */
/* Conventions of a context: read them, change a field, set them (in == NULL restores the defaults).  They apply to
 * everything the context evaluates afterwards (kernels receive the threshold as an argument) and to later
 * ccal_apply_reference_bounds / ccal_convert_model calls. */
int ccal_get_model_conventions(const ccal_ctx* ctx, ccal_model_conventions* out);
int ccal_set_model_conventions(ccal_ctx* ctx, const ccal_model_conventions* in);

/* Host memory the caller hands to ccal_solve* / ccal_upload_params / ccal_download_params can be PINNED once (page-locked and
 * mapped for the GPU): the library then moves it without its intermediate staging copy - the kernels read the starting poses where
 * the caller keeps them, the result is written there by one DMA (10 000 frames: ~85 us of a 240-us ccal_solve are the two staging
 * copies of 480 KB of poses; src/util.rs:384-390 hands such a map over per call).  ccal_pin_buffer registers [host_ptr, host_ptr +
 * bytes) (hipHostRegister; memory the caller allocated pinned itself - hipHostMalloc - is recognised without it); the range must
 * stay allocated until ccal_unpin_buffer or ccal_ctx_destroy.  Pinning never changes a result (same kernels, same bits). */
int ccal_pin_buffer(ccal_ctx* ctx, void* host_ptr, size_t bytes);
int ccal_unpin_buffer(ccal_ctx* ctx, void* host_ptr);

/* ---- problem ---------------------------------------------------------------------------- */
/* One problem holds at most 2^30 - 1 corners (CCAL_ERR_INVALID_ARG beyond: shard the frames, ccal_multi_problem_create). */
int ccal_problem_create(ccal_ctx* ctx, const ccal_problem_desc* desc, ccal_problem** out);
void ccal_problem_destroy(ccal_problem* p);
int ccal_set_defaults(ccal_solver_opts* opts);   /* tiny-solver OptimizerOptions::default() */
int ccal_set_bounds(ccal_problem* p, int cam, int eff_idx, double lo, double hi);
int ccal_clear_bounds(ccal_problem* p, int cam, int eff_idx);
int ccal_fix_param(ccal_problem* p, int cam, int eff_idx);
int ccal_unfix_param(ccal_problem* p, int cam, int eff_idx);
int ccal_apply_reference_bounds(ccal_problem* p);                 /* src/util.rs:29-49 for every camera */
int ccal_disable_distortions(ccal_problem* p, int n_disabled, double* intr_io /* zeroed in place */);
int ccal_set_allreduce(ccal_problem* p, ccal_allreduce_fn fn, void* user);
int ccal_set_rccl_comm(ccal_problem* p, void* nccl_comm /* ncclComm_t created by the host or by ccal_rccl_comm_create; NULL = none */);
/* Communicator bootstrap for hosts that do not link RCCL themselves: rank 0 draws an id, hands its 128 bytes to every
 * rank by any means (file, socket, MPI, the launcher's store), every rank creates its communicator on its context's GPU.
 * RCCL is resolved at run time (the instance already loaded into the process, else librccl.so.1). */
int ccal_rccl_available(void);                                    /* 1 if an RCCL library could be resolved */
int ccal_rccl_version(void);                                      /* ncclGetVersion, 0 if unavailable */
int ccal_rccl_unique_id(void* id_out /* CCAL_RCCL_UNIQUE_ID_BYTES */);
int ccal_rccl_comm_create(ccal_ctx* ctx, int world, int rank, const void* id /* 128 bytes */, void** comm_out);
int ccal_rccl_comm_destroy(void* comm);
int ccal_rccl_comm_count(void* comm);                             /* ncclCommCount: ranks of the communicator, -1 on error */

int64_t ccal_num_corners(const ccal_problem* p);
int ccal_reduced_dim(const ccal_problem* p);                      /* K */
int ccal_block_dim(const ccal_problem* p, int cam);               /* D of that camera's blocks */
int ccal_eff_num_params(const ccal_problem* p, int cam);          /* P_eff */
int64_t ccal_jacobian_len(const ccal_problem* p);                 /* doubles in J_out of ccal_eval */

/* ---- mode E: residual + Jacobian of every block ------------------------------------------
 * r_out [n_corners][2]; J_out: per observation frame, [n_i][2][D_cam] row-major blocks, frames
 * concatenated (offset of frame i = sum_{j<i} n_j * 2 * D_cam(j)).  apply_loss != 0 applies the
 * Huber corrector (r and J scaled by sqrt(rho')) the way tiny-solver does before assembly.
 * *_dev variants take device pointers and only enqueue work on the context stream. */
int ccal_eval(ccal_problem* p, const double* intr, const double* poses, const double* extr,
              int apply_loss, double* r_out, double* J_out);
int ccal_upload_params(ccal_problem* p, const double* intr, const double* poses, const double* extr);
int ccal_download_params(ccal_problem* p, double* intr, double* poses, double* extr);
int ccal_eval_dev(ccal_problem* p, int apply_loss, double* r_out_dev, double* J_out_dev);
int ccal_sync(ccal_ctx* ctx);

/* ---- mode N: fused normal equations + exact per-frame Schur complement -------------------
 * S [K][K] (row-major, symmetric, full), b [K] = reduced J^T r (so S dx = -b), cost = sum rho' s.
 * lambda > 0 adds Marquardt damping lambda * clamp(diag) to every block before elimination. */
int ccal_build_normal(ccal_problem* p, const double* intr, const double* poses, const double* extr,
                      double lambda, double* S, double* b, double* cost);
int ccal_build_normal_dev(ccal_problem* p, double lambda);   /* uses uploaded params; result stays on device */

/* ---- the optimizer loop --------------------------------------------------------------------
 * ccal_solve: host pointers in and out (poses staged through pinned memory both ways - unless the caller's arrays are pinned
 * themselves, ccal_pin_buffer: then they are read and written in place).
 * ccal_solve_dev: the starting point is what ccal_upload_params (or a previous solve) left on the device, the result
 * stays there (ccal_download_params fetches it); the call itself moves ~1 KB to the device and polls a status word. */
int ccal_solve(ccal_problem* p, const ccal_solver_opts* opts,
               double* intr_io, double* poses_io, double* extr_io, ccal_report* report);
int ccal_solve_dev(ccal_problem* p, const ccal_solver_opts* opts, ccal_report* report);
/* n INDEPENDENT problems solved side by side in ONE call (the per-camera calib_camera calls of a rig, the retries of
 * src/bin/camera_calibration.rs:205-246, many sessions of a service): a session-sized problem (a few hundred frames) leaves
 * the GPU almost idle - one latency-bound launch per optimizer step.  Session-sized single-camera problems of one
 * GPU, model and focal mode advance in LOCKSTEP: ONE launch per step serves all of them (the problems' workgroups side by side), so
 * a batch of eight costs the host what one solve costs.  Everything else (rigs, large problems) is driven per context by a
 * host thread of its own (create such problems on contexts of their own - ccal_ctx_create with stream NULL - to make them overlap;
 * those that share a context are solved one after the other).  A problem's launches are sized for its share of the GPU (the batch's
 * problems on that device): fewer, longer wavefronts than a lone ccal_solve takes - results equal to n ccal_solve calls up to the
 * order of summation (<= 1e-11 relative; the lane mapping, hence the order, depends on how many problems of the batch share the
 * GPU: the same problem in another batch may differ in the last bits); verdict and iteration count can differ from a lone
 * ccal_solve only where a stop threshold is met to within that rounding.  A problem that needs bit-reproducible results
 * whatever runs beside it is solved with ccal_solve.  intr_io == NULL: device-resident like ccal_solve_dev (poses_io / extr_io ignored);
 * else intr_io[i] / poses_io[i] / extr_io[i] as in ccal_solve.  Every problem's verdict goes to reports[i].status; the return
 * value is CCAL_OK unless a call failed for another reason (then the first such code).  Sharded problems are refused. */
int ccal_solve_batch(ccal_problem** problems, int n, const ccal_solver_opts* opts,
                     double** intr_io, double** poses_io, double** extr_io, ccal_report* reports);

/* ---- one process, several GPUs ------------------------------------------------------------------
 * The reference is ONE process: calib_camera is one blocking call of the tool's main (src/util.rs:384-390,
 * src/bin/camera_calibration.rs:70).  These entry points keep it that way on a multi-GPU node: the library shards the
 * frame slots, drives every GPU from a host thread of its own and owns the transport of the step's one all-reduce.
 *
 * ccal_solve_sharded: n shards of ONE problem - problems created by the caller on n DIFFERENT contexts (one per GPU;
 *   contexts on the same GPU work too), each holding a contiguous range of the frame slots with every camera's
 *   observations of those slots, the same cameras, bounds and fixed parameters.  intr_io / extr_io: the shared camera
 *   block (in: starting point, out: result - bit-identical on every shard, checked); poses_io[i]: shard i's poses
 *   [n_slots_i][6].  Transport: whatever is set on ALL shards (ccal_set_rccl_comm / ccal_set_allreduce); none set = the
 *   library's in-process transport for the duration of the call (HIP events order the shards' streams, a kernel adds the
 *   shards' buffers in shard order; needs the shards on one GPU or peer access between their GPUs).  The report is the
 *   solve's (every shard reaches the same verdict and iteration count; solve_ms = the slowest shard).
 * ccal_multi_*: the same with the sharding done by the library.  ccal_multi_create makes one context per listed device
 *   and the transport between them - RCCL (ncclCommInitAll, one communicator per device) when the devices are all
 *   different and RCCL can be resolved, the in-process transport otherwise (a device listed twice: two shards on one GPU).
 *   ccal_multi_problem_create takes the SAME description as ccal_problem_create and cuts it into contiguous slot ranges
 *   balanced by corner count; poses / poses_obs / n_used keep the caller's slot and observation-frame order.
 * Threading: like a ccal_ctx, a ccal_multi (and its problems) is single-caller; the host threads that drive the devices inside a
 * call are the library's own.
 * Errors: as for sharded solves above.  A shard that fails raises a flag its peers see at their next look at the status word: they
 * stop waiting at once (not at the timeout).  After an error other than the solver's verdicts a ccal_multi with the in-process
 * transport recovers by itself; with RCCL its communicators are aborted and every later ccal_multi_solve on it fails
 * (CCAL_ERR_HIP): destroy it and create a new one. */
#define CCAL_MULTI_MAX_DEVICES 16
typedef enum { CCAL_TRANSPORT_NONE = 0 /* one device */, CCAL_TRANSPORT_RCCL = 1, CCAL_TRANSPORT_INPROC = 2 } ccal_transport;
typedef struct ccal_multi ccal_multi;                  /* a set of contexts (one per listed device) + their transport */
typedef struct ccal_multi_problem ccal_multi_problem;  /* one problem, frame slots sharded over the contexts of a ccal_multi */

int ccal_solve_sharded(ccal_problem** shards, int n, const ccal_solver_opts* opts,
                       double* intr_io, double** poses_io, double* extr_io, ccal_report* report);

int ccal_multi_create(const int* device_ids, int n_dev, ccal_multi** out);
/* the same with the transport named: -1 = automatic (ccal_multi_create), CCAL_TRANSPORT_RCCL = communicators even for ONE listed
 * device, CCAL_TRANSPORT_INPROC = the in-process transport even where RCCL is there.  A transport asked for by name is not
 * replaced by another one when it cannot be set up (CCAL_ERR_UNSUPPORTED, reason in ccal_create_last_error). */
int ccal_multi_create_transport(const int* device_ids, int n_dev, int transport, ccal_multi** out);
int ccal_multi_rccl_ranks(const ccal_multi* m);        /* ranks RCCL counts in the set's communicators (ncclCommCount); 0: not RCCL */
/* why the last ccal_ctx_create / ccal_multi_create* ON THIS THREAD failed (no handle exists to ask); "" after a success */
const char* ccal_create_last_error(void);
/* Host-only (no GPU touched): where ccal_multi_problem_create cuts a description into n_shards contiguous frame-slot ranges
 * balanced by corner count - first_out[i] .. first_out[i + 1] are shard i's slots (first_out has n_shards + 1 entries).  A host
 * that runs one process per GPU cuts its problem with the same function (bench.py --frames-total does). */
int ccal_partition_slots(const ccal_problem_desc* desc, int n_shards, int32_t* first_out);
void ccal_multi_destroy(ccal_multi* m);                /* deferred to the last ccal_multi_problem_destroy if problems are alive */
int ccal_multi_num_devices(const ccal_multi* m);
int ccal_multi_transport(const ccal_multi* m);         /* ccal_transport */
ccal_ctx* ccal_multi_ctx(ccal_multi* m, int i);        /* context of device i (ccal_last_error, conventions) */
const char* ccal_multi_last_error(const ccal_multi* m);
int ccal_multi_set_model_conventions(ccal_multi* m, const ccal_model_conventions* in);   /* on every context */
int ccal_multi_sync(ccal_multi* m);

int ccal_multi_problem_create(ccal_multi* m, const ccal_problem_desc* desc, ccal_multi_problem** out);
void ccal_multi_problem_destroy(ccal_multi_problem* mp);
int ccal_multi_problem_num_shards(const ccal_multi_problem* mp);
ccal_problem* ccal_multi_problem_shard(ccal_multi_problem* mp, int i);   /* borrowed: sizes, mode E buffers; do not destroy */
int ccal_multi_problem_slot_range(const ccal_multi_problem* mp, int i, int32_t* first_slot, int32_t* n_slots);
int ccal_multi_set_bounds(ccal_multi_problem* mp, int cam, int eff_idx, double lo, double hi);
int ccal_multi_clear_bounds(ccal_multi_problem* mp, int cam, int eff_idx);
int ccal_multi_fix_param(ccal_multi_problem* mp, int cam, int eff_idx);
int ccal_multi_unfix_param(ccal_multi_problem* mp, int cam, int eff_idx);
int ccal_multi_apply_reference_bounds(ccal_multi_problem* mp);
int ccal_multi_disable_distortions(ccal_multi_problem* mp, int n_disabled, double* intr_io);
int ccal_multi_init_poses(ccal_multi_problem* mp, const double* intr, int min_points, double* poses_obs, int32_t* n_used);
int ccal_multi_upload_params(ccal_multi_problem* mp, const double* intr, const double* poses /* [n_slots][6], all slots */, const double* extr);
int ccal_multi_eval_dev(ccal_multi_problem* mp, int apply_loss, double* const* r_dev, double* const* J_dev);   /* per shard device buffers; no collective */
/* validation() (src/util.rs:721-795) over the shards: the same statistics, bit for bit, as ccal_validation on one GPU */
int ccal_multi_validation(ccal_multi_problem* mp, int cam, const double* intr, const double* poses, const double* extr,
                          double* avg_99_percent, double* median);
int ccal_multi_reprojection_errors(ccal_multi_problem* mp, const double* intr, const double* poses, const double* extr,
                                   double* err_out /* [n_corners], the description's corner order */,
                                   const int64_t* obs_offsets /* the description's obs_offsets [n_obs + 1] */);
int ccal_multi_solve(ccal_multi_problem* mp, const ccal_solver_opts* opts,
                     double* intr_io, double* poses_io /* [n_slots][6], all slots */, double* extr_io, ccal_report* report);

/* ---- per-frame pose initialisation (src/util.rs:418-436) ---------------------------------
 * What calib_camera does before it builds the problem: unproject the detections with the current model,
 * keep the valid ones, divide by z, planar PnP (the reference calls sqpnp_simple; here a plane-induced
 * homography -- same basin, the joint solve refines it).  poses_obs [n_obs][6] = T_cam_board of every
 * observation frame; n_used [n_obs] = corners that entered the estimate, 0 = no pose (the reference skips
 * frames with fewer than 10 valid unprojections: pass min_points = 10). */
int ccal_init_poses(ccal_problem* p, const double* intr, int min_points, double* poses_obs, int32_t* n_used);

/* ---- init_camera_extrinsic (src/util.rs:511-561) ------------------------------------------
 * T_i_0 of camera i from the board poses camera 0 and camera i estimated for the same n_common frames:
 * one SE3Factor (src/optimization/factors.rs:234-272) per frame, HuberLoss(0.5), Gauss-Newton from
 * T_i_b[0] * T_0_b[0]^-1 (or from t_i_0_io when use_initial != 0).  A 6-unknown problem: host code of the
 * library, needs no context.  poses_* are [n_common][6] rvec,tvec. */
int ccal_init_camera_extrinsic(const double* poses_cam0, const double* poses_cami, int n_common,
                               double* t_i_0_io, int use_initial, ccal_report* report);
/* the same with the optimizer's stop rules given (max_iterations, min_error, min_abs / min_rel_error_decrease, error_metric);
 * opts == NULL: the defaults, i.e. the call above */
int ccal_init_camera_extrinsic_opts(const double* poses_cam0, const double* poses_cami, int n_common,
                                    double* t_i_0_io, int use_initial, const ccal_solver_opts* opts, ccal_report* report);
/* One SE3Factor block (src/optimization/factors.rs:248-271): r[6] = the rvec,tvec of T_i_b^-1 * (T_i_0 * T_0_b) and its
 * 6 x 6 Jacobian (row-major) with respect to x = rvec,tvec of T_i_0 - what tiny-solver's dual numbers evaluate to. */
int ccal_se3_factor(const double* pose_0_b, const double* pose_i_b, const double* x, double* r_out, double* J_out);

/* ---- convert_model (src/util.rs:224-282) -----------------------------------------------------
 * Fit the target model to the source model over the reference's pixel grid: ModelConvertFactor
 * (src/optimization/factors.rs:10-76) = ONE residual block source.project(ray) - target.project(ray) over the
 * rays the source model unprojects from rows/cols edge = max(w,h)/100 .. in steps of max(w,h)/30, 10000 where a
 * projection is undefined, HuberLoss(1.0) on the whole block; Gauss-Newton on all target intrinsics with the
 * reference's parameter bounds, the last `disabled_distortions` parameters fixed at 0; UCM -> EUCM is the closed
 * form beta = 1 (util.rs:229-235), UCM -> EUCMT the closed form beta = 1, t1 = t2 = 0 (util.rs:236-243).  tgt_params_io: in = the target's current parameters (its first four are
 * replaced by the source's fx, fy, cx, cy: util.rs:256-258), out = the fitted parameters.  opts NULL = the
 * reference's Gauss-Newton defaults.  Rays and the per-iteration Gram run on the device; the <= 9 x 9 solve
 * on the host. */
int ccal_convert_model(ccal_ctx* ctx, int src_model, const double* src_params, int tgt_model,
                       double* tgt_params_io, double width, double height, int disabled_distortions,
                       const ccal_solver_opts* opts, ccal_report* report);

/* ---- reference validation() statistics (src/util.rs:721-795) ---------------------------- */
int ccal_reprojection_errors(ccal_problem* p, const double* intr, const double* poses, const double* extr,
                             double* err_out /* [n_corners] Euclidean px error */);
int ccal_validation(ccal_problem* p, int cam, const double* intr, const double* poses, const double* extr,
                    double* avg_99_percent, double* median);

#ifdef __cplusplus
}
#endif
#endif /* CCAL_H */
