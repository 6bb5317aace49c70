/*
 * hode.h -- C ABI of libhode.so: batched hybrid-ODE (6-state GLP-1/glucose RHS + MLP residual)
 * integration and adjoint on AMD MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE path of OliverDOU776/Hybrid-ODE-for-GLP-1-and-Glucose.
 * The reference has no FFI of its own (pure Python): each entry point
 * below names the reference Python it replaces.  Citations are relative to the reference root.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller, contiguous row-major, 16-byte aligned
 *     where noted; nothing is allocated, freed or kept by the callee;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is enqueued
 *     on it, nothing synchronises, so calls may be captured into a hipGraph;
 *   - return value: 0 ok, HODE_EINVAL bad argument, HODE_EUNSUPPORTED shape outside the
 *     kernels' compiled range, HODE_ELAUNCH HIP launch error.  Integration failures are NEVER
 *     a return code: they are per-trajectory values in `status[B]`
 *     (reference: models/hybrid_ode_nn.py:243-256 logs a warning and zero-fills);
 *   - flat MLP parameter vector `nn_p` = PyTorch parameters() order of NNResidual.network
 *     (models/nn_residual.py:59-78):  W1[H,9] b1[H] (W[H,H] b[H])x(L-1) Wout[6,H] bout[6];
 *   - `ode_p[17]` = ODECore buffers in registration order (models/ode_core.py:44-71):
 *     a_GI k_I rho G_b I_b E_max EC_50 Glu_b V_max K_m k_L k_GE0 IGD_50 g p_7 p_8 p_9;
 *   - input "mode" of meal / tvns / gd (models/hybrid_ode_nn.py:217-231):
 *     0 absent (=0), 1 constant per patient [B], 2 time-varying on the grid [B,T] (lerp);
 *   - parameter sets: the B trajectories are n_sets equal contiguous groups, group s uses
 *     ode_p + 17*s and nn_p + P*s (n_sets = 1: one shared set; >1: VI samples / Sobol sets,
 *     inference/vi.py:88-100, plots/plot_all.py:171-196).  B % n_sets must be 0.
 */
#ifndef HODE_H
#define HODE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HODE_OK 0
#define HODE_EINVAL (-1)
#define HODE_EUNSUPPORTED (-2)
#define HODE_ELAUNCH (-3)

/* integrator (hybrid_ode_nn.py:174-181 maps names to SciPy; 'rk45' == DP5(4)) */
#define HODE_METHOD_DP54 0 /* adaptive Dormand-Prince 5(4), grid points are step boundaries */
#define HODE_METHOD_RK4 1  /* classic RK4, one step per grid interval (BASELINE config 1)   */

/* per-trajectory status */
#define HODE_ST_OK 0
#define HODE_ST_MAXSTEPS 1  /* accepted-step budget (max_steps) exhausted */
#define HODE_ST_UNDERFLOW 2 /* step size below 10 ulp(t) (scipy rk.py:128-129) */
#define HODE_ST_NONFINITE 3 /* state became non-finite */

/* Network envelope.  NNResidual (models/nn_residual.py:28-98) builds 9 -> H x L -> 6 for any H, L; the reference's configs
 * use 64 x 4 (default, 4gi_*, mimic_clinical) and 128 x 5 (configs/ablation_no_physics.yaml:11-12).
 *   H <= 64 and L <= 4   tuned kernels: all weights register-resident, one hidden unit per wavefront lane;
 *   otherwise            generic kernels (two hidden units per lane, weights streamed from L2, gradients by coalesced
 *                        atomics), same results contract, sized for the batches such shapes are trained with.
 * Activation: ReLU in the tuned kernels; tanh / elu / leaky_relu(0.1) through the generic kernels (HODE_LAYERS below).  Dropout
 * (also offered by NNResidual, never passed by HybridODENN, models/hybrid_ode_nn.py:57-58) is a training-time random mask with
 * no meaning inside an ODE right-hand side that is evaluated six times per step: the host class raises NotImplementedError for
 * it instead of silently computing something else. */
/* Activation codes.  The `L` argument of every entry point is HODE_LAYERS(hidden layers, activation): layers in bits 0..7, the
 * code in bits 8..15 -- a plain layer count means ReLU, which is all HybridODENN ever builds.  tanh / elu (alpha 1) /
 * leaky_relu (0.1) of NNResidual (models/nn_residual.py:50-56) are reached by replacing `model.nn_residual`; they run through
 * the generic kernels whatever the shape. */
#define HODE_ACT_RELU 0
#define HODE_ACT_TANH 1
#define HODE_ACT_ELU 2
#define HODE_ACT_LEAKY_RELU 3
#define HODE_LAYERS(L, act) (((L) & 0xff) | ((act) << 8))
/* hode_solve_fwd_* only: OR into `L` when nn_p holds ONE network shared by all n_sets sets of mechanistic constants (ode_p stays
 * [n_sets][17]).  The Sobol study of plots/plot_all.py:139-196 integrates 16 384 constant sets through one trained network: shared,
 * the launch reads 54 KB of weights instead of 885 MB of copies. */
#define HODE_LAYERS_NN_SHARED (1 << 16)
#define HODE_MAX_HIDDEN 128
#define HODE_MAX_LAYERS 8
#define HODE_TUNED_HIDDEN 64 /* envelope of the register-resident kernels */
#define HODE_TUNED_LAYERS 4

const char *hode_version(void);

/* number of MLP parameters for (H hidden, L hidden layers); 13510 for (64,4), 68102 for (128,5) */
int hode_nn_param_count(int H, int L);

/* bytes of the tape the solve writes for the adjoint: per accepted step {t, h, t0, 1/(t1-t0), v0, v1-v0, d0, d1-d0}
 * (the step and the constants of its grid interval: time, tVNS and gastric-distension inputs), the grid
 * interval (bit 30 set when the step ended exactly on the grid point closing that interval), and the "stage tape" -- the
 * MLP activations and stage state of every Runge-Kutta stage (tuned path 6 x (L x 64 + 8) reals per step, generic path
 * 6 x (2L x 64 + 8)), so that the adjoint never re-runs the forward.  This is a memory-for-compute trade sized for
 * 288 GB of HBM: 6.3 KB per step in fp32 for (64,4), i.e. 1.9 MB per trajectory at max_steps = 300.  Tuned shapes: + the
 * adjoint's gradient rows, min(B, 1024) x (P + 17 reals, padded to a multiple of 64).
 * hode_tape_bytes(.., L) == hode_tape_bytes_hl(.., 64, L). */
size_t hode_tape_bytes_hl(int B, int max_steps, int elem_size /* 4 or 8 */, int H, int L);
size_t hode_tape_bytes(int B, int max_steps, int elem_size /* 4 or 8 */, int L);

/* ---- K1: RHS forward.  Replaces HybridODENN.ode_residual (models/hybrid_ode_nn.py:108-134)
 *      = ODECore.forward (models/ode_core.py:81-166) + NNResidual.forward
 *      (models/nn_residual.py:100-151).  t/meal/tvns/gd: [B] or NULL.  out[B,6].            */
int hode_rhs_fwd_f32(void *stream, int B, const float *x, const float *t, const float *meal,
                     const float *tvns, const float *gd, const float *ode_p, const float *nn_p,
                     int H, int L, float *out);
int hode_rhs_fwd_f64(void *stream, int B, const double *x, const double *t, const double *meal,
                     const double *tvns, const double *gd, const double *ode_p, const double *nn_p,
                     int H, int L, double *out);

/* ---- K5: RHS backward (VJP).  Replaces torch autograd over ode_residual in the physics loss
 *      (models/hybrid_ode_nn.py:318-330).  gout[B,6] -> gx[B,6] (written), gt[B] (written, may
 *      be NULL), gnn[P] and gode[17] (ACCUMULATED with atomics: zero them first; may be NULL). */
int hode_rhs_bwd_f32(void *stream, int B, const float *x, const float *t, const float *meal,
                     const float *tvns, const float *gd, const float *ode_p, const float *nn_p,
                     int H, int L, const float *gout, float *gx, float *gt, float *gnn, float *gode);
int hode_rhs_bwd_f64(void *stream, int B, const double *x, const double *t, const double *meal,
                     const double *tvns, const double *gd, const double *ode_p, const double *nn_p,
                     int H, int L, const double *gout, double *gx, double *gt, double *gnn, double *gode);

/* ---- K2+K3: forward solve.  Replaces HybridODENN.forward (models/hybrid_ode_nn.py:136-261):
 *      the per-patient scipy.integrate.solve_ivp loop (:184-256) incl. input interpolation
 *      (:210-231) and the RK45 stepper (scipy/integrate/_ivp/rk.py:14-72,111-176).
 *      t: [T] (t_batched=0) or [B,T] (t_batched=1).  y[B,T,6] written (rows after a failure
 *      are zero).  status/nsteps/nfev: int32[B] (nsteps/nfev may be NULL).
 *      tape: NULL, or hode_tape_bytes_hl(B,max_steps,sizeof(real),H,L) bytes (256-byte aligned) that
 *      receive the accepted steps and their stage activations (needed by hode_solve_bwd_*).     */
int hode_solve_fwd_f32(void *stream, int B, int T, const float *x0, const float *t, int t_batched,
                       const float *meal, int meal_mode, const float *tvns, int tvns_mode,
                       const float *gd, int gd_mode, const float *ode_p, const float *nn_p,
                       int n_sets, int H, int L, int method, double rtol, double atol,
                       int max_steps, float *y, int32_t *status, int32_t *nsteps, int32_t *nfev,
                       void *tape);
int hode_solve_fwd_f64(void *stream, int B, int T, const double *x0, const double *t, int t_batched,
                       const double *meal, int meal_mode, const double *tvns, int tvns_mode,
                       const double *gd, int gd_mode, const double *ode_p, const double *nn_p,
                       int n_sets, int H, int L, int method, double rtol, double atol,
                       int max_steps, double *y, int32_t *status, int32_t *nsteps, int32_t *nfev,
                       void *tape);

/* ---- K4: reverse-time discrete adjoint of the solve above (no reference counterpart: the
 *      reference detaches the solve; north_star requires it).
 *      gy[B,T,6] = dLoss/dy  ->  gx0[B,6] (written), gnn[n_sets,P] and gode[n_sets,17]
 *      (ACCUMULATED: added to what is there -- zero them first; either may be NULL).  Tuned shapes (H <= 64, L <= 4): no
 *      floating-point atomics -- every workgroup writes one gradient row into the tail of the tape and a second, fixed-order
 *      pass adds the rows: the same call gives the same bits (the reference's CPU training is deterministic).  Generic shapes:
 *      coalesced atomics, reproducible to rounding only.  tape: the buffer the forward filled.  It is not const: the adjoint
 *      uses its tail (the gradient rows) as scratch -- and its LDS-DMA reads whole 256-byte rows: up to 224 bytes beyond the last stage
 *      record and up to 224 beyond a step entry, i.e. into the regions that FOLLOW them inside the same buffer (never beyond
 *      hode_tape_bytes_hl() bytes: the gradient rows, at least 256 bytes, close the buffer) --; what the forward recorded stays intact, the same tape may be walked
 *      again.
 *      nsteps[b], status[b]: what the forward returned for this tape.  The adjoint walks min(nsteps[b], max_steps) steps:
 *      a count larger than the tape it is handed (the caller merged the bookkeeping of a re-integration with a larger
 *      budget, say) is CLAMPED SILENTLY -- never an out-of-bounds read, but then the gradient is that of the truncated
 *      trajectory; pass the nsteps of the forward call that filled THIS tape.  A trajectory with status != 0 contributes the
 *      cotangents of the rows it still wrote (rows after the failure are zero in y and their gy is ignored); a non-finite
 *      step is never on the tape (status 3 trajectories end BEFORE the step that blew up).                  */
int hode_solve_bwd_f32(void *stream, int B, int T, const float *t, int t_batched,
                       const float *meal, int meal_mode, const float *tvns, int tvns_mode,
                       const float *gd, int gd_mode, const float *ode_p, const float *nn_p,
                       int n_sets, int H, int L, int method, int max_steps, const int32_t *nsteps,
                       const int32_t *status, void *tape, const float *gy, float *gx0,
                       float *gnn, float *gode);
int hode_solve_bwd_f64(void *stream, int B, int T, const double *t, int t_batched,
                       const double *meal, int meal_mode, const double *tvns, int tvns_mode,
                       const double *gd, int gd_mode, const double *ode_p, const double *nn_p,
                       int n_sets, int H, int L, int method, int max_steps, const int32_t *nsteps,
                       const int32_t *status, void *tape, const double *gy, double *gx0,
                       double *gnn, double *gode);

/* ---- K6: fused global-norm clip + Adam.  Replaces clip_grad_norm_(...,5.0) + torch.optim.Adam
 *      .step() (train/train_hybrid.py:255-261, 438-441).  g is first multiplied by grad_scale
 *      (e.g. 1/world_size after an all-reduce(sum)); max_norm <= 0 disables clipping.
 *      scratch: >= 8 bytes of device memory (holds the squared norm), zeroed by the call.       */
int hode_adam_step_f32(void *stream, int64_t n, float *p, const float *g, float *m, float *v,
                       float lr, float beta1, float beta2, float eps, int step, float max_norm,
                       float grad_scale, float weight_decay, void *scratch);

/* ---- loss helper: sum((y - obs)^2) and dLoss/dy = 2*scale*(y-obs) in one pass
 *      (models/hybrid_ode_nn.py:294 F.mse_loss).  loss_sum: double[1], ACCUMULATED.            */
int hode_mse_fwd_bwd_f32(void *stream, int64_t n, const float *y, const float *obs, float scale,
                         double *loss_sum, float *gy);

/* ---- self test of the cross-lane primitives (DPP / permlane swaps); out: int32[64*8].        */
int hode_selftest_xlane(void *stream, int32_t *out);

/* =====================================================================================================
 * Data side: the cohort generator and the dataset windows, on the device.
 * ===================================================================================================== */
#define HODE_4GI_NPAR 26       /* CLglc CLglci Qglc VCglc VPglc CLins VCins Ke0ins VCglp VM_GLP KM_GLP CLglg VCglg
                                  CLgip VCgip Qgip VPgip GLCINS_S EMAX_1 EC50_1 HILL_1 EMAX_4 EC50_4 FDGLP FDGIP FDGLG */
#define HODE_4GI_T2DM 0
#define HODE_4GI_HV 1
#define HODE_4GI_TABLE_COLS 9  /* subject_id time_hours time_minutes glucose_mmol_L insulin_pmol_L glp1_pmol_L
                                  glucagon_pmol_L gip_pmol_L meal_indicator  (data/generate4GI.py:246-257) */
#define HODE_4GI_SCRATCH_BYTES 98304
#define HODE_4GI_NORM_NONE 0   /* mean 0 / std 1 */
#define HODE_4GI_NORM_LOCAL 1  /* statistics of the windows passed in this call */
#define HODE_4GI_NORM_GIVEN 2  /* mean_std[12] is an INPUT (statistics combined over the shards of a multi-GPU dataset) */

/* the reference's parameter set (data/generate4GI.py:15-64) for a patient type -> par[HODE_4GI_NPAR] (HOST memory) */
int hode_4gi_default_params(int patient_type, double *par_host);

/* ---- K7: 4GI cohort generator.  Replaces FourGIModel.simulate (data/generate4GI.py:159-212: one scipy odeint call
 *      per subject per grid interval) and the table assembly of generate_dataset (:221-271) for B subjects at once.
 *      One subject per wavefront lane, fp64, DP5(4) at (rtol, atol) inside every grid interval.
 *      T grid points at k*interval_min minutes; bsl[B,5] = the subject's baselines (glucose, insulin, GLP-1, glucagon,
 *      GIP); meals: meal_time/meal_size [n_meals] shared by all subjects (meals_per_subject = 0) or [B,n_meals];
 *      par_host: HOST pointer to HODE_4GI_NPAR doubles or NULL (= hode_4gi_default_params(patient_type));
 *      z[T,5,B]: standard-normal draws for the measurement noise (biomarker order glucose, insulin, glp1, glucagon,
 *      gip; subject index fastest so that a wavefront reads contiguous memory) or NULL / noise_cv = 0 for the clean
 *      solution;
 *      table[B*T, 9] written, subject ids start at subject0; status[B] (may be NULL) as in hode_solve_fwd.          */
int hode_4gi_generate_f64(void *stream, int B, int T, double interval_min, int patient_type, const double *par_host,
                          const double *bsl, int n_meals, const double *meal_time, const double *meal_size,
                          int meals_per_subject, const double *z, double noise_cv, int64_t subject0, double rtol,
                          double atol, int max_steps, double *table, int32_t *status);

/* 4GI right-hand side alone (FourGIModel.model_equations, data/generate4GI.py:73-157): y[B,8], meal[B] -> d[B,8] */
int hode_4gi_rhs_f64(void *stream, int B, int patient_type, const double *par_host, const double *bsl, const double *y,
                     const double *meal, double *d);

/* ---- K8: sliding windows + z-scoring.  Replaces GlucoseDataset.__init__/__getitem__ (train/train_hybrid.py:43-155).
 *      table[rows, ncols] fp64 row-major; col_* = column indices (col_ge / col_ffa / col_meal / col_tvns may be -1:
 *      0, 1, 0, 0 as in :76-91); time = table[:, col_time] / time_div (60 for time_minutes, :93-94);
 *      row0[N] (device int64) = first table row of every window (subject by subject, start += stride, :105-121);
 *      normalize = HODE_4GI_NORM_LOCAL: mean/std over ALL window rows (overlaps counted as often as they occur, population std
 *      + 1e-6; :124-127), HODE_4GI_NORM_NONE: mean 0 / std 1, HODE_4GI_NORM_GIVEN: as found in mean_std.  Written: states[N,S,6] fp32 (observations; initial_state = [:,0]),
 *      meal[N,S], tvns[N,S], time[N,S] fp32, mean_std[12] fp64 (6 means, 6 stds).
 *      scratch: HODE_4GI_SCRATCH_BYTES of device memory.  Deterministic (no atomics).                              */
int hode_4gi_windows_f32(void *stream, const double *table, int ncols, int col_time, double time_div, int col_glucose,
                         int col_insulin, int col_glucagon, int col_glp1, int col_ge, int col_ffa, int col_meal,
                         int col_tvns, const int64_t *row0, int64_t N, int64_t S, int normalize, float *states,
                         float *meal, float *tvns, float *time, double *mean_std, void *scratch);

/* Mergeable statistics of the windows of ONE shard: moments[13] = {count, mean[6], M2[6]} (M2 = sum of squared
 * deviations from the shard mean).  A dataset sharded over GPUs all-gathers these 13 doubles, merges them in rank
 * order (Chan et al.: M2 = M2a + M2b + d^2 na nb / n), sets std = sqrt(M2/n) + 1e-6 and calls hode_4gi_windows_f32
 * with HODE_4GI_NORM_GIVEN, so every rank normalises with the statistics of the whole dataset. */
int hode_4gi_window_moments_f64(void *stream, const double *table, int ncols, int col_glucose, int col_insulin,
                                int col_glucagon, int col_glp1, int col_ge, int col_ffa, const int64_t *row0, int64_t N,
                                int64_t S, double *moments, void *scratch);

#ifdef __cplusplus
}
#endif
#endif /* HODE_H */
