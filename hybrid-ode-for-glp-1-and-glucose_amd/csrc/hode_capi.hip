// hode_capi.hip -- extern "C" entry points of libhode.so (declared in include/hode.h).
// Plain pointers and sizes only; argument validation happens here, on the host, BEFORE any
// launch: a kernel is never started on shapes it was not compiled for.
#include "hode_kernels.h"
#include <math.h>
#include <string.h>
#include "hode_device.h"

using namespace hode;

namespace {

inline bool mode_ok(int mode, const void *p) { return mode == 0 || (p != nullptr && (mode == 1 || mode == 2)); }

template <typename R>
int solve_fwd(void *stream, int B, int T, const R *x0, const R *t, int t_batched, const R *meal, int meal_mode,
              const R *tvns, int tvns_mode, const R *gd, int gd_mode, const R *ode_p, const R *nn_p, int n_sets,
              int H, int L, int method, double rtol, double atol, int max_steps, R *y, int32_t *status,
              int32_t *nsteps, int32_t *nfev, void *tape)
{
    if (B == 0 && T >= 1) return HODE_OK;                     // an empty batch is valid (empty tensors have null data)
    if (B < 0 || T < 1 || !x0 || !t || !ode_p || !nn_p || !y || !status) return HODE_EINVAL;
    if (!mode_ok(meal_mode, meal) || !mode_ok(tvns_mode, tvns) || !mode_ok(gd_mode, gd)) return HODE_EINVAL;
    if (n_sets < 1 || (B % n_sets) != 0 || max_steps < 1) return HODE_EINVAL;
    if (method != HODE_METHOD_DP54 && method != HODE_METHOD_RK4) return HODE_EINVAL;
    if (!(rtol >= 0) || !(atol >= 0) || (method == HODE_METHOD_DP54 && rtol == 0 && atol == 0)) return HODE_EINVAL;
    if (H < 1 || H > HODE_MAX_HIDDEN || layers_of(L) < 1 || layers_of(L) > HODE_MAX_LAYERS || act_of(L) > HODE_ACT_LEAKY_RELU || (L >> 17) != 0)
        return HODE_EUNSUPPORTED;
    const bool nn_shared = (L & HODE_LAYERS_NN_SHARED) != 0;   // one network for every set of constants (forward only)
    if (nn_shared && tape) return HODE_EUNSUPPORTED;          // (the adjoint writes one gradient row per parameter set)
    L &= 0xffff;
    if (B == 0) return HODE_OK;
    SolveArgs<R> a;
    a.B = B; a.T = T; a.t_batched = t_batched ? 1 : 0;
    a.meal_mode = meal_mode; a.tvns_mode = tvns_mode; a.gd_mode = gd_mode;
    a.n_sets = n_sets; a.H = H; a.P = nn_param_count(H, L); a.max_steps = max_steps;
    a.x0 = x0; a.t = t; a.meal = meal; a.tvns = tvns; a.gd = gd; a.ode_p = ode_p; a.nn_p = nn_p;
    a.rtol = (R)rtol; a.atol = (R)atol;
    a.y = y; a.status = status; a.nsteps = nsteps; a.nfev = nfev;
    a.tape = (R *)tape;
    a.tape_seg = tape ? (int32_t *)((char *)tape + tape_seg_offset(B, max_steps, sizeof(R))) : nullptr;
    a.tape_stage = tape ? (R *)((char *)tape + tape_stage_offset(B, max_steps, sizeof(R))) : nullptr;
    a.L = layers_of(L);
    a.act = act_of(L);
    a.nn_stride = nn_shared ? 0 : a.P;
    if (!tuned_shape(H, L)) return launch_solve_fwd_generic<R>((hipStream_t)stream, a, method);
    return launch_solve_fwd<R>((hipStream_t)stream, a, layers_of(L), method);
}

template <typename R>
int rhs_fwd(void *stream, int B, const R *x, const R *t, const R *meal, const R *tvns, const R *gd, const R *ode_p,
            const R *nn_p, int H, int L, R *out)
{
    if (B == 0) return HODE_OK;
    if (B < 0 || !x || !ode_p || !nn_p || !out) return HODE_EINVAL;
    if (H < 1 || H > HODE_MAX_HIDDEN || layers_of(L) < 1 || layers_of(L) > HODE_MAX_LAYERS || act_of(L) > HODE_ACT_LEAKY_RELU || (L >> 16) != 0)
        return HODE_EUNSUPPORTED;
    if (B == 0) return HODE_OK;
    RhsArgs<R> a{};
    a.B = B; a.H = H; a.P = nn_param_count(H, L);
    a.x = x; a.t = t; a.meal = meal; a.tvns = tvns; a.gd = gd; a.ode_p = ode_p; a.nn_p = nn_p; a.out = out;
    a.act = act_of(L);
    if (!tuned_shape(H, L)) return launch_rhs_fwd_generic<R>((hipStream_t)stream, a, layers_of(L));
    return launch_rhs_fwd<R>((hipStream_t)stream, a, layers_of(L));
}

template <typename R>
int solve_bwd(void *stream, int B, int T, const R *t, int t_batched, const R *meal, int meal_mode, const R *tvns,
              int tvns_mode, const R *gd, int gd_mode, const R *ode_p, const R *nn_p, int n_sets, int H, int L,
              int method, int max_steps, const int32_t *nsteps, const int32_t *status, void *tape, const R *gy,
              R *gx0, R *gnn, R *gode)
{
    if (B == 0 && T >= 1) return HODE_OK;
    if (B < 0 || T < 1 || !t || !ode_p || !nn_p || !nsteps || !status || !tape || !gy || !gx0) return HODE_EINVAL;
    if (!mode_ok(meal_mode, meal) || !mode_ok(tvns_mode, tvns) || !mode_ok(gd_mode, gd)) return HODE_EINVAL;
    if (n_sets < 1 || (B % n_sets) != 0 || max_steps < 1) return HODE_EINVAL;
    if (method != HODE_METHOD_DP54 && method != HODE_METHOD_RK4) return HODE_EINVAL;
    if (H < 1 || H > HODE_MAX_HIDDEN || layers_of(L) < 1 || layers_of(L) > HODE_MAX_LAYERS || act_of(L) > HODE_ACT_LEAKY_RELU || (L >> 16) != 0)
        return HODE_EUNSUPPORTED;
    if (B == 0) return HODE_OK;
    AdjArgs<R> a;
    a.B = B; a.T = T; a.t_batched = t_batched ? 1 : 0;
    a.meal_mode = meal_mode; a.tvns_mode = tvns_mode; a.gd_mode = gd_mode;
    a.n_sets = n_sets; a.H = H; a.P = nn_param_count(H, L); a.max_steps = max_steps;
    a.t = t; a.meal = meal; a.tvns = tvns; a.gd = gd; a.ode_p = ode_p; a.nn_p = nn_p;
    a.nsteps = nsteps; a.status = status;
    a.tape = (const R *)tape;
    a.tape_seg = (const int32_t *)((const char *)tape + tape_seg_offset(B, max_steps, sizeof(R)));
    a.tape_stage = (const R *)((const char *)tape + tape_stage_offset(B, max_steps, sizeof(R)));
    a.gy = gy; a.gx0 = gx0; a.gnn = gnn; a.gode = gode;
    a.tape_delta = has_delta_tape(sizeof(R), H, L) ? (R *)((char *)tape + tape_delta_offset(B, max_steps, sizeof(R), H, L)) : nullptr;
    a.partials = (tuned_shape(H, L) || sizeof(R) == 4) ? (R *)((char *)tape + tape_partials_offset(B, max_steps, sizeof(R), H, L)) : nullptr;
    a.partial_rows = adj_partial_rows(B);
    a.act = act_of(L);
    if (!tuned_shape(H, L)) return launch_solve_bwd_generic<R>((hipStream_t)stream, a, layers_of(L), method);
    return launch_solve_bwd<R>((hipStream_t)stream, a, layers_of(L), method);
}

template <typename R>
int rhs_bwd(void *stream, int B, const R *x, const R *t, const R *meal, const R *tvns, const R *gd, const R *ode_p,
            const R *nn_p, int H, int L, const R *gout, R *gx, R *gt, R *gnn, R *gode)
{
    if (B == 0) return HODE_OK;
    if (B < 0 || !x || !ode_p || !nn_p || !gout || !gx) return HODE_EINVAL;
    if (H < 1 || H > HODE_MAX_HIDDEN || layers_of(L) < 1 || layers_of(L) > HODE_MAX_LAYERS || act_of(L) > HODE_ACT_LEAKY_RELU || (L >> 16) != 0)
        return HODE_EUNSUPPORTED;
    if (B == 0) return HODE_OK;
    RhsArgs<R> a{};
    a.B = B; a.H = H; a.P = nn_param_count(H, L);
    a.x = x; a.t = t; a.meal = meal; a.tvns = tvns; a.gd = gd; a.ode_p = ode_p; a.nn_p = nn_p;
    a.gout = gout; a.gx = gx; a.gt = gt; a.gnn = gnn; a.gode = gode;
    a.act = act_of(L);
    if (!tuned_shape(H, L)) return launch_rhs_bwd_generic<R>((hipStream_t)stream, a, layers_of(L));
    return launch_rhs_bwd<R>((hipStream_t)stream, a, layers_of(L));
}

}  // namespace

extern "C" {

const char *hode_version(void) { return "hode 0.3.0 (gfx950; wave-per-trajectory DP5(4) + wave-specialised adjoint; MLP up to 8 x 128)"; }

int hode_nn_param_count(int H, int L) { return (H < 1 || layers_of(L) < 1) ? HODE_EINVAL : nn_param_count(H, L); }

size_t hode_tape_bytes_hl(int B, int max_steps, int elem_size, int H, int L)
{
    if (B < 0 || max_steps < 0 || (elem_size != 4 && elem_size != 8) || H < 1 || H > HODE_MAX_HIDDEN || layers_of(L) < 1 ||
        layers_of(L) > HODE_MAX_LAYERS || act_of(L) > HODE_ACT_LEAKY_RELU || (L >> 16) != 0)
        return 0;
    return tape_total_bytes(B, max_steps, (size_t)elem_size, H, L);
}
size_t hode_tape_bytes(int B, int max_steps, int elem_size, int L) { return hode_tape_bytes_hl(B, max_steps, elem_size, 64, L); }

int hode_rhs_fwd_f32(void *stream, int B, const float *x, const float *t, const float *meal, const float *tvns,
                     const float *gd, const float *ode_p, const float *nn_p, int H, int L, float *out)
{
    return rhs_fwd<float>(stream, B, x, t, meal, tvns, gd, ode_p, nn_p, H, L, out);
}
int hode_rhs_fwd_f64(void *stream, int B, const double *x, const double *t, const double *meal, const double *tvns,
                     const double *gd, const double *ode_p, const double *nn_p, int H, int L, double *out)
{
    return rhs_fwd<double>(stream, B, x, t, meal, tvns, gd, ode_p, nn_p, H, L, out);
}

int hode_solve_fwd_f32(void *stream, int B, int T, const float *x0, const float *t, int t_batched, const float *meal,
                       int meal_mode, const float *tvns, int tvns_mode, const float *gd, int gd_mode,
                       const float *ode_p, const float *nn_p, int n_sets, int H, int L, int method, double rtol,
                       double atol, int max_steps, float *y, int32_t *status, int32_t *nsteps, int32_t *nfev, void *tape)
{
    return solve_fwd<float>(stream, B, T, x0, t, t_batched, meal, meal_mode, tvns, tvns_mode, gd, gd_mode, ode_p, nn_p,
                            n_sets, H, L, method, rtol, atol, max_steps, y, status, nsteps, nfev, tape);
}
int hode_solve_fwd_f64(void *stream, int B, int T, const double *x0, const double *t, int t_batched, const double *meal,
                       int meal_mode, const double *tvns, int tvns_mode, const double *gd, int gd_mode,
                       const double *ode_p, const double *nn_p, int n_sets, int H, int L, int method, double rtol,
                       double atol, int max_steps, double *y, int32_t *status, int32_t *nsteps, int32_t *nfev, void *tape)
{
    return solve_fwd<double>(stream, B, T, x0, t, t_batched, meal, meal_mode, tvns, tvns_mode, gd, gd_mode, ode_p, nn_p,
                             n_sets, H, L, method, rtol, atol, max_steps, y, status, nsteps, nfev, tape);
}

int hode_rhs_bwd_f32(void *stream, int B, const float *x, const float *t, const float *meal, const float *tvns,
                     const float *gd, const float *ode_p, const float *nn_p, int H, int L, const float *gout, float *gx,
                     float *gt, float *gnn, float *gode)
{
    return rhs_bwd<float>(stream, B, x, t, meal, tvns, gd, ode_p, nn_p, H, L, gout, gx, gt, gnn, gode);
}
int hode_rhs_bwd_f64(void *stream, int B, const double *x, const double *t, const double *meal, const double *tvns,
                     const double *gd, const double *ode_p, const double *nn_p, int H, int L, const double *gout,
                     double *gx, double *gt, double *gnn, double *gode)
{
    return rhs_bwd<double>(stream, B, x, t, meal, tvns, gd, ode_p, nn_p, H, L, gout, gx, gt, gnn, gode);
}

int hode_solve_bwd_f32(void *stream, int B, int T, const float *t, int t_batched, const float *meal, int meal_mode,
                       const float *tvns, int tvns_mode, const float *gd, int gd_mode, const float *ode_p,
                       const float *nn_p, int n_sets, int H, int L, int method, int max_steps, const int32_t *nsteps,
                       const int32_t *status, void *tape, const float *gy, float *gx0, float *gnn, float *gode)
{
    return solve_bwd<float>(stream, B, T, t, t_batched, meal, meal_mode, tvns, tvns_mode, gd, gd_mode, ode_p, nn_p,
                            n_sets, H, L, method, max_steps, nsteps, status, tape, gy, gx0, gnn, gode);
}
int hode_solve_bwd_f64(void *stream, int B, int T, const double *t, int t_batched, const double *meal, int meal_mode,
                       const double *tvns, int tvns_mode, const double *gd, int gd_mode, const double *ode_p,
                       const double *nn_p, int n_sets, int H, int L, int method, int max_steps, const int32_t *nsteps,
                       const int32_t *status, void *tape, const double *gy, double *gx0, double *gnn, double *gode)
{
    return solve_bwd<double>(stream, B, T, t, t_batched, meal, meal_mode, tvns, tvns_mode, gd, gd_mode, ode_p, nn_p,
                             n_sets, H, L, method, max_steps, nsteps, status, tape, gy, gx0, gnn, gode);
}

int hode_adam_step_f32(void *stream, int64_t n, float *p, const float *g, float *m, float *v, float lr, float beta1,
                       float beta2, float eps, int step, float max_norm, float grad_scale, float weight_decay,
                       void *scratch)
{
    if (n < 0 || !p || !g || !m || !v || !scratch || step < 1) return HODE_EINVAL;
    return launch_adam((hipStream_t)stream, n, p, g, m, v, lr, beta1, beta2, eps, step, max_norm, grad_scale,
                       weight_decay, scratch);
}

int hode_mse_fwd_bwd_f32(void *stream, int64_t n, const float *y, const float *obs, float scale, double *loss_sum,
                         float *gy)
{
    if (n < 0 || !y || !obs || !loss_sum) return HODE_EINVAL;
    return launch_mse((hipStream_t)stream, n, y, obs, scale, loss_sum, gy);
}

int hode_selftest_xlane(void *stream, int32_t *out)
{
    if (!out) return HODE_EINVAL;
    return launch_selftest((hipStream_t)stream, out);
}

// ---- data side -----------------------------------------------------------------------------------------------
int hode_4gi_default_params(int patient_type, double *par)
{
    if (!par || (patient_type != HODE_4GI_T2DM && patient_type != HODE_4GI_HV)) return HODE_EINVAL;
    const bool hv = patient_type == HODE_4GI_HV;
    // data/generate4GI.py:15-64, in the order of HODE_4GI_NPAR
    const double v[HODE_4GI_NPAR] = {hv ? 5.36 : 1.72, hv ? 0.072 : 0.0256, 26.5, 9.33, 8.56, 73.2, 6.09, exp(-0.159),
                                     16.0, exp(7.97), exp(4.91), 453.2, 64.6, 86.8, 9.21, 49.4, 22.8, 2.46, exp(2.37),
                                     exp(3.29), 1.79, 6.73, exp(4.59), 0.0102, 0.0343, 0.00329};
    for (int i = 0; i < HODE_4GI_NPAR; ++i) par[i] = v[i];
    return HODE_OK;
}

static int fourgi_params(int patient_type, const double *par_host, FourGIPar *out)
{
    double v[HODE_4GI_NPAR];
    if (patient_type != HODE_4GI_T2DM && patient_type != HODE_4GI_HV) return HODE_EINVAL;
    if (par_host)
        for (int i = 0; i < HODE_4GI_NPAR; ++i) v[i] = par_host[i];
    else
        hode_4gi_default_params(patient_type, v);
    memcpy(out, v, sizeof v);
    return HODE_OK;
}

int hode_4gi_generate_f64(void *stream, int B, int T, double interval_min, int patient_type, const double *par_host,
                          const double *bsl, int n_meals, const double *meal_time, const double *meal_size,
                          int meals_per_subject, const double *z, double noise_cv, int64_t subject0, double rtol,
                          double atol, int max_steps, double *table, int32_t *status)
{
    if (B < 0 || T < 1 || n_meals < 0 || max_steps < 1) return HODE_EINVAL;
    if (!(interval_min > 0.0) || !(rtol > 0.0) || !(atol >= 0.0)) return HODE_EINVAL;
    if (B == 0) return HODE_OK;
    if (!bsl || !table || (n_meals > 0 && (!meal_time || !meal_size))) return HODE_EINVAL;
    GenArgs a{};
    if (int rc = fourgi_params(patient_type, par_host, &a.par)) return rc;
    a.B = B; a.T = T; a.hv = patient_type == HODE_4GI_HV; a.n_meals = n_meals; a.meals_per_subject = meals_per_subject != 0;
    a.max_steps = max_steps; a.subject0 = subject0; a.interval_min = interval_min; a.rtol = rtol; a.atol = atol;
    a.noise_cv = noise_cv; a.bsl = bsl; a.meal_time = meal_time; a.meal_size = meal_size; a.z = z; a.table = table;
    a.status = status;
    return launch_4gi_generate((hipStream_t)stream, a);
}

int hode_4gi_rhs_f64(void *stream, int B, int patient_type, const double *par_host, const double *bsl, const double *y,
                     const double *meal, double *d)
{
    if (B < 0) return HODE_EINVAL;
    if (B == 0) return HODE_OK;
    if (!bsl || !y || !meal || !d) return HODE_EINVAL;
    FourGIPar p;
    if (int rc = fourgi_params(patient_type, par_host, &p)) return rc;
    return launch_4gi_rhs((hipStream_t)stream, B, patient_type == HODE_4GI_HV, p, bsl, y, meal, d);
}

int hode_4gi_windows_f32(void *stream, const double *table, int ncols, int col_time, double time_div, int col_glucose,
                         int col_insulin, int col_glucagon, int col_glp1, int col_ge, int col_ffa, int col_meal,
                         int col_tvns, const int64_t *row0, int64_t N, int64_t S, int normalize, float *states,
                         float *meal, float *tvns, float *time, double *mean_std, void *scratch)
{
    if (N < 0 || S < 1 || ncols < 1 || !mean_std || !(time_div != 0.0)) return HODE_EINVAL;
    if (normalize < HODE_4GI_NORM_NONE || normalize > HODE_4GI_NORM_GIVEN) return HODE_EINVAL;
    const int need[5] = {col_time, col_glucose, col_insulin, col_glucagon, col_glp1};
    for (int c : need)
        if (c < 0 || c >= ncols) return HODE_EINVAL;
    const int opt[4] = {col_ge, col_ffa, col_meal, col_tvns};
    for (int c : opt)
        if (c < -1 || c >= ncols) return HODE_EINVAL;
    if (N > 0 && (!table || !row0 || !states || !meal || !tvns || !time || !scratch)) return HODE_EINVAL;
    WinArgs a{};
    a.table = table; a.ncols = ncols; a.col_time = col_time; a.col_meal = col_meal; a.col_tvns = col_tvns;
    a.col_state[0] = col_glucose; a.col_state[1] = col_insulin; a.col_state[2] = col_glucagon; a.col_state[3] = col_glp1;
    a.col_state[4] = col_ge; a.col_state[5] = col_ffa;
    a.time_div = time_div; a.row0 = row0; a.N = N; a.S = S; a.states = states; a.meal = meal; a.tvns = tvns; a.time = time;
    return launch_4gi_windows((hipStream_t)stream, a, normalize, mean_std, scratch);
}

int hode_4gi_window_moments_f64(void *stream, const double *table, int ncols, int col_glucose, int col_insulin,
                                int col_glucagon, int col_glp1, int col_ge, int col_ffa, const int64_t *row0, int64_t N,
                                int64_t S, double *moments, void *scratch)
{
    if (N < 0 || S < 1 || ncols < 1 || !moments) return HODE_EINVAL;
    const int need[4] = {col_glucose, col_insulin, col_glucagon, col_glp1};
    for (int c : need)
        if (c < 0 || c >= ncols) return HODE_EINVAL;
    if (col_ge < -1 || col_ge >= ncols || col_ffa < -1 || col_ffa >= ncols) return HODE_EINVAL;
    if (N > 0 && (!table || !row0 || !scratch)) return HODE_EINVAL;
    WinArgs a{};
    a.table = table; a.ncols = ncols; a.col_time = 0; a.col_meal = -1; a.col_tvns = -1;
    a.col_state[0] = col_glucose; a.col_state[1] = col_insulin; a.col_state[2] = col_glucagon; a.col_state[3] = col_glp1;
    a.col_state[4] = col_ge; a.col_state[5] = col_ffa;
    a.time_div = 1.0; a.row0 = row0; a.N = N; a.S = S;
    return launch_4gi_window_moments((hipStream_t)stream, a, moments, scratch);
}

}  // extern "C"
