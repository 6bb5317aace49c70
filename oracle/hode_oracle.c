/*
 * ORACLE -- test infrastructure only.  NOT part of the product path.
 *
 * CPU restatement (plain C) of the reference hot path: the hybrid GLP-1/glucose RHS, the
 * Dormand-Prince 5(4) integrator around it and a discrete adjoint.  Only tests/, the
 * cpu_baseline leg of bench.py and __graft_entry__.smoke() may load this library, and only
 * as the checker.  The product (libhode.so, HIP) never links or calls it.
 *
 * Parity pinning: every function is checked in tests/test_oracle_golden.py against vectors
 * generated in the build container by importing the reference itself
 * (tools/capture_golden.py writes tests/golden/ .npz files).
 *
 * Third-party arithmetic restated here: scipy.integrate.solve_ivp(method='RK45'), SciPy 1.15.3
 * (reference requirements.txt:3 says scipy>=1.10.0, unpinned; 1.15.3 is what the reference
 * runs on in the build container): scipy/integrate/_ivp/{ivp.py,rk.py,common.py}.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define HODE_MAXH 128
#define HODE_MAXL 8

/* ------------------------------------------------------------------ fp32 instantiation */
#define REAL float
#define ACC double
#define SFX(x) x##_f32
#define RPOW(a, b) powf((a), (b))
#define RLOG(a) logf((a))
#define RTANH(a) tanhf((a))
#define REXPM1(a) expm1f((a))
#include "hode_oracle_impl.h"
#undef REAL
#undef ACC
#undef SFX
#undef RPOW
#undef RLOG
#undef RTANH
#undef REXPM1

/* ------------------------------------------------------------------ fp64 instantiation */
#define REAL double
#define ACC double
#define SFX(x) x##_f64
#define RPOW(a, b) pow((a), (b))
#define RLOG(a) log((a))
#define RTANH(a) tanh((a))
#define REXPM1(a) expm1((a))
#include "hode_oracle_impl.h"
#undef REAL
#undef ACC
#undef SFX
#undef RPOW
#undef RLOG
#undef RTANH
#undef REXPM1

int hode_oracle_tape_entry_size_f32(void) { return (int)sizeof(tape_t_f32); }
int hode_oracle_tape_entry_size_f64(void) { return (int)sizeof(tape_t_f64); }

/*
 * "Reference mode": what HybridODENN.forward(solver='rk45') does at ANY tolerance
 * (models/hybrid_ode_nn.py:184-256): per patient, SciPy RK45 over the whole span in fp64,
 * NOT broken at grid points, t_eval served from the 4th-order dense output
 * (scipy rk.py:552-574, ivp.py:696-723), the RHS evaluated in fp32 on fp32-rounded (t, y)
 * (hybrid_ode_nn.py:207-208) with inputs interpolated in fp32 using np.searchsorted
 * side='left' (hybrid_ode_nn.py:217-229).  Output rounded to fp32 (:248).
 */
static const double dpP[7][4] = {
    {1, -8048581381.0 / 2820520608.0, 8663915743.0 / 2820520608.0, -12715105075.0 / 11282082432.0},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799.0, -68118460800.0 / 10900136933.0, 87487479700.0 / 32700410799.0},
    {0, -1754552775.0 / 470086768.0, 14199869525.0 / 1410260304.0, -10690763975.0 / 1880347072.0},
    {0, 127303824393.0 / 49829197408.0, -318862633887.0 / 49829197408.0, 701980252875.0 / 199316789632.0},
    {0, -282668133.0 / 205662961.0, 2019193451.0 / 616988883.0, -1453857185.0 / 822651844.0},
    {0, 40617522.0 / 29380423.0, -110615467.0 / 29380423.0, 69997945.0 / 29380423.0}};

void hode_oracle_dp_tableau(double *C, double *A, double *Bw, double *E, double *P)
{
    for (int i = 0; i < 6; ++i) C[i] = dpC_f64[i];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 5; ++j) A[i * 5 + j] = (j < i) ? dpA_f64[i][j] : 0.0;
    for (int i = 0; i < 6; ++i) Bw[i] = dpA_f64[6][i];
    for (int i = 0; i < 7; ++i) E[i] = dpE_f64[i];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 4; ++j) P[i * 4 + j] = dpP[i][j];
}

typedef struct {
    const mlp_t_f32 *m; const float *ode; const float *tgrid; int T;
    const float *meal; int meal_mode; const float *tvns; int tvns_mode; const float *gd; int gd_mode;
    int b; int nfev;
} refctx_t;

static float ref_input(const refctx_t *c, const float *v, int mode, float t32)
{
    if (!v || mode == 0) return 0.0f;
    if (mode == 1) return v[c->b];
    const float *row = v + (size_t)c->b * c->T;
    /* np.searchsorted(t_eval, t, side='left') */
    int idx = 0;
    while (idx < c->T && c->tgrid[idx] < t32) ++idx;
    if (idx == 0) return row[0];
    if (idx >= c->T) return row[c->T - 1];
    float t1 = c->tgrid[idx - 1], t2 = c->tgrid[idx];
    float alpha = (t32 - t1) / (t2 - t1);
    return row[idx - 1] + alpha * (row[idx] - row[idx - 1]);
}

/* optional debugging hook: called with (t, nfev) at every RHS evaluation of reference mode */
void (*hode_oracle_trace)(double, int) = 0;

static void ref_fun(refctx_t *c, double t, const double *y, double *f)
{
    if (hode_oracle_trace) hode_oracle_trace(t, c->nfev);
    float act[HODE_MAXL + 1][HODE_MAXH];
    float t32 = (float)t, y32[6], f32[6];
    for (int i = 0; i < 6; ++i) y32[i] = (float)y[i];
    float meal = ref_input(c, c->meal, c->meal_mode, t32);
    float tvns = ref_input(c, c->tvns, c->tvns_mode, t32);
    float gd = ref_input(c, c->gd, c->gd_mode, t32);
    rhs_one_f32(c->m, c->ode, t32, y32, meal, tvns, gd, c->gd != NULL && c->gd_mode != 0, f32, act);
    for (int i = 0; i < 6; ++i) f[i] = (double)f32[i];
    c->nfev++;
}

static double rms6(const double *v) { double s = 0; for (int i = 0; i < 6; ++i) s += v[i] * v[i]; return sqrt(s / 6.0); }

int hode_oracle_solve_scipy_rk45(int B, int T, const float *x0, const float *tg, int t_batched,
                                 const float *meal, int meal_mode, const float *tvns, int tvns_mode,
                                 const float *gd, int gd_mode, const float *ode, const float *nn_p,
                                 int H, int L, double rtol, double atol, float *yout, int *status,
                                 int *nsteps, int *nfev)
{
    mlp_t_f32 m;
    if (mlp_bind_f32(&m, nn_p, H, L) < 0) return -1;
    for (int b = 0; b < B; ++b) {
        refctx_t c = {&m, ode, t_batched ? tg + (size_t)b * T : tg, T, meal, meal_mode, tvns, tvns_mode, gd, gd_mode, b, 0};
        float *yb = yout + (size_t)b * T * 6;
        memset(yb, 0, sizeof(float) * 6 * T);
        double t = (double)c.tgrid[0], tf = (double)c.tgrid[T - 1];
        double y[6], f[6], K[7][6];
        for (int i = 0; i < 6; ++i) y[i] = (double)x0[6 * b + i];
        int st = 0, ns = 0, ev = 0;   /* ev = next t_eval index to serve */
        if (!(tf > t)) {              /* degenerate span: solve_ivp returns t_eval points == t0 */
            for (int k = 0; k < T; ++k) for (int i = 0; i < 6; ++i) yb[6 * k + i] = (float)y[i];
            if (status) status[b] = 0;
            if (nsteps) nsteps[b] = 0;
            if (nfev) nfev[b] = 0;
            continue;
        }
        ref_fun(&c, t, y, f);
        /* select_initial_step (common.py:68-135), order = 4 */
        double h_abs;
        {
            double sc[6], a0[6], a1[6], y1[6], f1[6], dd[6];
            double interval = fabs(tf - t);
            for (int i = 0; i < 6; ++i) { sc[i] = atol + fabs(y[i]) * rtol; a0[i] = y[i] / sc[i]; a1[i] = f[i] / sc[i]; }
            double d0 = rms6(a0), d1 = rms6(a1);
            double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
            if (h0 > interval) h0 = interval;
            for (int i = 0; i < 6; ++i) y1[i] = y[i] + h0 * f[i];
            ref_fun(&c, t + h0, y1, f1);
            for (int i = 0; i < 6; ++i) dd[i] = (f1[i] - f[i]) / sc[i];
            double d2 = rms6(dd) / h0;
            double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
            h_abs = fmin(fmin(100 * h0, h1), interval);
        }
        while (st == 0 && t < tf) {
            double min_step = 10.0 * fabs(nextafter(t, INFINITY) - t);
            if (h_abs < min_step) h_abs = min_step;     /* rk.py:121-126 clip into [min_step, max_step] */
            int accepted = 0, rejected = 0;
            double h = 0, tn = 0, yn[6], fn[6];
            while (!accepted) {
                if (h_abs < min_step) { st = 2; break; }
                h = h_abs; tn = t + h;
                if (tn - tf > 0) tn = tf;
                h = tn - t; h_abs = fabs(h);
                for (int i = 0; i < 6; ++i) K[0][i] = f[i];
                for (int s = 1; s < 6; ++s) {
                    double ys[6];
                    for (int i = 0; i < 6; ++i) {
                        double acc = 0;
                        for (int j = 0; j < s; ++j) acc += dpA_f64[s][j] * K[j][i];
                        ys[i] = y[i] + acc * h;
                    }
                    ref_fun(&c, t + dpC_f64[s] * h, ys, K[s]);
                }
                for (int i = 0; i < 6; ++i) {
                    double acc = 0;
                    for (int j = 0; j < 6; ++j) acc += dpA_f64[6][j] * K[j][i];
                    yn[i] = y[i] + h * acc;
                }
                ref_fun(&c, t + h, yn, fn);
                for (int i = 0; i < 6; ++i) K[6][i] = fn[i];
                double ev6[6];
                for (int i = 0; i < 6; ++i) {
                    double e = 0;
                    for (int j = 0; j < 7; ++j) e += dpE_f64[j] * K[j][i];
                    double scale = atol + fmax(fabs(y[i]), fabs(yn[i])) * rtol;
                    ev6[i] = e * h / scale;
                }
                double en = rms6(ev6);
                if (en < 1) {
                    double fac = (en == 0) ? 10.0 : fmin(10.0, 0.9 * pow(en, -0.2));
                    if (rejected) fac = fmin(1.0, fac);
                    h_abs *= fac;
                    accepted = 1;
                } else if (en >= 1) {
                    h_abs *= fmax(0.2, 0.9 * pow(en, -0.2));
                    rejected = 1;
                } else {            /* NaN error norm: SciPy accepts nothing and loops; fail instead */
                    st = 3; break;
                }
            }
            if (st) break;
            ns++;
            /* serve t_eval in (t_old, t_new] from the dense output; first call also serves t0 */
            int ev_new = ev;
            while (ev_new < T && (double)c.tgrid[ev_new] <= tn) ++ev_new;
            for (int k = ev; k < ev_new; ++k) {
                double x = ((double)c.tgrid[k] - t) / h, p[4] = {x, x * x, x * x * x, x * x * x * x};
                for (int i = 0; i < 6; ++i) {
                    double acc = 0;
                    for (int q = 0; q < 4; ++q) {
                        double Q = 0;
                        for (int s = 0; s < 7; ++s) Q += K[s][i] * dpP[s][q];
                        acc += Q * p[q];
                    }
                    yb[6 * k + i] = (float)(y[i] + h * acc);
                }
            }
            ev = ev_new;
            t = tn;
            for (int i = 0; i < 6; ++i) { y[i] = yn[i]; f[i] = fn[i]; }
        }
        if (status) status[b] = st;
        if (nsteps) nsteps[b] = ns;
        if (nfev) nfev[b] = c.nfev;
    }
    return 0;
}

const char *hode_oracle_version(void) { return "hode-oracle 0.1 (C restatement; test infrastructure only)"; }
