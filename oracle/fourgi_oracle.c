/*
 * fourgi_oracle.c -- ORACLE (test infrastructure, never shipped, never on the product path) for the
 * data side of the hot path: a CPU restatement of the reference's 4GI cohort
 * generator.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Follows (reference root = OliverDOU776/Hybrid-ODE-for-GLP-1-and-Glucose):
 *   data/generate4GI.py:15-64    parameter set (T2DM / HV)
 *   data/generate4GI.py:65-71    baselines
 *   data/generate4GI.py:73-157   model_equations (8 states, piecewise-constant meal input)
 *   data/generate4GI.py:159-212  simulate: grid, per-interval meal rate, one IVP per grid interval,
 *                                amounts -> concentrations
 * The reference integrates every interval with scipy.integrate.odeint (LSODA, rtol = atol = 1.49e-8);
 * the restatement integrates the same IVPs with DP5(4) in fp64 at tighter tolerances.  Pinned by
 * tests/golden/g8_* (outputs of the reference's own FourGIModel.simulate, tools/capture_golden_data.py).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define NS 8

typedef struct {
    /* fixed parameters */
    double CLglc, CLglci, Qglc, VCglc, VPglc, CLins, VCins, Ke0ins, VCglp, VM_GLP, KM_GLP, CLglg, VCglg, CLgip, VCgip,
        Qgip, VPgip, GLCINS_S, EMAX_1, EC50_1, HILL_1, EMAX_4, EC50_4, FDGLP, FDGIP, FDGLG;
    int hv;
    /* per-subject */
    double Bglc, Bins, Bglp, Bglg, Bgip;
    double S0ins, S0glg, KINglc, KINins, KINglp, KINglg, KINgip;
} Subject;

static void subject_init(Subject *s, int hv, const double *bsl) {
    s->hv = hv;
    s->CLglc = hv ? 5.36 : 1.72;
    s->CLglci = hv ? 0.072 : 0.0256;
    s->Qglc = 26.5; s->VCglc = 9.33; s->VPglc = 8.56;
    s->CLins = 73.2; s->VCins = 6.09; s->Ke0ins = exp(-0.159);
    s->VCglp = 16.0; s->VM_GLP = exp(7.97); s->KM_GLP = exp(4.91);
    s->CLglg = 453.2; s->VCglg = 64.6;
    s->CLgip = 86.8; s->VCgip = 9.21; s->Qgip = 49.4; s->VPgip = 22.8;
    s->GLCINS_S = 2.46; s->EMAX_1 = exp(2.37); s->EC50_1 = exp(3.29); s->HILL_1 = 1.79;
    s->EMAX_4 = 6.73; s->EC50_4 = exp(4.59);
    s->FDGLP = 0.0102; s->FDGIP = 0.0343; s->FDGLG = 0.00329;
    s->Bglc = bsl[0]; s->Bins = bsl[1]; s->Bglp = bsl[2]; s->Bglg = bsl[3]; s->Bgip = bsl[4];
    /* baseline-dependent constants of generate4GI.py:94-116 */
    double r0 = pow(s->Bglp / s->EC50_1, s->HILL_1);
    s->S0ins = s->EMAX_1 * r0 / (1.0 + r0);
    double q0 = s->Bglg / s->EC50_4;
    s->S0glg = s->EMAX_4 * q0 / (1.0 + q0);
    s->KINglc = s->Bglc * (s->CLglc + s->CLglci * s->Bins);
    s->KINins = s->Bins * s->CLins / (1.0 + s->S0ins * pow(s->Bglc, s->GLCINS_S));
    s->KINglp = s->VM_GLP * s->Bglp * s->VCglp / (s->KM_GLP + s->Bglp);
    s->KINglg = s->Bglg * s->CLglg;
    s->KINgip = s->Bgip * s->CLgip;
}

static void subject_y0(const Subject *s, double *y) { /* generate4GI.py:175-184 */
    y[0] = s->Bglc * s->VCglc; y[1] = s->Bins * s->VCins; y[2] = s->Bglp * s->VCglp; y[3] = s->Bglg * s->VCglg;
    y[4] = s->Bgip * s->VCgip; y[5] = s->Bglc * s->VPglc; y[6] = s->Bins; y[7] = s->Bgip * s->VPgip;
}

static void rhs(const Subject *s, const double *y, double meal, double *d) { /* generate4GI.py:73-157 */
    double Cglc = y[0] / s->VCglc, Cins = y[1] / s->VCins, Cglp = y[2] / s->VCglp, Cglg = y[3] / s->VCglg;
    double r = pow(Cglp / s->EC50_1, s->HILL_1);
    double Sins = s->EMAX_1 * r / (1.0 + r);
    double q = Cglg / s->EC50_4;
    double Sglg = s->EMAX_4 * q / (1.0 + q);
    double eff_glg_on_glc = (1.0 + Sglg) / (1.0 + s->S0glg);
    double p2 = Cglc >= s->Bglc ? 0.925 : (s->hv ? 0.327 : 0.0);
    double eff_glc_on_glg = Cglc > 0.0 ? pow(s->Bglc / Cglc, p2) : 1.0;
    double me = meal * 10.0;
    double fglp = me > 0.0 ? s->FDGLP * me : 0.0, fgip = me > 0.0 ? s->FDGIP * me : 0.0,
           fglg = me > 0.0 ? s->FDGLG * me : 0.0;
    double k27 = s->Qglc / s->VCglc, k72 = s->Qglc / s->VPglc, k612 = s->Qgip / s->VCgip, k126 = s->Qgip / s->VPgip;
    d[0] = meal + s->KINglc * eff_glg_on_glc - k27 * y[0] + k72 * y[5] - (s->CLglc / s->VCglc) * y[0] -
           (s->CLglci * y[6] / s->VCglc) * y[0];
    d[1] = s->KINins * (1.0 + Sins * pow(Cglc, s->GLCINS_S)) - (s->CLins / s->VCins) * y[1];
    d[2] = s->KINglp * (1.0 + fglp) - s->VM_GLP * Cglp / (s->KM_GLP + Cglp);
    d[3] = s->KINglg * (1.0 + fglg) * eff_glc_on_glg - (s->CLglg / s->VCglg) * y[3];
    d[4] = s->KINgip * (1.0 + fgip) - (s->CLgip / s->VCgip) * y[4] - k612 * y[4] + k126 * y[7];
    d[5] = k27 * y[0] - k72 * y[5];
    d[6] = s->Ke0ins * (Cins - y[6]);
    d[7] = k612 * y[4] - k126 * y[7];
}

/* Dormand-Prince 5(4) coefficients (scipy/integrate/_ivp/rk.py:377-401) */
static const double A21 = 1.0 / 5, A31 = 3.0 / 40, A32 = 9.0 / 40, A41 = 44.0 / 45, A42 = -56.0 / 15, A43 = 32.0 / 9,
                    A51 = 19372.0 / 6561, A52 = -25360.0 / 2187, A53 = 64448.0 / 6561, A54 = -212.0 / 729,
                    A61 = 9017.0 / 3168, A62 = -355.0 / 33, A63 = 46732.0 / 5247, A64 = 49.0 / 176,
                    A65 = -5103.0 / 18656, B1 = 35.0 / 384, B3 = 500.0 / 1113, B4 = 125.0 / 192, B5 = -2187.0 / 6784,
                    B6 = 11.0 / 84, E1 = -71.0 / 57600, E3 = 71.0 / 16695, E4 = -71.0 / 1920, E5 = 17253.0 / 339200,
                    E6 = -22.0 / 525, E7 = 1.0 / 40;

/* integrate one grid interval of length H with a constant meal rate; returns 0 ok / 1 budget / 2 underflow / 3 nan */
static int interval(const Subject *s, double *y, double H, double meal, double rtol, double atol, int max_steps,
                    double *h_io, int64_t *nsteps) {
    double k1[NS], k2[NS], k3[NS], k4[NS], k5[NS], k6[NS], k7[NS], w[NS], yn[NS];
    double tau = 0.0, h = *h_io;
    if (!(h > 0.0) || h > H) h = H;
    rhs(s, y, meal, k1);
    for (int it = 0; it < max_steps; ++it) {
        int last = 0;
        if (tau + h >= H * (1.0 - 1e-14)) { h = H - tau; last = 1; }
        if (h < 1e-14 * H) return 2;
        for (int i = 0; i < NS; ++i) w[i] = y[i] + h * (A21 * k1[i]);
        rhs(s, w, meal, k2);
        for (int i = 0; i < NS; ++i) w[i] = y[i] + h * (A31 * k1[i] + A32 * k2[i]);
        rhs(s, w, meal, k3);
        for (int i = 0; i < NS; ++i) w[i] = y[i] + h * (A41 * k1[i] + A42 * k2[i] + A43 * k3[i]);
        rhs(s, w, meal, k4);
        for (int i = 0; i < NS; ++i) w[i] = y[i] + h * (A51 * k1[i] + A52 * k2[i] + A53 * k3[i] + A54 * k4[i]);
        rhs(s, w, meal, k5);
        for (int i = 0; i < NS; ++i)
            w[i] = y[i] + h * (A61 * k1[i] + A62 * k2[i] + A63 * k3[i] + A64 * k4[i] + A65 * k5[i]);
        rhs(s, w, meal, k6);
        for (int i = 0; i < NS; ++i)
            yn[i] = y[i] + h * (B1 * k1[i] + B3 * k3[i] + B4 * k4[i] + B5 * k5[i] + B6 * k6[i]);
        rhs(s, yn, meal, k7);
        double acc = 0.0;
        int finite = 1;
        for (int i = 0; i < NS; ++i) {
            double e = h * (E1 * k1[i] + E3 * k3[i] + E4 * k4[i] + E5 * k5[i] + E6 * k6[i] + E7 * k7[i]);
            double sc = atol + rtol * fmax(fabs(y[i]), fabs(yn[i]));
            acc += (e / sc) * (e / sc);
            if (!isfinite(yn[i])) finite = 0;
        }
        if (!finite) return 3;
        double err = sqrt(acc / NS);
        if (err < 1.0) {
            double fac = err == 0.0 ? 10.0 : fmin(10.0, 0.9 * pow(err, -0.2));
            for (int i = 0; i < NS; ++i) { y[i] = yn[i]; k1[i] = k7[i]; }
            ++*nsteps;
            if (last) { *h_io = h * fac; return 0; }
            tau += h;
            h *= fac;
        } else {
            h *= fmax(0.2, 0.9 * pow(err, -0.2));
        }
    }
    return 1;
}

/* FourGIModel.simulate for a cohort (generate4GI.py:159-212): conc[B,T,5] = glucose, insulin, glp1, glucagon, gip.
 * bsl[B,5]; meal_time/meal_size: [n_meals] shared (meals_per_subject = 0) or [B,n_meals]. */
int fourgi_oracle_simulate(int B, int T, double interval_min, int patient_type, const double *bsl, int n_meals,
                           const double *meal_time, const double *meal_size, int meals_per_subject, double rtol,
                           double atol, int max_steps, double *conc, int32_t *status, int64_t *nsteps_total) {
    int64_t total = 0;
    for (int b = 0; b < B; ++b) {
        Subject s;
        double y[NS];
        subject_init(&s, patient_type, bsl + 5 * (size_t)b);
        subject_y0(&s, y);
        const double *mt = meal_time + (meals_per_subject ? (size_t)b * n_meals : 0);
        const double *ms = meal_size + (meals_per_subject ? (size_t)b * n_meals : 0);
        double h = 0.0;
        int st = 0;
        double *out = conc + (size_t)b * T * 5;
        for (int k = 0; k < T; ++k) {
            if (st == 0) {
                out[5 * k + 0] = y[0] / s.VCglc; out[5 * k + 1] = y[1] / s.VCins; out[5 * k + 2] = y[2] / s.VCglp;
                out[5 * k + 3] = y[3] / s.VCglg; out[5 * k + 4] = y[4] / s.VCgip;
            } else {
                for (int c = 0; c < 5; ++c) out[5 * k + c] = 0.0;
            }
            if (k == T - 1 || st != 0) continue;
            double t0 = (k * interval_min) / 60.0, t1 = ((k + 1) * interval_min) / 60.0;
            double rate = 0.0;
            for (int m = 0; m < n_meals; ++m)
                if (t0 <= mt[m] && mt[m] < t1) rate = ms[m] / (t1 - t0);
            st = interval(&s, y, t1 - t0, rate, rtol, atol, max_steps, &h, &total);
        }
        if (status) status[b] = st;
    }
    if (nsteps_total) *nsteps_total = total;
    return 0;
}

/* RHS alone (for kernel-level checks): y[B,8], meal[B] -> d[B,8] */
int fourgi_oracle_rhs(int B, int patient_type, const double *bsl, const double *y, const double *meal, double *d) {
    for (int b = 0; b < B; ++b) {
        Subject s;
        subject_init(&s, patient_type, bsl + 5 * (size_t)b);
        rhs(&s, y + NS * (size_t)b, meal[b], d + NS * (size_t)b);
    }
    return 0;
}
