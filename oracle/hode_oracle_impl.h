/*
 * ORACLE -- test infrastructure only.  NOT part of the product path.
 *
 * Type-generic body of the CPU restatement; included twice by hode_oracle.c with
 *   REAL = float  / SFX(x) = x##_f32   and   REAL = double / SFX(x) = x##_f64.
 *
 * What is restated (citations are into /root/reference unless prefixed scipy/):
 *   rhs            models/ode_core.py:104-161 (6 mechanistic terms)
 *                  models/nn_residual.py:114-151 (input row [t,x(6),glp1,tvns] -> MLP); activation :50-56:
 *                  relu (default; the only one HybridODENN ever builds), tanh, elu (alpha 1), leaky_relu (0.1)
 *                  models/hybrid_ode_nn.py:108-134 (sum; glp1 := x[3]; tvns from inputs)
 *   input lerp     models/hybrid_ode_nn.py:210-231 (piecewise linear on the grid; dim()==1 -> constant)
 *   DP5(4) step    scipy/integrate/_ivp/rk.py:14-72 (rk_step), :111-176 (_step_impl controller),
 *                  :377-401 (tableau), scipy/integrate/_ivp/common.py:63-135 (norm, initial step)
 * The "grid" integrator below is the integrator the HIP product implements: the same RK45
 * controller, but every grid point is a mandatory step boundary, the step
 * size proposal and the FSAL derivative are carried across grid points.
 */

/* ---- parameter layout helpers ------------------------------------------------------------ */
/* flat nn_p = PyTorch parameters() order (models/nn_residual.py:59-78):
 *   W1[H,9] b1[H] (W[H,H] b[H]) x (L-1)  Wout[6,H] bout[6]                                   */

/* Activation codes (include/hode.h HODE_ACT_*); the `L` argument of every entry point carries one in bits 8..15. */
#ifndef HODE_ORACLE_ACT_DEFINED
#define HODE_ORACLE_ACT_DEFINED
enum { ORACLE_ACT_RELU = 0, ORACLE_ACT_TANH = 1, ORACLE_ACT_ELU = 2, ORACLE_ACT_LEAKY = 3 };
#endif

typedef struct {
    int H, L, act;
    const REAL *W[HODE_MAXL + 1];
    const REAL *b[HODE_MAXL + 1];
    int in_dim[HODE_MAXL + 1], out_dim[HODE_MAXL + 1];
    int w_off[HODE_MAXL + 1], b_off[HODE_MAXL + 1];
} SFX(mlp_t);

static int SFX(mlp_bind)(SFX(mlp_t) *m, const REAL *p, int H, int L)
{
    const int act = (L >> 8) & 0xff;
    L &= 0xff;
    if (L < 1 || L > HODE_MAXL || H < 1 || H > HODE_MAXH || act > ORACLE_ACT_LEAKY) return -1;
    m->H = H; m->L = L; m->act = act;
    int off = 0;
    for (int l = 0; l <= L; ++l) {
        int in = (l == 0) ? 9 : H, out = (l == L) ? 6 : H;
        m->in_dim[l] = in; m->out_dim[l] = out;
        m->w_off[l] = off; m->W[l] = p + off; off += in * out;
        m->b_off[l] = off; m->b[l] = p + off; off += out;
    }
    return off;
}

/* activation and its derivative from the POST-activation value h (torch: nn.ReLU / Tanh / ELU(alpha 1) / LeakyReLU(0.1);
 * h > 0 iff the pre-activation is > 0 for all four; ELU'(x <= 0) = exp(x) = h + 1) */
static REAL SFX(act_f)(REAL s, int act)
{
    switch (act) {
    case ORACLE_ACT_TANH: return RTANH(s);
    case ORACLE_ACT_ELU: return s > 0 ? s : REXPM1(s);
    case ORACLE_ACT_LEAKY: return s > 0 ? s : (REAL)0.1 * s;
    default: return s > 0 ? s : (REAL)0;
    }
}
static REAL SFX(act_d)(REAL h, int act)
{
    switch (act) {
    case ORACLE_ACT_TANH: return (REAL)1 - h * h;
    case ORACLE_ACT_ELU: return h > 0 ? (REAL)1 : h + (REAL)1;
    case ORACLE_ACT_LEAKY: return h > 0 ? (REAL)1 : (REAL)0.1;
    default: return h > 0 ? (REAL)1 : (REAL)0;
    }
}

/* ---- RHS --------------------------------------------------------------------------------- */
/* act[l] (l=0..L) holds the INPUT of layer l: act[0] = 9 inputs, act[l>0] = relu outputs.    */
static void SFX(rhs_one)(const SFX(mlp_t) *m, const REAL *ode, REAL t, const REAL *x,
                         REAL meal, REAL tvns, REAL gd, int use_gd, REAL *f,
                         REAL act[HODE_MAXL + 1][HODE_MAXH])
{
    const REAL a_GI = ode[0], k_I = ode[1], rho = ode[2], G_b = ode[3], I_b = ode[4],
               E_max = ode[5], EC_50 = ode[6], Glu_b = ode[7], V_max = ode[8], K_m = ode[9],
               k_L = ode[10], k_GE0 = ode[11], IGD_50 = ode[12], g = ode[13], p_7 = ode[14],
               p_8 = ode[15], p_9 = ode[16];
    const REAL G = x[0], I = x[1], Glu = x[2], GLP1 = x[3], FFA = x[5];
    /* ode_core.py:124-125 */
    REAL Pi = (REAL)1 + rho * GLP1;
    REAL dI = Pi * a_GI * (G - G_b) - k_I * (I - I_b);
    /* ode_core.py:129-130 */
    REAL eff = E_max * (GLP1 / (EC_50 + GLP1));
    REAL dGlu = -eff * (Glu - Glu_b);
    /* ode_core.py:134-135 */
    REAL dGLP1 = V_max * (G / (K_m + G)) - k_L * GLP1;
    /* ode_core.py:139-140 (torch.pow) */
    REAL gde = 0;
    if (use_gd) {
        REAL u = RPOW(gd, g), v = RPOW(IGD_50, g);
        gde = u / (v + u);
    }
    REAL k_GE = k_GE0 * ((REAL)1 - gde);
    /* ode_core.py:144 */
    REAL dFFA = -p_7 * FFA - p_8 * I * FFA + p_9 * G * FFA;
    /* ode_core.py:148-150 */
    REAL dG = meal - (REAL)0.01 * (I - I_b) + (REAL)0.005 * (Glu - Glu_b) - k_GE * G;

    /* nn_residual.py:138-143: [t, G,I,Glu,GLP1,GE,FFA, glp1(=x[3]), tvns] */
    REAL *in = act[0];
    in[0] = t;
    for (int i = 0; i < 6; ++i) in[1 + i] = x[i];
    in[7] = x[3];
    in[8] = tvns;
    REAL out6[6];
    for (int l = 0; l <= m->L; ++l) {
        const REAL *W = m->W[l], *b = m->b[l];
        int nin = m->in_dim[l], nout = m->out_dim[l];
        REAL *dst = (l == m->L) ? out6 : act[l + 1];
        for (int j = 0; j < nout; ++j) {
            REAL s = b[j];
            for (int k = 0; k < nin; ++k) s += W[j * nin + k] * act[l][k];
            dst[j] = (l == m->L) ? s : SFX(act_f)(s, m->act);
        }
    }
    f[0] = dG + out6[0];
    f[1] = dI + out6[1];
    f[2] = dGlu + out6[2];
    f[3] = dGLP1 + out6[3];
    f[4] = (REAL)0 + out6[4];   /* ode_core.py:153 dGE = 0 */
    f[5] = dFFA + out6[5];
}

/* VJP of rhs_one: given lam (6) returns gx (6) += J^T lam, accumulates gnn, gode (may be NULL). */
static void SFX(rhs_vjp_one)(const SFX(mlp_t) *m, const REAL *ode, REAL t, const REAL *x,
                             REAL meal, REAL tvns, REAL gd, int use_gd, const REAL *lam,
                             REAL *gx, ACC *gnn, ACC *gode, REAL scale)
{
    REAL act[HODE_MAXL + 1][HODE_MAXH];
    REAL f[6];
    SFX(rhs_one)(m, ode, t, x, meal, tvns, gd, use_gd, f, act);
    const REAL a_GI = ode[0], k_I = ode[1], rho = ode[2], G_b = ode[3], I_b = ode[4],
               E_max = ode[5], EC_50 = ode[6], Glu_b = ode[7], V_max = ode[8], K_m = ode[9],
               k_L = ode[10], k_GE0 = ode[11], IGD_50 = ode[12], g = ode[13], p_7 = ode[14],
               p_8 = ode[15], p_9 = ode[16];
    const REAL G = x[0], I = x[1], Glu = x[2], GLP1 = x[3], FFA = x[5];
    const REAL lG = lam[0], lI = lam[1], lGlu = lam[2], lGLP = lam[3], lF = lam[5];
    REAL gde = 0, u = 0, v = 1;
    if (use_gd) { u = RPOW(gd, g); v = RPOW(IGD_50, g); gde = u / (v + u); }
    REAL k_GE = k_GE0 * ((REAL)1 - gde);
    REAL Pi = (REAL)1 + rho * GLP1;
    REAL den1 = EC_50 + GLP1, den2 = K_m + G;
    /* mechanistic J^T lam */
    REAL o[6];
    o[0] = -k_GE * lG + Pi * a_GI * lI + V_max * K_m / (den2 * den2) * lGLP + p_9 * FFA * lF;
    o[1] = (REAL)-0.01 * lG - k_I * lI - p_8 * FFA * lF;
    o[2] = (REAL)0.005 * lG - E_max * GLP1 / den1 * lGlu;
    o[3] = rho * a_GI * (G - G_b) * lI - E_max * EC_50 / (den1 * den1) * (Glu - Glu_b) * lGlu - k_L * lGLP;
    o[4] = 0;
    o[5] = (-p_7 - p_8 * I + p_9 * G) * lF;
    if (gode) {
        gode[0] += scale * lI * Pi * (G - G_b);
        gode[1] += scale * -lI * (I - I_b);
        gode[2] += scale * lI * GLP1 * a_GI * (G - G_b);
        gode[3] += scale * -lI * Pi * a_GI;
        gode[4] += scale * (lI * k_I + lG * (REAL)0.01);
        gode[5] += scale * -lGlu * GLP1 / den1 * (Glu - Glu_b);
        gode[6] += scale * lGlu * E_max * GLP1 / (den1 * den1) * (Glu - Glu_b);
        gode[7] += scale * (lGlu * E_max * GLP1 / den1 - lG * (REAL)0.005);
        gode[8] += scale * lGLP * G / den2;
        gode[9] += scale * -lGLP * V_max * G / (den2 * den2);
        gode[10] += scale * -lGLP * GLP1;
        gode[11] += scale * -lG * G * ((REAL)1 - gde);
        if (use_gd && gd > 0) {
            REAL s2 = (v + u) * (v + u);
            gode[12] += scale * lG * k_GE0 * G * (-u * g * RPOW(IGD_50, g - 1) / s2);
            gode[13] += scale * lG * k_GE0 * G * (u * v * (RLOG(gd) - RLOG(IGD_50)) / s2);
        }
        gode[14] += scale * -lF * FFA;
        gode[15] += scale * -lF * I * FFA;
        gode[16] += scale * lF * G * FFA;
    }
    /* MLP backward */
    REAL delta[HODE_MAXH], prev[HODE_MAXH];
    for (int j = 0; j < 6; ++j) delta[j] = lam[j];
    for (int l = m->L; l >= 0; --l) {
        const REAL *W = m->W[l];
        int nin = m->in_dim[l], nout = m->out_dim[l];
        for (int k = 0; k < nin; ++k) prev[k] = 0;
        for (int j = 0; j < nout; ++j) {
            REAL d = delta[j];
            if (gnn) {
                gnn[m->b_off[l] + j] += scale * d;
                for (int k = 0; k < nin; ++k) gnn[m->w_off[l] + j * nin + k] += scale * d * act[l][k];
            }
            for (int k = 0; k < nin; ++k) prev[k] += W[j * nin + k] * d;
        }
        if (l > 0)
            for (int k = 0; k < nin; ++k) delta[k] = prev[k] * SFX(act_d)(act[l][k], m->act);
    }
    /* d/dx through input row: x[i] -> in[1+i], x[3] also -> in[7] */
    for (int i = 0; i < 6; ++i) o[i] += prev[1 + i];
    o[3] += prev[7];
    for (int i = 0; i < 6; ++i) gx[i] += o[i];
}

/* ---- batched RHS entry points -------------------------------------------------------------- */
int SFX(hode_oracle_rhs)(int B, const REAL *x, const REAL *t, const REAL *meal, const REAL *tvns,
                         const REAL *gd, const REAL *ode, const REAL *nn_p, int H, int L, REAL *out)
{
    SFX(mlp_t) m;
    if (SFX(mlp_bind)(&m, nn_p, H, L) < 0) return -1;
    REAL act[HODE_MAXL + 1][HODE_MAXH];
    for (int b = 0; b < B; ++b)
        SFX(rhs_one)(&m, ode, t ? t[b] : 0, x + 6 * b, meal ? meal[b] : 0, tvns ? tvns[b] : 0,
                     gd ? gd[b] : 0, gd != NULL, out + 6 * b, act);
    return 0;
}

int SFX(hode_oracle_rhs_vjp)(int B, const REAL *x, const REAL *t, const REAL *meal, const REAL *tvns,
                             const REAL *gd, const REAL *ode, const REAL *nn_p, int H, int L,
                             const REAL *gout, REAL *gx, REAL *gnn, REAL *gode)
{
    SFX(mlp_t) m;
    int P = SFX(mlp_bind)(&m, nn_p, H, L);
    if (P < 0) return -1;
    ACC *an = (ACC *)calloc((size_t)P, sizeof(ACC));
    ACC ao[17] = {0};
    for (int b = 0; b < B; ++b) {
        REAL g6[6] = {0, 0, 0, 0, 0, 0};
        SFX(rhs_vjp_one)(&m, ode, t ? t[b] : 0, x + 6 * b, meal ? meal[b] : 0, tvns ? tvns[b] : 0,
                         gd ? gd[b] : 0, gd != NULL, gout + 6 * b, g6, gnn ? an : NULL,
                         gode ? ao : NULL, (REAL)1);
        for (int i = 0; i < 6; ++i) gx[6 * b + i] = g6[i];
    }
    if (gnn) for (int i = 0; i < P; ++i) gnn[i] = (REAL)an[i];
    if (gode) for (int i = 0; i < 17; ++i) gode[i] = (REAL)ao[i];
    free(an);
    return 0;
}

/* ---- forcing on one grid interval (hybrid_ode_nn.py:217-231) ------------------------------- */
typedef struct {
    REAL t0, t1;          /* interval ends */
    REAL m0, m1, v0, v1, d0, d1;   /* meal / tvns / gd at the ends */
    int use_gd;
} SFX(seg_t);

static inline void SFX(seg_eval)(const SFX(seg_t) *s, REAL t, REAL *meal, REAL *tvns, REAL *gd)
{
    REAL a = (t - s->t0) / (s->t1 - s->t0);
    *meal = s->m0 + a * (s->m1 - s->m0);
    *tvns = s->v0 + a * (s->v1 - s->v0);
    *gd = s->d0 + a * (s->d1 - s->d0);
}

static inline REAL SFX(inp_at)(const REAL *p, int mode, int b, int T, int k)
{
    if (!p || mode == 0) return 0;
    return (mode == 1) ? p[b] : p[(size_t)b * T + k];
}

/* ---- Dormand-Prince 5(4) tableau (scipy rk.py:377-401) -------------------------------------- */
static const double SFX(dpC)[7] = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1, 1};
static const double SFX(dpA)[7][6] = {
    {0},
    {1.0 / 5},
    {3.0 / 40, 9.0 / 40},
    {44.0 / 45, -56.0 / 15, 32.0 / 9},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656},
    {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};   /* row 6 = B */
static const double SFX(dpE)[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920,
                                   17253.0 / 339200, -22.0 / 525, 1.0 / 40};

#ifndef HODE_SEG_CLOSED
#define HODE_SEG_CLOSED (1 << 30)   /* bit 30 of tape seg: the step ended exactly on the grid point closing its interval */
#endif
typedef struct { REAL t, h, y[6]; int seg; } SFX(tape_t);

/*
 * Forward solve, "grid" integrator.   method: 0 = DP5(4) adaptive, 1 = classic RK4 with one
 * step per grid interval (BASELINE config 1).  status: 0 ok, 1 max_steps hit, 2 step underflow,
 * 3 non-finite state.  On failure the remaining rows of y stay zero (hybrid_ode_nn.py:243-256).
 * tape (optional): accepted steps of each trajectory, [B][max_steps].
 */
int SFX(hode_oracle_solve)(int B, int T, const REAL *x0, const REAL *tg, int t_batched,
                           const REAL *meal, int meal_mode, const REAL *tvns, int tvns_mode,
                           const REAL *gd, int gd_mode, const REAL *ode, const REAL *nn_p, int H,
                           int L, int method, double rtol, double atol, int max_steps, REAL *y,
                           int *status, int *nsteps, int *nfev, void *tape_v)
{
    SFX(mlp_t) m;
    if (SFX(mlp_bind)(&m, nn_p, H, L) < 0) return -1;
    if (T < 1 || B < 0) return -1;
    SFX(tape_t) *tape = (SFX(tape_t) *)tape_v;
    REAL act[HODE_MAXL + 1][HODE_MAXH];
    const REAL EPSR = (sizeof(REAL) == 4) ? (REAL)1.1920929e-7 : (REAL)2.220446049250313e-16;
    for (int b = 0; b < B; ++b) {
        const REAL *tb = t_batched ? tg + (size_t)b * T : tg;
        REAL *yb = y + (size_t)b * T * 6;
        REAL yc[6], K[7][6];
        int st = 0, ns = 0, nf = 0;
        memset(yb, 0, sizeof(REAL) * 6 * T);
        for (int i = 0; i < 6; ++i) yb[i] = yc[i] = x0[6 * b + i];
        REAL h_abs = 0;
        int have_f = 0;
        for (int k = 0; k + 1 < T && st == 0; ++k) {
            SFX(seg_t) s;
            s.t0 = tb[k]; s.t1 = tb[k + 1];
            s.m0 = SFX(inp_at)(meal, meal_mode, b, T, k); s.m1 = SFX(inp_at)(meal, meal_mode, b, T, k + 1);
            s.v0 = SFX(inp_at)(tvns, tvns_mode, b, T, k); s.v1 = SFX(inp_at)(tvns, tvns_mode, b, T, k + 1);
            s.d0 = SFX(inp_at)(gd, gd_mode, b, T, k); s.d1 = SFX(inp_at)(gd, gd_mode, b, T, k + 1);
            s.use_gd = (gd != NULL && gd_mode != 0);
            REAL len = s.t1 - s.t0;
            if (!(len > 0)) { for (int i = 0; i < 6; ++i) yb[6 * (k + 1) + i] = yc[i]; continue; }
            REAL tc = s.t0, mm, vv, dd;
            if (method == 1) {                       /* classic RK4, one step per interval */
                if (ns >= max_steps) { st = 1; break; }   /* budget < T-1: report, rows from here on stay zero */
                REAL k1[6], k2[6], k3[6], k4[6], yt[6];
                REAL hh = len;
                SFX(seg_eval)(&s, tc, &mm, &vv, &dd);
                SFX(rhs_one)(&m, ode, tc, yc, mm, vv, dd, s.use_gd, k1, act);
                for (int i = 0; i < 6; ++i) yt[i] = yc[i] + (REAL)0.5 * hh * k1[i];
                SFX(seg_eval)(&s, tc + (REAL)0.5 * hh, &mm, &vv, &dd);
                SFX(rhs_one)(&m, ode, tc + (REAL)0.5 * hh, yt, mm, vv, dd, s.use_gd, k2, act);
                for (int i = 0; i < 6; ++i) yt[i] = yc[i] + (REAL)0.5 * hh * k2[i];
                SFX(rhs_one)(&m, ode, tc + (REAL)0.5 * hh, yt, mm, vv, dd, s.use_gd, k3, act);
                for (int i = 0; i < 6; ++i) yt[i] = yc[i] + hh * k3[i];
                SFX(seg_eval)(&s, s.t1, &mm, &vv, &dd);
                SFX(rhs_one)(&m, ode, s.t1, yt, mm, vv, dd, s.use_gd, k4, act);
                REAL ynew[6];
                int fin4 = 1;
                for (int i = 0; i < 6; ++i) {
                    ynew[i] = yc[i] + hh / (REAL)6 * (k1[i] + (REAL)2 * k2[i] + (REAL)2 * k3[i] + k4[i]);
                    if (!isfinite((double)ynew[i])) fin4 = 0;
                }
                nf += 4;
                /* a step whose result is not finite is NOT an accepted step: it never reaches the tape (its row is never
                   written, so the adjoint must not walk its non-finite stage records) -- status 3, rows from here on zero */
                if (!fin4) { st = 3; break; }
                if (tape && ns < max_steps) {
                    SFX(tape_t) *e = tape + (size_t)b * max_steps + ns;
                    e->t = tc; e->h = hh; e->seg = k | HODE_SEG_CLOSED; for (int i = 0; i < 6; ++i) e->y[i] = yc[i];
                }
                for (int i = 0; i < 6; ++i) yc[i] = ynew[i];
                ns += 1;
            } else {
                if (!have_f) {                       /* first RHS + Hairer initial step (common.py:68-135) */
                    SFX(seg_eval)(&s, tc, &mm, &vv, &dd);
                    SFX(rhs_one)(&m, ode, tc, yc, mm, vv, dd, s.use_gd, K[0], act);
                    nf++;
                    double d0 = 0, d1 = 0, d2 = 0, sc[6];
                    for (int i = 0; i < 6; ++i) {
                        sc[i] = atol + fabs((double)yc[i]) * rtol;
                        d0 += ((double)yc[i] / sc[i]) * ((double)yc[i] / sc[i]);
                        d1 += ((double)K[0][i] / sc[i]) * ((double)K[0][i] / sc[i]);
                    }
                    d0 = sqrt(d0 / 6); d1 = sqrt(d1 / 6);
                    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
                    if (h0 > (double)len) h0 = (double)len;
                    REAL y1[6], f1[6];
                    for (int i = 0; i < 6; ++i) y1[i] = yc[i] + (REAL)h0 * K[0][i];
                    SFX(seg_eval)(&s, tc + (REAL)h0, &mm, &vv, &dd);
                    SFX(rhs_one)(&m, ode, tc + (REAL)h0, y1, mm, vv, dd, s.use_gd, f1, act);
                    nf++;
                    for (int i = 0; i < 6; ++i) {
                        double q = ((double)f1[i] - (double)K[0][i]) / sc[i];
                        d2 += q * q;
                    }
                    d2 = sqrt(d2 / 6) / h0;
                    double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3)
                                                             : pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
                    h_abs = (REAL)fmin(fmin(100 * h0, h1), (double)len);
                    have_f = 1;
                }
                while (tc < s.t1 && st == 0) {
                    int rejected = 0;
                    for (;;) {                       /* rk.py:126-176 */
                        if (ns >= max_steps) { st = 1; break; }
                        REAL min_step = (REAL)10 * EPSR * (REAL)fmax(fabs((double)tc), (double)1e-30);
                        if (h_abs < min_step) { st = 2; break; }
                        REAL h = h_abs, tn = tc + h;
                        int clipped = 0;
                        /* grid point = mandatory boundary; stretch by <=1% to avoid sliver steps */
                        if (tn >= s.t1 || (s.t1 - tn) < (REAL)0.01 * h) { tn = s.t1; h = tn - tc; clipped = 1; }
                        REAL yt[6], yn[6];
                        for (int sI = 1; sI < 6; ++sI) {
                            for (int i = 0; i < 6; ++i) {
                                REAL acc = 0;
                                for (int j = 0; j < sI; ++j) acc += (REAL)SFX(dpA)[sI][j] * K[j][i];
                                yt[i] = yc[i] + h * acc;
                            }
                            REAL ts = tc + (REAL)SFX(dpC)[sI] * h;
                            SFX(seg_eval)(&s, ts, &mm, &vv, &dd);
                            SFX(rhs_one)(&m, ode, ts, yt, mm, vv, dd, s.use_gd, K[sI], act);
                        }
                        for (int i = 0; i < 6; ++i) {
                            REAL acc = 0;
                            for (int j = 0; j < 6; ++j) acc += (REAL)SFX(dpA)[6][j] * K[j][i];
                            yn[i] = yc[i] + h * acc;
                        }
                        SFX(seg_eval)(&s, tn, &mm, &vv, &dd);
                        SFX(rhs_one)(&m, ode, tn, yn, mm, vv, dd, s.use_gd, K[6], act);
                        nf += 6;
                        double en = 0;
                        int finite = 1;
                        for (int i = 0; i < 6; ++i) {
                            REAL e = 0;
                            for (int j = 0; j < 7; ++j) e += (REAL)SFX(dpE)[j] * K[j][i];
                            double scl = atol + fmax(fabs((double)yc[i]), fabs((double)yn[i])) * rtol;
                            double q = (double)(e * h) / scl;
                            en += q * q;
                            if (!isfinite((double)yn[i])) finite = 0;
                        }
                        en = sqrt(en / 6);
                        if (!finite || !isfinite(en)) en = 1e30;   /* treat as rejection */
                        if (en < 1) {
                            double fac = (en == 0) ? 10.0 : fmin(10.0, 0.9 * pow(en, -0.2));
                            if (rejected) fac = fmin(1.0, fac);
                            if (tape) {
                                SFX(tape_t) *e = tape + (size_t)b * max_steps + ns;
                                e->t = tc; e->h = h; e->seg = k | (clipped ? HODE_SEG_CLOSED : 0); for (int i = 0; i < 6; ++i) e->y[i] = yc[i];
                            }
                            /* a clipped step must not shrink the carried proposal */
                            REAL hn = h * (REAL)fac;
                            h_abs = (clipped && hn < h_abs) ? h_abs : hn;
                            for (int i = 0; i < 6; ++i) { yc[i] = yn[i]; K[0][i] = K[6][i]; }
                            tc = tn; ns++;
                            break;
                        } else {
                            h_abs = h * (REAL)fmax(0.2, 0.9 * pow(en, -0.2));
                            rejected = 1;
                            if (en >= 1e30 && !(h_abs > min_step)) { st = 3; break; }
                        }
                    }
                }
            }
            if (st == 0) {
                int fin = 1;
                for (int i = 0; i < 6; ++i) if (!isfinite((double)yc[i])) fin = 0;
                if (!fin) st = 3;
                else for (int i = 0; i < 6; ++i) yb[6 * (k + 1) + i] = yc[i];
            }
        }
        if (status) status[b] = st;
        if (nsteps) nsteps[b] = ns;
        if (nfev) nfev[b] = nf;
    }
    return 0;
}

/*
 * Discrete adjoint ("discretise-then-differentiate") of hode_oracle_solve over the accepted
 * steps recorded on the tape (step sizes treated as constants).  Gradient accumulators are ACC.
 *   gy[B,T,6] = dLoss/dy  ->  gx0[B,6], gnn[P], gode[17]
 * No reference counterpart exists; checked against finite differences in tests.
 */
int SFX(hode_oracle_solve_bwd)(int B, int T, const REAL *tg, int t_batched, const REAL *meal,
                               int meal_mode, const REAL *tvns, int tvns_mode, const REAL *gd,
                               int gd_mode, const REAL *ode, const REAL *nn_p, int H, int L,
                               int method, int max_steps, const int *nsteps, const int *status, const void *tape_v,
                               const REAL *gy, REAL *gx0, REAL *gnn, REAL *gode)
{
    SFX(mlp_t) m;
    int P = SFX(mlp_bind)(&m, nn_p, H, L);
    if (P < 0) return -1;
    const SFX(tape_t) *tape = (const SFX(tape_t) *)tape_v;
    ACC *an = (ACC *)calloc((size_t)P, sizeof(ACC));
    ACC ao[17] = {0};
    REAL act[HODE_MAXL + 1][HODE_MAXH];
    for (int b = 0; b < B; ++b) {
        const REAL *tb = t_batched ? tg + (size_t)b * T : tg;
        REAL lam[6];
        int n = nsteps[b];
        int ok = (status == NULL) || (status[b] == 0);
        for (int i = 0; i < 6; ++i) lam[i] = 0;
        for (int st = n - 1; st >= 0; --st) {
            const SFX(tape_t) *e = tape + (size_t)b * max_steps + st;
            int k = e->seg & (HODE_SEG_CLOSED - 1);
            /* if this is the last step of segment k, y_{n+1} is grid row k+1 (and any following
             * zero-length rows): inject their dLoss/dy.  The last step of a FAILED trajectory: if it closed
             * its segment, row k+1 and the zero-length copies behind it were still written (up to the
             * segment that failed); if it did not, nothing after row k was. */
            int knext;
            if (st < n - 1) knext = (e + 1)->seg & (HODE_SEG_CLOSED - 1);
            else if (ok) knext = T - 1;
            else {
                knext = k;
                if (e->seg & HODE_SEG_CLOSED) {
                    knext = k + 1;
                    while (knext + 1 < T && !(tb[knext + 1] > tb[knext])) ++knext;
                }
            }
            for (int r = k + 1; r <= knext; ++r)
                for (int i = 0; i < 6; ++i) lam[i] += gy[((size_t)b * T + r) * 6 + i];
            SFX(seg_t) s;
            s.t0 = tb[k]; s.t1 = tb[k + 1];
            s.m0 = SFX(inp_at)(meal, meal_mode, b, T, k); s.m1 = SFX(inp_at)(meal, meal_mode, b, T, k + 1);
            s.v0 = SFX(inp_at)(tvns, tvns_mode, b, T, k); s.v1 = SFX(inp_at)(tvns, tvns_mode, b, T, k + 1);
            s.d0 = SFX(inp_at)(gd, gd_mode, b, T, k); s.d1 = SFX(inp_at)(gd, gd_mode, b, T, k + 1);
            s.use_gd = (gd != NULL && gd_mode != 0);
            REAL h = e->h, tc = e->t, mm, vv, dd;
            REAL Y[6][6], Kk[6][6], ts[6];
            int S;
            double Aloc[6][6] = {{0}}, bw[6] = {0}, cw[6] = {0};
            if (method == 1) {
                S = 4;
                Aloc[1][0] = 0.5; Aloc[2][1] = 0.5; Aloc[3][2] = 1.0;
                bw[0] = 1.0 / 6; bw[1] = 2.0 / 6; bw[2] = 2.0 / 6; bw[3] = 1.0 / 6;
                cw[0] = 0; cw[1] = 0.5; cw[2] = 0.5; cw[3] = 1.0;
            } else {
                S = 6;
                for (int i = 0; i < 6; ++i) { for (int j = 0; j < i; ++j) Aloc[i][j] = SFX(dpA)[i][j]; bw[i] = SFX(dpA)[6][i]; cw[i] = SFX(dpC)[i]; }
            }
            /* recompute the stages */
            for (int sI = 0; sI < S; ++sI) {
                for (int i = 0; i < 6; ++i) {
                    REAL acc = 0;
                    for (int j = 0; j < sI; ++j) acc += (REAL)Aloc[sI][j] * Kk[j][i];
                    Y[sI][i] = e->y[i] + h * acc;
                }
                ts[sI] = tc + (REAL)cw[sI] * h;
                if (method != 1 && sI == 0) ts[sI] = tc;
                SFX(seg_eval)(&s, ts[sI], &mm, &vv, &dd);
                SFX(rhs_one)(&m, ode, ts[sI], Y[sI], mm, vv, dd, s.use_gd, Kk[sI], act);
            }
            /* reverse sweep over stages */
            REAL Ybar[6][6], newlam[6];
            for (int i = 0; i < 6; ++i) newlam[i] = lam[i];
            for (int sI = S - 1; sI >= 0; --sI) {
                REAL kb[6];
                for (int i = 0; i < 6; ++i) {
                    REAL acc = (REAL)bw[sI] * lam[i];
                    for (int j = sI + 1; j < S; ++j) acc += (REAL)Aloc[j][sI] * Ybar[j][i];
                    kb[i] = h * acc;
                }
                for (int i = 0; i < 6; ++i) Ybar[sI][i] = 0;
                SFX(seg_eval)(&s, ts[sI], &mm, &vv, &dd);
                SFX(rhs_vjp_one)(&m, ode, ts[sI], Y[sI], mm, vv, dd, s.use_gd, kb, Ybar[sI],
                                 gnn ? an : NULL, gode ? ao : NULL, (REAL)1);
                for (int i = 0; i < 6; ++i) newlam[i] += Ybar[sI][i];
            }
            for (int i = 0; i < 6; ++i) lam[i] = newlam[i];
        }
        /* row 0 and the rows that are copies of it: the grid may start with repeated times; kf = first segment of
         * positive length (= segment of step 0; = the segment a trajectory without any accepted step failed in) */
        int kf = 0;
        while (kf + 1 < T && !(tb[kf + 1] > tb[kf])) ++kf;
        for (int r = 0; r <= kf; ++r)
            for (int i = 0; i < 6; ++i) lam[i] += gy[((size_t)b * T + r) * 6 + i];
        for (int i = 0; i < 6; ++i) gx0[6 * b + i] = lam[i];
    }
    if (gnn) for (int i = 0; i < P; ++i) gnn[i] = (REAL)an[i];
    if (gode) for (int i = 0; i < 17; ++i) gode[i] = (REAL)ao[i];
    free(an);
    return 0;
}
