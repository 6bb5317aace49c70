// hode_datagen.hip -- the data side of the hot path, on the device:
//
//   K7  fourgi_generate_kernel   4GI (8-state) cohort simulator + measurement noise + 9-column table
//                                replaces FourGIModel.simulate / generate_dataset (reference data/generate4GI.py:73-271:
//                                one scipy.odeint call per subject per 5-minute interval, ~6 ms each)
//   K8  win_* kernels            sliding windows + z-scoring of a table into fp32 training batches
//                                replaces GlucoseDataset (reference train/train_hybrid.py:43-155: pandas groupby +
//                                per-window numpy copies)
//
// K7 is a pure-ODE path (no MLP): ONE SUBJECT PER LANE (north_star's mapping), fp64, adaptive DP5(4) per lane with
// free-running lanes (a lane does not wait for its neighbours at grid points).  ~1.2 kflop per RHS (three pow), 72 B
// written and 40 B of noise draws read per grid point: fp64-VALU-bound, no LDS, no cross-lane traffic.  K8 is HBM-bound streaming: the table is
// read twice (shifted moments in one pass, then emit), outputs are written once, coalesced.  Reductions are deterministic (fixed block partials, fixed-order final sum), so the same table always
// gives the same batches.
#include "hode_kernels.h"

namespace hode {

// Dormand-Prince 5(4) error weights (the stage rows live in rk_stage as compile-time constants)
__constant__ double kDPE[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};

// x^p for the Hill / power-law terms.  exp(p*log x) costs ~1/3 of ocml's correctly rounded pow (which carries log x
// in double-double); its error is |p ln x| ulp <~ 10 ulp = 1e-15 here, far below the integration tolerance.
// Same special values as numpy's float power on this path: 0^p = 0 (p > 0), x < 0 -> NaN, x^0 = 1.
#ifdef HODE_4GI_EXACT_POW
__device__ __forceinline__ double powr_(double x, double p) { return pow(x, p); }
#else
__device__ __forceinline__ double powr_(double x, double p) { return p == 0.0 ? 1.0 : exp(p * log(x)); }
#endif

// per-subject constants of generate4GI.py:94-116 (they depend on the subject's baselines only)
struct Subject {
    double Bglc, S0glg, KINglc, KINins, KINglp, KINglg, KINgip;
    double inv_1pS0glg;   // 1 / (1 + S0glg): a per-subject constant divisor of the RHS, applied as a multiplication
};

__device__ inline Subject subject_init(const FourGIPar &p, const double *bsl)
{
    Subject s;
    const double Bglc = bsl[0], Bins = bsl[1], Bglp = bsl[2], Bglg = bsl[3], Bgip = bsl[4];
    const double r0 = pow(Bglp / p.EC50_1, p.HILL_1);
    const double S0ins = p.EMAX_1 * r0 / (1.0 + r0);
    const double q0 = Bglg / p.EC50_4;
    s.Bglc = Bglc;
    s.S0glg = p.EMAX_4 * q0 / (1.0 + q0);
    s.KINglc = Bglc * (p.CLglc + p.CLglci * Bins);
    s.KINins = Bins * p.CLins / (1.0 + S0ins * pow(Bglc, p.GLCINS_S));
    s.KINglp = p.VM_GLP * Bglp * p.VCglp / (p.KM_GLP + Bglp);
    s.KINglg = Bglg * p.CLglg;
    s.KINgip = Bgip * p.CLgip;
    s.inv_1pS0glg = 1.0 / (1.0 + s.S0glg);
    return s;
}

// generate4GI.py:73-157.  y = amounts (Gc, Ins, GLP, Glg, GIP, Gp, InsE, GIPp); meal = glucose input rate of the interval
__device__ __forceinline__ void fourgi_rhs(const FourGIPar &p, const Subject &s, int hv, const double *y, double meal,
                                           double *d)
{
    // Divisions by model constants are multiplications by reciprocals (wave-uniform, hoisted out of the stepping loop):
    // an fp64 division is ~14 instructions, and 7 of the 12 per evaluation have a constant divisor.  <= 1 ulp per
    // operation away from the reference's expression; the table's concentrations (emit_row) keep the true division.
    const double iVCglc = 1.0 / p.VCglc, iVCins = 1.0 / p.VCins, iVCglp = 1.0 / p.VCglp, iVCglg = 1.0 / p.VCglg;
    const double Cglc = y[0] * iVCglc, Cins = y[1] * iVCins, Cglp = y[2] * iVCglp, Cglg = y[3] * iVCglg;
    const double r = powr_(Cglp * (1.0 / p.EC50_1), p.HILL_1);
    const double Sins = p.EMAX_1 * r / (1.0 + r);
    const double q = Cglg * (1.0 / p.EC50_4);
    const double Sglg = p.EMAX_4 * q / (1.0 + q);
    const double glg_on_glc = (1.0 + Sglg) * s.inv_1pS0glg;
    const double p2 = Cglc >= s.Bglc ? 0.925 : (hv ? 0.327 : 0.0);
    const double glc_on_glg = Cglc > 0.0 ? powr_(s.Bglc / Cglc, p2) : 1.0;
    const double me = meal * 10.0;
    const bool fed = me > 0.0;
    const double fglp = fed ? p.FDGLP * me : 0.0, fgip = fed ? p.FDGIP * me : 0.0, fglg = fed ? p.FDGLG * me : 0.0;
    const double k27 = p.Qglc * iVCglc, k72 = p.Qglc / p.VPglc, k612 = p.Qgip / p.VCgip, k126 = p.Qgip / p.VPgip;
    d[0] = meal + s.KINglc * glg_on_glc - k27 * y[0] + k72 * y[5] - (p.CLglc * iVCglc) * y[0] -
           (p.CLglci * y[6] * iVCglc) * y[0];
    d[1] = s.KINins * (1.0 + Sins * powr_(Cglc, p.GLCINS_S)) - (p.CLins / p.VCins) * y[1];
    d[2] = s.KINglp * (1.0 + fglp) - p.VM_GLP * Cglp / (p.KM_GLP + Cglp);
    d[3] = s.KINglg * (1.0 + fglg) * glc_on_glg - (p.CLglg / p.VCglg) * y[3];
    d[4] = s.KINgip * (1.0 + fgip) - (p.CLgip / p.VCgip) * y[4] - k612 * y[4] + k126 * y[7];
    d[5] = k27 * y[0] - k72 * y[5];
    d[6] = p.Ke0ins * (Cins - y[6]);
    d[7] = k612 * y[4] - k126 * y[7];
}

__global__ __launch_bounds__(64) void fourgi_rhs_kernel(int B, int hv, FourGIPar p, const double *__restrict__ bsl,
                                                        const double *__restrict__ y, const double *__restrict__ meal,
                                                        double *__restrict__ d)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const Subject s = subject_init(p, bsl + 5 * (size_t)b);
    double yy[8], dd[8];
    for (int i = 0; i < 8; ++i) yy[i] = y[8 * (size_t)b + i];
    fourgi_rhs(p, s, hv, yy, meal[b], dd);
    for (int i = 0; i < 8; ++i) d[8 * (size_t)b + i] = dd[i];
}

__device__ inline double grid_hours(int k, double interval_min) { return ((double)k * interval_min) / 60.0; }

// numpy: data + normal(0, cv * |data|) = data + ((cv) * |data|) * z -- every operation rounded, no fma contraction
__device__ inline double add_noise(double v, double cv, double z)
{
#pragma clang fp contract(off)
    const double scale = cv * fabs(v);
    const double noise = scale * z;
    return v + noise;
}

// Rows leave the chip in bursts: a lane's row is 72 B and its next row comes an integration interval later, so direct
// stores leave every 32-byte sector of the table half written twice (measured: 0.49 GB written for a 0.29 GB table).
// Each lane therefore parks kRowBurst rows in its private slice of LDS -- [row][column][lane], lane fastest: conflict
// free -- and flushes them as one contiguous 576-byte burst that the L2 merges into whole sectors.
constexpr int kRowBurst = 8;
constexpr int kRowLds = kRowBurst * 9 * 64;   // doubles per wave (36 KB): 4 single-wave workgroups per CU

__device__ inline void flush_rows(const GenArgs &a, const double *stage, int b, int k_last)
{
    const int n = k_last % kRowBurst + 1, k0 = k_last - (n - 1);
    double *dst = a.table + ((size_t)b * a.T + k0) * 9;
    for (int i = 0; i < n * 9; ++i) dst[i] = stage[i * 64 + threadIdx.x];
}

// one row of the table (generate4GI.py:198-205 amounts -> concentrations, :214-219 noise, :246-257 columns)
__device__ inline void emit_row(const GenArgs &a, double *stage, int b, int k, const double *y, bool ok, const double *mt,
                                int n_meals)
{
    const FourGIPar &p = a.par;
    double *row = stage + (size_t)(k % kRowBurst) * 9 * 64 + threadIdx.x;
    const double th = grid_hours(k, a.interval_min);
    row[0 * 64] = (double)(a.subject0 + b);
    row[1 * 64] = th;
    row[2 * 64] = th * 60.0;
    const double conc[5] = {y[0] / p.VCglc, y[1] / p.VCins, y[2] / p.VCglp, y[3] / p.VCglg, y[4] / p.VCgip};
    const double cvs[5] = {1.0, 1.5, 1.5, 1.2, 1.3};
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        double v = ok ? conc[c] : 0.0;
        if (a.z != nullptr && a.noise_cv != 0.0) {
            const double zz = a.z[((size_t)k * 5 + c) * a.B + b];   // [T][5][B]: a wave reads 512 contiguous bytes
            v = add_noise(v, a.noise_cv * cvs[c], zz);   // generate4GI.py:239-243: cv = noise_cv * {1, 1.5, 1.5, 1.2, 1.3}
        }
        row[(3 + c) * 64] = v;
    }
    bool ind = false;
    for (int m = 0; m < n_meals; ++m) ind = ind || (fabs(th - mt[m]) < 0.01);
    row[8 * 64] = ind ? 1.0 : 0.0;
    if (k % kRowBurst == kRowBurst - 1 || k == a.T - 1) flush_rows(a, stage, b, k);
}

// One Runge-Kutta stage, unrolled per stage index: the tableau row is a compile-time constant, so stage ST reads only the
// ST stage derivatives it needs (20 array reads per step instead of 42 with a rolled loop over zero-padded rows).
template <int ST>
__device__ __forceinline__ void rk_stage(const FourGIPar &p, const Subject &s, int hv, const double (&y)[8], double h, double rate,
                                         double (&K)[7][8], double (&w)[8])
{
    constexpr double A[7][6] = {
        {0, 0, 0, 0, 0, 0},
        {1.0 / 5, 0, 0, 0, 0, 0},
        {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
        {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
        {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
        {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
        {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double acc = 0.0;
        bool first = true;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (A[ST][j] != 0.0) {            // folded at compile time
                acc = first ? A[ST][j] * K[j][i] : acc + A[ST][j] * K[j][i];
                first = false;
            }
        }
        w[i] = first ? y[i] : y[i] + h * acc;
    }
    double d[8];
    fourgi_rhs(p, s, hv, w, rate, d);
#pragma unroll
    for (int i = 0; i < 8; ++i) K[ST][i] = d[i];
}

__global__ __launch_bounds__(64) void fourgi_generate_kernel(GenArgs a)
{
    __shared__ double stage[kRowLds];
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    const FourGIPar &p = a.par;
    const Subject s = subject_init(p, a.bsl + 5 * (size_t)b);
    const double *mt = a.meal_time + (a.meals_per_subject ? (size_t)b * a.n_meals : 0);
    const double *ms = a.meal_size + (a.meals_per_subject ? (size_t)b * a.n_meals : 0);
    const double Bins = a.bsl[5 * (size_t)b + 1], Bglp = a.bsl[5 * (size_t)b + 2], Bglg = a.bsl[5 * (size_t)b + 3],
                 Bgip = a.bsl[5 * (size_t)b + 4];
    // generate4GI.py:175-184
    double y[8] = {s.Bglc * p.VCglc, Bins * p.VCins, Bglp * p.VCglp, Bglg * p.VCglg,
                   Bgip * p.VCgip,   s.Bglc * p.VPglc, Bins,         Bgip * p.VPgip};
    double K[7][8], w[8];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) K[j][i] = 0.0;

    auto meal_rate = [&](int k, double &H) {  // generate4GI.py:191-197: the meal is spread over its grid interval
        const double t0 = grid_hours(k, a.interval_min), t1 = grid_hours(k + 1, a.interval_min);
        H = t1 - t0;
        double rate = 0.0;
        for (int m = 0; m < a.n_meals; ++m)
            if (t0 <= mt[m] && mt[m] < t1) rate = ms[m] / H;
        return rate;
    };

    int k = 0, status = 0, it = 0;
    emit_row(a, stage, b, 0, y, true, mt, a.n_meals);
    if (a.T > 1) {
        double H, tau = 0.0;
        double rate = meal_rate(0, H);
        double h = H;
        bool need0 = true;
        while (true) {
            bool last = false;
            if (tau + h >= H * (1.0 - 1e-14)) {
                h = H - tau;
                last = true;
            }
            if (h < 1e-14 * H) {
                status = HODE_ST_UNDERFLOW;
                break;
            }
            if (need0) rk_stage<0>(p, s, a.hv, y, h, rate, K, w);
            rk_stage<1>(p, s, a.hv, y, h, rate, K, w);
            rk_stage<2>(p, s, a.hv, y, h, rate, K, w);
            rk_stage<3>(p, s, a.hv, y, h, rate, K, w);
            rk_stage<4>(p, s, a.hv, y, h, rate, K, w);
            rk_stage<5>(p, s, a.hv, y, h, rate, K, w);
            rk_stage<6>(p, s, a.hv, y, h, rate, K, w);
            need0 = false;
            // after stage 6: w = y_new, K[6] = f(y_new)
            double acc = 0.0;
            bool finite = true;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                double e = kDPE[0] * K[0][i];
#pragma unroll
                for (int j = 2; j < 7; ++j) e += kDPE[j] * K[j][i];
                e *= h;
                const double sc = a.atol + a.rtol * fmax(fabs(y[i]), fabs(w[i]));
                acc += (e / sc) * (e / sc);
                finite = finite && isfinite(w[i]);
            }
            if (!finite) {
                status = HODE_ST_NONFINITE;
                break;
            }
            const double err = sqrt(acc / 8.0);
            if (err < 1.0) {
                const double fac = err == 0.0 ? 10.0 : fmin(10.0, 0.9 * pow(err, -0.2));
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    y[i] = w[i];
                    K[0][i] = K[6][i];
                }
                if (last) {
                    ++k;
                    emit_row(a, stage, b, k, y, true, mt, a.n_meals);
                    if (k == a.T - 1) break;
                    const double nr = meal_rate(k, H);
                    need0 = nr != rate;  // same input => the FSAL stage IS the first stage of the next interval
                    rate = nr;
                    tau = 0.0;
                    h *= fac;
                    if (!(h > 0.0) || h > H) h = H;
                    it = 0;
                    continue;
                }
                tau += h;
                h *= fac;
            } else {
                h *= fmax(0.2, 0.9 * pow(err, -0.2));
            }
            if (++it >= a.max_steps) {
                status = HODE_ST_MAXSTEPS;
                break;
            }
        }
    }
    for (int kk = k + 1; kk < a.T; ++kk) emit_row(a, stage, b, kk, y, false, mt, a.n_meals);  // rows after a failure: zeros
    if (a.status) a.status[b] = status;
}

int launch_4gi_generate(hipStream_t s, const GenArgs &a)
{
    if (a.B <= 0) return HODE_OK;
    hipLaunchKernelGGL(fourgi_generate_kernel, dim3((a.B + 63) / 64), dim3(64), 0, s, a);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

int launch_4gi_rhs(hipStream_t s, int B, int hv, const FourGIPar &p, const double *bsl, const double *y, const double *meal,
                   double *d)
{
    if (B <= 0) return HODE_OK;
    hipLaunchKernelGGL(fourgi_rhs_kernel, dim3((B + 63) / 64), dim3(64), 0, s, B, hv, p, bsl, y, meal, d);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

// --------------------------------------------------------------------------------------------- K8: windows
constexpr int kWinBlocks = 1024;  // fixed number of partial sums => deterministic reduction, >= 4 workgroups per CU
constexpr int kWinThreads = 256;
static_assert(kWinBlocks * 12 * sizeof(double) <= HODE_4GI_SCRATCH_BYTES, "scratch holds one partial per block");

__device__ inline double win_value(const WinArgs &a, int64_t row, int c)
{
    const int col = a.col_state[c];
    // train_hybrid.py:76-80: absent 'ge' -> 0.0, absent 'ffa' -> 1.0
    return col >= 0 ? a.table[row * a.ncols + col] : (c == 5 ? 1.0 : 0.0);
}

// ONE pass over the windows for both moments: partial[block][c] = sum (x - ref_c), partial[block][6+c] = sum (x - ref_c)^2
// with ref = the first window's first row.  Shifting by a sample of the data keeps sum^2/n small against the second
// moment, so the variance is as accurate as numpy's two-pass value (to ~1e-15 relative) at one table read instead of two.
__global__ __launch_bounds__(kWinThreads) void win_moment_kernel(WinArgs a, double *__restrict__ partial)
{
    __shared__ double red[kWinThreads / 64][12];
    double s1[6] = {0, 0, 0, 0, 0, 0}, s2[6] = {0, 0, 0, 0, 0, 0}, ref[6];
    const int64_t first = a.row0[0];
#pragma unroll
    for (int c = 0; c < 6; ++c) ref[c] = win_value(a, first, c);
    const int64_t total = a.N * a.S;
    const int64_t per = (total + kWinBlocks - 1) / kWinBlocks;   // contiguous chunk per block, fixed by (N, S) alone
    const int64_t lo = per * blockIdx.x, hi = lo + per < total ? lo + per : total;
    for (int64_t e = lo + threadIdx.x; e < hi; e += kWinThreads) {
        const int64_t row = a.row0[e / a.S] + e % a.S;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            const double x = win_value(a, row, c) - ref[c];
            s1[c] += x;
            s2[c] += x * x;
        }
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) {
        double v = c < 6 ? s1[c % 6] : s2[c % 6];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);   // butterfly: the same order in every run
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        double v = 0.0;
        for (int wv = 0; wv < kWinThreads / 64; ++wv) v += red[wv][threadIdx.x];
        partial[blockIdx.x * 12 + threadIdx.x] = v;
    }
}

// mean_c = ref_c + S1/n ; std_c = sqrt((S2 - S1^2/n)/n) + 1e-6   (train_hybrid.py:124-127).  12 waves, one per sum.
// moments != 0: write {n, mean[6], M2[6]} (M2 = sum of squared deviations) instead -- the mergeable form a multi-GPU
// dataset exchanges (13 doubles per rank, combined with Chan's formula on the host).
__global__ __launch_bounds__(768) void win_finish_kernel(WinArgs a, const double *__restrict__ partial,
                                                         double *__restrict__ mean_std, int moments)
{
    __shared__ double tot[12];
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double v = 0.0;
    for (int blk = lane; blk < kWinBlocks; blk += 64) v += partial[blk * 12 + q];   // fixed order per lane
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) tot[q] = v;
    __syncthreads();
    const int c = threadIdx.x;
    if (c < 6) {
        const double n = (double)(a.N * a.S);
        const double ref = win_value(a, a.row0[0], c);
        const double m2 = tot[6 + c] - tot[c] * tot[c] / n;
        if (moments) {
            if (c == 0) mean_std[0] = n;
            mean_std[1 + c] = ref + tot[c] / n;
            mean_std[7 + c] = m2 > 0.0 ? m2 : 0.0;
        } else {
            const double var = m2 / n;
            mean_std[c] = ref + tot[c] / n;
            mean_std[6 + c] = sqrt(var > 0.0 ? var : 0.0) + 1e-6;
        }
    }
}

__global__ void win_identity_kernel(double *mean_std)
{
    if (threadIdx.x < 6) {
        mean_std[threadIdx.x] = 0.0;
        mean_std[6 + threadIdx.x] = 1.0;
    }
}

// one thread per (window, position): train_hybrid.py:113-121 (slices) and :133-154 (normalise, cast to fp32)
__global__ __launch_bounds__(256) void win_emit_kernel(WinArgs a, const double *__restrict__ mean_std)
{
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= a.N * a.S) return;
    const int64_t row = a.row0[e / a.S] + e % a.S;
    float st[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) st[c] = (float)((win_value(a, row, c) - mean_std[c]) / mean_std[6 + c]);
    float2 *o = reinterpret_cast<float2 *>(a.states + e * 6);   // 24-byte records, 8-byte aligned
    o[0] = make_float2(st[0], st[1]);
    o[1] = make_float2(st[2], st[3]);
    o[2] = make_float2(st[4], st[5]);
    a.meal[e] = a.col_meal >= 0 ? (float)a.table[row * a.ncols + a.col_meal] : 0.f;
    a.tvns[e] = a.col_tvns >= 0 ? (float)a.table[row * a.ncols + a.col_tvns] : 0.f;
    a.time[e] = (float)(a.table[row * a.ncols + a.col_time] / a.time_div);
}

int launch_4gi_windows(hipStream_t s, const WinArgs &a, int normalize, double *mean_std, void *scratch)
{
    const int64_t total = a.N * a.S;
    if (total <= 0) {
        hipLaunchKernelGGL(win_identity_kernel, dim3(1), dim3(64), 0, s, mean_std);
        return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
    }
    double *partial = (double *)scratch;
    if (normalize == HODE_4GI_NORM_GIVEN) {
        // mean_std already holds the statistics (e.g. combined over the ranks of a sharded dataset)
    } else if (normalize) {
        hipLaunchKernelGGL(win_moment_kernel, dim3(kWinBlocks), dim3(kWinThreads), 0, s, a, partial);
        hipLaunchKernelGGL(win_finish_kernel, dim3(1), dim3(768), 0, s, a, partial, mean_std, 0);
    } else {
        hipLaunchKernelGGL(win_identity_kernel, dim3(1), dim3(64), 0, s, mean_std);
    }
    const int64_t blocks = (total + 255) / 256;
    if (blocks > 0x7fffffff) return HODE_EUNSUPPORTED;
    hipLaunchKernelGGL(win_emit_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, mean_std);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

__global__ void win_zero_moments_kernel(double *m)
{
    if (threadIdx.x < 13) m[threadIdx.x] = 0.0;
}

int launch_4gi_window_moments(hipStream_t s, const WinArgs &a, double *moments, void *scratch)
{
    if (a.N * a.S <= 0) {
        hipLaunchKernelGGL(win_zero_moments_kernel, dim3(1), dim3(64), 0, s, moments);
        return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
    }
    hipLaunchKernelGGL(win_moment_kernel, dim3(kWinBlocks), dim3(kWinThreads), 0, s, a, (double *)scratch);
    hipLaunchKernelGGL(win_finish_kernel, dim3(1), dim3(768), 0, s, a, (const double *)scratch, moments, 1);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

}  // namespace hode
