// hode_solve_body.h -- the integration of ONE trajectory by ONE wavefront, shared by the two forward kernels
// (register-resident weights: hode_solve_fwd.hip; hidden matrices in a workgroup-shared LDS image: hode_solve_fwd_wg.hip).
//
// Replaces the body of the per-patient loop of HybridODENN.forward (reference models/hybrid_ode_nn.py:184-256): the
// scipy.integrate.solve_ivp call, the RHS round trip through NumPy and the input interpolation (:210-231).
// Integrator = SciPy RK45's controller (scipy/integrate/_ivp/rk.py:111-176) with every grid point a mandatory step
// boundary (the meal forcing is piecewise linear with kinks there); FSAL derivative and step-size
// proposal are carried across grid points.  CPU restatement: oracle/hode_oracle_impl.h (hode_oracle_solve).
#pragma once
#include "hode_device.h"
#include "hode_kernels.h"

namespace hode {

template <typename R> __device__ __forceinline__ R inp_at(const R *__restrict__ p, int mode, int b, int T, int k)
{
    if (mode == 0) return R(0);
    return (mode == 1) ? p[b] : p[(size_t)b * T + k];
}

// The same value through the SCALAR data cache: b and k are wave-uniform (one trajectory per wave) and the time grid and the forcing rows
// are read-only for the whole launch, so the load may take the constant address space -- s_load_dword, counted by lgkmcnt like an LDS
// read, not by vmcnt where it would queue behind the taping kernel's stores.  solve_one issues it ONE GRID INTERVAL AHEAD.
template <typename R> struct UniformInput {
    typedef const __attribute__((address_space(4))) R CR;
    CR *row;            // element 0 of this trajectory's row (mode 0: any readable element -- the value is discarded)
    int stride;         // 1: a value per grid point; 0: one value per trajectory
    bool on;
    // all of it wave-uniform and fixed for the trajectory: the grid loop pays one multiply-add, one load and one select per input --
    // as a mode switch inside the loop it was four scalar branches per input and interval on the path between two steps
    __device__ __forceinline__ UniformInput(const R *p, int mode, int b, int T, const R *any)
        : row((CR *)(uintptr_t)(mode == 0 ? any : p) + (mode == 0 ? (size_t)0 : mode == 1 ? (size_t)b : (size_t)b * T)),
          stride(mode == 2 ? 1 : 0), on(mode != 0) {}
    __device__ __forceinline__ R at(int k) const
    {
        const R v = row[k * stride];
        return on ? v : R(0);
    }
};

template <typename R> struct Eps;
template <> struct Eps<float> { static constexpr float v = 1.1920929e-7f; };
template <> struct Eps<double> { static constexpr double v = 2.220446049250313e-16; };

// The right-hand side as a functor: F = rhs(t, Y, meal, tvns, gde, rec) evaluates f(t, x, u) in the replicated state
// layout and, when rec != nullptr, records what the adjoint needs for this stage (layer activations + stage state) at rec.
//   RhsRegs  : the tuned path -- every weight in registers (MlpRegs) or hidden matrices in an LDS image (MlpLds)
//   RhsStream: the generic path (hode_generic.h) -- H <= 128, any depth, weights streamed from L2
template <typename R, int NL, typename WT> struct RhsRegs {
    const WT &W;
    const OdeP<R> &o;
    int lane;
    static constexpr bool kKeep = true;
    static constexpr bool kUnrollStages = true;      // solve_one: six copies of this right-hand side (~230 instructions each) are worth it
    // stage record of the tuned path: h_1 .. h_NL (rows of 64) | the 6-vector stage state in 8 reals (the replicated
    // 64-lane copy of the state would be another 256-byte row): 1 056 B per stage for (64,4) in fp32
    __device__ __forceinline__ int slot_elems() const { return NL * kWave + 8; }
    __device__ __forceinline__ R operator()(R ts, R Ys, R meal, R tvns, R gde, R *__restrict__ rec) const
    {
        if (rec != nullptr) {
            ActsToRecord<R> ac{rec + lane};
            const R F = rhs_eval<R, NL, true>(W, o, ts, Ys, meal, tvns, gde, lane, &ac);
            // the stage state: every octet of the replicated layout holds the same six values (slots 6, 7: zeros), so all 64 lanes
            // store to the 8 reals -- identical bits per address -- instead of masking 56 lanes off (exec save / branch / restore)
            int l7 = lane & 7;
            asm volatile("" : "+v"(l7));                    // (recomputed per stage: a hoisted copy costs a VGPR the kernel does not have)
#ifdef HODE_TAPE_NT
            __builtin_nontemporal_store(Ys, rec + NL * kWave + l7);
#else
            rec[NL * kWave + l7] = Ys;
#endif
            return F;
        }
        return rhs_eval<R, NL, false>(W, o, ts, Ys, meal, tvns, gde, lane, (MlpActs<R, NL> *)nullptr);
    }
};

// en^(-1/5) for the step-size factor (scipy/integrate/_ivp/rk.py:150-176).  fp32: v_log_f32 / v_exp_f32 as they are (the
// library forms wrap each in a denormal-range rescaling, ~8 instructions; a denormal en gives the factor its cap, 10).
template <typename R> __device__ __forceinline__ float pow_m02(float en)
{
    if constexpr (sizeof(R) == 4) return __builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf(en));
    else return __builtin_exp2f(-0.2f * __builtin_log2f(en));
}

// sqrt(sum / 6): the RMS norm of the step controller (scipy/integrate/_ivp/common.py:63-66).  The fp32 instantiation takes
// v_sqrt_f32 and a multiplication (1 ulp each) instead of the IEEE square root and division hipcc expands to ~25
// instructions per step; the fp64 parity instantiation keeps the exact forms.
template <typename R> __device__ __forceinline__ float rms6(float sum)
{
    if constexpr (sizeof(R) == 4) return __builtin_amdgcn_sqrtf(sum * (1.0f / 6.0f));
    else return sqrtf(sum / 6.0f);
}

// Dormand-Prince 5(4) as compile-time constants (the values of kTableau[HODE_METHOD_DP54], hode_device.h): the fp32 solve unrolls its
// six stages with one register per stage derivative, so a stage combination is a chain of FMAs with literal coefficients
namespace dp54c {
constexpr double A[7][6] = {{0, 0, 0, 0, 0, 0},
                            {1.0 / 5, 0, 0, 0, 0, 0},
                            {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
                            {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
                            {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
                            {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
                            {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
constexpr double C[7] = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1, 1};
constexpr double E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
// sum_j A[S][j] K[j] (row 6 = the fifth-order weights), ascending j, zero coefficients skipped, one FMA per term
template <int S, int J = 1> __device__ __forceinline__ float row_sum(const float (&K)[7], float acc)
{
    if constexpr (J < (S < 6 ? S : 6)) {
        constexpr float a = (float)A[S][J];
        if constexpr (a != 0.f) acc = __builtin_fmaf(a, K[J], acc);
        return row_sum<S, J + 1>(K, acc);
    } else {
        return acc;
    }
}
template <int S> __device__ __forceinline__ float stage_sum(const float (&K)[7]) { return row_sum<S>(K, (float)A[S][0] * K[0]); }
template <int J = 1> __device__ __forceinline__ float err_sum(const float (&K)[7], float acc)
{
    if constexpr (J < 7) {
        constexpr float e = (float)E[J];
        if constexpr (e != 0.f) acc = __builtin_fmaf(e, K[J], acc);
        return err_sum<J + 1>(K, acc);
    } else {
        return acc;
    }
}
}  // namespace dp54c

// rows  [8][64] tableau coefficient rows, cvec [8] tableau nodes (LDS, shared by the workgroup)
// ybuf  [64 + 8] output staging of THIS wave (LDS)
// rhs   the right-hand side functor of trajectory b's parameter set; o = its 17 mechanistic constants (Hill term)
template <typename R, int METHOD, bool TAPE, bool GD, typename RHS>
__device__ __forceinline__ void solve_one(const SolveArgs<R> &a, const int b, const RHS &rhs, const OdeP<R> &o,
                                          const R *__restrict__ rows, const R *__restrict__ cvec, R *__restrict__ ybuf,
                                          const int lane)
{
    const int c8 = lane & 7, grp = lane >> 3;
    const int T = a.T;
    R *__restrict__ yb = a.y + (size_t)b * T * 6;
    R *__restrict__ tape = TAPE ? a.tape + (size_t)b * a.max_steps * 8 : nullptr;
    int *__restrict__ tseg = TAPE ? a.tape_seg + (size_t)b * a.max_steps : nullptr;
    // stage tape: [step][stage 0..5][record]; record = layer activations + stage state (rhs.slot_elems() reals)
    const int kSlot = rhs.slot_elems();
    R *__restrict__ stg = TAPE ? a.tape_stage + (size_t)b * a.max_steps * 6 * kSlot : nullptr;
    constexpr bool use_gd = GD;               // the Hill term (two pow calls) only exists in the GD instantiation
    const TableauData &tab = kTableau[METHOD];

    // state, replicated over the eight 8-lane groups: lane l holds y_{l&7}
    R Y = (c8 < 6) ? a.x0[(size_t)b * 6 + c8] : R(0);
    // y[b, k, 0:6] is a contiguous stream of 6 T reals: rows are staged in LDS and leave as full 64-lane
    // stores (a 24-byte store per grid point costs a partial cache line each: measured 1.9x write traffic)
    int ypos = 0;                             // staged reals
    size_t ybase = 0;                         // reals already written
    // (the lane index is laundered in both: hipcc otherwise hoists the per-lane addresses ybuf + lane and yb + lane out of
    //  the integration -- three VGPRs for the whole kernel, which the taping instantiation spills and reloads from scratch,
    //  with a full wait, at every grid point; recomputing them is two instructions per grid point)
    auto y_put = [&](R v) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (ln < 6) ybuf[ypos + ln] = v;
        ypos += 6;
        if (ypos >= kWave) {
            __builtin_amdgcn_wave_barrier();
            yb[ybase + ln] = ybuf[ln];
            const R carry = (ln < 8) ? ybuf[kWave + ln] : R(0);
            __builtin_amdgcn_wave_barrier();
            if (ln < 8) ybuf[ln] = carry;
            ybase += kWave;
            ypos -= kWave;
        }
    };
    auto y_flush = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        __builtin_amdgcn_wave_barrier();
        if (ln < ypos) yb[ybase + ln] = ybuf[ln];
        ybase += ypos;
        ypos = 0;
    };
    y_put(Y);

    int st = HODE_ST_OK, ns = 0, nf = 0, k = 0;
    R h_abs = R(0);
    R KK = R(0);                              // packed stage derivatives: lanes 8s..8s+7 = K_{s+1}
    // fp32 DP5(4): the stages are unrolled and every stage derivative has a register of its own (replicated layout, like Y):
    // a stage combination is <= 6 FMAs with literal coefficients -- no packed register, no coefficient row from LDS, no 7-instruction
    // cross-lane sum, no slot select (12 vector instructions per stage in the rolled form, ~4.5 here).  KF = the FSAL derivative.
    // (not with the Hill term: two pow calls x 6 stages end in scratch; not for the generic path's functors: six copies of a team's
    //  layer loop are more code than the instruction cache holds)
    constexpr bool kUnrolled = METHOD == HODE_METHOD_DP54 && sizeof(R) == 4 && !GD && RHS::kUnrollStages;
    R KF = R(0);
    bool have_f = false;

    // Grid point k + 2 is fetched at the top of interval k and first touched when interval k + 1 begins (every trajectory of the
    // benchmark takes ONE step per interval: fetched where they are used, the six loads of an interval and their full wait -- an L2
    // round trip, with the taping kernel's stores in the same queue -- opened every step of the integration; profiles/r04_pmc_fwd.log)
    const UniformInput<R> in_t(a.t, 2, a.t_batched ? b : 0, T, a.t), in_m(a.meal, a.meal_mode, b, T, a.t), in_v(a.tvns, a.tvns_mode, b, T, a.t),
        in_d(a.gd, GD ? a.gd_mode : 0, b, T, a.t);
    auto grid_at = [&](int kk, R &tq, R &mq, R &vq, R &dq) {
        kk = kk < T ? kk : T - 1;
        tq = in_t.at(kk);
        mq = in_m.at(kk);
        vq = in_v.at(kk);
        dq = GD ? in_d.at(kk) : R(0);
    };
    R t0, m0, v0, d0, t1, m1, v1, d1, t2, m2, v2, d2;
    grid_at(0, t0, m0, v0, d0);
    grid_at(1, t1, m1, v1, d1);
    t2 = t1, m2 = m1, v2 = v1, d2 = d1;
    for (; k + 1 < T && st == HODE_ST_OK; ++k, t0 = t1, m0 = m1, v0 = v1, d0 = d1, t1 = t2, m1 = m2, v1 = v2, d1 = d2) {
        const R len = t1 - t0;
        if (!(len > R(0))) {                  // repeated grid time: copy the state
            grid_at(k + 2, t2, m2, v2, d2);
            y_put(Y);
            continue;
        }
        // fp32: v_rcp_f32 + one Newton step (<= 1 ulp; the IEEE division is ten dependent instructions between two steps)
        R inv_len;
        if constexpr (sizeof(R) == 4) {
            const float r0 = __builtin_amdgcn_rcpf(len);
            inv_len = first_lane(__builtin_fmaf(__builtin_fmaf(-len, r0, 1.0f), r0, r0));
        } else {
            inv_len = first_lane(R(1) / len);
        }
        const R dm = first_lane(m1 - m0), dv = first_lane(v1 - v0), dd = first_lane(d1 - d0);
        // piecewise-linear forcing on this interval (models/hybrid_ode_nn.py:217-229)
        // slot >= 0 (TAPE): also record the layer activations and the stage state for the adjoint
        auto f_at = [&](R ts, R Ys, R *__restrict__ rec) -> R {
            const R al = (ts - t0) * inv_len;
            R gde = R(0);
            if constexpr (use_gd) gde = gd_effect(o, rfma(al, dd, d0));
            return rhs(ts, Ys, rfma(al, dm, m0), rfma(al, dv, v0), gde, TAPE ? rec : nullptr);
        };
        // stage record `slot` of this trajectory (TAPE): the caller guarantees 0 <= slot < 6 max_steps
        auto rec_at = [&](int slot) -> R * { return TAPE ? stg + (size_t)slot * kSlot : nullptr; };
        // `closes`: the step ends exactly on the grid point t1 (bit 30 of the interval index; the adjoint needs it to know
        // which grid rows a FAILED trajectory still wrote)
        auto tape_put = [&](R tc, R h, bool closes) {
            if constexpr (TAPE) {
                // entry = {t, h, t0, 1 / (t1 - t0), v0, v1 - v0, d0, d1 - d0}: the step and the constants of its grid interval as THIS
                // kernel used them -- all the adjoint's step header needs, in one 32-byte record it can fetch without knowing the
                // interval index first (rounds 1-3 stored the state y0..y5 in slots 2..7; no adjoint kernel ever read it)
                // (the lane's slot is laundered: hipcc otherwise keeps the per-lane pointer tape + slot live across the whole
                //  integration -- two VGPRs the kernel does not have: it was spilled and reloaded from scratch every step)
                int slot_l = lane;
                asm volatile("" : "+v"(slot_l));
                const R e = (slot_l == 0) ? tc : (slot_l == 1) ? h : (slot_l == 2) ? t0 : (slot_l == 3) ? inv_len : (slot_l == 4) ? v0
                          : (slot_l == 5) ? dv : (slot_l == 6) ? (use_gd ? d0 : R(0)) : (use_gd ? dd : R(0));   // (no GD input: both are zero anyway)
                if (slot_l < 8) tape[(size_t)ns * 8 + slot_l] = e;
                if (lane == 0) tseg[ns] = k | (closes ? kSegClosed : 0);
            }
        };
        R tc = t0;
        grid_at(k + 2, t2, m2, v2, d2);       // (behind the interval's own constants: hipcc waits for every scalar load before the division)

        if constexpr (METHOD == HODE_METHOD_RK4) {
            if (ns >= a.max_steps) { st = HODE_ST_MAXSTEPS; break; }     // budget < T-1: report, never overrun the tape
            const R hh = len;
            KK = R(0);
#pragma unroll 1
            for (int s = 0; s < 4; ++s) {
                const R Ys = rfma(hh, group_sum8(rows[s * kWave + lane] * KK), Y);
                const R F = f_at(rfma((R)tab.c[s], hh, t0), Ys, rec_at(ns * 6 + s));       // ns < max_steps (checked above)
                KK = stage_put(KK, F, s);
            }
            const R Yn4 = rfma(hh, group_sum8(rows[7 * kWave + lane] * KK), Y);
            nf += 4;
            // a step whose result is not finite is NOT an accepted step: it never reaches the tape (its grid row is never
            // written, so the adjoint must not walk its non-finite stage records and poison the batch's shared gradient)
            if (!(fabsf((float)first_lane(oct_allsum(Yn4))) <= 3.0e38f)) { st = HODE_ST_NONFINITE; break; }
            tape_put(t0, hh, true);
            Y = Yn4;
            ns += 1;
        } else {
            if (!have_f) {
                // first derivative + Hairer's initial step (scipy/integrate/_ivp/common.py:68-135)
                const R K1 = f_at(t0, Y, a.max_steps > 0 ? rec_at(0) : nullptr);
                const R sc = (c8 < 6) ? (a.atol + rabs(Y) * a.rtol) : R(1);
                const R q0 = Y / sc, q1 = K1 / sc;
                const float dn0 = sqrtf((float)first_lane(oct_allsum(q0 * q0)) / 6.0f);
                const float dn1 = sqrtf((float)first_lane(oct_allsum(q1 * q1)) / 6.0f);
                float h0 = (dn0 < 1e-5f || dn1 < 1e-5f) ? 1e-6f : 0.01f * dn0 / dn1;
                h0 = fminf(h0, (float)len);
                const R f1 = f_at(t0 + (R)h0, rfma((R)h0, K1, Y), nullptr);
                const R q2 = (f1 - K1) / sc;
                const float dn2 = sqrtf((float)first_lane(oct_allsum(q2 * q2)) / 6.0f) / h0;
                const float h1 = (dn1 <= 1e-15f && dn2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f)
                                                                   : powf(0.01f / fmaxf(dn1, dn2), 0.2f);
                h_abs = first_lane((R)fminf(fminf(100.0f * h0, h1), (float)len));
                if constexpr (kUnrolled) KF = K1;
                else KK = (grp == 0) ? K1 : R(0);
                nf += 2;
                have_f = true;
            }
            while (tc < t1 && st == HODE_ST_OK) {
                bool rejected = false;
                for (;;) {                     // scipy/integrate/_ivp/rk.py:126-176
                    // (between two steps the wave is ONE dependent chain of wave-uniform decisions, each a vector compare -> scalar
                    //  branch round trip, while its SIMD partner has the pipe to itself (profiles/r04_fwd_trace.log): the two exits share one
                    //  branch and the clip is a select -- `|`, not `||`: no branch between the two comparisons)
                    const R atc = rabs(tc);
                    const R min_step = R(10) * Eps<R>::v * (atc > R(1e-30) ? atc : R(1e-30));
                    const bool out_of_steps = ns >= a.max_steps;
                    if (out_of_steps | (h_abs < min_step)) { st = out_of_steps ? HODE_ST_MAXSTEPS : HODE_ST_UNDERFLOW; break; }
                    const R tn_free = tc + h_abs;
                    const bool clipped = (tn_free >= t1) | ((t1 - tn_free) < R(0.01) * h_abs);
                    const R tn = first_lane(clipped ? t1 : tn_free);
                    const R h = first_lane(clipped ? t1 - tc : h_abs);
                    const bool fsal_fits = ns + 1 < a.max_steps;
                    R Ys = Y, F = R(0), err;
                    if constexpr (kUnrolled) {
                        float K[7];
                        K[0] = KF;
#define HODE_DP_STAGE(S)                                                                                                        \
                        Ys = rfma(h, dp54c::stage_sum<S>(K), Y);                                                               \
                        K[S] = f_at((S) >= 5 ? tn : first_lane(rfma((float)dp54c::C[S], h, tc)), Ys,                            \
                                    ((S) < 6 || fsal_fits) ? rec_at(ns * 6 + (S)) : nullptr);
                        HODE_DP_STAGE(1) HODE_DP_STAGE(2) HODE_DP_STAGE(3) HODE_DP_STAGE(4) HODE_DP_STAGE(5) HODE_DP_STAGE(6)
#undef HODE_DP_STAGE
                        F = K[6];
                        err = h * dp54c::err_sum(K, (float)dp54c::E[0] * K[0]);
                    } else {
                    KK = (grp == 0) ? KK : R(0);          // drop stale stages (0 * NaN would poison the sums)
                    // the coefficient row and node of stage s+1 are fetched from LDS BEFORE the RHS of stage s,
                    // so the LDS latency hides behind the MLP instead of opening every stage
                    R coef = rows[1 * kWave + lane], cs = cvec[1];
                    // stage s writes record 6 ns + s (s = 6, the FSAL stage, is record 0 of the NEXT step).  ns < max_steps holds
                    // here, so only the FSAL stage can fall off the tape: one flag per step instead of a range check of the slot
                    // at every stage (twelve scalar instructions, every one an issue slot of the wave)
#pragma unroll 1
                    for (int s = 1; s <= 6; ++s) {        // stages 2..6 and the FSAL stage (row 6 = 5th-order weights)
                        Ys = rfma(h, group_sum8(coef * KK), Y);
                        // both candidates are wave-uniform: select on the scalar unit.  (A v_cndmask_b32 whose VCC mask was
                        // written by the scalar unit costs 12-19 cycles instead of 3, tools/ubench/vcc_ubench.hip.)
                        const R ts = (s >= 5) ? tn : first_lane(rfma(cs, h, tc));
                        coef = rows[(s + 1) * kWave + lane];   // s = 6 fetches row 7 = error weights
                        cs = cvec[(s + 1) & 7];
                        // stage s of this step; the FSAL stage (s == 6) is stage 0 of the NEXT step
                        F = f_at(ts, Ys, (s < 6 || fsal_fits) ? rec_at(ns * 6 + s) : nullptr);
                        KK = stage_put(KK, F, s);
                    }
                    err = h * group_sum8(coef * KK);
                    }
                    const R Yn = Ys;                      // 5th-order solution
                    nf += 6;
                    const R ymax = rabs(Y) > rabs(Yn) ? rabs(Y) : rabs(Yn);
                    // (the quotient on all lanes, then the select: as one conditional expression hipcc builds an exec-masked region --
                    //  save, branch, restore -- on the path between two steps)
                    R qall = rdiv(err, a.atol + ymax * a.rtol);
                    asm volatile("" : "+v"(qall));
                    const R qe = (c8 < 6) ? qall : R(0);
                    float en = first_lane(rms6<R>((float)first_lane(oct_allsum(qe * qe))));   // scalar from here on
                    const float ysum = (float)first_lane(oct_allsum(Yn));
                    if (!(en == en) || !(fabsf(ysum) <= 3.0e38f) || !(fabsf(en) <= 3.0e38f)) en = 1e30f;
                    if (en < 1.0f) {
                        float fac = (en == 0.0f) ? 10.0f : fminf(10.0f, 0.9f * pow_m02<R>(en));
                        if (rejected) fac = fminf(1.0f, fac);
                        tape_put(tc, h, clipped);
                        const R hn = first_lane(h * (R)fac);
                        h_abs = (clipped && hn < h_abs) ? h_abs : hn;               // a clipped step never shrinks the proposal
                        Y = Yn;
                        if constexpr (kUnrolled) KF = F;
                        else KK = (grp == 0) ? F : KK;    // FSAL: K7 becomes K1
                        tc = tn;
                        ns++;
                        break;
                    } else {
                        h_abs = first_lane(h * (R)fmaxf(0.2f, 0.9f * pow_m02<R>(en)));
                        rejected = true;
                        if (en >= 1e30f && !(h_abs > min_step)) { st = HODE_ST_NONFINITE; break; }
                    }
                }
            }
        }
        if (st == HODE_ST_OK) {
            if constexpr (METHOD == HODE_METHOD_RK4) {
                const float ysum = (float)first_lane(oct_allsum(Y));
                if (!(fabsf(ysum) <= 3.0e38f)) st = HODE_ST_NONFINITE;
                else y_put(Y);
            } else {
                y_put(Y);        // DP5(4): Y is the result of an ACCEPTED step, and a step with a non-finite result is never accepted (en = 1e30)
            }
        }
        if (st != HODE_ST_OK) break;
    }
    y_flush();
    if (st != HODE_ST_OK) {
        // rows from the failed interval on stay zero (models/hybrid_ode_nn.py:243-256)
        for (size_t i = ybase + lane; i < (size_t)T * 6; i += kWave) yb[i] = R(0);
    }
    if (lane == 0) {
        a.status[b] = st;
        if (a.nsteps) a.nsteps[b] = ns;
        if (a.nfev) a.nfev[b] = nf;
    }
}

}  // namespace hode
