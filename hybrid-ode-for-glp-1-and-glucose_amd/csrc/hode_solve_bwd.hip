// hode_solve_bwd.hip -- K4: reverse-time discrete adjoint of the forward solve; K5: RHS backward.
//
// No reference counterpart: the reference detaches the solve (models/hybrid_ode_nn.py:186,
// 234-237, 248).  north_star asks for adjoint backprop, so this kernel
// differentiates the discrete scheme the forward kernel ran ("discretise-then-differentiate"):
// it walks the tape of accepted steps backwards, recomputes the Runge-Kutta stages of each step
// and pulls the cotangent through them.  Step sizes are treated as constants.
// CPU restatement: oracle/hode_oracle_impl.h (hode_oracle_solve_bwd).
//
// Mapping: one trajectory per wavefront, one hidden unit per lane (see hode_device.h).
//   * lane j keeps row j of the weights AND row j of the gradient accumulators in VGPRs
//     (2 x 211 registers in fp32 -> one wave per SIMD);
//   * dW_l[j][:] += delta_l[j] * h_{l-1}[:]  is 64 FMAs with h broadcast by v_readlane;
//   * delta_{l-1} = W_l^T delta_l reads the transposed matrices from an LDS image shared by the
//     4 waves of the workgroup (lane k fetches W_l[4jj..4jj+3][k] with one 16-byte read);
//   * a wave loops over several trajectories and keeps accumulating, so the cross-trajectory
//     reduction costs one atomic flush per wave at the end (13.5k atomics per wave).
#include "hode_device.h"
#include "hode_kernels.h"

namespace hode {

template <typename R> __device__ __forceinline__ R inp_at_b(const R *__restrict__ p, int mode, int b, int T, int k)
{
    if (mode == 0) return R(0);
    return (mode == 1) ? p[b] : p[(size_t)b * T + k];
}

// LDS layout of the adjoint workgroup (4 waves):
//   wt    [(NL-1)][64*64]   transposed hidden matrices (shared)
//   rows  [8][64]           tableau rows     A[s][lane>>3]          (shared)
//   rowsT [8][64]           transposed rows  A[lane>>3][s], row 7 = 1 (shared)
//   acts  [4 waves][6 stages][NL][64]   activations of the recomputed stages (per wave)
template <typename R, int NL> __host__ __device__ constexpr size_t bwd_lds_elems()
{
    return (size_t)(NL > 1 ? NL - 1 : 1) * kMaxH * kMaxH + 2 * 8 * kWave + (size_t)4 * 6 * NL * kWave;
}

template <typename R, int NL, bool GODE>
__global__ __launch_bounds__(256, 1) void solve_bwd_kernel(const AdjArgs<R> a, const int method)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *wt = reinterpret_cast<R *>(smem_raw);
    R *rows = wt + (size_t)(NL > 1 ? NL - 1 : 1) * kMaxH * kMaxH;
    R *rowsT = rows + 8 * kWave;

    const int lane = threadIdx.x & 63;
    const int c8 = lane & 7, grp = lane >> 3;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    R *acts_lds = rowsT + 8 * kWave + (size_t)wave * 6 * NL * kWave;
    const int set = blockIdx.y;
    const int T = a.T;
    const int per_set = a.B / a.n_sets;
    const TableauData &tab = kTableau[method];
    const int S = tab.S;

    MlpRegs<R, NL> W;
    mlp_load<R, NL>(W, a.nn_p + (size_t)set * a.P, a.H, lane);
    OdeP<R> o;
    ode_load(o, a.ode_p + 17 * set);
    if (wave == 0) wt_store<R, NL>(wt, W, lane);
    tableau_rows_store<R>(rows, method, threadIdx.x, 256);
    tableau_rowsT_store<R>(rowsT, method, threadIdx.x, 256);
    __syncthreads();

    MlpGrads<R, NL> g;
    grads_zero(g);
    R go[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) go[i] = R(0);
    const bool use_gd = a.gd_mode != 0;

    for (int bi = blockIdx.x * 4 + wave; bi < per_set; bi += gridDim.x * 4) {
        const int b = set * per_set + bi;
        const R *__restrict__ tg = a.t + (a.t_batched ? (size_t)b * T : 0);
        const R *__restrict__ tape = a.tape + (size_t)b * a.max_steps * 8;
        const int *__restrict__ tseg = a.tape_seg + (size_t)b * a.max_steps;
        const R *__restrict__ gyb = a.gy + (size_t)b * T * 6;
        const int n = a.nsteps[b];
        const bool ok = a.status[b] == HODE_ST_OK;
        R lam = R(0);                              // cotangent of the state, replicated per 8-lane group
        int knext = T - 1;                         // grid interval of the step after the current one
#pragma unroll 1
        for (int st = n - 1; st >= 0; --st) {
            const int k = tseg[st];
            // cotangents of the grid rows this step produced (row k+1 and any repeated rows).  A failed
            // trajectory's unfinished last interval was never written to y: it injects nothing.
            const int hi = (st == n - 1) ? (ok ? T - 1 : k) : knext;
            for (int r = k + 1; r <= hi; ++r) lam += (c8 < 6) ? gyb[(size_t)r * 6 + c8] : R(0);
            knext = k;
            const R tc = tape[(size_t)st * 8 + 0], h = tape[(size_t)st * 8 + 1];
            const R Y0 = (c8 < 6) ? tape[(size_t)st * 8 + 2 + c8] : R(0);
            const R t0 = tg[k], t1 = tg[k + 1];
            const R m0 = inp_at_b(a.meal, a.meal_mode, b, T, k), m1 = inp_at_b(a.meal, a.meal_mode, b, T, k + 1);
            const R v0 = inp_at_b(a.tvns, a.tvns_mode, b, T, k), v1 = inp_at_b(a.tvns, a.tvns_mode, b, T, k + 1);
            const R d0 = inp_at_b(a.gd, a.gd_mode, b, T, k), d1 = inp_at_b(a.gd, a.gd_mode, b, T, k + 1);
            const R inv_len = first_lane(R(1) / (t1 - t0));
            const R dm = first_lane(m1 - m0), dv = first_lane(v1 - v0), dd = first_lane(d1 - d0);

            // ---- pass 1: recompute the stages, keep the derivatives packed and the activations in LDS
            R KK = R(0);
#pragma unroll 1
            for (int s = 0; s < S; ++s) {
                const R Ys = rfma(h, group_sum8(rows[s * kWave + lane] * KK), Y0);
                const R ts = rfma((R)tab.c[s], h, tc);
                const R al = (ts - t0) * inv_len;
                const R gde = use_gd ? gd_effect(o, rfma(al, dd, d0)) : R(0);
                MlpActs<R, NL> ac;
                const R F = rhs_eval<R, NL, true>(W, o, ts, Ys, rfma(al, dm, m0), rfma(al, dv, v0), gde, lane, &ac);
                KK = (grp == s) ? F : KK;
#pragma unroll
                for (int l = 0; l < NL; ++l) acts_lds[((size_t)s * NL + l) * kWave + lane] = ac.h[l];
            }
            // ---- pass 2: reverse sweep.  kb_s = h (b_s lam + sum_{j>s} a_js Z_j),  Z_s = J_s^T kb_s
            R ZZ = R(0);
#pragma unroll 1
            for (int s = S - 1; s >= 0; --s) {
                const R Ys = rfma(h, group_sum8(rows[s * kWave + lane] * KK), Y0);
                const R kb = h * rfma((R)tab.bw[s], lam, group_sum8(rowsT[s * kWave + lane] * ZZ));
                const R ts = rfma((R)tab.c[s], h, tc);
                const R al = (ts - t0) * inv_len;
                const R gdv = rfma(al, dd, d0);
                const R gde = use_gd ? gd_effect(o, gdv) : R(0);
                MlpActs<R, NL> ac;
#pragma unroll
                for (int l = 0; l < NL; ++l) ac.h[l] = acts_lds[((size_t)s * NL + l) * kWave + lane];
                const R Z = rhs_vjp<R, NL, GODE, false>(W, wt, o, ts, Ys, rfma(al, dm, m0), rfma(al, dv, v0), gde, gdv,
                                                        use_gd, lane, ac, kb, g, go, nullptr);
                ZZ = (grp == s) ? Z : ZZ;
            }
            lam += group_sum8(rowsT[7 * kWave + lane] * ZZ);
        }
        if (lane < 6) a.gx0[(size_t)b * 6 + lane] = lam + gyb[lane];
    }

    if (a.gnn) grads_flush<R, NL>(g, a.gnn + (size_t)set * a.P, a.H, lane);
    if constexpr (GODE) {
        if (a.gode && lane == 0) {
#pragma unroll
            for (int i = 0; i < 17; ++i) atomic_add(a.gode + 17 * set + i, go[i]);
        }
    }
}

template <typename R, int NL> static int launch_bwd_nl(hipStream_t s, const AdjArgs<R> &a, int method)
{
    const int per_set = a.B / a.n_sets;
    int blocks = (per_set + 3) / 4;
    if (blocks > 256) blocks = 256;             // one 4-wave workgroup per CU, waves loop over trajectories
    if (a.n_sets > 1 && blocks * a.n_sets > 256) blocks = (256 + a.n_sets - 1) / a.n_sets;
    if (blocks < 1) blocks = 1;
    const size_t lds = bwd_lds_elems<R, NL>() * sizeof(R);
    dim3 grid(blocks, a.n_sets), block(256);
    if (a.gode) {
        auto kern = solve_bwd_kernel<R, NL, true>;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return HODE_ELAUNCH;
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, method);
    } else {
        auto kern = solve_bwd_kernel<R, NL, false>;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return HODE_ELAUNCH;
        hipLaunchKernelGGL(kern, grid, block, lds, s, a, method);
    }
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <typename R> int launch_solve_bwd(hipStream_t s, const AdjArgs<R> &a, int L, int method)
{
    switch (L) {
    case 1: return launch_bwd_nl<R, 1>(s, a, method);
    case 2: return launch_bwd_nl<R, 2>(s, a, method);
    case 3: return launch_bwd_nl<R, 3>(s, a, method);
    case 4: return launch_bwd_nl<R, 4>(s, a, method);
    }
    return HODE_EUNSUPPORTED;
}
template int launch_solve_bwd<float>(hipStream_t, const AdjArgs<float> &, int, int);
template int launch_solve_bwd<double>(hipStream_t, const AdjArgs<double> &, int, int);

// ------------------------------------------------------------------------------------------
// K5: RHS backward.  Replaces torch autograd over ode_residual in the physics loss
// (reference models/hybrid_ode_nn.py:318-330).  Same mapping; one sample per wave.
template <typename R, int NL, bool GODE>
__global__ __launch_bounds__(256, 1) void rhs_bwd_kernel(const RhsArgs<R> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *wt = reinterpret_cast<R *>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    MlpRegs<R, NL> W;
    mlp_load<R, NL>(W, a.nn_p, a.H, lane);
    OdeP<R> o;
    ode_load(o, a.ode_p);
    if (wave == 0) wt_store<R, NL>(wt, W, lane);
    __syncthreads();
    MlpGrads<R, NL> g;
    grads_zero(g);
    R go[17];
#pragma unroll
    for (int i = 0; i < 17; ++i) go[i] = R(0);
    for (int s = blockIdx.x * 4 + wave; s < a.B; s += gridDim.x * 4) {
        const R Y = (lane < 6) ? a.x[(size_t)s * 6 + lane] : R(0);
        const R kb = (lane < 6) ? a.gout[(size_t)s * 6 + lane] : R(0);
        const R t = a.t ? a.t[s] : R(0);
        const R meal = a.meal ? a.meal[s] : R(0);
        const R tvns = a.tvns ? a.tvns[s] : R(0);
        const R gdv = a.gd ? a.gd[s] : R(0);
        const R gde = a.gd ? gd_effect(o, gdv) : R(0);
        MlpActs<R, NL> ac;
        (void)rhs_eval<R, NL, true>(W, o, t, Y, meal, tvns, gde, lane, &ac);
        R gt;
        const R Z = rhs_vjp<R, NL, GODE, true>(W, wt, o, t, Y, meal, tvns, gde, gdv, a.gd != nullptr, lane, ac, kb, g,
                                               go, &gt);
        if (lane < 6) a.gx[(size_t)s * 6 + lane] = Z;
        if (a.gt && lane == 0) a.gt[s] = gt;
    }
    if (a.gnn) grads_flush<R, NL>(g, a.gnn, a.H, lane);
    if constexpr (GODE) {
        if (a.gode && lane == 0) {
#pragma unroll
            for (int i = 0; i < 17; ++i) atomic_add(a.gode + i, go[i]);
        }
    }
}

template <typename R, int NL> static int launch_rhs_bwd_nl(hipStream_t s, const RhsArgs<R> &a)
{
    int blocks = (a.B + 3) / 4;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) return HODE_OK;
    const size_t lds = (size_t)(NL > 1 ? NL - 1 : 1) * kMaxH * kMaxH * sizeof(R);
    if (a.gode) {
        auto kern = rhs_bwd_kernel<R, NL, true>;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return HODE_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a);
    } else {
        auto kern = rhs_bwd_kernel<R, NL, false>;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return HODE_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a);
    }
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <typename R> int launch_rhs_bwd(hipStream_t s, const RhsArgs<R> &a, int L)
{
    switch (L) {
    case 1: return launch_rhs_bwd_nl<R, 1>(s, a);
    case 2: return launch_rhs_bwd_nl<R, 2>(s, a);
    case 3: return launch_rhs_bwd_nl<R, 3>(s, a);
    case 4: return launch_rhs_bwd_nl<R, 4>(s, a);
    }
    return HODE_EUNSUPPORTED;
}
template int launch_rhs_bwd<float>(hipStream_t, const RhsArgs<float> &, int);
template int launch_rhs_bwd<double>(hipStream_t, const RhsArgs<double> &, int);

}  // namespace hode
