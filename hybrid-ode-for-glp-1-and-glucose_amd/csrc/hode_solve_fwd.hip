// hode_solve_fwd.hip -- K2+K3: batched forward integration, one trajectory per wavefront, ALL weights in registers.
//
// Replaces HybridODENN.forward (reference models/hybrid_ode_nn.py:136-261).  The integration itself is
// hode_solve_body.h (solve_one).  This kernel -- one wave per workgroup, 211 weight registers, 2 waves per SIMD -- is the
// production kernel for fp32 and fp64.  Three experiment kernels that reach 4 waves per SIMD and measured SLOWER (DESIGN.md
// section 6.2) live in csrc/lab/ and are compiled into the lab library only.
#include "hode_solve_body.h"
#include <cstdlib>

namespace hode {

// MULTI: launches of MANY SHORT trajectories of ONE parameter set (the two-point solves of the physics loss: 81 920 per 4 096-patient
// step, a dozen evaluations each) hand every wave `chunk` consecutive trajectories, so that the 54 KB of weights are gathered from L2
// once per wave and not once per trajectory (half of such a launch's time).  A separate instantiation: the loop is kept out of the
// kernels every other launch runs.
template <typename R, int NL, int METHOD, int LB, bool TAPE, bool GD, bool MULTI>
__global__ __launch_bounds__(64, LB) void solve_fwd_kernel(const SolveArgs<R> a, const int chunk)
{
    __shared__ R rows[8 * kWave];             // tableau coefficient rows (hode_device.h)
    __shared__ R cvec[8];                     // tableau nodes c[s] as reals
    __shared__ R ybuf[kWave + 8];             // output staging: rows of 6 reals are gathered into 256-byte stores
    __shared__ R wstage[(sizeof(R) == 4) ? kStageElems : 1];   // weight-row permutation scratch (prologue only)
    const int lane = threadIdx.x;
    const int b0 = MULTI ? blockIdx.x * chunk : blockIdx.x;                 // one wave == one trajectory (MULTI: one after the other)
    const int set = b0 / (a.B / a.n_sets);

    tableau_rows_store<R>(rows, METHOD, lane, 64);
    if (lane < 8) cvec[lane] = (R)kTableau[METHOD].c[lane];
    MlpRegs<R, NL> W;
    mlp_load<R, NL>(W, a.nn_p + (size_t)set * a.nn_stride, a.H, lane, wstage);
    OdeP<R> o;
    ode_load(o, a.ode_p + 17 * set);
    __syncthreads();
    const RhsRegs<R, NL, MlpRegs<R, NL>> rhs{W, o, lane};
    if constexpr (MULTI) {
        const int b1 = (b0 + chunk < a.B) ? b0 + chunk : a.B;
#pragma unroll 1
        for (int b = b0; b < b1; ++b) solve_one<R, METHOD, TAPE, GD>(a, b, rhs, o, rows, cvec, ybuf, lane);
    } else {
        solve_one<R, METHOD, TAPE, GD>(a, b0, rhs, o, rows, cvec, ybuf, lane);
    }
}

template <typename R, int NL, int METHOD, bool TAPE, bool GD>
static void launch_one(hipStream_t s, const SolveArgs<R> &a)
{
    // waves per SIMD the register budget allows: 211 weight registers for 3 hidden matrices -> 2; fewer layers -> more
    constexpr int LB = (sizeof(R) == 4) ? (NL >= 3 ? 2 : (NL == 2 ? 3 : 4)) : 1;
    if constexpr (!TAPE && !GD && METHOD == HODE_METHOD_DP54 && sizeof(R) == 4) {
        // grid points <= 4, one parameter set, more trajectories than four rounds of the chip's 2 048 wave slots: see the kernel
        // (the benchmark batch as ONE round of 2 048 waves with two trajectories each was measured too: 3.10 against 3.01 ms)
        if (a.T <= 4 && a.n_sets == 1 && a.B > 8192) {
            const int chunk = (a.B + 8191) / 8192;
            hipLaunchKernelGGL((solve_fwd_kernel<R, NL, METHOD, LB, TAPE, GD, true>), dim3((a.B + chunk - 1) / chunk), dim3(64), 0, s, a, chunk);
            return;
        }
    }
    hipLaunchKernelGGL((solve_fwd_kernel<R, NL, METHOD, LB, TAPE, GD, false>), dim3(a.B), dim3(64), 0, s, a, 1);
}

template <typename R, int NL>
static int launch_nl(hipStream_t s, const SolveArgs<R> &a, int method)
{
    const bool tape = a.tape != nullptr, gd = a.gd_mode != 0;
    if (method == HODE_METHOD_DP54) {
        if (tape) { if (gd) launch_one<R, NL, HODE_METHOD_DP54, true, true>(s, a); else launch_one<R, NL, HODE_METHOD_DP54, true, false>(s, a); }
        else { if (gd) launch_one<R, NL, HODE_METHOD_DP54, false, true>(s, a); else launch_one<R, NL, HODE_METHOD_DP54, false, false>(s, a); }
    } else {
        if (tape) { if (gd) launch_one<R, NL, HODE_METHOD_RK4, true, true>(s, a); else launch_one<R, NL, HODE_METHOD_RK4, true, false>(s, a); }
        else { if (gd) launch_one<R, NL, HODE_METHOD_RK4, false, true>(s, a); else launch_one<R, NL, HODE_METHOD_RK4, false, false>(s, a); }
    }
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

#ifdef HODE_LAB
// Lab library only (make lab -> hode/lab/libhode_lab.so).  HODE_FWD=wg: the LDS-image workgroup kernel
// (lab/hode_solve_fwd_wg.hip); HODE_FWD=quad: four trajectories per four waves with column-split weights
// (lab/hode_solve_fwd_quad.hip); HODE_FWD=rows: four trajectories per four waves, weights split by output rows over the waves
// and by input blocks over the 16-lane rows (lab/hode_solve_fwd_rows.hip); anything else: this file's kernel
static char fwd_mode()
{
    static const char v = [] {
        const char *e = getenv("HODE_FWD");
        if (e == nullptr) return '\0';
        return (e[0] == 'r' && e[1] == 'o') ? 'R' : e[0];
    }();
    return v;
}
#endif

template <typename R> int launch_solve_fwd(hipStream_t s, const SolveArgs<R> &a, int L, int method)
{
#ifdef HODE_LAB
    if constexpr (sizeof(R) == 4) {
        if (L >= 2 && L <= 4 && fwd_mode() == 'w') return launch_solve_fwd_wg(s, a, L, method);
        if (L >= 2 && L <= 4 && fwd_mode() == 'q') return launch_solve_fwd_quad(s, a, L, method);
        if (L >= 2 && L <= 4 && fwd_mode() == 'R') return launch_solve_fwd_rows(s, a, L, method);
    }
#endif
    switch (L) {
    case 1: return launch_nl<R, 1>(s, a, method);
    case 2: return launch_nl<R, 2>(s, a, method);
    case 3: return launch_nl<R, 3>(s, a, method);
    case 4: return launch_nl<R, 4>(s, a, method);
    }
    return HODE_EUNSUPPORTED;
}

template int launch_solve_fwd<float>(hipStream_t, const SolveArgs<float> &, int, int);
template int launch_solve_fwd<double>(hipStream_t, const SolveArgs<double> &, int, int);

}  // namespace hode

#ifdef HODE_FWD_TRACE
// experiment build only: the stamps of hode_device.h's g_ft (this translation unit's copy: the forward kernels are instantiated here)
extern "C" int hode_lab_fwd_trace(unsigned long long *dst, int n_words, unsigned *count)
{
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(hode::g_ft_n), sizeof(unsigned)) != hipSuccess) return -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(hode::g_ft), sizeof(unsigned long long) * (size_t)n_words) == hipSuccess ? 0 : -1;
}
#endif
