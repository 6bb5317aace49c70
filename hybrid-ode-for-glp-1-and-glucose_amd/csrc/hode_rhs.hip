// hode_rhs.hip -- K1 (RHS forward) and the cross-lane self test.
//
// K1 replaces HybridODENN.ode_residual (reference models/hybrid_ode_nn.py:108-134) =
// ODECore.forward (models/ode_core.py:81-166) + NNResidual.forward (models/nn_residual.py:100-151)
// for a batch of independent samples.  Same mapping as the solver: one sample per wavefront, one
// hidden unit per lane; a workgroup of 4 waves loads the weights once and loops over samples.
#include "hode_device.h"
#include "hode_kernels.h"

namespace hode {

template <typename R, int NL>
__global__ __launch_bounds__(256) void rhs_fwd_kernel(const RhsArgs<R> a)
{
    const int lane = threadIdx.x & 63;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    __shared__ R wstage[(sizeof(R) == 4) ? 4 * kStageElems : 1];
    MlpRegs<R, NL> W;
    mlp_load<R, NL>(W, a.nn_p, a.H, lane, wstage + ((sizeof(R) == 4) ? wave * kStageElems : 0));
    OdeP<R> o;
    ode_load(o, a.ode_p);
    const int stride = gridDim.x * 4;
    for (int s = blockIdx.x * 4 + wave; s < a.B; s += stride) {
        const R Y = ((lane & 7) < 6) ? a.x[(size_t)s * 6 + (lane & 7)] : R(0);     // replicated layout (rhs_eval)
        const R t = a.t ? a.t[s] : R(0);
        const R meal = a.meal ? a.meal[s] : R(0);
        const R tvns = a.tvns ? a.tvns[s] : R(0);
        const R gde = a.gd ? gd_effect(o, a.gd[s]) : R(0);
        const R F = rhs_eval<R, NL, false>(W, o, t, Y, meal, tvns, gde, lane, (MlpActs<R, NL> *)nullptr);
        if (lane < 6) a.out[(size_t)s * 6 + lane] = F;
    }
}

template <typename R, int NL> static int launch_rhs_nl(hipStream_t s, const RhsArgs<R> &a)
{
    int blocks = (a.B + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) return HODE_OK;
    hipLaunchKernelGGL((rhs_fwd_kernel<R, NL>), dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <typename R> int launch_rhs_fwd(hipStream_t s, const RhsArgs<R> &a, int L)
{
    switch (L) {
    case 1: return launch_rhs_nl<R, 1>(s, a);
    case 2: return launch_rhs_nl<R, 2>(s, a);
    case 3: return launch_rhs_nl<R, 3>(s, a);
    case 4: return launch_rhs_nl<R, 4>(s, a);
    }
    return HODE_EUNSUPPORTED;
}
template int launch_rhs_fwd<float>(hipStream_t, const RhsArgs<float> &, int);
template int launch_rhs_fwd<double>(hipStream_t, const RhsArgs<double> &, int);

// ---- self test of the cross-lane primitives: each column is compared on the host with the
//      value the primitive is documented to produce (tests/test_hip_parity.py::test_xlane)
__global__ __launch_bounds__(64) void selftest_kernel(int32_t *out)
{
    const int lane = threadIdx.x;
    const float v = (float)(lane * lane + 1);
    out[lane * 12 + 0] = (int)xlane_xor1(v);
    out[lane * 12 + 1] = (int)xlane_xor2(v);
    out[lane * 12 + 2] = (int)xlane_xor4(v);
    out[lane * 12 + 3] = (int)xlane_xor8(v);
    out[lane * 12 + 4] = (int)allsum_x16(v);
    out[lane * 12 + 5] = (int)allsum_x32(v);
    out[lane * 12 + 6] = (int)wave_allsum(v);
    float p[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) p[q] = (float)((lane + 1) * (q + 1) + (lane % 3));
    out[lane * 12 + 7] = (int)wave_reduce6_to_lanes(p, lane);
    out[lane * 12 + 8] = (int)dpp_mov<0x112, 0xF, true>(0.0f, v);   // row_shr:2
    out[lane * 12 + 9] = (int)lane_bcast(v, 37);
    const double vd = (double)(lane * lane + 1);
    out[lane * 12 + 10] = (int)wave_allsum(vd);
    double pd[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) pd[q] = (double)((lane + 1) * (q + 1) + (lane % 3));
    out[lane * 12 + 11] = (int)wave_reduce6_to_lanes(pd, lane);
}

int launch_selftest(hipStream_t s, int32_t *out)
{
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, s, out);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

}  // namespace hode
