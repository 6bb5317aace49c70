// hode_solve_fwd_wg.hip -- K2+K3, fp32 EXPERIMENT (HODE_FWD=wg): one trajectory per wavefront, WPB wavefronts per workgroup,
// hidden weight matrices in ONE LDS image shared by the workgroup (hode_device.h: MlpLds).
//
// Measured on MI355X, 4 096 x 241: 4.46-4.65 ms against 4.12 ms for the register kernel at every tuning point below
// (all LDS / one matrix in registers / two; 8 or 16 waves per workgroup), bit-identical results.  The LDS pipe delivers
// ~120 B/clk per CU to this access pattern (16 x ds_read_b128 per 64 DPP FMAs per wave), not the 244 B/clk four SIMDs of
// DPP FMAs ask for, and a single matrix read from LDS at unchanged occupancy already costs +9 %.  Not the default.
//
// The idea: with every weight in registers (hode_solve_fwd.hip) a wave needs 256 VGPRs, so a SIMD holds two waves; the per-RHS
// fixed part (first / last layer, 6-value reduction, mechanistic terms, stage algebra: ~200 plain VALU instructions)
// then runs at one instruction per 4 cycles per wave and cannot fill the 2-cycle issue slots.  Sharing the hidden
// matrices through LDS frees 128-192 registers per wave: 4 waves per SIMD (16 per CU = 16 trajectories per CU, so
// the 4 096-patient batch is exactly one workgroup per CU).  The arithmetic and its order are those of the register
// kernel: both give the same bits (tests/test_hip_parity.py::test_fwd_workgroup_kernel_is_bitwise_the_register_kernel).
#include "../hode_solve_body.h"
#include <cstdlib>

namespace hode {

template <int NL, int METHOD, bool TAPE, bool GD, int NREG, int WPB>
__global__ __launch_bounds__(64 * WPB, (WPB >= 4) ? WPB / 4 : 1) void solve_fwd_wg_kernel(const SolveArgs<float> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *img = reinterpret_cast<float *>(smem_raw);                       // [(NL-1)][16][64][4]
    float *rows = img + (size_t)(NL - 1) * kMaxH * kMaxH;                   // [8][64]
    float *cvec = rows + 8 * kWave;                                         // [8]
    const int lane = threadIdx.x & 63;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    float *ybuf = cvec + 8 + (size_t)wave * (kWave + 8);                    // [WPB][64 + 8]
    const int set = blockIdx.y;
    const int per_set = a.B / a.n_sets;
    const float *__restrict__ nn_set = a.nn_p + (size_t)set * a.nn_stride;

    wimg_store(img, nn_set, a.H, NL - 1, threadIdx.x, 64 * WPB);
    tableau_rows_store<float>(rows, METHOD, threadIdx.x, 64 * WPB);
    if (threadIdx.x < 8) cvec[threadIdx.x] = (float)kTableau[METHOD].c[threadIdx.x];
    MlpLds<NL, NREG> W;
    W.img = reinterpret_cast<const float4 *>(img);
    W.lane = lane;
    mlp_load_edges<float, NL>(W, nn_set, a.H, lane);
    OdeP<float> o;
    ode_load(o, a.ode_p + 17 * set);
    __syncthreads();
    W.load_regs();
    // no workgroup barrier below this line: the waves of a workgroup integrate independent trajectories
    const int bi = blockIdx.x * WPB + wave;
    const RhsRegs<float, NL, MlpLds<NL, NREG>> rhs{W, o, lane};
    if (bi < per_set) solve_one<float, METHOD, TAPE, GD>(a, set * per_set + bi, rhs, o, rows, cvec, ybuf, lane);
}

template <int NL> constexpr size_t fwd_wg_lds_bytes(int wpb)
{
    return ((size_t)(NL - 1) * kMaxH * kMaxH + 8 * kWave + 8 + (size_t)wpb * (kWave + 8)) * sizeof(float);
}

template <int NL, int METHOD, bool TAPE, bool GD, int NREG, int WPB>
static int launch_wg_one(hipStream_t s, const SolveArgs<float> &a)
{
    const int per_set = a.B / a.n_sets;
    const size_t lds = fwd_wg_lds_bytes<NL>(WPB);
    auto kern = solve_fwd_wg_kernel<NL, METHOD, TAPE, GD, NREG, WPB>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return HODE_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3((per_set + WPB - 1) / WPB, a.n_sets), dim3(64 * WPB), lds, s, a);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

// HODE_FWD_CFG=<NREG><WPB/4> picks a tuning point for measurements (tools/fwd_variants.py): 4, 22, default 14.
static int fwd_cfg()
{
    static const int v = [] { const char *e = getenv("HODE_FWD_CFG"); return e ? atoi(e) : -1; }();
    return v;
}

template <int NL, int METHOD, bool TAPE, bool GD> static int launch_wg_cfg(hipStream_t s, const SolveArgs<float> &a)
{
    constexpr int kMaxReg = NL - 1;
    const int cfg = fwd_cfg();
    switch (cfg) {
    case 4: return launch_wg_one<NL, METHOD, TAPE, GD, 0, 16>(s, a);
    case 22: return launch_wg_one<NL, METHOD, TAPE, GD, (kMaxReg >= 2 ? 2 : kMaxReg), 8>(s, a);
    default: return launch_wg_one<NL, METHOD, TAPE, GD, (kMaxReg >= 1 ? 1 : 0), 16>(s, a);      // "14"
    }
}

template <int NL> static int launch_wg_nl(hipStream_t s, const SolveArgs<float> &a, int method)
{
    const bool tape = a.tape != nullptr, gd = a.gd_mode != 0;
    if (method == HODE_METHOD_DP54) {
        if (tape) return gd ? launch_wg_cfg<NL, HODE_METHOD_DP54, true, true>(s, a) : launch_wg_cfg<NL, HODE_METHOD_DP54, true, false>(s, a);
        return gd ? launch_wg_cfg<NL, HODE_METHOD_DP54, false, true>(s, a) : launch_wg_cfg<NL, HODE_METHOD_DP54, false, false>(s, a);
    }
    if (tape) return gd ? launch_wg_cfg<NL, HODE_METHOD_RK4, true, true>(s, a) : launch_wg_cfg<NL, HODE_METHOD_RK4, true, false>(s, a);
    return gd ? launch_wg_cfg<NL, HODE_METHOD_RK4, false, true>(s, a) : launch_wg_cfg<NL, HODE_METHOD_RK4, false, false>(s, a);
}

int launch_solve_fwd_wg(hipStream_t s, const SolveArgs<float> &a, int L, int method)
{
    switch (L) {
    case 2: return launch_wg_nl<2>(s, a, method);
    case 3: return launch_wg_nl<3>(s, a, method);
    case 4: return launch_wg_nl<4>(s, a, method);
    }
    return HODE_EUNSUPPORTED;
}

}  // namespace hode
