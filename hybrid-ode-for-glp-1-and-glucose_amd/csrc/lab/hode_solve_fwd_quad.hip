// hode_solve_fwd_quad.hip -- K2+K3, fp32 EXPERIMENT (HODE_FWD=quad): FOUR trajectories per workgroup of FOUR waves, hidden
// matrices split by columns.  Bit-identical to the production kernel; measured no faster (see the end of this comment).
//
// The idea.  Micro-benchmarks (tools/ubench/inst_cost_ubench.hip) say a single wave issues one VALU instruction every
// ~7.5 cycles whatever the instruction, so the cost of an instruction to the SIMD falls with occupancy: v_fmac_f32_dpp
// 6.7 / 3.8 / 2.9 cycles at 1 / 2 / 4 waves per SIMD, plain v_fma_f32 6.8 / 2.8 / 1.9.  211 weight registers allow two
// waves.  Reading the weights from an LDS image instead (hode_solve_fwd_wg.hip) buys the occupancy and loses it again to the
// LDS pipe: 48 KB of reads per RHS and wave.
//
// Here the four waves of a workgroup integrate four trajectories and SHARE the weights by columns: wave w keeps, of every
// hidden matrix, the 16 columns 16w .. 16w+15 (48 registers for three matrices, in the rotating-operand order of
// hode_device.h), and computes that quarter of the matrix-vector product for ALL FOUR trajectories -- the same 64
// v_fmac_f32_dpp per layer and wave as before.  Per layer the waves exchange through LDS
//     1. their activation vectors (256 B each; wave w reads back, already replicated, the 16-lane row w of each of them),
//     2. the partial sums for the three trajectories they do not own (3 x 256 B each),
// 2 KB written and 1.75 KB read per wave instead of 16 KB, two workgroup barriers.  Everything else -- first / last layer,
// mechanistic terms, Runge-Kutta algebra, step-size control, output staging -- stays private to the wave that owns the
// trajectory (solve_one, hode_solve_body.h).  128 VGPRs, no scratch: four waves per SIMD; the 4 096-patient batch is
// 1 024 workgroups = 4 per CU.
//
// The partial sums are combined in the order of the register kernel, ((P0 + P1) + (P2 + P3)) + bias: results are
// BIT-IDENTICAL to hode_solve_fwd.hip (tests/test_hip_parity.py).
//
// Lock step.  A layer needs all four waves, so the four trajectories evaluate their right-hand sides in rounds: every RHS
// evaluation is one round = a flag exchange (who is still integrating) + NL - 1 layer exchanges.  A wave whose trajectory
// has finished (or that has none: ragged last workgroup) keeps serving rounds with a zero activation until all four flags
// are down; all waves leave together.
//
// MEASURED (MI355X, fp32, T = 241): 4.40 ms at 4 096 trajectories against 4.13 for the register kernel, 7.96 ms at 8 192
// against 7.96 -- the SAME 1.03 M trajectories/s at twice the occupancy; a lone trajectory takes 1.8x longer (seven
// barriers per RHS).  Starting the workgroups of a CU a fraction of a round apart changes nothing.  Three structurally
// different VALU-only kernels (registers at 2 waves per SIMD, LDS image at 4, column split at 4) end at the same
// ~1.0 M trajectories/s, i.e. ~3.8 cycles per VALU instruction and SIMD at the ~2.3 GHz the chip holds: the ceiling is
// not occupancy, LDS or the DPP rate of one of them but what they share -- ~550 instructions (363 VALU) per RHS and wave
// through the same instruction front end.
#include "../hode_solve_body.h"
#include <cstdlib>

namespace hode {

namespace {

constexpr int kQuad = 4;            // waves = trajectories per workgroup

template <int NL> struct MlpQuad {
    float w1[9];
    float w1g;
    float b[NL];
    float w5[6];
    float w5r[8];
    float b5;
    float wq[(NL > 1) ? NL - 1 : 1][16];      // wq[l][n] on lane j = W_l[j][16 wave + ((j - n) & 15)]
    float *xh;                                 // LDS [4][64]     activation of each trajectory (input of the current layer)
    float *xp;                                 // LDS [4][4][64]  xp[src wave][trajectory]: partial sums
    int *flags;                                // LDS [4]         trajectory still integrating?
    int lane, wave;

    __device__ __forceinline__ void load(const float *__restrict__ p, int H)
    {
        mlp_load_edges<float, NL>(*this, p, H, lane);
        const float *Wl = p + 9 * H + H;
        const bool row_ok = lane < H;
#pragma unroll
        for (int l = 0; l < NL - 1; ++l) {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int col = 16 * wave + ((lane - n) & 15);
                const bool ok = row_ok && col < H;
                wq[l][n] = ok ? Wl[(size_t)(ok ? lane : 0) * H + (ok ? col : 0)] : 0.f;
            }
            Wl += (size_t)H * H + H;
        }
    }

    // who is still integrating?  (one barrier; the flags are rewritten only after the layer exchanges of this round)
    __device__ __forceinline__ bool round_begin(bool active) const
    {
        if (lane == 0) flags[wave] = active ? 1 : 0;
        __syncthreads();
        return (flags[0] | flags[1] | flags[2] | flags[3]) != 0;
    }

    // pre-activation of hidden layer l + 2 for this wave's trajectory; all four waves call it together
    __device__ __forceinline__ float hidden(int l, float h) const
    {
        xh[wave * kWave + lane] = h;
        __syncthreads();
        const int p16 = 16 * wave + (lane & 15);
        float R[kQuad], acc[kQuad];
#pragma unroll
        for (int t = 0; t < kQuad; ++t) {
            R[t] = xh[t * kWave + p16];            // row `wave` of trajectory t's activation, replicated over the four rows
            acc[t] = 0.f;
        }
        asm volatile("" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]));
        quarter<0>(wq[l], R, acc);
#pragma unroll
        for (int t = 0; t < kQuad; ++t)
            if (t != wave) xp[(wave * kQuad + t) * kWave + lane] = acc[t];
        __syncthreads();
        float P[kQuad];
#pragma unroll
        for (int q = 0; q < kQuad; ++q) P[q] = xp[(q * kQuad + wave) * kWave + lane];
        // own quarter from the register (its xp slot is never written), wave-uniform selects
        P[0] = (wave == 0) ? acc[0] : P[0];
        P[1] = (wave == 1) ? acc[1] : P[1];
        P[2] = (wave == 2) ? acc[2] : P[2];
        P[3] = (wave == 3) ? acc[3] : P[3];
        return ((P[0] + P[1]) + (P[2] + P[3])) + b[l + 1];      // bias last: the order of every fp32 forward kernel
    }

    template <int N> static __device__ __forceinline__ void quarter(const float (&w)[16], const float (&R)[kQuad], float (&acc)[kQuad])
    {
        acc[0] = fmac_ror<N>(acc[0], R[0], w[N]);
        acc[1] = fmac_ror<N>(acc[1], R[1], w[N]);
        acc[2] = fmac_ror<N>(acc[2], R[2], w[N]);
        acc[3] = fmac_ror<N>(acc[3], R[3], w[N]);
        if constexpr (N < 15) quarter<N + 1>(w, R, acc);
    }
};

// RHS functor of the quad kernel: one lock-step round per evaluation
template <int NL> struct RhsQuad {
    const MlpQuad<NL> &W;
    const OdeP<float> &o;
    int lane;
    static constexpr bool kUnrollStages = true;      // the arithmetic of the production kernel (bitwise comparisons)
    __device__ __forceinline__ int slot_elems() const { return NL * kWave + 8; }
    __device__ __forceinline__ float operator()(float ts, float Ys, float meal, float tvns, float gde, float *__restrict__ rec) const
    {
        (void)W.round_begin(true);
        if (rec != nullptr) {
            ActsToRecord<float> ac{rec + lane};
            const float F = rhs_eval<float, NL, true>(W, o, ts, Ys, meal, tvns, gde, lane, &ac);
            if (lane < 8) ac.dst[NL * kWave] = Ys;
            return F;
        }
        return rhs_eval<float, NL, false>(W, o, ts, Ys, meal, tvns, gde, lane, (MlpActs<float, NL> *)nullptr);
    }
};

template <int NL, int METHOD, bool TAPE, bool GD>
__global__ __launch_bounds__(64 * kQuad, kQuad) void solve_fwd_quad_kernel(const SolveArgs<float> a)
{
    __shared__ float rows[8 * kWave];
    __shared__ float cvec[8];
    __shared__ float ybufs[kQuad * (kWave + 8)];
    __shared__ float xh[kQuad * kWave];
    __shared__ float xp[kQuad * kQuad * kWave];
    __shared__ int flags[kQuad];
    const int lane = threadIdx.x & 63;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    const int set = blockIdx.y;
    const int per_set = a.B / a.n_sets;

    tableau_rows_store<float>(rows, METHOD, threadIdx.x, 64 * kQuad);
    if (threadIdx.x < 8) cvec[threadIdx.x] = (float)kTableau[METHOD].c[threadIdx.x];
    MlpQuad<NL> W;
    W.xh = xh; W.xp = xp; W.flags = flags; W.lane = lane; W.wave = wave;
    W.load(a.nn_p + (size_t)set * a.nn_stride, a.H);
    OdeP<float> o;
    ode_load(o, a.ode_p + 17 * set);
    __syncthreads();
    const int bi = blockIdx.x * kQuad + wave;
    if (bi < per_set) {
        const RhsQuad<NL> rhs{W, o, lane};
        solve_one<float, METHOD, TAPE, GD>(a, set * per_set + bi, rhs, o, rows, cvec, ybufs + wave * (kWave + 8), lane);
    }
    // serve the partners until every trajectory of the workgroup is done (all four waves see the same flags: they leave together)
    while (W.round_begin(false)) {
#pragma unroll
        for (int l = 0; l < NL - 1; ++l) (void)W.hidden(l, 0.f);
    }
}

template <int NL, int METHOD, bool TAPE, bool GD> int launch_quad_one(hipStream_t s, const SolveArgs<float> &a)
{
    const int per_set = a.B / a.n_sets;
    hipLaunchKernelGGL((solve_fwd_quad_kernel<NL, METHOD, TAPE, GD>), dim3((per_set + kQuad - 1) / kQuad, a.n_sets), dim3(64 * kQuad), 0,
                       s, a);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <int NL> int launch_quad_nl(hipStream_t s, const SolveArgs<float> &a, int method)
{
    const bool tape = a.tape != nullptr, gd = a.gd_mode != 0;
    if (method == HODE_METHOD_DP54) {
        if (tape) return gd ? launch_quad_one<NL, HODE_METHOD_DP54, true, true>(s, a) : launch_quad_one<NL, HODE_METHOD_DP54, true, false>(s, a);
        return gd ? launch_quad_one<NL, HODE_METHOD_DP54, false, true>(s, a) : launch_quad_one<NL, HODE_METHOD_DP54, false, false>(s, a);
    }
    if (tape) return gd ? launch_quad_one<NL, HODE_METHOD_RK4, true, true>(s, a) : launch_quad_one<NL, HODE_METHOD_RK4, true, false>(s, a);
    return gd ? launch_quad_one<NL, HODE_METHOD_RK4, false, true>(s, a) : launch_quad_one<NL, HODE_METHOD_RK4, false, false>(s, a);
}

}  // namespace

int launch_solve_fwd_quad(hipStream_t s, const SolveArgs<float> &a, int L, int method)
{
    switch (L) {
    case 2: return launch_quad_nl<2>(s, a, method);
    case 3: return launch_quad_nl<3>(s, a, method);
    case 4: return launch_quad_nl<4>(s, a, method);
    }
    return HODE_EUNSUPPORTED;
}

}  // namespace hode
