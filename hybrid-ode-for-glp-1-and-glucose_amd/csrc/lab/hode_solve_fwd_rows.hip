// hode_solve_fwd_rows.hip -- K2+K3, fp32 (HODE_FWD=rows): FOUR trajectories per workgroup of FOUR waves, every hidden matrix
// split by OUTPUT rows over the waves and by INPUT blocks over the four 16-lane rows of a wave.  Bit-identical to the
// register kernel (hode_solve_fwd.hip).
//
// Why.  At two waves per SIMD the VALU is priced 2.8 (plain) / 3.8 (DPP) cycles per instruction, at four waves 1.9 / 2.9
// (tools/ubench/inst_cost_ubench.hip) -- and 211 weight registers allow two.  The column split of hode_solve_fwd_quad.hip
// reaches four waves with 48 weight registers but pays two LDS exchanges + two barriers per layer (activations out, partial
// sums back): measured no faster.  Here ONE exchange per layer is enough:
//
//   lane (r, i) of wave w  (r = lane >> 4 the 16-lane row, i = lane & 15)  keeps  wq[l][n] = W_l[16 w + i][16 r + ((i - n) & 15)]
//
// i.e. wave w owns output units 16 w .. 16 w + 15 and row r of the wave owns the input block 16 r .. 16 r + 15 -- 16 weight
// registers per matrix.  The activation vector h of a trajectory in its NATURAL layout (unit per lane) is exactly the DPP
// operand: row_ror:n within row r walks the 16 inputs of block r, no row replication (no v_permlane swaps on the way in).
// With the four trajectories' vectors H[0..3] in four registers a layer is
//     64 FMAs      acc[t] += row_ror:n(H[t]) * wq[l][n]                (t = 0..3, n = 0..15; every weight used four times)
//     3 swaps + 3 adds   the partial sums of the four rows are added AND transposed: row t ends with trajectory t's sums
//                        ((q0 + q1) + (q2 + q3) + bias, the register kernel's order)
//     relu, one ds_write_b32: row t, lanes i -> xh[t][16 w + i]        (the natural layout of the next layer's input)
//     one barrier, four ds_read_b32: H[t] = xh[t][lane]
// Everything else -- first / last layer, mechanistic terms, Runge-Kutta algebra, step-size control, output staging -- stays
// private to the wave that owns the trajectory (solve_one, hode_solve_body.h).  128 VGPRs: four waves per SIMD.
//
// MEASURED (MI355X, fp32, T = 241, profiles/r02_fwd_variants_final.log): 3.87 ms at 4 096 trajectories against 3.57 for the
// register kernel, 7.00 against 6.90 at 8 192 -- within 2 % at twice the occupancy.  The SQ counters (tools/pmc_fwd_variant.sh)
// say why: both kernels issue the same ~4.4 G vector instructions per launch at ~4.3 shader cycles each; the forward solve is
// bound by its instruction count, not by occupancy, and four barriers per RHS cost what the missing
// v_permlane swaps save.  Kept opt-in.
//
// Lock step.  A layer needs all four waves, so the four trajectories evaluate their right-hand sides in rounds: one round =
// NL exchanges (h_1 of every trajectory out, then one per hidden matrix).  The "still integrating" flags travel with the
// first exchange.  A wave whose trajectory has finished (or that has none: ragged last workgroup) keeps serving rounds with
// a zero activation until all four flags are down; all waves leave together.  The exchange area is double-buffered: a wave
// can be at most one barrier ahead of its partners.
#include "../hode_solve_body.h"
#include <cstdlib>

namespace hode {

namespace {

constexpr int kRowsWaves = 4;       // waves = trajectories per workgroup
constexpr int kXhStride = 80;       // floats per trajectory in the exchange area: rows t and t+1 of a write hit different banks

template <int NL> struct MlpRows {
    float w1[9];
    float w1g;
    float b[NL];
    float w5[6];
    float w5r[8];
    float b5;
    float wq[(NL > 1) ? NL - 1 : 1][16];      // wq[l][n] on lane (r, i) = W_l[16 wave + i][16 r + ((i - n) & 15)]
    float bq[(NL > 1) ? NL - 1 : 1];          // b_{l+1}[16 wave + i] (every row: the bias is added behind the row reduction)
    float *xh;                                 // LDS [2][4][kXhStride]
    int *flags;                                // LDS [2][4]
    int lane, wave;
    mutable float H[kRowsWaves];               // the four trajectories' activation vectors (input of the next hidden layer)
    mutable int buf;                           // exchange buffer of the next exchange (wave-uniform)
    mutable bool active;                       // this wave's trajectory is still integrating
    mutable bool any_active;                   // ... any of the four (as of the current round)

    __device__ __forceinline__ void load(const float *__restrict__ p, int Hd)
    {
        mlp_load_edges<float, NL>(*this, p, Hd, lane);
        const float *Wl = p + 9 * Hd + Hd;
        const int i = lane & 15, r = lane >> 4;
        const int j = 16 * wave + i;
#pragma unroll
        for (int l = 0; l < NL - 1; ++l) {
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const int col = 16 * r + ((i - n) & 15);
                const bool ok = j < Hd && col < Hd;
                wq[l][n] = ok ? Wl[(size_t)(ok ? j : 0) * Hd + (ok ? col : 0)] : 0.f;
            }
            Wl += (size_t)Hd * Hd;
            bq[l] = (j < Hd) ? Wl[(j < Hd) ? j : 0] : 0.f;
            Wl += Hd;
        }
        buf = 0;
        active = false;
        any_active = false;
    }

    // 64 FMAs of one layer as one asm statement (see mlp_hidden in hode_device.h for why)
    __device__ __forceinline__ void fma64(const float (&w)[16], float (&acc)[kRowsWaves]) const
    {
#define HODE_RW(n)                                                                               \
    "v_fmac_f32_dpp %[a0], %[h0], %[w" #n "] row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"      \
    "v_fmac_f32_dpp %[a1], %[h1], %[w" #n "] row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"      \
    "v_fmac_f32_dpp %[a2], %[h2], %[w" #n "] row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"      \
    "v_fmac_f32_dpp %[a3], %[h3], %[w" #n "] row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"
        asm("v_fmac_f32 %[a0], %[h0], %[w0]\n\tv_fmac_f32 %[a1], %[h1], %[w0]\n\t"
            "v_fmac_f32 %[a2], %[h2], %[w0]\n\tv_fmac_f32 %[a3], %[h3], %[w0]\n\t"
            HODE_RW(1) HODE_RW(2) HODE_RW(3) HODE_RW(4) HODE_RW(5) HODE_RW(6) HODE_RW(7) HODE_RW(8) HODE_RW(9) HODE_RW(10)
            HODE_RW(11) HODE_RW(12) HODE_RW(13) HODE_RW(14) HODE_RW(15)
            : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
            : [h0] "v"(H[0]), [h1] "v"(H[1]), [h2] "v"(H[2]), [h3] "v"(H[3]), [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]),
              [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]), [w6] "v"(w[6]), [w7] "v"(w[7]), [w8] "v"(w[8]), [w9] "v"(w[9]),
              [w10] "v"(w[10]), [w11] "v"(w[11]), [w12] "v"(w[12]), [w13] "v"(w[13]), [w14] "v"(w[14]), [w15] "v"(w[15]));
#undef HODE_RW
    }

    static constexpr bool kHiddenRelu = true;
    __device__ __forceinline__ float hidden_relu(int l, float h) const { return hidden(l, h); }
    // post-activation of hidden layer l + 2 for this wave's trajectory (h = its post-activation of layer l + 1, used for
    // l = 0 only: later layers find it in the exchange area).  All four waves call it together, l = 0 .. NL - 2 in order.
    __device__ __forceinline__ float hidden(int l, float h) const
    {
        if (l == 0) {                                       // a round begins: h_1 of every trajectory + the flags
            float *x = xh + buf * (kRowsWaves * kXhStride);
            x[wave * kXhStride + lane] = h;
            if (lane == 0) flags[buf * kRowsWaves + wave] = active ? 1 : 0;
            __syncthreads();
#pragma unroll
            for (int t = 0; t < kRowsWaves; ++t) H[t] = x[t * kXhStride + lane];
            const int *f = flags + buf * kRowsWaves;
            any_active = first_lane(f[0] | f[1] | f[2] | f[3]) != 0;
            buf ^= 1;
        }
        float acc[kRowsWaves] = {0.f, 0.f, 0.f, 0.f};
        fma64(wq[l], acc);
        // add the four rows' partial sums and transpose: row t <- trajectory t
        auto s01 = __builtin_amdgcn_permlane16_swap((unsigned)f2i(acc[0]), (unsigned)f2i(acc[1]), false, false);
        auto s23 = __builtin_amdgcn_permlane16_swap((unsigned)f2i(acc[2]), (unsigned)f2i(acc[3]), false, false);
        const float u01 = i2f((int)s01[0]) + i2f((int)s01[1]);   // rows: [t0 q0+q1, t1 q0+q1, t0 q2+q3, t1 q2+q3]
        const float u23 = i2f((int)s23[0]) + i2f((int)s23[1]);   //       [t2 q0+q1, t3 q0+q1, t2 q2+q3, t3 q2+q3]
        auto sw = __builtin_amdgcn_permlane32_swap((unsigned)f2i(u01), (unsigned)f2i(u23), false, false);
        const float pre = (i2f((int)sw[0]) + i2f((int)sw[1])) + bq[l];   // row t: (q0 + q1) + (q2 + q3) + b of trajectory t
        float *x = xh + buf * (kRowsWaves * kXhStride);
        x[(lane >> 4) * kXhStride + 16 * wave + (lane & 15)] = rmax0(pre);
        __syncthreads();
        float own;
        if (l + 2 < NL) {
#pragma unroll
            for (int t = 0; t < kRowsWaves; ++t) H[t] = x[t * kXhStride + lane];
        }
        own = x[wave * kXhStride + lane];
        buf ^= 1;
        return own;
    }
};

// RHS functor of the rows kernel: one lock-step round per evaluation
template <int NL> struct RhsRows {
    const MlpRows<NL> &W;
    const OdeP<float> &o;
    int lane;
    static constexpr bool kUnrollStages = true;      // the arithmetic of the production kernel (bitwise comparisons)
    __device__ __forceinline__ int slot_elems() const { return NL * kWave + 8; }
    __device__ __forceinline__ float operator()(float ts, float Ys, float meal, float tvns, float gde, float *__restrict__ rec) const
    {
        W.active = true;
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
__global__ __launch_bounds__(64 * kRowsWaves, kRowsWaves) void solve_fwd_rows_kernel(const SolveArgs<float> a)
{
    static_assert(NL >= 2, "the rows kernel shares hidden matrices; NL = 1 has none");
    __shared__ float rows[8 * kWave];
    __shared__ float cvec[8];
    __shared__ float ybufs[kRowsWaves * (kWave + 8)];
    __shared__ float xh[2 * kRowsWaves * kXhStride];
    __shared__ int flags[2 * kRowsWaves];
    const int lane = threadIdx.x & 63;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    const int set = blockIdx.y;
    const int per_set = a.B / a.n_sets;

    tableau_rows_store<float>(rows, METHOD, threadIdx.x, 64 * kRowsWaves);
    if (threadIdx.x < 8) cvec[threadIdx.x] = (float)kTableau[METHOD].c[threadIdx.x];
    MlpRows<NL> W;
    W.xh = xh; W.flags = flags; W.lane = lane; W.wave = wave;
    W.load(a.nn_p + (size_t)set * a.nn_stride, a.H);
    OdeP<float> o;
    ode_load(o, a.ode_p + 17 * set);
    __syncthreads();
    const int bi = blockIdx.x * kRowsWaves + wave;
    if (bi < per_set) {
        const RhsRows<NL> rhs{W, o, lane};
        solve_one<float, METHOD, TAPE, GD>(a, set * per_set + bi, rhs, o, rows, cvec, ybufs + wave * (kWave + 8), lane);
    }
    // serve the partners until every trajectory of the workgroup is done.  The flags of a round are the same for all four
    // waves, and a wave that sees them all down has only servers for partners: they leave together, after the round.
    W.active = false;
    do {
#pragma unroll
        for (int l = 0; l < NL - 1; ++l) (void)W.hidden(l, 0.f);
    } while (W.any_active);
}

template <int NL, int METHOD, bool TAPE, bool GD> int launch_rows_one(hipStream_t s, const SolveArgs<float> &a)
{
    const int per_set = a.B / a.n_sets;
    hipLaunchKernelGGL((solve_fwd_rows_kernel<NL, METHOD, TAPE, GD>), dim3((per_set + kRowsWaves - 1) / kRowsWaves, a.n_sets),
                       dim3(64 * kRowsWaves), 0, s, a);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <int NL> int launch_rows_nl(hipStream_t s, const SolveArgs<float> &a, int method)
{
    const bool tape = a.tape != nullptr, gd = a.gd_mode != 0;
    if (method == HODE_METHOD_DP54) {
        if (tape) return gd ? launch_rows_one<NL, HODE_METHOD_DP54, true, true>(s, a) : launch_rows_one<NL, HODE_METHOD_DP54, true, false>(s, a);
        return gd ? launch_rows_one<NL, HODE_METHOD_DP54, false, true>(s, a) : launch_rows_one<NL, HODE_METHOD_DP54, false, false>(s, a);
    }
    if (tape) return gd ? launch_rows_one<NL, HODE_METHOD_RK4, true, true>(s, a) : launch_rows_one<NL, HODE_METHOD_RK4, true, false>(s, a);
    return gd ? launch_rows_one<NL, HODE_METHOD_RK4, false, true>(s, a) : launch_rows_one<NL, HODE_METHOD_RK4, false, false>(s, a);
}

}  // namespace

int launch_solve_fwd_rows(hipStream_t s, const SolveArgs<float> &a, int L, int method)
{
    switch (L) {
    case 2: return launch_rows_nl<2>(s, a, method);
    case 3: return launch_rows_nl<3>(s, a, method);
    case 4: return launch_rows_nl<4>(s, a, method);
    }
    return HODE_EUNSUPPORTED;
}

}  // namespace hode
