// hode_device.h -- device-side building blocks shared by all kernels (gfx950 / CDNA4 only).
//
// Execution model used throughout: ONE TRAJECTORY PER WAVEFRONT, ONE HIDDEN UNIT PER LANE.
//   * lane j keeps row j of every hidden weight matrix in VGPRs (weights are loaded once per
//     trajectory and stay register-resident for all ~1440 RHS evaluations);
//   * a 64x64 layer is 64 FMAs per lane, the activation of lane k is broadcast to the wave
//     through v_readlane (SGPR operand of the FMA) -- no LDS round trip, no barrier;
//   * the 6-vector state and the 7 Runge-Kutta stage derivatives are "lane-distributed":
//     component i lives in lane i of ONE VGPR, so the stage algebra is one FMA per tableau
//     entry for all six components;
//   * step-size control is per trajectory == per wave: accept/reject is a wave-uniform branch,
//     there is no lane divergence and no cross-trajectory coupling.
// MFMA is deliberately not used (north_star): every layer is a matrix-VECTOR product per
// trajectory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hode {

constexpr int kWave = 64;
constexpr int kMaxH = 64;

// ------------------------------------------------------------------------------------------
// bit casts
__device__ __forceinline__ int f2i(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ float i2f(int v) { return __builtin_bit_cast(float, v); }

// ------------------------------------------------------------------------------------------
// lane broadcast: value of lane k (k wave-uniform) to every lane, via SGPR
__device__ __forceinline__ float lane_bcast(float v, int k)
{
    return i2f(__builtin_amdgcn_readlane(f2i(v), k));
}
__device__ __forceinline__ double lane_bcast(double v, int k)
{
    uint64_t u = __builtin_bit_cast(uint64_t, v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, k);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), k);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ float first_lane(float v) { return i2f(__builtin_amdgcn_readfirstlane(f2i(v))); }
__device__ __forceinline__ double first_lane(double v)
{
    uint64_t u = __builtin_bit_cast(uint64_t, v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int first_lane(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ------------------------------------------------------------------------------------------
// DPP moves.  CTRL: quad_perm 0x00-0xFF, row_shl:n 0x100+n, row_shr:n 0x110+n, row_ror:n 0x120+n
template <int CTRL, int BANK, bool BOUND>
__device__ __forceinline__ float dpp_mov(float old, float v)
{
    return i2f(__builtin_amdgcn_update_dpp(f2i(old), f2i(v), CTRL, 0xF, BANK, BOUND));
}
template <int CTRL, int BANK, bool BOUND>
__device__ __forceinline__ double dpp_mov(double old, double v)
{
    uint64_t o = __builtin_bit_cast(uint64_t, old), u = __builtin_bit_cast(uint64_t, v);
    uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)o, (int)(uint32_t)u, CTRL, 0xF, BANK, BOUND);
    uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(o >> 32), (int)(uint32_t)(u >> 32), CTRL, 0xF, BANK, BOUND);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// value of lane (l ^ 1), (l ^ 2), (l ^ 4), (l ^ 8)
template <typename R> __device__ __forceinline__ R xlane_xor1(R v) { return dpp_mov<0xB1, 0xF, true>(v, v); }  // quad_perm [1,0,3,2]
template <typename R> __device__ __forceinline__ R xlane_xor2(R v) { return dpp_mov<0x4E, 0xF, true>(v, v); }  // quad_perm [2,3,0,1]
template <typename R> __device__ __forceinline__ R xlane_xor4(R v)
{
    R t = dpp_mov<0x104, 0x5, false>(v, v);   // row_shl:4 into banks 0,2  (lane i <- i+4)
    return dpp_mov<0x114, 0xA, false>(t, v);  // row_shr:4 into banks 1,3  (lane i <- i-4)
}
template <typename R> __device__ __forceinline__ R xlane_xor8(R v) { return dpp_mov<0x128, 0xF, true>(v, v); }  // row_ror:8

// v(l) + v(l ^ 16) and v(l) + v(l ^ 32) on every lane: gfx950 v_permlane16_swap / v_permlane32_swap
__device__ __forceinline__ float allsum_x16(float v)
{
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)f2i(v), (unsigned)f2i(v), false, false);
    return i2f((int)r[0]) + i2f((int)r[1]);
}
__device__ __forceinline__ float allsum_x32(float v)
{
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)f2i(v), (unsigned)f2i(v), false, false);
    return i2f((int)r[0]) + i2f((int)r[1]);
}
__device__ __forceinline__ double allsum_x16(double v)
{
    uint64_t u = __builtin_bit_cast(uint64_t, v);
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)u, (unsigned)u, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
    double a = __builtin_bit_cast(double, ((uint64_t)hi[0] << 32) | lo[0]);
    double b = __builtin_bit_cast(double, ((uint64_t)hi[1] << 32) | lo[1]);
    return a + b;
}
__device__ __forceinline__ double allsum_x32(double v)
{
    uint64_t u = __builtin_bit_cast(uint64_t, v);
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)u, (unsigned)u, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(u >> 32), (unsigned)(u >> 32), false, false);
    double a = __builtin_bit_cast(double, ((uint64_t)hi[0] << 32) | lo[0]);
    double b = __builtin_bit_cast(double, ((uint64_t)hi[1] << 32) | lo[1]);
    return a + b;
}

// sum over the whole wave, result on every lane
template <typename R> __device__ __forceinline__ R wave_allsum(R v)
{
    v += xlane_xor1(v);
    v += xlane_xor2(v);
    v += xlane_xor4(v);
    v += xlane_xor8(v);
    v = allsum_x16(v);
    return allsum_x32(v);
}
// sum over lanes 0..7 (on each aligned group of 8), result on every lane of the group
template <typename R> __device__ __forceinline__ R oct_allsum(R v)
{
    v += xlane_xor1(v);
    v += xlane_xor2(v);
    v += xlane_xor4(v);
    return v;
}

// Transpose-reduce: every lane holds p[0..5]; returns on each lane l the wave-wide sum of
// p[l & 7] (zero for (l & 7) >= 6).  The register count halves at every exchange, so the whole
// 6-value reduction costs ~30 VALU instead of 6 x 7 for six separate wave reductions.
template <typename R> __device__ __forceinline__ R wave_reduce6_to_lanes(const R (&p)[6], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    R q0 = (b0 ? p[1] : p[0]) + xlane_xor1(b0 ? p[0] : p[1]);
    R q1 = (b0 ? p[3] : p[2]) + xlane_xor1(b0 ? p[2] : p[3]);
    R q2 = (b0 ? p[5] : p[4]) + xlane_xor1(b0 ? p[4] : p[5]);
    R r0 = (b1 ? q1 : q0) + xlane_xor2(b1 ? q0 : q1);
    R r1 = (b1 ? R(0) : q2) + xlane_xor2(b1 ? q2 : R(0));
    R s = (b2 ? r1 : r0) + xlane_xor4(b2 ? r0 : r1);
    s += xlane_xor8(s);
    s = allsum_x16(s);
    return allsum_x32(s);
}

// ------------------------------------------------------------------------------------------
// Dormand-Prince 5(4) tableau (Dormand & Prince 1980; the pair scipy's RK45 and torchdiffeq's
// dopri5 implement; scipy/integrate/_ivp/rk.py:377-401)
template <typename R> struct DP {
    static constexpr R c2 = R(1) / 5, c3 = R(3) / 10, c4 = R(4) / 5, c5 = R(8) / 9;
    static constexpr R a21 = R(1) / 5;
    static constexpr R a31 = R(3) / 40, a32 = R(9) / 40;
    static constexpr R a41 = R(44) / 45, a42 = R(-56) / 15, a43 = R(32) / 9;
    static constexpr R a51 = R(19372) / 6561, a52 = R(-25360) / 2187, a53 = R(64448) / 6561, a54 = R(-212) / 729;
    static constexpr R a61 = R(9017) / 3168, a62 = R(-355) / 33, a63 = R(46732) / 5247, a64 = R(49) / 176, a65 = R(-5103) / 18656;
    static constexpr R b1 = R(35) / 384, b3 = R(500) / 1113, b4 = R(125) / 192, b5 = R(-2187) / 6784, b6 = R(11) / 84;
    static constexpr R e1 = R(-71) / 57600, e3 = R(71) / 16695, e4 = R(-71) / 1920, e5 = R(17253) / 339200, e6 = R(-22) / 525, e7 = R(1) / 40;
};

// ------------------------------------------------------------------------------------------
// MLP parameters of ONE parameter set, register-resident.  NL = number of hidden layers (1..4).
// Lane j owns hidden unit j of every layer.  H < 64 is zero-padded (relu(0) = 0 keeps it exact).
template <typename R, int NL> struct MlpRegs {
    R w1[9];                          // W1[j][0..8]
    R b[NL];                          // b_l[j]
    R wh[(NL > 1) ? NL - 1 : 1][kMaxH]; // W_l[j][0..63], l = 2..NL
    R w5[6];                          // Wout[o][j]
    R b5;                             // lane o < 6: bout[o]
};

__host__ __device__ inline int nn_param_count(int H, int L) { return 9 * H + H + (L - 1) * (H * H + H) + 6 * H + 6; }

template <typename R, int NL>
__device__ __forceinline__ void mlp_load(MlpRegs<R, NL> &W, const R *__restrict__ p, int H, int lane)
{
    // branch-free: out-of-range lanes / columns read a clamped (valid) address and are zeroed
    const R live = (lane < H) ? R(1) : R(0);
    const int j = (lane < H) ? lane : H - 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) W.w1[i] = live * p[j * 9 + i];
    p += 9 * H;
    W.b[0] = live * p[j];
    p += H;
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
        const R *row = p + (size_t)j * H;
        if (H == kMaxH) {
            if constexpr (sizeof(R) == 4) {
                const float4 *r4 = reinterpret_cast<const float4 *>(row);
#pragma unroll
                for (int k = 0; k < kMaxH / 4; ++k) {
                    float4 v = r4[k];
                    W.wh[l][4 * k + 0] = v.x; W.wh[l][4 * k + 1] = v.y; W.wh[l][4 * k + 2] = v.z; W.wh[l][4 * k + 3] = v.w;
                }
            } else {
                const double2 *r2 = reinterpret_cast<const double2 *>(row);
#pragma unroll
                for (int k = 0; k < kMaxH / 2; ++k) {
                    double2 v = r2[k];
                    W.wh[l][2 * k + 0] = v.x; W.wh[l][2 * k + 1] = v.y;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < kMaxH; ++k) W.wh[l][k] = ((k < H) ? live : R(0)) * row[(k < H) ? k : H - 1];
        }
        p += (size_t)H * H;
        W.b[l + 1] = live * p[j];
        p += H;
    }
#pragma unroll
    for (int o = 0; o < 6; ++o) W.w5[o] = live * p[o * H + j];
    p += 6 * H;
    W.b5 = (lane < 6) ? p[(lane < 6) ? lane : 0] : R(0);
}

// ------------------------------------------------------------------------------------------
// The 17 mechanistic constants (models/ode_core.py:44-71), wave-uniform (scalar loads).
template <typename R> struct OdeP {
    R a_GI, k_I, rho, G_b, I_b, E_max, EC_50, Glu_b, V_max, K_m, k_L, k_GE0, IGD_50, g, p_7, p_8, p_9;
};
template <typename R> __device__ __forceinline__ void ode_load(OdeP<R> &o, const R *__restrict__ p)
{
    o.a_GI = p[0]; o.k_I = p[1]; o.rho = p[2]; o.G_b = p[3]; o.I_b = p[4]; o.E_max = p[5];
    o.EC_50 = p[6]; o.Glu_b = p[7]; o.V_max = p[8]; o.K_m = p[9]; o.k_L = p[10]; o.k_GE0 = p[11];
    o.IGD_50 = p[12]; o.g = p[13]; o.p_7 = p[14]; o.p_8 = p[15]; o.p_9 = p[16];
}

template <typename R> __device__ __forceinline__ R rmax0(R v) { return v > R(0) ? v : R(0); }
__device__ __forceinline__ float rfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double rfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float rpow(float a, float b) { return powf(a, b); }
__device__ __forceinline__ double rpow(double a, double b) { return pow(a, b); }
__device__ __forceinline__ float rlog(float a) { return logf(a); }
__device__ __forceinline__ double rlog(double a) { return log(a); }
__device__ __forceinline__ float rabs(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double rabs(double a) { return __builtin_fabs(a); }

// gastric-distension Hill term (models/ode_core.py:139-140); only evaluated when a GD input exists
template <typename R> __device__ __forceinline__ R gd_effect(const OdeP<R> &o, R gd)
{
    R u = rpow(gd, o.g), v = rpow(o.IGD_50, o.g);
    return u / (v + u);
}

// Activations kept by the backward pass: h[l] = relu output of hidden layer l+1 on lane j.
template <typename R, int NL> struct MlpActs { R h[NL]; };

// ------------------------------------------------------------------------------------------
// RHS  f(t, x, u) = ODECore + NNResidual  (models/hybrid_ode_nn.py:108-134)
//   Y   lane-distributed state (lane i < 6 holds x_i)
//   returns lane-distributed derivative (lanes >= 6 hold 0)
template <typename R, int NL, bool KEEP>
__device__ __forceinline__ R rhs_eval(const MlpRegs<R, NL> &W, const OdeP<R> &o, R t, R Y, R meal, R tvns,
                                      R gde /* Hill term, 0 without GD */, int lane, MlpActs<R, NL> *acts)
{
    const R G = lane_bcast(Y, 0), I = lane_bcast(Y, 1), Glu = lane_bcast(Y, 2), GLP1 = lane_bcast(Y, 3),
            GE = lane_bcast(Y, 4), FFA = lane_bcast(Y, 5);
    // ---- mechanistic part (models/ode_core.py:124-153), evaluated redundantly on every lane
    const R Pi = R(1) + o.rho * GLP1;
    const R dI = Pi * o.a_GI * (G - o.G_b) - o.k_I * (I - o.I_b);
    const R dGlu = -(o.E_max * (GLP1 / (o.EC_50 + GLP1))) * (Glu - o.Glu_b);
    const R dGLP1 = o.V_max * (G / (o.K_m + G)) - o.k_L * GLP1;
    const R k_GE = o.k_GE0 * (R(1) - gde);
    const R dFFA = -o.p_7 * FFA - o.p_8 * I * FFA + o.p_9 * G * FFA;
    const R dG = meal - R(0.01) * (I - o.I_b) + R(0.005) * (Glu - o.Glu_b) - k_GE * G;
    R mech = (lane == 0) ? dG : (lane == 1) ? dI : (lane == 2) ? dGlu : (lane == 3) ? dGLP1 : (lane == 5) ? dFFA : R(0);
    // ---- MLP (models/nn_residual.py:138-147): input row [t, G, I, Glu, GLP1, GE, FFA, glp1:=GLP1, tvns]
    R h = W.b[0];
    h = rfma(W.w1[0], t, h);
    h = rfma(W.w1[1], G, h);
    h = rfma(W.w1[2], I, h);
    h = rfma(W.w1[3], Glu, h);
    h = rfma(W.w1[4], GLP1, h);
    h = rfma(W.w1[5], GE, h);
    h = rfma(W.w1[6], FFA, h);
    h = rfma(W.w1[7], GLP1, h);
    h = rfma(W.w1[8], tvns, h);
    h = rmax0(h);
    if constexpr (KEEP) acts->h[0] = h;
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
        R acc0 = W.b[l + 1], acc1 = R(0);
#pragma unroll
        for (int k = 0; k < kMaxH; k += 2) {
            acc0 = rfma(W.wh[l][k], lane_bcast(h, k), acc0);
            acc1 = rfma(W.wh[l][k + 1], lane_bcast(h, k + 1), acc1);
        }
        h = rmax0(acc0 + acc1);
        if constexpr (KEEP) acts->h[l + 1] = h;
    }
    R p[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) p[q] = W.w5[q] * h;
    const R nn = wave_reduce6_to_lanes(p, lane);
    return (lane < 6) ? (mech + nn + W.b5) : R(0);
}

}  // namespace hode
