// hode_device.h -- device-side building blocks shared by all kernels (gfx950 / CDNA4 only).
//
// Execution model used throughout: ONE TRAJECTORY PER WAVEFRONT, ONE HIDDEN UNIT PER LANE.
//   * the hidden weight matrices live in VGPRs, 64 registers per matrix and lane (loaded once per trajectory, register-
//     resident for all ~1440 RHS evaluations); activations are kept one unit per lane;
//   * a 64x64 layer is 64 FMAs per lane; the activation of lane k reaches lane j as the DPP row_ror:n operand of the FMA
//     itself ("rotating operand") -- no v_readlane per element, no LDS round trip, no barrier.  fp32 forward kernels: lane
//     16 r + i keeps W[16 w + i][16 r + ((i - n) & 15)] and reduces four row-partial accumulators with a 3-swap transpose
//     (mlp_hidden_blk); the adjoint and the LDS-image experiment replicate the 16-lane rows first (rows_replicate);
//   * the 6-vector state is replicated per 8-lane group (lane l holds component l & 7) and the
//     Runge-Kutta stage derivatives are packed into ONE VGPR (lanes 8s..8s+7 = stage s), so a stage
//     combination is one multiply by a per-lane coefficient row + a 7-instruction cross-lane sum;
//   * step-size control is per trajectory == per wave: accept/reject is a wave-uniform branch,
//     there is no lane divergence and no cross-trajectory coupling.
// MFMA is deliberately not used (north_star): every layer is a matrix-VECTOR product per
// trajectory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hode.h"

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
// value of lane (l ^ 7) within each 8-lane half row (row_half_mirror).  For a value that is already uniform over the quads
// -- a sum after the xor1 / xor2 exchanges -- this IS the other quad's value, in one DPP operand instead of the two masked
// moves + copy of xlane_xor4.
template <typename R> __device__ __forceinline__ R xlane_hmirror(R v) { return dpp_mov<0x141, 0xF, true>(v, v); }

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
    v += xlane_hmirror(v);          // quad-uniform by now
    v += xlane_xor8(v);
    v = allsum_x16(v);
    return allsum_x32(v);
}
// sum over lanes 0..7 (on each aligned group of 8), result on every lane of the group
template <typename R> __device__ __forceinline__ R oct_allsum(R v)
{
    v += xlane_xor1(v);
    v += xlane_xor2(v);
    v += xlane_hmirror(v);          // quad-uniform by now
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
// sum over the 8 lanes that share (lane & 7), result on all of them
template <typename R> __device__ __forceinline__ R group_sum8(R v)
{
    if constexpr (sizeof(R) == 4) {
        // v + row_ror:8(v) as ONE instruction (hipcc otherwise re-fuses the producer of v into a copy + v_fmac pair)
        float t = v;
        asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(t));
        v = t;
    } else {
        v += xlane_xor8(v);
    }
    v = allsum_x16(v);
    return allsum_x32(v);
}

// ------------------------------------------------------------------------------------------
// Runge-Kutta tableaux as data.  The stage derivatives of a step are PACKED into one VGPR:
// lanes 8s..8s+7 hold stage s (component = lane & 7), so a stage combination
//   Y_s = Y + h * sum_j a_sj K_j   is   Y + h * group_sum8(coef_s * KK)
// with a per-lane coefficient row coef_s[lane] = a[s][lane >> 3] fetched from an LDS table.
// One copy of the RHS code serves every stage (the stage loop is NOT unrolled).
struct TableauData {
    double A[8][8];   // A[s][j]; for DP5(4) row 6 = the 5th-order weights (FSAL stage)
    double bw[8];     // solution weights
    double c[8];      // nodes
    double E[8];      // error-estimate weights (DP5(4) only)
    int S;            // stages that carry the solution (backward sweeps these)
};
__constant__ TableauData kTableau[2] = {
    // HODE_METHOD_DP54: Dormand-Prince 5(4) (scipy/integrate/_ivp/rk.py:377-401)
    {{{0},
      {1.0 / 5},
      {3.0 / 40, 9.0 / 40},
      {44.0 / 45, -56.0 / 15, 32.0 / 9},
      {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
      {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656},
      {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84},
      {0}},
     {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0, 0},
     {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1, 1, 0},
     {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40, 0},
     6},
    // HODE_METHOD_RK4: the classic 4-stage scheme
    {{{0}, {0.5}, {0, 0.5}, {0, 0, 1.0}, {0}, {0}, {0}, {0}},
     {1.0 / 6, 2.0 / 6, 2.0 / 6, 1.0 / 6, 0, 0, 0, 0},
     {0, 0.5, 0.5, 1.0, 0, 0, 0, 0},
     {0},
     4}};

// LDS coefficient rows (per workgroup): rowsA[s][lane] = A[s][lane>>3]; row 7 = E (DP) / bw (RK4)
template <typename R> __device__ __forceinline__ void tableau_rows_store(R *rows, int method, int tid, int nthreads)
{
    for (int i = tid; i < 8 * kWave; i += nthreads) {
        const int s = i >> 6, l = i & 63;
        double v = kTableau[method].A[s][l >> 3];
        if (s == 7) v = (method == HODE_METHOD_DP54) ? kTableau[method].E[l >> 3] : kTableau[method].bw[l >> 3];
        rows[i] = (R)v;
    }
}
// transposed rows for the adjoint: rowsT[s][lane] = A[lane>>3][s] (only stages < S); row 7 = 1 for stages < S;
// row 6 carries the solution weights bw[0..7] (lanes 0..7) and the nodes c[0..7] (lanes 8..15) as reals
template <typename R> __device__ __forceinline__ void tableau_rowsT_store(R *rows, int method, int tid, int nthreads)
{
    const int S = kTableau[method].S;
    for (int i = tid; i < 8 * kWave; i += nthreads) {
        const int s = i >> 6, l = i & 63, j = l >> 3;
        double v = (j < S && s < S) ? kTableau[method].A[j][s] : 0.0;
        if (s == 7) v = (j < S) ? 1.0 : 0.0;
        if (s == 6) v = (l < 8) ? kTableau[method].bw[l] : (l < 16) ? kTableau[method].c[l - 8] : 0.0;   // scalars as reals
        rows[i] = (R)v;
    }
}

// ------------------------------------------------------------------------------------------
// MLP parameters of ONE parameter set, register-resident.  NL = number of hidden layers (1..4).
// Activations: lane j = hidden unit j of every layer.  H < 64 is zero-padded (relu(0) = 0 keeps it exact).
__device__ __forceinline__ float mlp_hidden(const float (&w)[64], float bias, float h);
__device__ __forceinline__ double mlp_hidden(const double (&w)[64], double bias, double h);
__device__ __forceinline__ float mlp_hidden_relu(const float (&w)[64], float bias, float h);
typedef float f2_t __attribute__((ext_vector_type(2)));       // an even-aligned VGPR pair: operand of the packed fp32 instructions
template <bool RELU> __device__ __forceinline__ float mlp_hidden_blk(const f2_t (&wp)[32], float bias, float h);
__device__ __forceinline__ double mlp_hidden_relu(const double (&w)[64], double bias, double h);
template <typename R, int NL> struct MlpRegs {
    R w1[9];                          // W1[j][0..8]
    R w1g;                            // W1[j][4] + W1[j][7]: the weight of GLP1, which the input row holds twice
    R b[NL];                          // b_l[j]
    // fp64: W_l[j][0..63], l = 2..NL.  fp32: the row-block order of mlp_hidden_blk as the register PAIRS its packed FMAs take --
    // wh[l][2 n] = (w0_n, w2_n), wh[l][2 n + 1] = (w1_n, w3_n) -- filled pair by pair in mlp_load (gathered into single registers first
    // and paired up afterwards, hipcc shuffled the 192 weights through 456 B of scratch per lane: 240 MB of HBM traffic per launch)
    using HW = std::conditional_t<sizeof(R) == 4, f2_t, R>;
    HW wh[(NL > 1) ? NL - 1 : 1][sizeof(R) == 4 ? kMaxH / 2 : kMaxH];
    R w5[6];                          // Wout[o][j]
    R w5r[8];                         // fp32: Wout[lane & 7] in rotating order (out_rot_fill); unused in fp64
    R b5;                             // lane l: bout[l & 7] (0 for slots 6,7)
    // pre-activation of hidden layer l + 2 (l is a compile-time constant at every call site: unrolled layer loop)
    __device__ __forceinline__ R hidden(int l, R h) const
    {
        if constexpr (sizeof(R) == 4) return mlp_hidden_blk<false>(wh[l], b[l + 1], h);
        else return mlp_hidden(wh[l], b[l + 1], h);
    }
    // fp32: the ReLU is the last instruction of the layer's asm statement (hidden_relu); applied to the asm's result from
    // outside it costs two instructions -- hipcc canonicalises a value it did not compute itself before a max
    static constexpr bool kHiddenRelu = sizeof(R) == 4;
    __device__ __forceinline__ R hidden_relu(int l, R h) const
    {
        if constexpr (sizeof(R) == 4) return mlp_hidden_blk<true>(wh[l], b[l + 1], h);
        else return mlp_hidden_relu(wh[l], b[l + 1], h);
    }
};
// weight holders whose hidden_relu(l, h) returns the POST-activation of hidden layer l + 2
template <typename T, typename = void> struct applies_relu { static constexpr bool value = false; };
template <typename T> struct applies_relu<T, decltype((void)T::kHiddenRelu)> { static constexpr bool value = T::kHiddenRelu; };

__host__ __device__ inline int nn_param_count(int H, int L) { L &= 0xff; return 9 * H + H + (L - 1) * (H * H + H) + 6 * H + 6; }   // (L may carry an activation code in bits 8..15)

// fp32 hidden layers use the "rotating operand" form (see mlp_hidden): register r = 16q + n of lane j holds
// W_l[j][16q + ((j - n) & 15)], so that the activation can be fetched with a DPP row_ror:n operand of
// the FMA itself instead of a v_readlane per element.  fp64 keeps the natural order (readlane path).
template <typename R> __device__ __forceinline__ int wcol(int r, int lane)
{
    if constexpr (sizeof(R) == 4) return (r & 48) | ((lane - r) & 15);
    else return r;
}

// Row stride of the LDS staging area used to permute a weight row into rotating-operand order.
constexpr int kStageStride = kMaxH + 1;
constexpr int kStageElems = (kMaxH / 2) * kStageStride;      // two passes of 32 lanes: 8.3 KB instead of 16.6 KB per wave

// Load one parameter set into registers.  `stage` = wave-private LDS scratch of kStageElems reals (fp32 only;
// may be nullptr for fp64): every lane reads its weight row with coalesced 16-byte loads, drops it into LDS
// and reads it back in the lane-dependent rotated order -- no per-lane gather from global memory (which
// cost 630 B/lane of scratch spills = 170 MB of extra HBM traffic per launch) and no long-lived temporaries.
// Output layer in rotating order (fp32): lane (r, i) = lane 16 r + i keeps, for output o = i & 7,
//     w5r[n] = Wout[o][16 r + ((i - n) & 15)]   n = 0..7        (zero for o >= 6)
// so that sum_n row_ror:n(h) * w5r[n] -- 8 FMAs on the activation vector in its natural layout -- is one half of the lane's
// row-r part of output o; lane i + 8 of the same row (same output) holds the other half (its eight sources are the other
// eight lanes), a row_ror:8 add joins them, and an all-reduce over the four rows (2 swaps + 2 adds) leaves out_o on every lane
// with i & 7 == o: the replicated layout of the state.  15 vector instructions instead of the 35 of six products +
// wave_reduce6_to_lanes, for two more weight registers.
template <typename WT> __device__ __forceinline__ void out_rot_fill(WT &W, const float *__restrict__ pout, int H, int lane)
{
    const int i = lane & 15, r = lane >> 4, o = lane & 7;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int k = 16 * r + ((i - n) & 15);
        const bool ok = o < 6 && k < H;
        W.w5r[n] = ok ? pout[(ok ? o : 0) * H + (ok ? k : 0)] : 0.f;
    }
}
template <typename WT> __device__ __forceinline__ void out_rot_fill(WT &, const double *__restrict__, int, int) {}

// first / last layer weights and all biases ("edge" parameters) of one parameter set -> registers of lane j
template <typename R, int NL, typename WT>
__device__ __forceinline__ void mlp_load_edges(WT &W, const R *__restrict__ p, int H, int lane)
{
    const R live = (lane < H) ? R(1) : R(0);
    const int j = (lane < H) ? lane : H - 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) W.w1[i] = live * p[j * 9 + i];
    W.w1g = W.w1[4] + W.w1[7];        // input row = [t, G, I, Glu, GLP1, GE, FFA, GLP1, tvns]: the two GLP1 columns act as one
    p += 9 * H;
    W.b[0] = live * p[j];
    p += H;
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
        p += (size_t)H * H;
        W.b[l + 1] = live * p[j];
        p += H;
    }
#pragma unroll
    for (int o = 0; o < 6; ++o) W.w5[o] = live * p[o * H + j];
    out_rot_fill(W, p, H, lane);
    p += 6 * H;
    W.b5 = ((lane & 7) < 6) ? p[((lane & 7) < 6) ? (lane & 7) : 0] : R(0);   // replicated per 8-lane group
    if constexpr (sizeof(R) == 4) W.b5 = (lane < 8) ? W.b5 : R(0);            // fp32: enters out_rot once, before the row sums
}

template <typename R, int NL>
__device__ __forceinline__ void mlp_load(MlpRegs<R, NL> &W, const R *__restrict__ p, int H, int lane, R *stage)
{
    // branch-free: out-of-range lanes / columns read a clamped (valid) address and are zeroed
    const R live = (lane < H) ? R(1) : R(0);
    const int j = (lane < H) ? lane : H - 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) W.w1[i] = live * p[j * 9 + i];
    W.w1g = W.w1[4] + W.w1[7];        // input row = [t, G, I, Glu, GLP1, GE, FFA, GLP1, tvns]: the two GLP1 columns act as one
    p += 9 * H;
    W.b[0] = live * p[j];
    p += H;
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
        const R *row = p + (size_t)j * H;
        if constexpr (sizeof(R) == 4) {
            // row-block order (mlp_hidden_blk): lane (r, i) = lane 16 r + i keeps, for w = 0..3 and n = 0..15,
            //     weight (w, n) = W_l[16 w + i][16 r + ((i - n) & 15)]    -> half (w >> 1) of the pair wh[l][2 n + (w & 1)]
            // gathered straight from L2 (192 dword loads per lane and trajectory, ~0.5 % of a 241-point solve)
            (void)stage;
            (void)row;
            const int i = lane & 15, r = lane >> 4;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int u = 16 * w + i;
#pragma unroll
                for (int n = 0; n < 16; ++n) {
                    const int c = 16 * r + ((i - n) & 15);
                    const bool ok = u < H && c < H;
                    const R v = ok ? p[(size_t)(ok ? u : 0) * H + (ok ? c : 0)] : R(0);
                    if (w >> 1) W.wh[l][2 * n + (w & 1)].y = v;
                    else W.wh[l][2 * n + (w & 1)].x = v;
                }
            }
        } else {
            (void)stage;
            if (H == kMaxH) {
                const double2 *r2 = reinterpret_cast<const double2 *>(row);
#pragma unroll
                for (int k = 0; k < kMaxH / 2; ++k) {
                    double2 v = r2[k];
                    W.wh[l][2 * k + 0] = v.x; W.wh[l][2 * k + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < kMaxH; ++k) W.wh[l][k] = ((k < H) ? live : R(0)) * row[(k < H) ? k : H - 1];
            }
        }
        p += (size_t)H * H;
        W.b[l + 1] = live * p[j];
        p += H;
    }
#pragma unroll
    for (int o = 0; o < 6; ++o) W.w5[o] = live * p[o * H + j];
    out_rot_fill(W, p, H, lane);
    p += 6 * H;
    W.b5 = ((lane & 7) < 6) ? p[((lane & 7) < 6) ? (lane & 7) : 0] : R(0);   // replicated per 8-lane group
    if constexpr (sizeof(R) == 4) W.b5 = (lane < 8) ? W.b5 : R(0);            // fp32: enters out_rot once, before the row sums
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
// a / b.  fp32: a * v_rcp_f32(b) (the reciprocal is good to 1 ulp; 2 VALU instead of the ~10 of an IEEE division -- a
// Newton step on the reciprocal, 2 more, bought nothing the fp32 parity bars can see); fp64 (parity runs): exact division.
__device__ __forceinline__ float rdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ double rdiv(double a, double b) { return a / b; }
__device__ __forceinline__ float rabs(float a) { return __builtin_fabsf(a); }
__device__ __forceinline__ double rabs(double a) { return __builtin_fabs(a); }

// gastric-distension Hill term (models/ode_core.py:139-140); only evaluated when a GD input exists
template <typename R> __device__ __forceinline__ R gd_effect(const OdeP<R> &o, R gd)
{
    R u = rpow(gd, o.g), v = rpow(o.IGD_50, o.g);
    return u / (v + u);
}

// ------------------------------------------------------------------------------------------
// One 64x64 hidden layer: out_j = b_j + sum_k W[j][k] h_k, lane j owns row j, h_k lives in lane k.
//  fp32: the four 16-lane rows of h are first replicated to every row (1 v_permlane16_swap +
//        2 v_permlane32_swap), then each FMA takes its activation through a DPP row_ror:n operand
//        (lane i reads lane (i-n)&15 of its row): 64 v_fmac_f32_dpp + ~8 instead of 64 v_readlane +
//        32 v_pk_fma.  Four independent accumulators.
//  fp64: v_readlane broadcast (no 64-bit DPP FMA).
// acc += row_ror:N(x) * w as ONE instruction.  hipcc (ROCm 7.2) selects the VOP3 v_fma_f32 for fmaf() and
// its DPP-combine pass cannot fold a v_mov_b32_dpp into a VOP3 op on gfx9, so the VOP2 form is written out.
// Hazard note (cdna_hip_programming.md 5.7: hipcc pads nothing around asm): a DPP operand needs 2 wait
// states after the VALU that wrote it -- rows_replicate() ends with an explicit s_nop 1, and the
// accumulator / weight operands are ordinary (interlocked) VALU operands.
#define HODE_FMAC_ROR(N)                                                                                    \
    template <> __device__ __forceinline__ float fmac_ror<N>(float acc, float x, float w)                   \
    {                                                                                                       \
        asm("v_fmac_f32_dpp %0, %1, %2 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w)); \
        return acc;                                                                                         \
    }
template <int N> __device__ __forceinline__ float fmac_ror(float acc, float x, float w);
template <> __device__ __forceinline__ float fmac_ror<0>(float acc, float x, float w) { return __builtin_fmaf(x, w, acc); }
HODE_FMAC_ROR(1) HODE_FMAC_ROR(2) HODE_FMAC_ROR(3) HODE_FMAC_ROR(4) HODE_FMAC_ROR(5) HODE_FMAC_ROR(6) HODE_FMAC_ROR(7)
HODE_FMAC_ROR(8) HODE_FMAC_ROR(9) HODE_FMAC_ROR(10) HODE_FMAC_ROR(11) HODE_FMAC_ROR(12) HODE_FMAC_ROR(13)
HODE_FMAC_ROR(14) HODE_FMAC_ROR(15)
#undef HODE_FMAC_ROR
__device__ __forceinline__ void rows_replicate(float h, float (&R)[4])
{
    auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)f2i(h), (unsigned)f2i(h), false, false);   // [r0 r0 r2 r2] , [r1 r1 r3 r3]
    auto a = __builtin_amdgcn_permlane32_swap(s16[0], s16[0], false, false);                           // [r0 x4] , [r2 x4]
    auto b = __builtin_amdgcn_permlane32_swap(s16[1], s16[1], false, false);                           // [r1 x4] , [r3 x4]
    R[0] = i2f((int)a[0]); R[2] = i2f((int)a[1]); R[1] = i2f((int)b[0]); R[3] = i2f((int)b[1]);
    // 2 wait states between the swaps (VALU writes) and the first DPP read of R[] in the asm FMAs
    asm volatile("s_nop 1" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]));
}
// ---- one hidden layer as ONE asm statement (fp32, weights in registers) ----------------------------------------------------
// Written as 64 separate asm statements hipcc's hazard recognizer puts an s_nop between any two of them that touch the same
// register (it assumes an opaque asm may have the dst_sel forwarding hazard and does not count other asm statements as wait
// states): one s_nop per four FMAs, ~50 per right-hand side.  As one statement the layer is 75 instructions instead of 94
// (measured: neutral for the forward, -1 % for forward-with-tape and adjoint -- at two waves per SIMD an s_nop of one wave is
// an issue slot of the other):
//     v_mov + s_nop 1 + v_permlane16_swap, 2 v_mov + s_nop 0 + 2 v_permlane32_swap     rows of h replicated (rows_replicate)
//     v_fma + 3 v_mul                                                                  rotation 0, bias folded in
//     60 v_fmac_f32_dpp                                                                rotations 1..15
//     3 v_add                                                                          (a0 + a1) + (a2 + a3)
// Hazards (the compiler pads nothing inside an asm): a VALU result needs 2 wait states before a v_permlane*_swap reads it
// (the s_nops, as hipcc emits them for the builtins) and before a DPP operand reads it (the four rotation-0 instructions
// stand between the swaps and the first DPP read); accumulators and weights are ordinary interlocked operands.
#define HODE_MV_ROW(n) \
    "v_fmac_f32_dpp %[a0], %[r0], %[w0_" #n "] row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %[a1], %[r1], %[w1_" #n "] row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %[a2], %[r2], %[w2_" #n "] row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t" \
    "v_fmac_f32_dpp %[a3], %[r3], %[w3_" #n "] row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define HODE_MV_ROWS_1_15 \
    HODE_MV_ROW(1) HODE_MV_ROW(2) HODE_MV_ROW(3) HODE_MV_ROW(4) HODE_MV_ROW(5) HODE_MV_ROW(6) HODE_MV_ROW(7) HODE_MV_ROW(8) \
    HODE_MV_ROW(9) HODE_MV_ROW(10) HODE_MV_ROW(11) HODE_MV_ROW(12) HODE_MV_ROW(13) HODE_MV_ROW(14) HODE_MV_ROW(15)
#define HODE_MV_W(n) [w0_##n] "v"(w[n]), [w1_##n] "v"(w[16 + n]), [w2_##n] "v"(w[32 + n]), [w3_##n] "v"(w[48 + n])
#define HODE_MV_WEIGHTS \
    HODE_MV_W(0), HODE_MV_W(1), HODE_MV_W(2), HODE_MV_W(3), HODE_MV_W(4), HODE_MV_W(5), HODE_MV_W(6), HODE_MV_W(7), HODE_MV_W(8), \
    HODE_MV_W(9), HODE_MV_W(10), HODE_MV_W(11), HODE_MV_W(12), HODE_MV_W(13), HODE_MV_W(14), HODE_MV_W(15)
// bias + sum_k W[j][k] h_k with w[16 q + n] on lane j = W[j][16 q + ((j - n) & 15)]; TAIL = "" or the ReLU
#define HODE_MV_LAYER(TAIL)                                                                                                   \
    float r0 = h, r1, r2, r3, a0, a1, a2, a3;                                                                                 \
    asm("v_mov_b32 %[r1], %[r0]\n\t"                                                                                          \
        "s_nop 1\n\t"                                                                                                         \
        "v_permlane16_swap_b32 %[r0], %[r1]\n\t" /* r0 = [h0 h0 h2 h2]   r1 = [h1 h1 h3 h3]   (16-lane rows of h) */          \
        "v_mov_b32 %[r2], %[r0]\n\t"                                                                                          \
        "v_mov_b32 %[r3], %[r1]\n\t"                                                                                          \
        "s_nop 0\n\t"                                                                                                         \
        "v_permlane32_swap_b32 %[r0], %[r2]\n\t" /* r0 = h0 x 4, r2 = h2 x 4 */                                               \
        "v_permlane32_swap_b32 %[r1], %[r3]\n\t" /* r1 = h1 x 4, r3 = h3 x 4 */                                               \
        "v_mul_f32 %[a0], %[r0], %[w0_0]\n\t"                                                                                 \
        "v_mul_f32 %[a1], %[r1], %[w1_0]\n\t"                                                                                 \
        "v_mul_f32 %[a2], %[r2], %[w2_0]\n\t"                                                                                 \
        "v_mul_f32 %[a3], %[r3], %[w3_0]\n\t"                                                                                 \
        HODE_MV_ROWS_1_15                                                                                                     \
        "v_add_f32 %[a0], %[a0], %[a1]\n\t"                                                                                   \
        "v_add_f32 %[a2], %[a2], %[a3]\n\t"                                                                                   \
        "v_add_f32 %[a0], %[a0], %[a2]\n\t"                                                                                   \
        "v_add_f32 %[a0], %[a0], %[bias]" TAIL                                                                                \
        : [r0] "+v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2),      \
          [a3] "=&v"(a3)                                                                                                      \
        : [bias] "v"(bias), HODE_MV_WEIGHTS);                                                                                 \
    return a0;
__device__ __forceinline__ float mlp_hidden(const float (&w)[kMaxH], float bias, float h) { HODE_MV_LAYER("") }
// max(0, .) of the above: NaN -> 0 like rmax0 (v_max_f32 returns the non-NaN operand)
__device__ __forceinline__ float mlp_hidden_relu(const float (&w)[kMaxH], float bias, float h)
{
    HODE_MV_LAYER("\n\tv_max_f32 %[a0], 0, %[a0]")
}
#undef HODE_MV_LAYER
// The four accumulators of a row-block product -- pairs (a0, a2), (a1, a3) -- summed over the four 16-lane rows and transposed
// (row t <- unit 16 t + i): two v_permlane16_swap, ONE packed add, one v_permlane32_swap, one add (+ bias, + ReLU).
// (q0 + q1) + (q2 + q3) + b: the order every fp32 kernel of this library uses.  Every swap reads data at least two instructions old.
template <bool RELU, bool BIAS> __device__ __forceinline__ float blk_rows_finish(f2_t a02, f2_t a13, float bias)
{
#ifndef HODE_FINISH_SPLIT
    asm("s_nop 1\n\t"                          /* (hipcc may have just COPIED an accumulator) */
        "v_permlane16_swap_b32 %[a2], %[a3]\n\t" /* a2 = [u2.q0 u3.q0 u2.q2 u3.q2]   a3 = [u2.q1 u3.q1 u2.q3 u3.q3] */
        "s_nop 0\n\t"
        "v_permlane16_swap_b32 %[a0], %[a1]"      /* a0 = [u0.q0 u1.q0 u0.q2 u1.q2]   a1 = [u0.q1 u1.q1 u0.q3 u1.q3] */
        : [a0] "+v"(a02.x), [a2] "+v"(a02.y), [a1] "+v"(a13.x), [a3] "+v"(a13.y));
    asm("s_nop 0\n\tv_pk_add_f32 %0, %0, %1" : "+v"(a02) : "v"(a13));     // rows: u0 / u2 q0+q1, u1 / u3 q0+q1, q2+q3, q2+q3
    float a0 = a02.x, a2 = a02.y;
#else
    // On the four accumulators as FOUR 32-bit operands of one statement (the halves are free to read; as read-write halves of the two
    // pairs, hipcc wrapped the swaps of two layers out of three in a copy out of and back into a pair): two swaps, two adds
    float a0 = a02.x, a2 = a02.y, a1 = a13.x, a3 = a13.y;
    asm("s_nop 1\n\t"
        "v_permlane16_swap_b32 %[a2], %[a3]\n\t" /* a2 = [u2.q0 u3.q0 u2.q2 u3.q2]   a3 = [u2.q1 u3.q1 u2.q3 u3.q3] */
        "s_nop 0\n\t"
        "v_permlane16_swap_b32 %[a0], %[a1]\n\t" /* a0 = [u0.q0 u1.q0 u0.q2 u1.q2]   a1 = [u0.q1 u1.q1 u0.q3 u1.q3] */
        "v_add_f32 %[a2], %[a2], %[a3]\n\t"      /* rows: u2 / u3 q0+q1, q2+q3 */
        "v_add_f32 %[a0], %[a0], %[a1]"           /* rows: u0 / u1 q0+q1, q2+q3  (the same sums as the packed add: (q0 + q1), (q2 + q3)) */
        : [a0] "+v"(a0), [a2] "+v"(a2), [a1] "+v"(a1), [a3] "+v"(a3));
#endif
    if constexpr (BIAS) {
        if constexpr (RELU) {
            asm("s_nop 1\n\tv_permlane32_swap_b32 %[a0], %[a2]\n\tv_add_f32 %[a0], %[a0], %[a2]\n\tv_add_f32 %[a0], %[a0], %[bias]\n\tv_max_f32 %[a0], 0, %[a0]"
                : [a0] "+v"(a0), [a2] "+v"(a2) : [bias] "v"(bias));
        } else {
            asm("s_nop 1\n\tv_permlane32_swap_b32 %[a0], %[a2]\n\tv_add_f32 %[a0], %[a0], %[a2]\n\tv_add_f32 %[a0], %[a0], %[bias]"
                : [a0] "+v"(a0), [a2] "+v"(a2) : [bias] "v"(bias));
        }
    } else {
        asm("s_nop 1\n\tv_permlane32_swap_b32 %[a0], %[a2]\n\tv_add_f32 %[a0], %[a0], %[a2]" : [a0] "+v"(a0), [a2] "+v"(a2));
    }
    return a0;
}

// ---- one hidden layer WITHOUT row replication (fp32, register kernel) ---------------------------------------------------------
// Lane (r, i) = lane 16 r + i keeps w[16 w + n] = W[16 w + i][16 r + ((i - n) & 15)]: accumulator a_w of the lane is the part
// of unit 16 w + i that comes from the lane's OWN 16-lane row of the activation vector, so h in its natural layout (unit per
// lane) is the DPP operand as it is.  The four accumulators are then added over the rows AND transposed -- row t <- unit
// 16 t + i -- by two v_permlane16_swap, one v_permlane32_swap and three adds (the trick of hode_solve_fwd_rows.hip inside
// one wave):   72 vector instructions per layer instead of the 75 of mlp_hidden, no copies, h back in the natural layout.
// The bias (natural layout: b[16 t + i] on lane (t, i)) is added behind the reduction: (q0 + q1) + (q2 + q3) + b -- the order
// every fp32 forward kernel of this library uses, so that they stay comparable bit for bit.
// Hazards: h is a VALU result (the previous layer's v_max): four plain multiplications stand before the first DPP read; every
// v_permlane*_swap reads accumulators at least two instructions old (the s_nops where nothing else fits).
// Instruction selection (round 3): per rotation n ONE v_mov_b32_dpp materialises row_ror:n(h) and TWO v_pk_fma_f32 update the
// accumulator pairs (a0, a1), (a2, a3) from the weight pairs (w0_n, w1_n), (w2_n, w3_n), both halves taking the low half of the
// moved operand (op_sel_hi:[1,0,1]) -- 47 instructions per layer instead of 64 v_fmac_f32_dpp, 12.3 against 14.7 SIMD cycles per
// rotation at two waves per SIMD (tools/ubench/inst_cost_ubench.hip: k_step_pk2 / k_step_dpp4).  Same products, same order per
// accumulator, one rounding per FMA: bit-identical to the DPP form.
// One asm statement per instruction group: the moved operand must be an asm OPERAND (a 64-bit pair cannot be named as its low half
// for the 32-bit v_mov_b32_dpp inside one statement; naming a fixed pair such as v[0:1] and declaring it clobbered was tried: hipcc
// then spills whatever lived there and reloads it -- with a full vmcnt wait -- inside the stage loop of the taping kernel).  The
// s_nops hipcc puts between the statements are issue slots of the SIMD's other wave and cost nothing measurable.
template <bool RELU> __device__ __forceinline__ float mlp_hidden_blk(const f2_t (&wp)[kMaxH / 2], float bias, float h)
{
    // accumulator pairs (a0, a2) and (a1, a3): after the two 16-lane swaps the sums a0 + a1 and a2 + a3 are ONE packed add
    f2_t a02, a13, hr;
    // h is a FRESH vector result (the previous layer's v_max) and a DPP read needs two wait states behind its producer.  The two
    // products of rotation 0 used to BE those wait states -- by source order; but a DPP move depends on h only, and in some
    // instantiations (RK4 x three layers x tape) hipcc scheduled moves in front of the products: stale lanes, trajectories off by
    // 1e-1, silently (found by tools/soak_tuned.py, pinned down and now checked by tools/dpp_hazard_check.py).  So every reader of h
    // takes it from this statement: two wait states, one issue slot, and the moves stay free to be scheduled early.
    asm("s_nop 1" : "+v"(h));
#ifndef HODE_BLK_UNPINNED
    // The accumulators live in v[4:7] BY NAME (constraint {v[..]} on every statement of the layer, so that the register
    // allocator keeps them there from the first product to the last swap): the finish can then swap HALVES of the pairs and add
    // the PAIRS -- 2 swaps + 1 packed add.  With allocator-chosen pairs the halves are separate operands, and hipcc wrapped the
    // swaps of two layers out of three in a copy out of a pair and back (5 instructions; tools/fwd_valu.py).  Kernels that hold
    // the weights in registers run at two waves per SIMD (256 registers), so v4..255 exist wherever this function is used.
#define HODE_A02 "+{v[4:5]}"(a02)
#define HODE_A13 "+{v[6:7]}"(a13)
    {
        f2_t hh;
        hh.x = h;
        asm("v_pk_mul_f32 v[4:5], %2, %4 op_sel_hi:[1,0]\n\tv_pk_mul_f32 v[6:7], %3, %4 op_sel_hi:[1,0]"
            : "=&{v[4:5]}"(a02), "=&{v[6:7]}"(a13) : "v"(wp[0]), "v"(wp[1]), "v"(hh));
    }
#define HODE_BK_STEP(n)                                                                                                        \
    {                                                                                                                          \
        float lo;                                                                                                              \
        asm("v_mov_b32_dpp %0, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf" : "=v"(lo) : "v"(h));                             \
        hr.x = lo;                                                                                                             \
        asm("v_pk_fma_f32 v[4:5], %2, %4, v[4:5] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 v[6:7], %3, %4, v[6:7] op_sel_hi:[1,0,1]" \
            : HODE_A02, HODE_A13 : "v"(wp[2 * n]), "v"(wp[2 * n + 1]), "v"(hr));                                               \
    }
    HODE_BK_STEP(1) HODE_BK_STEP(2) HODE_BK_STEP(3) HODE_BK_STEP(4) HODE_BK_STEP(5) HODE_BK_STEP(6) HODE_BK_STEP(7) HODE_BK_STEP(8)
    HODE_BK_STEP(9) HODE_BK_STEP(10) HODE_BK_STEP(11) HODE_BK_STEP(12) HODE_BK_STEP(13) HODE_BK_STEP(14) HODE_BK_STEP(15)
#undef HODE_BK_STEP
    // a02 = v4 (a0), v5 (a2); a13 = v6 (a1), v7 (a3).  Same sums as blk_rows_finish: (q0 + q1) + (q2 + q3) + b
    asm("s_nop 1\n\t"
        "v_permlane16_swap_b32 v5, v7\n\t"   /* a2 = [u2.q0 u3.q0 u2.q2 u3.q2]   a3 = [u2.q1 u3.q1 u2.q3 u3.q3] */
        "s_nop 0\n\t"
        "v_permlane16_swap_b32 v4, v6\n\t"   /* a0 = [u0.q0 u1.q0 u0.q2 u1.q2]   a1 = [u0.q1 u1.q1 u0.q3 u1.q3] */
        "s_nop 0\n\t"
        "v_pk_add_f32 v[4:5], v[4:5], v[6:7]\n\t"     /* rows: u0 / u2 q0+q1, u1 / u3 q0+q1, q2+q3, q2+q3 */
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 v4, v5\n\t"
        "v_add_f32 v4, v4, v5\n\t"
        "v_add_f32 v4, v4, %[bias]"
        : HODE_A02, HODE_A13 : [bias] "v"(bias));
#undef HODE_A02
#undef HODE_A13
    float r = a02.x;
    if constexpr (RELU) asm("v_max_f32 %0, 0, %0" : "+v"(r));
    return r;
#else
    // n = 0: the lane's own activation (no rotation); plain products start the sums (h is a fresh VALU result: these two
    // instructions are also the wait states its first DPP read needs)
    {
        f2_t hh;
        hh.x = h;
        asm("v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %3, %4 op_sel_hi:[1,0]" : "=&v"(a02), "=&v"(a13) : "v"(wp[0]), "v"(wp[1]), "v"(hh));
    }
#define HODE_BK_STEP(n)                                                                                                        \
    {                                                                                                                          \
        float lo;                                                                                                              \
        asm("v_mov_b32_dpp %0, %1 row_ror:" #n " row_mask:0xf bank_mask:0xf" : "=v"(lo) : "v"(h));                             \
        hr.x = lo;                                                                                                             \
        asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]"                  \
            : "+v"(a02), "+v"(a13) : "v"(wp[2 * n]), "v"(wp[2 * n + 1]), "v"(hr));                                             \
    }
    HODE_BK_STEP(1) HODE_BK_STEP(2) HODE_BK_STEP(3) HODE_BK_STEP(4) HODE_BK_STEP(5) HODE_BK_STEP(6) HODE_BK_STEP(7) HODE_BK_STEP(8)
    HODE_BK_STEP(9) HODE_BK_STEP(10) HODE_BK_STEP(11) HODE_BK_STEP(12) HODE_BK_STEP(13) HODE_BK_STEP(14) HODE_BK_STEP(15)
#undef HODE_BK_STEP
    return blk_rows_finish<RELU, true>(a02, a13, bias);
#endif
}
// acc[q] += sum_n row_ror:n(R[q]) * w[16 q + n], n ascending within each accumulator; R[] must be two wait states old
__device__ __forceinline__ void rot_matvec64(const float (&w)[kMaxH], const float (&R)[4], float (&acc)[4])
{
    asm("v_fmac_f32 %[a0], %[r0], %[w0_0]\n\tv_fmac_f32 %[a1], %[r1], %[w1_0]\n\t"
        "v_fmac_f32 %[a2], %[r2], %[w2_0]\n\tv_fmac_f32 %[a3], %[r3], %[w3_0]\n\t"
        HODE_MV_ROWS_1_15
        : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
        : [r0] "v"(R[0]), [r1] "v"(R[1]), [r2] "v"(R[2]), [r3] "v"(R[3]), HODE_MV_WEIGHTS);
}
#undef HODE_MV_ROW
#undef HODE_MV_ROWS_1_15
#undef HODE_MV_W
#undef HODE_MV_WEIGHTS
__device__ __forceinline__ double mlp_hidden_relu(const double (&w)[kMaxH], double bias, double h)
{
    const double v = mlp_hidden(w, bias, h);
    return v > 0.0 ? v : 0.0;
}
__device__ __forceinline__ double mlp_hidden(const double (&w)[kMaxH], double bias, double h)
{
    double acc0 = bias, acc1 = 0.0;
#pragma unroll
    for (int k = 0; k < kMaxH; k += 2) {
        acc0 = rfma(w[k], lane_bcast(h, k), acc0);
        acc1 = rfma(w[k + 1], lane_bcast(h, k + 1), acc1);
    }
    return acc0 + acc1;
}

// ---- hidden matrices in a workgroup-shared LDS image (forward solve, fp32) ---------------------------------------
// 211 weight registers per wave cap the register-resident forward kernel at 2 waves per SIMD, where the ~200 plain
// (2-cycle) VALU instructions of the per-RHS fixed part cannot overlap: one wave issues at most one VALU per 4 cycles.
// The image keeps the SAME rotating-operand order, four weights per 16-byte word:
//     img[l][n][lane j] = { W_l[j][16 q + ((j - n) & 15)] : q = 0..3 }          (n = 0..15)
// i.e. exactly the operands of FMA group n of mlp_hidden_step, so the arithmetic (and its order) is bit-identical to the
// register kernel; a layer is 16 conflict-free ds_read_b128 + 64 v_fmac_f32_dpp.  NREG of the NL-1 matrices may still be
// copied to registers (fewer LDS reads, fewer waves): the LDS pipe moves 256 B/clk per CU, four SIMDs of DPP FMAs
// fed from LDS alone would ask for 244 B/clk.
constexpr int kImgVec = 16 * kWave;                     // float4 words per hidden matrix
__device__ __forceinline__ void wimg_store(float *__restrict__ img, const float *__restrict__ nn_p, int H, int NLm1, int tid,
                                           int nthreads)
{
    const float *Wl = nn_p + 9 * H + H;
    for (int l = 0; l < NLm1; ++l) {
        for (int i = tid; i < kMaxH * kMaxH; i += nthreads) {
            const int q = i & 3, j = (i >> 2) & 63, n = i >> 8;
            const int col = 16 * q + ((j - n) & 15);
            img[(size_t)l * kMaxH * kMaxH + i] = (j < H && col < H) ? Wl[(size_t)j * H + col] : 0.f;
        }
        Wl += (size_t)H * H + H;
    }
}
template <int N>
__device__ __forceinline__ void mlp_hidden_lds_step(const float4 *__restrict__ img, int lane, const float (&R)[4], float (&acc)[4])
{
    const float4 w = img[N * kWave + lane];
    acc[0] = fmac_ror<N>(acc[0], R[0], w.x);
    acc[1] = fmac_ror<N>(acc[1], R[1], w.y);
    acc[2] = fmac_ror<N>(acc[2], R[2], w.z);
    acc[3] = fmac_ror<N>(acc[3], R[3], w.w);
    if constexpr (N < 15) mlp_hidden_lds_step<N + 1>(img, lane, R, acc);
}
__device__ __forceinline__ float mlp_hidden_lds(const float4 *__restrict__ img, int lane, float bias, float h)
{
    float R[4];
    rows_replicate(h, R);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    mlp_hidden_lds_step<0>(img, lane, R, acc);
    return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + bias;      // bias last: the order of every fp32 forward kernel
}
template <int NL, int NREG> struct MlpLds {
    static_assert(NREG >= 0 && NREG <= ((NL > 1) ? NL - 1 : 0), "NREG counts hidden matrices");
    float w1[9];
    float w1g;
    float b[NL];
    float w5[6];
    float w5r[8];
    float b5;
    float whr[(NREG > 0) ? NREG : 1][kMaxH];      // the first NREG hidden matrices, register-resident
    const float4 *img;                            // all NL-1 matrices (LDS)
    int lane;
    __device__ __forceinline__ void load_regs()
    {
#pragma unroll
        for (int l = 0; l < NREG; ++l)
#pragma unroll
            for (int n = 0; n < 16; ++n) {
                const float4 w = img[(l * 16 + n) * kWave + lane];
                whr[l][n] = w.x; whr[l][16 + n] = w.y; whr[l][32 + n] = w.z; whr[l][48 + n] = w.w;
            }
    }
    __device__ __forceinline__ float hidden(int l, float h) const
    {
        if (l < NREG) return mlp_hidden(whr[(l < NREG) ? l : 0], b[l + 1], h);
        return mlp_hidden_lds(img + (size_t)l * kImgVec, lane, b[l + 1], h);
    }
};

// dW[j][k] += d_j * h_k in the register order of the weights (rotated for fp32, natural for fp64)
template <int N> __device__ __forceinline__ void mlp_outer_step(float (&gw)[kMaxH], float d, const float (&R)[4])
{
    gw[0 * 16 + N] = fmac_ror<N>(gw[0 * 16 + N], R[0], d);
    gw[1 * 16 + N] = fmac_ror<N>(gw[1 * 16 + N], R[1], d);
    gw[2 * 16 + N] = fmac_ror<N>(gw[2 * 16 + N], R[2], d);
    gw[3 * 16 + N] = fmac_ror<N>(gw[3 * 16 + N], R[3], d);
    if constexpr (N < 15) mlp_outer_step<N + 1>(gw, d, R);
}
__device__ __forceinline__ void mlp_outer_acc(float (&gw)[kMaxH], float d, float hin)
{
    float R[4];
    rows_replicate(hin, R);
    mlp_outer_step<0>(gw, d, R);
}
__device__ __forceinline__ void mlp_outer_acc(double (&gw)[kMaxH], double d, double hin)
{
#pragma unroll
    for (int k = 0; k < kMaxH; ++k) gw[k] = rfma(d, lane_bcast(hin, k), gw[k]);
}

// Activations kept by the backward pass: h[l] = relu output of hidden layer l+1 on lane j.
template <typename R, int NL> struct MlpActs {
    R h[NL];
    __device__ __forceinline__ void put(int l, R v) { h[l] = v; }
};
// ... or written straight to the stage record (row l of 64 reals) the moment a layer is done: the forward solve with a tape
// has no register to hold four rows until the end of the evaluation
template <typename R> struct ActsToRecord {
    R *__restrict__ dst;                  // record + lane
#ifdef HODE_TAPE_NT
    __device__ __forceinline__ void put(int l, R v) { __builtin_nontemporal_store(v, dst + l * kWave); }   // write-once stream: nt policy
#else
    __device__ __forceinline__ void put(int l, R v) { dst[l * kWave] = v; }
#endif
};

// x_K of the replicated state layout on every lane.  fp32: a DPP row broadcast into a VGPR (lane K of each 16-lane row holds
// x_K) instead of a v_readlane into an SGPR: the mechanistic terms combine the state with the 17 ODE constants, which live
// in SGPRs, and a VALU instruction reads at most one SGPR -- every (state, constant) pair cost a v_mov_b32 before.
template <int K> __device__ __forceinline__ float state_bcast(float Y)
{
#ifndef HODE_STATE_SGPR
    return i2f(__builtin_amdgcn_update_dpp(0, f2i(Y), 0x150 + K, 0xF, 0xF, true));       // row_newbcast:K (every lane written)
#else
    return lane_bcast(Y, K);
#endif
}
template <int K> __device__ __forceinline__ double state_bcast(double Y) { return lane_bcast(Y, K); }
// acc + x_K * w with the broadcast folded into the FMA (fp32: v_fmac_f32_dpp row_newbcast:K) -- for a component only one
// instruction reads.  Y must be two wait states old (it is the stage state, computed well before the RHS starts).
template <int K> __device__ __forceinline__ float fmac_state(float acc, float Y, float w);
#define HODE_FMAC_STATE(K)                                                                                              \
    template <> __device__ __forceinline__ float fmac_state<K>(float acc, float Y, float w)                             \
    {                                                                                                                   \
        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(Y), "v"(w));   \
        return acc;                                                                                                     \
    }
HODE_FMAC_STATE(0) HODE_FMAC_STATE(1) HODE_FMAC_STATE(2) HODE_FMAC_STATE(3) HODE_FMAC_STATE(4) HODE_FMAC_STATE(5)
#undef HODE_FMAC_STATE
template <int K> __device__ __forceinline__ double fmac_state(double acc, double Y, double w) { return rfma(w, lane_bcast(Y, K), acc); }

// KK with the eight lanes of stage slot s (lanes 8 s .. 8 s + 7) replaced by F.  fp32: the lane mask 0xff << 8 s is scalar
// arithmetic and feeds v_cndmask_b32 as an SGPR pair -- (lane >> 3) == s costs a shift and a compare on the vector ALU in
// every stage.  s must be wave-uniform.
__device__ __forceinline__ float stage_put(float KK, float F, int s)
{
    const unsigned long long m = 0xffull << (8 * s);
    float out;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(out) : "v"(KK), "v"(F), "s"(m));
    return out;
}
__device__ __forceinline__ double stage_put(double KK, double F, int s) { return ((int)(threadIdx.x & 63) >> 3) == s ? F : KK; }

// sel ? term : other, with `term` (a wave-uniform value every lane can compute) evaluated on ALL lanes first.  Left alone,
// hipcc sinks the arithmetic of each term of a select chain into an exec-masked region of the lanes that keep it: the same
// VALU instructions plus a v_cmp / s_and_saveexec / s_cbranch_execz round trip per term (measured: 4 % of the forward solve).
template <typename R> __device__ __forceinline__ R keep_term(bool sel, R term, R other)
{
    asm volatile("" : "+v"(term));
    return sel ? term : other;
}

// ------------------------------------------------------------------------------------------
// Mechanistic part (models/ode_core.py:124-153), evaluated redundantly on every lane from the broadcast state; the lane
// keeps the component of its slot c8 = lane & 7 (GE, slot 4, has no dynamics; slots 6, 7 are padding).
template <typename R>
__device__ __forceinline__ R mech_eval(const OdeP<R> &o, R G, R I, R Glu, R GLP1, R FFA, R meal, R gde, int c8)
{
    // Every product / sum is written out (fused where one rounding is saved) and contraction is off: the bits do not depend
    // on which kernel this is inlined into (the forward variants are compared bit for bit, tests/test_hip_parity.py).
#pragma clang fp contract(off)
    const R u = G - o.G_b, v = I - o.I_b, w = Glu - o.Glu_b;
    const R Pi = rfma(o.rho, GLP1, R(1));
    R dI = rfma(Pi * o.a_GI, u, -(o.k_I * v));                                     // ode_core.py:124-125
    R dGlu = -(o.E_max * rdiv(GLP1, o.EC_50 + GLP1)) * w;                          // :129-130
    R dGLP1 = rfma(o.V_max, rdiv(G, o.K_m + G), -(o.k_L * GLP1));                  // :134-135
    const R k_GE = o.k_GE0 * (R(1) - gde);                                         // :139-140
    R dFFA = rfma(o.p_9, G, rfma(-o.p_8, I, -o.p_7)) * FFA;                        // :144  (-p7 - p8 I + p9 G) F
    R dG = rfma(-k_GE, G, rfma(R(0.005), w, rfma(R(-0.01), v, meal)));             // :148-150
    R r = keep_term(c8 == 0, dG, R(0));
    r = keep_term(c8 == 1, dI, r);
    r = keep_term(c8 == 2, dGlu, r);
    r = keep_term(c8 == 3, dGLP1, r);
    r = keep_term(c8 == 5, dFFA, r);
    return r;
}

// the output layer of out_rot_fill: 1 v_mul + 7 v_fmac_f32_dpp on h (natural layout; the s_nop gives the DPP read of h its
// second wait state after the v_max that produced it), the row_ror:8 add of the two half sums, the all-reduce over the rows.
__device__ __forceinline__ float out_rot(const float (&w)[8], float b5m, float h)
{
    float a, t;
#define HODE_OR(n) "v_fmac_f32_dpp %[a], %[h], %[w" #n "] row_ror:" #n " row_mask:0xf bank_mask:0xf\n\t"
    asm("v_fma_f32 %[a], %[h], %[w0], %[b]\n\t"     // b = bout[o] on lanes 0..7 only: it passes the reductions once
        "s_nop 0\n\t"
        HODE_OR(1) HODE_OR(2) HODE_OR(3) HODE_OR(4) HODE_OR(5) HODE_OR(6) HODE_OR(7)
        "s_nop 1\n\t"
        "v_add_f32_dpp %[a], %[a], %[a] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
#ifndef HODE_OUT_ROT_SWIZZLE
        "v_mov_b32 %[t], %[a]\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %[a], %[t]\n\t"       // a = [r0 r0 r2 r2], t = [r1 r1 r3 r3]
#else
        // experiment: the partner row (lane ^ 16) through the LDS crossbar -- ds_swizzle_b32, BITMASK_PERM xor 0x10 -- saves
        // two vector instructions per right-hand side and measured 1.5 % SLOWER: its latency sits on the stage's dependent chain
        "ds_swizzle_b32 %[t], %[a] offset:0x401f\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
#endif
        "v_add_f32 %[a], %[a], %[t]\n\t"             // [r0+r1 x2, r2+r3 x2]
        "v_mov_b32 %[t], %[a]\n\t"
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %[a], %[t]\n\t"       // a = [r0+r1 x4], t = [r2+r3 x4]
        "v_add_f32 %[a], %[a], %[t]"
        : [a] "=&v"(a), [t] "=&v"(t)
        : [h] "v"(h), [b] "v"(b5m), [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[3]), [w4] "v"(w[4]), [w5] "v"(w[5]),
          [w6] "v"(w[6]), [w7] "v"(w[7]));
#undef HODE_OR
    return a;
}

// ------------------------------------------------------------------------------------------
// RHS  f(t, x, u) = ODECore + NNResidual  (models/hybrid_ode_nn.py:108-134)
//   Y   lane-distributed state: lane l holds x_{l&7} (replicated over the eight 8-lane groups;
//       fp32 reads lane k of EVERY 16-lane row for x_k, fp64 lane k of the wave)
//   returns the derivative in the same replicated layout (component slots 6,7 hold 0)
//   W   weights holder: MlpRegs (everything in VGPRs) or MlpLds (hidden matrices in a workgroup-shared LDS image)
#ifdef HODE_FWD_TRACE
// experiment build only (tools/build_variant.sh fwdtrace -DHODE_FWD_TRACE=<workgroup>; tools/fwd_trace.py): shader-clock stamps of ONE
// wave at six points of every right-hand side it evaluates -- entry | mechanistic terms | first layer | hidden layers 1..3 | return --, in a ring of 4 096 records
static __device__ unsigned long long g_ft[4096 * 8];
static __device__ unsigned g_ft_n;
#define HODE_FT(i, v) if (ft_on) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ft[i]), "+v"(v))
#else
#define HODE_FT(i, v)
#endif
template <typename R, int NL, bool KEEP, typename WT, typename ACTS = MlpActs<R, NL>>
__device__ __forceinline__ R rhs_eval(const WT &W, const OdeP<R> &o, R t, R Y, R meal, R tvns,
                                      R gde /* Hill term, 0 without GD */, int lane, ACTS *acts)
{
#ifdef HODE_FWD_TRACE
    const bool ft_on = sizeof(R) == 4 && NL == 4 && blockIdx.x == HODE_FWD_TRACE;
    unsigned long long ft[8] = {};
#endif
    HODE_FT(0, Y);
    const R G = state_bcast<0>(Y), I = state_bcast<1>(Y), Glu = state_bcast<2>(Y), GLP1 = state_bcast<3>(Y),
            FFA = state_bcast<5>(Y);
    const int c8 = lane & 7;
    R mech = mech_eval(o, G, I, Glu, GLP1, FFA, meal, gde, c8);
    HODE_FT(6, mech);
    // ---- MLP (models/nn_residual.py:138-147): input row [t, G, I, Glu, GLP1, GE, FFA, glp1:=GLP1, tvns]
    R h = W.b[0];
    h = rfma(W.w1[0], t, h);
    h = rfma(W.w1[1], G, h);
    h = rfma(W.w1[2], I, h);
    h = rfma(W.w1[3], Glu, h);
    h = rfma(W.w1g, GLP1, h);                    // columns 4 and 7 (both GLP1), folded by the loaders
    h = fmac_state<4>(h, Y, W.w1[5]);            // GE: only the first layer reads it
    h = rfma(W.w1[6], FFA, h);
    h = rfma(W.w1[8], tvns, h);
    h = rmax0(h);
    HODE_FT(1, h);
    if constexpr (KEEP) acts->put(0, h);
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
        if constexpr (applies_relu<WT>::value) h = W.hidden_relu(l, h);
        else h = rmax0(W.hidden(l, h));
        HODE_FT(2 + l, h);
        if constexpr (KEEP) acts->put(l + 1, h);
    }
    if constexpr (sizeof(R) == 4) {
        // out_rot leaves Wout h + bout on the lanes of slot c8 < 6 and exact zeros on slots 6, 7 (zero weights, zero bias);
        // the mechanistic select chain ends in zero there as well: no final select
        R res = mech + out_rot(W.w5r, W.b5, h);
        HODE_FT(5, res);
#ifdef HODE_FWD_TRACE
        if (ft_on && lane == 0) {
            // slot = a hash of the entry time (an evaluation takes > 2 000 cycles: consecutive ones fall into different slots); a counter
            // in memory would put a load round trip between every two evaluations of the wave that is being measured
            const unsigned n = (unsigned)(ft[0] >> 10);
            for (int i = 0; i < 8; ++i) g_ft[(n & 4095) * 8 + i] = ft[i];
        }
#endif
        return res;
    }
    R nn;
    {
        R p[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) p[q] = W.w5[q] * h;
        nn = wave_reduce6_to_lanes(p, lane);
    }
    return (c8 < 6) ? (mech + nn + W.b5) : R(0);
}

// ------------------------------------------------------------------------------------------
// Backward (VJP) building blocks -- used by K4 (adjoint) and K5 (RHS backward).
__device__ __forceinline__ void atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double *p, double v) { unsafeAtomicAdd(p, v); }

// LDS image of the TRANSPOSED hidden matrices in "rotating operand" order (shared by a workgroup):
//   wt[l][r >> 2][k][r & 3] = W_l[ (r & 48) | ((k - r) & 15) ][k]        r = 16q + n, k = lane
// so that delta_{l-1}[k] = sum_j W_l[j][k] delta_l[j] becomes, on lane k,
//   sum_r  row_ror:n( rows-replicated delta_l )[k] * wt[l][r][k]
// i.e. 64 v_fmac_f32_dpp fed by 16 conflict-free 16-byte LDS reads per layer (fp32).  The fp64
// instantiation (parity runs) uses the same image with v_readlane broadcasts.
template <typename R>
__device__ __forceinline__ void wt_rot_store(R *__restrict__ wt, const R *__restrict__ nn_p, int H, int NLm1, int tid,
                                             int nthreads)
{
    const R *Wl = nn_p + 9 * H + H;
    for (int l = 0; l < NLm1; ++l) {
        for (int i = tid; i < kMaxH * kMaxH; i += nthreads) {
            const int r = ((i >> 8) << 2) | (i & 3), k = (i >> 2) & 63;
            const int j = (r & 48) | ((k - r) & 15);
            wt[(size_t)l * kMaxH * kMaxH + i] = (j < H && k < H) ? Wl[(size_t)j * H + k] : R(0);
        }
        Wl += (size_t)H * H + H;
    }
}

template <typename R> struct alignas(sizeof(R) * 4) Vec4 { R v[4]; };

// delta_prev = W^T delta for one hidden layer from the LDS image above
template <int RR> __device__ __forceinline__ void wt_mul_step(const Vec4<float> *__restrict__ wt4, int lane, const float (&Rr)[4],
                                                              float (&acc)[4])
{
    const Vec4<float> w = wt4[RR * kMaxH + lane];
    constexpr int q = RR >> 2, n0 = (RR & 3) * 4;
    acc[0] = fmac_ror<n0 + 0>(acc[0], Rr[q], w.v[0]);
    acc[1] = fmac_ror<n0 + 1>(acc[1], Rr[q], w.v[1]);
    acc[2] = fmac_ror<n0 + 2>(acc[2], Rr[q], w.v[2]);
    acc[3] = fmac_ror<n0 + 3>(acc[3], Rr[q], w.v[3]);
    if constexpr ((RR & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep at most 4 LDS reads hoisted
    if constexpr (RR < 15) wt_mul_step<RR + 1>(wt4, lane, Rr, acc);
}
__device__ __forceinline__ float wt_mul(const float *__restrict__ wt, int lane, float d)
{
    float Rr[4];
    rows_replicate(d, Rr);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    wt_mul_step<0>(reinterpret_cast<const Vec4<float> *>(wt), lane, Rr, acc);
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
__device__ __forceinline__ double wt_mul(const double *__restrict__ wt, int lane, double d)
{
    const Vec4<double> *wt4 = reinterpret_cast<const Vec4<double> *>(wt);
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 2
    for (int rr = 0; rr < kMaxH / 4; ++rr) {
        const Vec4<double> w = wt4[rr * kMaxH + lane];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = 4 * rr + c;
            const int j = (r & 48) | ((lane - r) & 15);            // the lane whose delta this entry multiplies
            const double dj = __shfl(d, j);
            if (c & 1) acc1 = rfma(w.v[c], dj, acc1); else acc0 = rfma(w.v[c], dj, acc0);
        }
    }
    return acc0 + acc1;
}

// Where the transposed hidden matrices live for the delta propagation:
//   WtLds  : the LDS image above (2 waves/SIMD fit, every group of 4 reads is an LDS-latency wait)
//   WtRegs : 64 more registers per hidden matrix in the same rotating-operand order (1 wave/SIMD,
//            no memory wait inside the 64-FMA loop); fp32 only
template <typename R> struct WtLds {
    const R *wt;
    __device__ __forceinline__ R mul(int l, int lane, R d) const { return wt_mul(wt + (size_t)l * kMaxH * kMaxH, lane, d); }
};
template <int NL> struct WtRegs {
    float w[(NL > 1) ? NL - 1 : 1][kMaxH];     // w[l][16q+n] on lane k = W_l[16q + ((k - n) & 15)][k]
    __device__ __forceinline__ void load(const float *__restrict__ nn_p, int H, int lane)
    {
        const float *Wl = nn_p + 9 * H + H;
#pragma unroll
        for (int l = 0; l < NL - 1; ++l) {
#pragma unroll
            for (int r = 0; r < kMaxH; ++r) {
                const int j = (r & 48) | ((lane - r) & 15);
                const bool in = (j < H) && (lane < H);
                w[l][r] = in ? Wl[(size_t)(in ? j : 0) * H + (in ? lane : 0)] : 0.f;
            }
            Wl += (size_t)H * H + H;
        }
    }
    __device__ __forceinline__ float mul(int l, int lane, float d) const
    {
        (void)lane;
        float Rr[4];
        rows_replicate(d, Rr);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        // l is a compile-time constant at every call site (unrolled layer loop)
        rot_matvec64(w[l], Rr, acc);
        return (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
};

// One hidden layer of the backward pass: gw += d (x) hin  and  returns W^T d.
// Generic form: the two products one after the other.
template <typename R, typename Wt>
__device__ __forceinline__ R layer_bwd(R (&gw)[kMaxH], const Wt &wt, int l, int lane, R d, R hin, const R *__restrict__ hrow = nullptr)
{
    (void)hrow;
    mlp_outer_acc(gw, d, hin);
    return wt.mul(l, lane, d);
}
// fp32 + LDS-resident transposed matrix: the outer-product FMAs (which need no memory) are interleaved
// with the W^T product so that each group of four 16-byte LDS reads has 16 independent FMAs between its
// issue and its first use -- the LDS latency hides behind work of the same wave.
template <int G>
__device__ __forceinline__ void layer_bwd_group(float (&gw)[kMaxH], const Vec4<float> *__restrict__ wt4, int lane, float d,
                                                const float (&Rh)[4], const float (&Rd)[4], float (&acc)[4])
{
    const Vec4<float> w0 = wt4[(4 * G + 0) * kMaxH + lane], w1 = wt4[(4 * G + 1) * kMaxH + lane],
                      w2 = wt4[(4 * G + 2) * kMaxH + lane], w3 = wt4[(4 * G + 3) * kMaxH + lane];
    __builtin_amdgcn_sched_barrier(0);          // reads are issued HERE, the 16 outer-product FMAs below cover their latency
    constexpr int n = 4 * G;
    gw[0 * 16 + n + 0] = fmac_ror<n + 0>(gw[0 * 16 + n + 0], Rh[0], d);
    gw[1 * 16 + n + 0] = fmac_ror<n + 0>(gw[1 * 16 + n + 0], Rh[1], d);
    gw[2 * 16 + n + 0] = fmac_ror<n + 0>(gw[2 * 16 + n + 0], Rh[2], d);
    gw[3 * 16 + n + 0] = fmac_ror<n + 0>(gw[3 * 16 + n + 0], Rh[3], d);
    gw[0 * 16 + n + 1] = fmac_ror<n + 1>(gw[0 * 16 + n + 1], Rh[0], d);
    gw[1 * 16 + n + 1] = fmac_ror<n + 1>(gw[1 * 16 + n + 1], Rh[1], d);
    gw[2 * 16 + n + 1] = fmac_ror<n + 1>(gw[2 * 16 + n + 1], Rh[2], d);
    gw[3 * 16 + n + 1] = fmac_ror<n + 1>(gw[3 * 16 + n + 1], Rh[3], d);
    gw[0 * 16 + n + 2] = fmac_ror<n + 2>(gw[0 * 16 + n + 2], Rh[0], d);
    gw[1 * 16 + n + 2] = fmac_ror<n + 2>(gw[1 * 16 + n + 2], Rh[1], d);
    gw[2 * 16 + n + 2] = fmac_ror<n + 2>(gw[2 * 16 + n + 2], Rh[2], d);
    gw[3 * 16 + n + 2] = fmac_ror<n + 2>(gw[3 * 16 + n + 2], Rh[3], d);
    gw[0 * 16 + n + 3] = fmac_ror<n + 3>(gw[0 * 16 + n + 3], Rh[0], d);
    gw[1 * 16 + n + 3] = fmac_ror<n + 3>(gw[1 * 16 + n + 3], Rh[1], d);
    gw[2 * 16 + n + 3] = fmac_ror<n + 3>(gw[2 * 16 + n + 3], Rh[2], d);
    gw[3 * 16 + n + 3] = fmac_ror<n + 3>(gw[3 * 16 + n + 3], Rh[3], d);
    __builtin_amdgcn_sched_barrier(0);          // (hipcc otherwise hoists the dependent FMAs right behind the reads;
                                                //  a two-deep pipeline of groups was tried: 32 more live registers spill)
    // rows 4G..4G+3 of the image: r = 16 G + 4 i + c  ->  q = G, n = 4 i + c.  ONE asm statement: between separate
    // statements that share an accumulator hipcc inserts an s_nop (see mlp_hidden above)
    static_assert(G >= 0 && G < 4, "four groups of sixteen rotations");
    if constexpr (G == 0) {
        // the first group starts the four sums with products: no zero-initialised accumulators (four v_mov_b32 per layer)
        asm("v_mul_f32 %[a0], %[r], %[w0]\n\t"
            "v_mul_f32_dpp %[a1], %[r], %[w1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %[a2], %[r], %[w2] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32_dpp %[a3], %[r], %[w3] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a0], %[r], %[w4] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a1], %[r], %[w5] row_ror:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[r], %[w6] row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[r], %[w7] row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a0], %[r], %[w8] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a1], %[r], %[w9] row_ror:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[r], %[w10] row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[r], %[w11] row_ror:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a0], %[r], %[w12] row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a1], %[r], %[w13] row_ror:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[r], %[w14] row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[r], %[w15] row_ror:15 row_mask:0xf bank_mask:0xf"
            : [a0] "=&v"(acc[0]), [a1] "=&v"(acc[1]), [a2] "=&v"(acc[2]), [a3] "=&v"(acc[3])
        : [r] "v"(Rd[G]), [w0] "v"(w0.v[0]), [w1] "v"(w0.v[1]), [w2] "v"(w0.v[2]), [w3] "v"(w0.v[3]), [w4] "v"(w1.v[0]),
          [w5] "v"(w1.v[1]), [w6] "v"(w1.v[2]), [w7] "v"(w1.v[3]), [w8] "v"(w2.v[0]), [w9] "v"(w2.v[1]), [w10] "v"(w2.v[2]),
          [w11] "v"(w2.v[3]), [w12] "v"(w3.v[0]), [w13] "v"(w3.v[1]), [w14] "v"(w3.v[2]), [w15] "v"(w3.v[3]));
    } else {
    asm("v_fmac_f32 %[a0], %[r], %[w0]\n\t"
        "v_fmac_f32_dpp %[a1], %[r], %[w1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[r], %[w2] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[r], %[w3] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a0], %[r], %[w4] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a1], %[r], %[w5] row_ror:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[r], %[w6] row_ror:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[r], %[w7] row_ror:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a0], %[r], %[w8] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a1], %[r], %[w9] row_ror:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[r], %[w10] row_ror:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[r], %[w11] row_ror:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a0], %[r], %[w12] row_ror:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a1], %[r], %[w13] row_ror:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a2], %[r], %[w14] row_ror:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[a3], %[r], %[w15] row_ror:15 row_mask:0xf bank_mask:0xf"
        : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])
        : [r] "v"(Rd[G]), [w0] "v"(w0.v[0]), [w1] "v"(w0.v[1]), [w2] "v"(w0.v[2]), [w3] "v"(w0.v[3]), [w4] "v"(w1.v[0]),
          [w5] "v"(w1.v[1]), [w6] "v"(w1.v[2]), [w7] "v"(w1.v[3]), [w8] "v"(w2.v[0]), [w9] "v"(w2.v[1]), [w10] "v"(w2.v[2]),
          [w11] "v"(w2.v[3]), [w12] "v"(w3.v[0]), [w13] "v"(w3.v[1]), [w14] "v"(w3.v[2]), [w15] "v"(w3.v[3]));
    }
    __builtin_amdgcn_sched_barrier(0);
}
// hrow != nullptr: the 64 activations h_in[0..63] also sit in LDS (the stage record the DMA delivered): their four 16-lane
// rows are read back replicated -- four conflict-free broadcast reads -- instead of being replicated through
// v_permlane swaps (3 swaps + 3 copies of VALU time per layer)
__device__ __forceinline__ float layer_bwd(float (&gw)[kMaxH], const WtLds<float> &wt, int l, int lane, float d, float hin,
                                           const float *__restrict__ hrow = nullptr)
{
    const Vec4<float> *wt4 = reinterpret_cast<const Vec4<float> *>(wt.wt + (size_t)l * kMaxH * kMaxH);
    float Rh[4], Rd[4];
    if (hrow != nullptr) {
        const int p16 = lane & 15;
        Rh[0] = hrow[p16]; Rh[1] = hrow[16 + p16]; Rh[2] = hrow[32 + p16]; Rh[3] = hrow[48 + p16];
        // the DPP reads of Rh[] in the asm FMAs need no wait states after an LDS return (not a VALU write), but keep the
        // compiler from sinking the loads below the asm block boundary
        asm volatile("" : "+v"(Rh[0]), "+v"(Rh[1]), "+v"(Rh[2]), "+v"(Rh[3]));
    } else {
        rows_replicate(hin, Rh);
    }
    rows_replicate(d, Rd);
    float acc[4];                                   // started by group 0
    __builtin_amdgcn_sched_barrier(0);
    layer_bwd_group<0>(gw, wt4, lane, d, Rh, Rd, acc);
    layer_bwd_group<1>(gw, wt4, lane, d, Rh, Rd, acc);
    layer_bwd_group<2>(gw, wt4, lane, d, Rh, Rd, acc);
    layer_bwd_group<3>(gw, wt4, lane, d, Rh, Rd, acc);
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// ------------------------------------------------------------------------------------------
// First/last-layer weights and their gradient accumulators ("edge" parameters: W1[.,9], b_1..b_NL,
// Wout[6,.], bout) behind a small policy, so that the adjoint kernel can keep them in LDS and spend
// its registers on the 3 x 64 hidden-matrix accumulators:
//   EdgeRegs : everything in VGPRs (K5, fp64 parity builds)
//   EdgeLds  : weights in a workgroup-shared LDS table, accumulators in a wave-private LDS table
//              (read-modify-write; the table is private to the wave, so no atomics are needed)
// slots: 0..8 W1 columns, 9..9+NL-1 hidden biases, then 6 Wout rows, then bout (lane o < 6).
template <int NL> struct EdgeSlots { static constexpr int w1 = 0, b = 9, w5 = 9 + NL, b5 = 15 + NL, count = 16 + NL; };

template <typename R, int NL> struct EdgeRegs {
    R w[EdgeSlots<NL>::count];
    R gacc[EdgeSlots<NL>::count];
    __device__ __forceinline__ R W(int slot) const { return w[slot]; }
    __device__ __forceinline__ void add(int slot, R v) { gacc[slot] += v; }
    __device__ __forceinline__ void fma(int slot, R a, R b) { gacc[slot] = rfma(a, b, gacc[slot]); }
    __device__ __forceinline__ R G(int slot) const { return gacc[slot]; }
    // gacc[slot] += v[slot] for the first N slots
    template <int N> __device__ __forceinline__ void add_all(const R (&v)[EdgeSlots<NL>::count])
    {
#pragma unroll
        for (int i = 0; i < N; ++i) gacc[i] += v[i];
    }
};
template <typename R, int NL> struct EdgeLds {
    const R *w;      // [slots][64] shared by the workgroup
    R *gacc;         // [slots][64] private to the wave
    int lane;
    __device__ __forceinline__ R W(int slot) const { return w[slot * kWave + lane]; }
    // wave-private table: a plain read-modify-write is safe and runs at the full LDS rate
    // (ds_add_f32 serialises per lane: measured 2.5x slower for the whole kernel)
    __device__ __forceinline__ void add(int slot, R v) { gacc[slot * kWave + lane] += v; }
    __device__ __forceinline__ void fma(int slot, R a, R b) { gacc[slot * kWave + lane] = rfma(a, b, gacc[slot * kWave + lane]); }
    __device__ __forceinline__ R G(int slot) const { return gacc[slot * kWave + lane]; }
    // all reads first (one LDS wait instead of one per slot), then the adds, then all writes
    template <int N> __device__ __forceinline__ void add_all(const R (&v)[EdgeSlots<NL>::count])
    {
        R cur[N];
#pragma unroll
        for (int i = 0; i < N; ++i) cur[i] = gacc[i * kWave + lane];
#pragma unroll
        for (int i = 0; i < N; ++i) gacc[i * kWave + lane] = cur[i] + v[i];
    }
};
// edge weights of one parameter set into a [slots][64] table (weights only; accumulators start at 0)
template <typename R, int NL>
__device__ __forceinline__ void edge_table_store(R *__restrict__ tab, const R *__restrict__ p, int H, int tid, int nthreads)
{
    using S = EdgeSlots<NL>;
    const R *pout = p + 9 * H + H + (size_t)(NL - 1) * ((size_t)H * H + H);
    for (int i = tid; i < S::count * kWave; i += nthreads) {
        const int slot = i >> 6, j = i & 63;
        R v = R(0);
        if (j < H) {
            if (slot < S::b) v = p[j * 9 + slot];
            else if (slot >= S::w5 && slot < S::b5) v = pout[(slot - S::w5) * H + j];
        }
        tab[i] = v;            // bias slots hold no weight the VJP needs
    }
}
// flush the edge accumulators into the flat gradient vector
template <typename R, int NL, typename Edge>
__device__ __forceinline__ void edge_flush(const Edge &e, R *__restrict__ gp, int H, int lane)
{
    using S = EdgeSlots<NL>;
    const bool live = lane < H;
    if (live) {
#pragma unroll
        for (int i = 0; i < 9; ++i) atomic_add(gp + lane * 9 + i, e.G(S::w1 + i));
        atomic_add(gp + 9 * H + lane, e.G(S::b + 0));
    }
    R *q = gp + 9 * H + H;
#pragma unroll
    for (int l = 1; l < NL; ++l) {
        q += (size_t)H * H;
        if (live) atomic_add(q + lane, e.G(S::b + l));
        q += H;
    }
    if (live) {
#pragma unroll
        for (int o = 0; o < 6; ++o) atomic_add(q + o * H + lane, e.G(S::w5 + o));
    }
    if (lane < 6) atomic_add(q + 6 * H + lane, e.G(S::b5));
}
// hidden-matrix accumulators only
template <typename R, int NL>
__device__ __forceinline__ void hidden_flush(const R (&wh)[(NL > 1) ? NL - 1 : 1][kMaxH], R *__restrict__ gp, int H, int lane)
{
    const bool live = lane < H;
    R *q = gp + 9 * H + H;
#pragma unroll
    for (int l = 0; l < NL - 1; ++l) {
        if (live) {
#pragma unroll
            for (int k = 0; k < kMaxH; ++k) {
                const int c = wcol<R>(k, lane);            // register k of lane j is column c (rotated order in fp32)
                if (c < H) atomic_add(q + (size_t)lane * H + c, wh[l][k]);
            }
        }
        q += (size_t)H * H + H;
    }
}

// J_mech^T kb (analytic Jacobian of models/ode_core.py:124-153) in the replicated layout; GODE: also d f / d(ode constant p) . kb
// for the 17 constants, accumulated LANE-DISTRIBUTED: lane p < 17 of the single register `go` holds the running sum for
// constant p (17 separate uniform accumulators cost 16 more VGPRs, which the adjoint kernel does not have).
template <typename R, bool GODE>
__device__ __forceinline__ R mech_vjp(const OdeP<R> &o, R G, R I, R Glu, R GLP1, R FFA, R lG, R lI, R lGlu, R lGLP, R lF, R gde,
                                      R gd_in, bool use_gd, int lane, R &go)
{
    const R Pi = R(1) + o.rho * GLP1;
    const R den1 = o.EC_50 + GLP1, den2 = o.K_m + G;
    const R k_GE = o.k_GE0 * (R(1) - gde);
    const R r1 = rdiv(R(1), den1), r2 = rdiv(R(1), den2);
    const R oG = -k_GE * lG + Pi * o.a_GI * lI + o.V_max * o.K_m * r2 * r2 * lGLP + o.p_9 * FFA * lF;
    const R oI = R(-0.01) * lG - o.k_I * lI - o.p_8 * FFA * lF;
    const R oGlu = R(0.005) * lG - o.E_max * GLP1 * r1 * lGlu;
    const R oGLP = o.rho * o.a_GI * (G - o.G_b) * lI - o.E_max * o.EC_50 * r1 * r1 * (Glu - o.Glu_b) * lGlu - o.k_L * lGLP;
    const R oF = (-o.p_7 - o.p_8 * I + o.p_9 * G) * lF;
    const int c8 = lane & 7;
    // (select chains that keep hipcc from sinking the terms into exec-masked regions -- keep_term as in mech_eval, or
    //  v_cndmask_b32 with literal lane masks -- were tried here: the adjoint kernel, at its register limit, answers with
    //  60-90 B of scratch instead of 28-36 and reloads inside the stage loop)
    const R mech = (c8 == 0) ? oG : (c8 == 1) ? oI : (c8 == 2) ? oGlu : (c8 == 3) ? oGLP : (c8 == 5) ? oF : R(0);
    if constexpr (GODE) {
        R c[17];
        c[0] = lI * Pi * (G - o.G_b);
        c[1] = -lI * (I - o.I_b);
        c[2] = lI * GLP1 * o.a_GI * (G - o.G_b);
        c[3] = -lI * Pi * o.a_GI;
        c[4] = lI * o.k_I + lG * R(0.01);
        c[5] = -lGlu * GLP1 * r1 * (Glu - o.Glu_b);
        c[6] = lGlu * o.E_max * GLP1 * r1 * r1 * (Glu - o.Glu_b);
        c[7] = lGlu * o.E_max * GLP1 * r1 - lG * R(0.005);
        c[8] = lGLP * G * r2;
        c[9] = -lGLP * o.V_max * G * r2 * r2;
        c[10] = -lGLP * GLP1;
        c[11] = -lG * G * (R(1) - gde);
        c[12] = R(0);
        c[13] = R(0);
        if (use_gd && gd_in > R(0)) {
            const R u = rpow(gd_in, o.g), v = rpow(o.IGD_50, o.g), s2 = (v + u) * (v + u);
            c[12] = lG * o.k_GE0 * G * (-u * o.g * rpow(o.IGD_50, o.g - R(1)) / s2);
            c[13] = lG * o.k_GE0 * G * (u * v * (rlog(gd_in) - rlog(o.IGD_50)) / s2);
        }
        c[14] = -lF * FFA;
        c[15] = -lF * I * FFA;
        c[16] = lF * G * FFA;
        R sel = R(0);
#pragma unroll
        for (int p = 0; p < 17; ++p) sel = (lane == p) ? c[p] : sel;
        go += sel;
    }
    return mech;
}

// VJP of rhs_eval.  kb = cotangent of f (replicated layout); returns the cotangent of the state in the
// same layout and accumulates parameter gradients.  acts = activations of this evaluation (from
// rhs_eval<KEEP> or from the stage tape).  Needs only the first/last layer weights in registers;
// the hidden matrices come transposed from LDS (wt).   GODE: also d/d(ode constants) (wave-uniform values).
//   Y     the stage state in the replicated layout (lane l holds x_{l&7}; from the compact stage record / the input batch)
//   acts  h_1 .. h_NL of this evaluation.  (Recomputing h_1 here from (t, x, tvns) -- 9 FMAs instead of a 256-byte tape row
//         per stage -- was built and measured: it costs the adjoint kernel its last registers, 80 B of scratch with
//         reloads inside the stage loop, whose vmcnt waits serialise behind the record DMA: 8.1 -> 12.5 ms.)
template <typename R, int NL, bool GODE, bool GT, typename Edge, typename Wt>
__device__ __forceinline__ R rhs_vjp(Edge &e, R (&gwh)[(NL > 1) ? NL - 1 : 1][kMaxH], const Wt &wt,
                                     const OdeP<R> &o, R t, R Y, R tvns, R gde, R gd_in, bool use_gd, int lane,
                                     const MlpActs<R, NL> &acts, R kb, R &go, R *gt_out, const R *__restrict__ hrows = nullptr)
{
    // hrows: LDS copy of the record rows h_1 .. h_NL ([NL][64]) or nullptr
    using S = EdgeSlots<NL>;
    // increments of the edge-parameter gradients are collected and applied in ONE batch at the end
    R inc[S::count];
    const R G = lane_bcast(Y, 0), I = lane_bcast(Y, 1), Glu = lane_bcast(Y, 2), GLP1 = lane_bcast(Y, 3),
            GE = lane_bcast(Y, 4), FFA = lane_bcast(Y, 5);
    // h_1 exactly as rhs_eval computes it (same operation order)
    const R h1 = acts.h[0];
    const R lG = lane_bcast(kb, 0), lI = lane_bcast(kb, 1), lGlu = lane_bcast(kb, 2), lGLP = lane_bcast(kb, 3),
            lGE = lane_bcast(kb, 4), lF = lane_bcast(kb, 5);
    const int c8 = lane & 7;
    const R mech = mech_vjp<R, GODE>(o, G, I, Glu, GLP1, FFA, lG, lI, lGlu, lGLP, lF, gde, gd_in, use_gd, lane, go);
    // ---- MLP backward
    const R hl = (NL > 1) ? acts.h[NL - 1] : h1;
    // fetch the six output-layer weights in one batch (LDS policy: six reads in flight, one wait)
    const R w50 = e.W(S::w5 + 0), w51 = e.W(S::w5 + 1), w52 = e.W(S::w5 + 2), w53 = e.W(S::w5 + 3),
            w54 = e.W(S::w5 + 4), w55 = e.W(S::w5 + 5);
    R d = w50 * lG;
    d = rfma(w51, lI, d);
    d = rfma(w52, lGlu, d);
    d = rfma(w53, lGLP, d);
    d = rfma(w54, lGE, d);
    d = rfma(w55, lF, d);
    inc[S::b5] = kb;                             // lane o < 6 holds d bout[o] (other groups hold copies)
    inc[S::w5 + 0] = lG * hl;
    inc[S::w5 + 1] = lI * hl;
    inc[S::w5 + 2] = lGlu * hl;
    inc[S::w5 + 3] = lGLP * hl;
    inc[S::w5 + 4] = lGE * hl;
    inc[S::w5 + 5] = lF * hl;
    d = (hl > R(0)) ? d : R(0);
#pragma unroll
    for (int l = NL - 1; l >= 1; --l) {           // hidden matrix l-1 maps acts.h[l-1] -> acts.h[l]
        const R hin = (l > 1) ? acts.h[l - 1] : h1;
        inc[S::b + l] = d;
        const R dp = layer_bwd(gwh[l - 1], wt, l - 1, lane, d, hin, hrows ? hrows + (l - 1) * kWave : nullptr);   // dW_l += d (x) h_{l-1};  dp = W_l^T d
        d = (hin > R(0)) ? dp : R(0);
    }
    inc[S::b + 0] = d;
    inc[S::w1 + 0] = d * t;
    inc[S::w1 + 1] = d * G;
    inc[S::w1 + 2] = d * I;
    inc[S::w1 + 3] = d * Glu;
    inc[S::w1 + 4] = d * GLP1;
    inc[S::w1 + 5] = d * GE;
    inc[S::w1 + 6] = d * FFA;
    inc[S::w1 + 7] = d * GLP1;
    inc[S::w1 + 8] = d * tvns;
    e.template add_all<S::count>(inc);
    const R w11 = e.W(S::w1 + 1), w12 = e.W(S::w1 + 2), w13 = e.W(S::w1 + 3), w14 = e.W(S::w1 + 4), w15 = e.W(S::w1 + 5),
            w16 = e.W(S::w1 + 6), w17 = e.W(S::w1 + 7);
    R p[6];
    p[0] = w11 * d;
    p[1] = w12 * d;
    p[2] = w13 * d;
    p[3] = (w14 + w17) * d;                       // GLP1 feeds inputs 4 and 7
    p[4] = w15 * d;
    p[5] = w16 * d;
    const R nn = wave_reduce6_to_lanes(p, lane);
    if constexpr (GT) *gt_out = wave_allsum(e.W(S::w1 + 0) * d);
    return (c8 < 6) ? (mech + nn) : R(0);
}

}  // namespace hode
