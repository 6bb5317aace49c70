// hode_generic.hip -- the GENERIC network path: K1 / K2+K3 / K4 / K5 for every MLP shape outside the tuned envelope,
// up to hidden width 128 and 8 hidden layers (reference configs/ablation_no_physics.yaml:11-12 trains nn_hidden 128,
// nn_layers 5; models/nn_residual.py:28-98 builds any width / depth).
//
// The tuned kernels (hode_solve_fwd.hip, hode_solve_bwd_ws.hip) keep a 4 x 64 network in registers: 211 weight registers
// per lane, gradient accumulators in registers, matrices compiled in as template parameters.  A 5 x 128 network has
// 67 k parameters (270 KB): it fits neither the register file of a wave nor the 160 KB of LDS.  So here
//   * depth is a run-time loop, every lane owns TWO hidden units (j and j + 64) in the "natural" layout of a layer's activations,
//     and the stage tape records 2 L rows of 64 + the state in 8 reals per stage;
//   * K1 / K5 (one evaluation per sample) and one-wave solves walk the matrices in their PyTorch layout, h_k by v_readlane
//     (rhs_stream / rhs_vjp_stream), parameter gradients through coalesced global atomics;
//   * the SOLVE kernels give every trajectory a TEAM of 4 / 8 waves (one workgroup) -- a trajectory of such a network is one long
//     chain of dependent evaluations, and the batches these shapes are trained on (32 windows) would leave the chip empty:
//       forward  rhs_rows: hidden layers split by output rows, activations through LDS, one barrier per layer, the wave's rows
//                register-resident for small batches, streamed from L2 (16-byte loads) for large ones;
//       adjoint  rhs_vjp_stream with TeamAccT: the wave's rows of W^T delta as partial sums (LDS exchange), ALL parameter gradients
//                in registers (compile-time indexed), one flush per workgroup and parameter set;
//     the edge parameters (first / output matrix, biases) come from an LDS image (EdgeImage).
// DESIGN.md section 4.6 has the measurements that led here.  The integrator, controller, tape format and state layout are those
// of the tuned path (hode_solve_body.h); CPU restatement: oracle/hode_oracle_impl.h (HODE_MAXH 128, HODE_MAXL 8).
#include "hode_solve_body.h"
#include <type_traits>

namespace hode {

// one parameter set in the flat PyTorch parameters() layout (include/hode.h)
// activation (wave-uniform code, include/hode.h HODE_ACT_*) and "cotangent times its derivative" from the POST-activation value:
// h > 0 iff the pre-activation is > 0 for all four; ELU'(x <= 0) = exp(x) = h + 1  (torch: nn.ReLU / Tanh / ELU / LeakyReLU(0.1),
// reference models/nn_residual.py:50-56)
__device__ __forceinline__ float act_f(float s, int act)
{
    if (act == HODE_ACT_RELU) return rmax0(s);
    if (act == HODE_ACT_TANH) return tanhf(s);
    if (act == HODE_ACT_ELU) return s > 0.f ? s : expm1f(s);
    return s > 0.f ? s : 0.1f * s;
}
__device__ __forceinline__ double act_f(double s, int act)
{
    if (act == HODE_ACT_RELU) return rmax0(s);
    if (act == HODE_ACT_TANH) return tanh(s);
    if (act == HODE_ACT_ELU) return s > 0.0 ? s : expm1(s);
    return s > 0.0 ? s : 0.1 * s;
}
template <typename R> __device__ __forceinline__ R act_bwd(R d, R h, int act)
{
    if (act == HODE_ACT_RELU) return h > R(0) ? d : R(0);            // (a select, not a product: 0 * inf stays 0)
    if (act == HODE_ACT_TANH) return d * (R(1) - h * h);
    if (act == HODE_ACT_ELU) return h > R(0) ? d : d * (h + R(1));
    return h > R(0) ? d : R(0.1) * d;
}

template <typename R> struct StreamNet {
    const R *p;
    int H, L;
    int act = HODE_ACT_RELU;
    __device__ __forceinline__ const R *W1() const { return p; }                       // [H][9]
    __device__ __forceinline__ const R *b1() const { return p + 9 * H; }
    __device__ __forceinline__ size_t hid_off(int l) const { return (size_t)9 * H + H + (size_t)l * ((size_t)H * H + H); }
    __device__ __forceinline__ const R *Wh(int l) const { return p + hid_off(l); }       // hidden matrix l = 0..L-2: [H][H]
    __device__ __forceinline__ const R *bh(int l) const { return Wh(l) + (size_t)H * H; }
    __device__ __forceinline__ size_t out_off() const { return hid_off(L - 1); }
    __device__ __forceinline__ const R *Wo() const { return p + out_off(); }             // [6][H]
    __device__ __forceinline__ const R *bo() const { return Wo() + 6 * H; }
};

// Where the EDGE parameters -- first matrix, output matrix, every bias: (16 + L) H + 6 values, 11.5 KB for 5 x 128 -- are read from.
//   EdgeFromParams  the flat parameter vector in global memory (K1 / K5: one evaluation per sample, nothing to amortise)
//   EdgeImage       a copy in the workgroup's LDS, made once per trajectory (solve kernels).  These ~35 loads per evaluation were
//                   issued again at every one of the 360+ evaluations of a trajectory -- hipcc cannot hoist them over the kernel's
//                   stores -- and each group of them is an L2 round trip in front of a dependent computation.
// Image layout: W1[H][9] | b1[H] | b_hidden[L-1][H] | Wout[6][H] | bout[6].
template <typename R> struct EdgeFromParams {
    const StreamNet<R> &n;
    __device__ __forceinline__ const R *W1() const { return n.W1(); }
    __device__ __forceinline__ const R *b1() const { return n.b1(); }
    __device__ __forceinline__ const R *bh(int l) const { return n.bh(l); }
    __device__ __forceinline__ const R *Wo() const { return n.Wo(); }
    __device__ __forceinline__ const R *bo() const { return n.bo(); }
};
constexpr int kEdgeImageMax = (16 + 8) * 128 + 8;        // H <= 128, L <= 8
template <typename R> struct EdgeImage {
    const R *img;
    int H, L;
    __device__ __forceinline__ const R *W1() const { return img; }
    __device__ __forceinline__ const R *b1() const { return img + 9 * H; }
    __device__ __forceinline__ const R *bh(int l) const { return img + (10 + l) * H; }
    __device__ __forceinline__ const R *Wo() const { return img + (9 + L) * H; }
    __device__ __forceinline__ const R *bo() const { return img + (15 + L) * H; }
    // cooperative copy by the whole workgroup (nthreads threads); the caller synchronises before the first use
    static __device__ __forceinline__ void fill(R *__restrict__ img, const StreamNet<R> &n, int tid, int nthreads)
    {
        const int H = n.H, L = n.L;
        for (int i = tid; i < 10 * H; i += nthreads) img[i] = n.p[i];                         // W1 | b1: contiguous
        for (int i = tid; i < (L - 1) * H; i += nthreads) img[10 * H + i] = n.bh(i / H)[i % H];
        for (int i = tid; i < 6 * H + 6; i += nthreads) img[(9 + L) * H + i] = n.Wo()[i];     // Wout | bout: contiguous
    }
};

__device__ __forceinline__ float bcast_dyn(float v, int k) { return i2f(__builtin_amdgcn_readlane(f2i(v), k)); }
__device__ __forceinline__ double bcast_dyn(double v, int k) { return lane_bcast(v, k); }
// activation k of a layer whose units live two per lane (k < 64: register a of lane k; else register b of lane k - 64)
template <typename R> __device__ __forceinline__ R unit_bcast(R a, R b, int k) { return k < 64 ? bcast_dyn(a, k) : bcast_dyn(b, k - 64); }

// f(t, x, u) for an arbitrary network.  rec != nullptr: record h_1..h_L (two rows of 64 each) and the stage state (8 reals).
// NW > 1: a team of NW waves evaluates the same f on identical data (see rhs_vjp_stream); the columns k of every hidden matrix
// are split over the waves, partial sums exchanged through xch[NW][2][64] and added in wave order on every wave.
template <typename R, int NW = 1, typename EW = EdgeFromParams<R>>
__device__ __forceinline__ R rhs_stream(const StreamNet<R> &n, const EW &ew, const OdeP<R> &o, R t, R Y, R meal, R tvns, R gde, int lane,
                                        R *__restrict__ rec, int part = 0, R *__restrict__ xch = nullptr)
{
    const int H = n.H, L = n.L;
    const R G = lane_bcast(Y, 0), I = lane_bcast(Y, 1), Glu = lane_bcast(Y, 2), GLP1 = lane_bcast(Y, 3),
            GE = lane_bcast(Y, 4), FFA = lane_bcast(Y, 5);
    const int c8 = lane & 7;
    const R mech = mech_eval(o, G, I, Glu, GLP1, FFA, meal, gde, c8);
    const bool vA = lane < H, vB = lane + 64 < H;
    const int jA = vA ? lane : 0, jB = vB ? lane + 64 : 0;              // clamped: masked lanes read a valid address
    const R in[9] = {t, G, I, Glu, GLP1, GE, FFA, GLP1, tvns};          // models/nn_residual.py:138-143
    R hA = ew.b1()[jA], hB = ew.b1()[jB];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        hA = rfma(ew.W1()[jA * 9 + i], in[i], hA);
        hB = rfma(ew.W1()[jB * 9 + i], in[i], hB);
    }
    hA = vA ? act_f(hA, n.act) : R(0);
    hB = vB ? act_f(hB, n.act) : R(0);
    if (rec) { rec[lane] = hA; rec[kWave + lane] = hB; }
    const int cols_per = (((H + NW - 1) / NW) + 7) & ~7;             // this wave's columns: a multiple of 8 (the chunked loads)
    const int k0 = (part * cols_per < H) ? part * cols_per : H, k1 = (k0 + cols_per < H) ? k0 + cols_per : H;
    for (int l = 0; l + 1 < L; ++l) {
        const R *__restrict__ rowA = n.Wh(l) + (size_t)jA * H, *__restrict__ rowB = n.Wh(l) + (size_t)jB * H;
        R aA = (NW == 1) ? ew.bh(l)[jA] : R(0), aB = (NW == 1) ? ew.bh(l)[jB] : R(0);
        // chunks of 8 columns: 16 independent loads in flight, then the 16 FMAs (one load, one dependent FMA at a time left
        // a lone wave waiting out an L2 round trip per column: 21.9 ms per 32 x 61 forward of the 5 x 128 network, 7.7 ms now)
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            R wA[8], wB[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { wA[u] = rowA[k + u]; wB[u] = rowB[k + u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const R hk = unit_bcast(hA, hB, k + u);        // k + u < 64 for the whole chunk or >= 64 for the whole chunk
                aA = rfma(wA[u], hk, aA);
                aB = rfma(wB[u], hk, aB);
            }
        }
        for (; k < k1; ++k) {
            const R hk = unit_bcast(hA, hB, k);
            aA = rfma(rowA[k], hk, aA);
            aB = rfma(rowB[k], hk, aB);
        }
        if constexpr (NW > 1) {
            xch[(part * 2 + 0) * kWave + lane] = aA;
            xch[(part * 2 + 1) * kWave + lane] = aB;
            __syncthreads();
            aA = ew.bh(l)[jA]; aB = ew.bh(l)[jB];
#pragma unroll
            for (int w = 0; w < NW; ++w) { aA += xch[(w * 2 + 0) * kWave + lane]; aB += xch[(w * 2 + 1) * kWave + lane]; }
            __syncthreads();
        }
        hA = vA ? act_f(aA, n.act) : R(0);
        hB = vB ? act_f(aB, n.act) : R(0);
        if (rec) { rec[(2 * (l + 1)) * kWave + lane] = hA; rec[(2 * (l + 1) + 1) * kWave + lane] = hB; }
    }
    R p[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) p[q] = ew.Wo()[q * H + jA] * hA + ew.Wo()[q * H + jB] * hB;     // h is 0 on masked lanes
    const R nn = wave_reduce6_to_lanes(p, lane);
    if (rec && lane < 8) rec[2 * L * kWave + lane] = Y;
    const R bout = (c8 < 6) ? ew.bo()[(c8 < 6) ? c8 : 0] : R(0);
    return (c8 < 6) ? (mech + nn + bout) : R(0);
}

// f(t, x, u) by a TEAM of NW waves, hidden layers split by OUTPUT ROWS (solve_fwd_generic_kernel, NW > 1, L >= 2).
// The column split above makes every wave add up NW partial vectors per layer (2 NW LDS reads and adds per wave, two barriers): at 16
// waves a wave issued ~1 100 instructions per evaluation for 128 useful FMAs, and four such waves share a SIMD's issue slots
// (tools/pmc_generic.sh: the forward of 32 x 61 was issue-bound).  Here the activation vector lives in LDS (hb[2][128], zero beyond
// H) and lane (r, c) = (lane >> 3, lane & 7) of wave w owns row 8 (w + NW blk) + r and the 16-column chunk c of it: 16 FMAs, a
// 3-step sum over the eight lanes of the row, bias + activation, one 4-byte LDS write per row, ONE barrier per layer -- an
// all-gather of eight finished outputs per wave instead of an all-reduce of 128 partial sums.  The first layer (9 inputs) and the
// output layer (6 rows: lane (component, chunk), summed over the chunks into the replicated state layout) are computed by every
// wave from identical data, so every wave ends with identical bits.
constexpr int kHbStride = 160;       // 128 values + the skew below
// Position of activation k in an LDS vector read in chunks of CC by eight lanes at a time: every chunk start moves 4 banks on
// (CC = 16: chunk c starts at dword 20 c; CC = 8: the upper four chunks shift by 4), so the eight 16-byte reads of a ds_read_b128
// fall into eight different bank quads.  Unskewed, chunk starts 16 c hit two bank quads: 68 % of the LDS cycles of the forward
// were bank conflicts (tools/pmc_generic.sh).
template <int CC> __device__ __forceinline__ int hx(int k) { return k + 4 * (k >> (CC == 16 ? 4 : 5)); }
constexpr int kGenAccMats = 4;       // hidden matrices whose weights (forward) / gradients (adjoint) a team can keep in registers
// CC = columns per lane (16: H <= 128; 8: H <= 64, all eight lanes of a row busy).  NB > 0: the wave's NB row blocks of up to
// kGenAccMats hidden matrices are REGISTER-RESIDENT (wres, loaded once per trajectory): 8-wave teams, NB = 2, CC = 16 for 128 hidden
// units = 128 weight registers per lane and no weight load left in the evaluation; NB = 0 streams them from L2 as above.
// CC consecutive weights of one matrix row.  fp32 rows that start on a 16-byte boundary (H a multiple of 4 and the parameter set
// itself aligned: always for a single set) are read as dwordx4; left to itself hipcc vectorises the scalar loop with a peeled first
// element -- dwordx4 loads at offsets 4, 20, 36 that straddle every 16-byte boundary (8.1 against 5.6 ms for the 1 024 x 61 forward).
template <typename R, int CC> __device__ __forceinline__ void load_chunk(const R *__restrict__ wrow, R (&w)[CC], bool aligned16)
{
    if constexpr (sizeof(R) == 4) {
        if (aligned16) {
            typedef float __attribute__((ext_vector_type(4))) f4;
            const f4 *__restrict__ q = reinterpret_cast<const f4 *>(wrow);
#pragma unroll
            for (int v = 0; v < CC / 4; ++v) {
                const f4 x = q[v];
                w[4 * v] = x.x; w[4 * v + 1] = x.y; w[4 * v + 2] = x.z; w[4 * v + 3] = x.w;
            }
            return;
        }
    }
#pragma unroll
    for (int u = 0; u < CC; ++u) w[u] = wrow[u];
}

template <typename R, int CC, int NB> struct RowWeights { R w[kGenAccMats][NB > 0 ? NB : 1][CC]; };

template <typename R, int NW, int CC, int NB>
__device__ __forceinline__ void rows_preload(RowWeights<R, CC, NB> &rw, const StreamNet<R> &n, int lane, int part)
{
    if constexpr (NB > 0) {
        const int H = n.H, k0 = CC * (lane & 7), r8 = lane >> 3;
#pragma unroll
        for (int l = 0; l < kGenAccMats; ++l) {
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int row = 8 * (part + NW * blk) + r8;
                const bool ok = l + 1 < n.L && row < H;
                const R *__restrict__ wrow = n.Wh(ok ? l : 0) + (size_t)(ok ? row : 0) * H;
#pragma unroll
                for (int u = 0; u < CC; ++u) rw.w[l][blk][u] = (ok && k0 + u < H) ? wrow[(k0 + u < H) ? k0 + u : 0] : R(0);
            }
        }
    }
}

template <typename R, int NW, typename EW, int CC, int NB>
__device__ __forceinline__ R rhs_rows(const StreamNet<R> &n, const EW &ew, const RowWeights<R, CC, NB> &rw, const OdeP<R> &o, R t, R Y, R meal,
                                      R tvns, R gde, int lane, R *__restrict__ rec, int part, R *__restrict__ hb)
{
    const int H = n.H, L = n.L;
    const R G = lane_bcast(Y, 0), I = lane_bcast(Y, 1), Glu = lane_bcast(Y, 2), GLP1 = lane_bcast(Y, 3),
            GE = lane_bcast(Y, 4), FFA = lane_bcast(Y, 5);
    const int c8 = lane & 7, r8 = lane >> 3;
    const R mech = mech_eval(o, G, I, Glu, GLP1, FFA, meal, gde, c8);
    const bool vA = lane < H, vB = lane + 64 < H;
    const int jA = vA ? lane : 0, jB = vB ? lane + 64 : 0;
    const R in[9] = {t, G, I, Glu, GLP1, GE, FFA, GLP1, tvns};          // models/nn_residual.py:138-143
    R hA = ew.b1()[jA], hB = ew.b1()[jB];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        hA = rfma(ew.W1()[jA * 9 + i], in[i], hA);
        hB = rfma(ew.W1()[jB * 9 + i], in[i], hB);
    }
    hA = vA ? act_f(hA, n.act) : R(0);
    hB = vB ? act_f(hB, n.act) : R(0);
    // every wave writes the SAME h_1: a wave's own LDS writes are visible to its later reads, no barrier needed
    hb[hx<CC>(lane)] = hA;
    hb[hx<CC>(kWave + lane)] = hB;
    if (rec) { rec[lane] = hA; rec[kWave + lane] = hB; }
    const int k0 = CC * c8;                                  // this lane's column chunk [k0, k0 + CC)
    // one hidden layer; slot = its index as a compile-time constant when its weights are register-resident (NB > 0), else unused
    auto layer = [&](const int l, auto slot) {
        const R *__restrict__ hin = hb + (l & 1) * kHbStride;
        R *__restrict__ hout = hb + ((l + 1) & 1) * kHbStride;
        R hc[CC];
#pragma unroll
        for (int u = 0; u < CC; ++u) hc[u] = hin[hx<CC>(k0) + u];    // (zero beyond H; a chunk is contiguous under the skew)
        auto finish = [&](R acc, int row, bool vr) {
            acc = oct_allsum(acc);
            const R v = act_f(acc + ew.bh(l)[vr ? row : 0], n.act);
            if (c8 == 0 && vr) hout[hx<CC>(row)] = v;
        };
        if constexpr (NB > 0) {
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int row = 8 * (part + NW * blk) + r8;
                R acc = R(0);
#pragma unroll
                for (int u = 0; u < CC; ++u) acc = rfma(rw.w[decltype(slot)::value][blk][u], hc[u], acc);
                finish(acc, row, row < H);
            }
        } else {
            const R *__restrict__ W = n.Wh(l);
            // vector path: whole chunks only (H a multiple of CC: a chunk lies inside the row or, clamped to column 0, is discarded)
            const bool al16 = (H % CC) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;      // wave-uniform
            for (int row0 = 8 * part; row0 < H; row0 += 8 * NW) {
                const int row = row0 + r8;
                const bool vr = row < H;
                const R *__restrict__ wr0 = W + (size_t)(vr ? row : 0) * H;
                R w[CC];
                if (al16) {
                    load_chunk<R, CC>(wr0 + (k0 < H ? k0 : 0), w, true);
                } else {
#pragma unroll
                    for (int u = 0; u < CC; ++u) w[u] = wr0[(k0 + u < H) ? k0 + u : 0];      // never past the row
                }
                R acc = R(0);
#pragma unroll
                for (int u = 0; u < CC; ++u) acc = rfma((k0 + u < H) ? w[u] : R(0), hc[u], acc);
                finish(acc, row, vr);
            }
        }
        __syncthreads();
        if (rec) { rec[(2 * (l + 1)) * kWave + lane] = hout[hx<CC>(lane)]; rec[(2 * (l + 1) + 1) * kWave + lane] = hout[hx<CC>(kWave + lane)]; }
    };
    if constexpr (NB > 0) {
        // (a fixed-trip loop, fully unrolled: the register-resident weights need a compile-time layer index)
#pragma unroll
        for (int i = 0; i < kGenAccMats; ++i) {
            if (i + 1 >= L) break;
            if (i == 0) layer(0, std::integral_constant<int, 0>{});
            else if (i == 1) layer(1, std::integral_constant<int, 1>{});
            else if (i == 2) layer(2, std::integral_constant<int, 2>{});
            else layer(3, std::integral_constant<int, 3>{});
        }
    } else {
        for (int l = 0; l + 1 < L; ++l) layer(l, std::integral_constant<int, 0>{});
    }
    // output layer: lane (component c8 < 6, chunk r8) takes 16 columns of row c8; group_sum8 adds the eight chunks and leaves the
    // sum on every lane with that component -- the replicated layout of the state
    const R *__restrict__ hf = hb + ((L - 1) & 1) * kHbStride;
    const int q = (c8 < 6) ? c8 : 0, ko = 16 * r8;
    R acc = R(0);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int k = ko + u;
        acc = rfma((k < H) ? ew.Wo()[q * H + ((k < H) ? k : 0)] : R(0), hf[hx<CC>(k)], acc);
    }
    acc = (c8 < 6) ? acc : R(0);
    const R nn = group_sum8(acc);
    if (rec && lane < 8) rec[2 * L * kWave + lane] = Y;
    const R bout = (c8 < 6) ? ew.bo()[q] : R(0);
    __syncthreads();                                         // the next evaluation overwrites hb
    return (c8 < 6) ? (mech + nn + bout) : R(0);
}

template <typename R, int NW = 1, typename EW = EdgeFromParams<R>, int CC = 16, int NB = 0> struct RhsStream {
    StreamNet<R> n;
    EW ew;                  // where the edge parameters come from (the workgroup's LDS image in the solve kernels)
    const OdeP<R> &o;
    int lane;
    int part;               // this wave's index in its team (NW > 1: hode_generic.hip solve_fwd_generic_kernel)
    R *xch;                 // the team's exchange area (column split: partial sums; row split: hb[2][128])
    RowWeights<R, CC, NB> rw;   // row split with NB > 0: this wave's rows of the hidden matrices (rows_preload)
    static constexpr bool kUnrollStages = false;
    __device__ __forceinline__ int slot_elems() const { return 2 * n.L * kWave + 8; }
    __device__ __forceinline__ R operator()(R ts, R Ys, R meal, R tvns, R gde, R *__restrict__ rec) const
    {
        // every wave of a team computes the same f; only the first one writes the stage record
        if constexpr (NW > 1) {
            if (n.L >= 2) return rhs_rows<R, NW, EW, CC, NB>(n, ew, rw, o, ts, Ys, meal, tvns, gde, lane, part == 0 ? rec : nullptr, part, xch);
        }
        return rhs_stream<R, NW, EW>(n, ew, o, ts, Ys, meal, tvns, gde, lane, part == 0 ? rec : nullptr, part, xch);
    }
};

// ---- SEVERAL trajectories per team (solve_fwd_generic_multi_kernel; fp32, register-resident rows) ------------------------------------
// rhs_rows gives a whole team to ONE trajectory: every wave repeats the integrator, the first and the output layer, and a layer costs a
// barrier per trajectory; above a few hundred trajectories the teams stream the hidden matrices from L2 instead of keeping them (262 KB
// per evaluation for 5 x 128: 17 TB/s at 1 024 x 61, the bound of that launch).  Here a workgroup of eight waves serves TB trajectories:
// wave w INTEGRATES trajectory w % TB (controller, first layer, output layer, tape: once per trajectory, not once per wave) and owns its
// row blocks of every hidden matrix in registers FOR ALL of them -- per layer one barrier, TB x (4 LDS reads of 16 bytes + 32 FMAs + the
// 8-lane sums).  The trajectories of a team need not be in step: a round is "every wave has published the first-layer output of its
// next evaluation", whatever stage, step or grid interval that evaluation belongs to.  A wave whose trajectory is finished (or that has
// none) keeps serving rounds until all are (done counter in LDS, read behind the round's first barrier by the waves that are serving:
// they all see the same value and leave together).
template <int TB, int CC, int NB> struct RhsMulti {
    static constexpr int kNW = 8;
    StreamNet<float> n;
    EdgeImage<float> ew;
    const OdeP<float> &o;
    int lane, part;
    float *hbm;                  // [TB][3][kHbStride] activation vectors (zero beyond H): h_1 in buffer 2, h_{l+2} in buffer l & 1 -- the
                                 // first layer of a trajectory's NEXT evaluation never writes what a slower wave still reads
    RowWeights<float, CC, NB> rw;
    static constexpr bool kUnrollStages = false;
    __device__ __forceinline__ int slot_elems() const { return 2 * n.L * kWave + 8; }
    // the hidden layers of one round for all TB trajectories; barriers: one behind every layer.  rec != nullptr: this wave's own
    // trajectory is being taped -- h_{l+1} is copied out of LDS right behind the layer's barrier (the buffer is next written two
    // layers on, behind a barrier this wave has not passed yet)
    __device__ __forceinline__ void layers(float *__restrict__ rec) const
    {
        const float *__restrict__ hb_own = hbm + (part % TB) * 3 * kHbStride;
        const int H = n.H, L = n.L;
        const int c8 = lane & 7, r8 = lane >> 3, k0 = CC * c8;
        auto layer = [&](const int l, auto slot) {
            float bias[NB > 0 ? NB : 1];
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const int row = 8 * (part + kNW * blk) + r8;
                bias[blk] = ew.bh(l)[row < H ? row : 0];
            }
#pragma unroll 2
            for (int tt = 0; tt < TB; ++tt) {
                const float *__restrict__ hin = hbm + (tt * 3 + (l == 0 ? 2 : (l - 1) & 1)) * kHbStride;
                float *__restrict__ hout = hbm + (tt * 3 + (l & 1)) * kHbStride;
                float hc[CC];
#pragma unroll
                for (int u = 0; u < CC; ++u) hc[u] = hin[hx<CC>(k0) + u];
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) {
                    const int row = 8 * (part + kNW * blk) + r8;
                    float acc = 0.f;
#pragma unroll
                    for (int u = 0; u < CC; ++u) acc = rfma(rw.w[decltype(slot)::value][blk][u], hc[u], acc);
                    acc = oct_allsum(acc);
                    const float v = act_f(acc + bias[blk], n.act);
                    if (c8 == 0 && row < H) hout[hx<CC>(row)] = v;
                }
            }
            __syncthreads();
            if (rec) {
                const float *__restrict__ ho = hb_own + (l & 1) * kHbStride;
                rec[(2 * (l + 1)) * kWave + lane] = ho[hx<CC>(lane)];
                rec[(2 * (l + 1) + 1) * kWave + lane] = ho[hx<CC>(kWave + lane)];
            }
        };
#pragma unroll
        for (int i = 0; i < kGenAccMats; ++i) {
            if (i + 1 >= L) break;
            if (i == 0) layer(0, std::integral_constant<int, 0>{});
            else if (i == 1) layer(1, std::integral_constant<int, 1>{});
            else if (i == 2) layer(2, std::integral_constant<int, 2>{});
            else layer(3, std::integral_constant<int, 3>{});
        }
    }
    __device__ __forceinline__ float operator()(float t, float Y, float meal, float tvns, float gde, float *__restrict__ rec) const
    {
        const int H = n.H, L = n.L;
        rec = part < TB ? rec : nullptr;                    // (TB < 8: waves w and w + TB integrate the same trajectory; one writes)
        float *__restrict__ hb = hbm + (part % TB) * 3 * kHbStride;
        const float G = lane_bcast(Y, 0), I = lane_bcast(Y, 1), Glu = lane_bcast(Y, 2), GLP1 = lane_bcast(Y, 3), GE = lane_bcast(Y, 4),
                    FFA = lane_bcast(Y, 5);
        const int c8 = lane & 7, r8 = lane >> 3;
        const float mech = mech_eval(o, G, I, Glu, GLP1, FFA, meal, gde, c8);
        const bool vA = lane < H, vB = lane + 64 < H;
        const int jA = vA ? lane : 0, jB = vB ? lane + 64 : 0;
        const float in[9] = {t, G, I, Glu, GLP1, GE, FFA, GLP1, tvns};          // models/nn_residual.py:138-143
        float hA = ew.b1()[jA], hB = ew.b1()[jB];
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            hA = rfma(ew.W1()[jA * 9 + i], in[i], hA);
            hB = rfma(ew.W1()[jB * 9 + i], in[i], hB);
        }
        hA = vA ? act_f(hA, n.act) : 0.f;
        hB = vB ? act_f(hB, n.act) : 0.f;
        hb[2 * kHbStride + hx<CC>(lane)] = hA;
        hb[2 * kHbStride + hx<CC>(kWave + lane)] = hB;
        if (rec) { rec[lane] = hA; rec[kWave + lane] = hB; }
        __syncthreads();                                    // the round's first barrier: every trajectory's h_1 is in LDS
        layers(rec);
        const float *__restrict__ hf = hb + (L >= 2 ? (L - 2) & 1 : 2) * kHbStride;
        const int q = (c8 < 6) ? c8 : 0, ko = 16 * r8;
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = ko + u;
            acc = rfma((k < H) ? ew.Wo()[q * H + ((k < H) ? k : 0)] : 0.f, hf[hx<CC>(k)], acc);
        }
        acc = (c8 < 6) ? acc : 0.f;
        const float nn = group_sum8(acc);
        if (rec && lane < 8) rec[2 * L * kWave + lane] = Y;
        const float bout = (c8 < 6) ? ew.bo()[q] : 0.f;
        return (c8 < 6) ? (mech + nn + bout) : 0.f;
    }
};

// J^T kb for an arbitrary network from a stage record.  g: gradient vector of this parameter set (or nullptr).
// NW > 1: a TEAM of NW waves works on the same stage of the same trajectory (solve_bwd_generic_kernel).  Every wave runs the whole
// function on identical data -- identical results, identical control flow -- except for the two loops over the rows of a hidden
// matrix, which are split: wave `part` takes rows [j0, j1) of W^T delta (partial sums exchanged through xch[NW][2][64] in LDS and
// added in wave order) and of the atomics dW += delta (x) h_in.  A trajectory of such a network is one long chain of L2 round
// trips; with the batches these shapes are trained on (32 trajectories) one wave per trajectory left 97 % of the chip idle.
// Register accumulators of a team wave for its eight rows of up to kGenAccMats hidden matrices (fp32 adjoint of the solve, H <=
// 8 NW, L - 1 <= kGenAccMats): dW[j0 + u][lane], dW[j0 + u][lane + 64].  With them the hidden-matrix gradients leave the chip once
// per workgroup instead of once per stage (280 KB of atomics per stage of the 5 x 128 network: what bound the adjoint above a few
// hundred trajectories).  NoAcc: the atomics of round 2 (K5, fp64, deeper networks).
struct NoAcc {};
template <int RPW> struct TeamAccT {
    static constexpr int kRows = RPW;          // rows of every hidden matrix this wave owns: 8 (H <= 8 NW) or 16 (H <= 16 NW)
    // four separate arrays, not one [4][8][2]: hipcc folds a switch over identical case bodies into a dynamically indexed access,
    // and a dynamically indexed register array lives in scratch
    float m0[RPW][2], m1[RPW][2], m2[RPW][2], m3[RPW][2];
    // The gradients of everything that is NOT a hidden matrix -- first layer, output layer, every bias: 2 566 values for 5 x 128 --
    // were the other half of the adjoint's time as atomics (every team adds to the same 10 KB at every stage: 92 ms -> 46 ms for
    // 1 024 x 61 without them, 8.7 -> 6.7 ms for 32 x 61).  All waves of a team hold the same cotangents, so the pieces are dealt
    // out over the waves, twelve registers each (units j = lane and lane + 64 in [.][0] and [.][1]):
    //   wave 0: the six rows of the output matrix             wave 1: columns 0..4 of the first matrix, and the output bias
    //   wave 2: columns 5..8 of the first matrix, its bias     wave 3 + l: the bias of hidden matrix l
    float e[6][2];
    static constexpr int kEdgeOut = 0, kEdgeIn0 = 1, kEdgeIn1 = 2, kEdgeBias = 3;
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int u = 0; u < RPW; ++u) m0[u][0] = m0[u][1] = m1[u][0] = m1[u][1] = m2[u][0] = m2[u][1] = m3[u][0] = m3[u][1] = 0.f;
#pragma unroll
        for (int u = 0; u < 6; ++u) e[u][0] = e[u][1] = 0.f;
    }
    // ST: the workgroup owns g (its gradient ROW, solve_bwd_generic_kernel): plain stores, no atomics
    template <bool ST> static __device__ __forceinline__ void emit(float *p, float v)
    {
        if constexpr (ST) *p = v; else atomic_add(p, v);
    }
    // one atomic per entry and WORKGROUP for this wave's edge piece
    template <bool ST> __device__ __forceinline__ void flush_edge(const StreamNet<float> &n, float *__restrict__ g, int part, int lane)
    {
#define atomic_add emit<ST>
        const int H = n.H;
        const bool vA = lane < H, vB = lane + 64 < H;
        if (part == kEdgeOut) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                if (vA) atomic_add(g + n.out_off() + q * H + lane, e[q][0]);
                if (vB) atomic_add(g + n.out_off() + q * H + lane + 64, e[q][1]);
            }
        } else if (part == kEdgeIn0) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                if (vA) atomic_add(g + lane * 9 + i, e[i][0]);
                if (vB) atomic_add(g + (lane + 64) * 9 + i, e[i][1]);
            }
            if (lane < 6) atomic_add(g + n.out_off() + 6 * H + lane, e[5][0]);
        } else if (part == kEdgeIn1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (vA) atomic_add(g + lane * 9 + 5 + i, e[i][0]);
                if (vB) atomic_add(g + (lane + 64) * 9 + 5 + i, e[i][1]);
            }
            if (vA) atomic_add(g + 9 * H + lane, e[4][0]);
            if (vB) atomic_add(g + 9 * H + lane + 64, e[4][1]);
        } else if (part - kEdgeBias < n.L - 1) {
            float *__restrict__ gb = g + n.hid_off(part - kEdgeBias) + (size_t)H * H;
            if (vA) atomic_add(gb + lane, e[0][0]);
            if (vB) atomic_add(gb + lane + 64, e[0][1]);
        }
#undef atomic_add
    }
    // accumulators of the matrix at compile-time position SLOT (0 = the LAST hidden matrix, see rhs_vjp_stream), row u of this wave
    template <int SLOT> __device__ __forceinline__ void fma(int u, float dj, float inA, float inB)
    {
        static_assert(SLOT >= 0 && SLOT < 4, "kGenAccMats");
        float(&m)[RPW][2] = SLOT == 0 ? m0 : SLOT == 1 ? m1 : SLOT == 2 ? m2 : m3;
        m[u][0] = __builtin_fmaf(dj, inA, m[u][0]);
        m[u][1] = __builtin_fmaf(dj, inB, m[u][1]);
    }
    template <bool ST> __device__ __forceinline__ void flush_one(const float (&m)[RPW][2], float *__restrict__ gW, int H, int j0, int lane)
    {
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const int j = j0 + u;
            if (j < H) {
                if (lane < H) emit<ST>(gW + (size_t)j * H + lane, m[u][0]);
                if (lane + 64 < H) emit<ST>(gW + (size_t)j * H + lane + 64, m[u][1]);
            }
        }
    }
    // one atomic per entry and WORKGROUP (after all its trajectories of a parameter set)
    template <bool ST = false> __device__ __forceinline__ void flush(const StreamNet<float> &n, float *__restrict__ g, int j0, int lane, int part)
    {
        if (g == nullptr) return;
        flush_edge<ST>(n, g, part, lane);
        const int nm = n.L - 1;                            // slot i holds hidden matrix nm - 1 - i
        if (nm > 0) flush_one<ST>(m0, g + n.hid_off(nm - 1), n.H, j0, lane);
        if (nm > 1) flush_one<ST>(m1, g + n.hid_off(nm - 2), n.H, j0, lane);
        if (nm > 2) flush_one<ST>(m2, g + n.hid_off(nm - 3), n.H, j0, lane);
        if (nm > 3) flush_one<ST>(m3, g + n.hid_off(nm - 4), n.H, j0, lane);
    }
};

template <typename T> struct IsTeamAcc { static constexpr bool v = false; static constexpr int rows = 8; };
template <int RPW> struct IsTeamAcc<TeamAccT<RPW>> { static constexpr bool v = true; static constexpr int rows = RPW; };

template <typename R, bool GODE, bool GT, int NW = 1, typename ACC = NoAcc, typename EW = EdgeFromParams<R>>
__device__ __forceinline__ R rhs_vjp_stream(const StreamNet<R> &n, const EW &ew, R *__restrict__ g, const OdeP<R> &o, R t, R tvns, R gde, R gd_in,
                                            bool use_gd, int lane, const R *__restrict__ rec, R kb, R &go, R *gt_out, int part,
                                            R *__restrict__ xch, ACC &acc, int &xpar)
{
    const int H = n.H, L = n.L;
    const R *__restrict__ sx = rec + 2 * L * kWave;            // the stage state: wave-uniform loads
    const R G = sx[0], I = sx[1], Glu = sx[2], GLP1 = sx[3], GE = sx[4], FFA = sx[5];
    const R lq[6] = {lane_bcast(kb, 0), lane_bcast(kb, 1), lane_bcast(kb, 2), lane_bcast(kb, 3), lane_bcast(kb, 4), lane_bcast(kb, 5)};
    const int c8 = lane & 7;
    const R mech = mech_vjp<R, GODE>(o, G, I, Glu, GLP1, FFA, lq[0], lq[1], lq[2], lq[3], lq[5], gde, gd_in, use_gd, lane, go);
    const bool vA = lane < H, vB = lane + 64 < H;
    const int jA = vA ? lane : 0, jB = vB ? lane + 64 : 0;
    // this wave's rows of every hidden matrix (multiples of 8: the chunked loads below), and who adds the non-matrix gradients
    const int rows_per = (((H + NW - 1) / NW) + 7) & ~7;
    const int j0 = (part * rows_per < H) ? part * rows_per : H, j1 = (j0 + rows_per < H) ? j0 + rows_per : H;
    constexpr bool kAcc = IsTeamAcc<ACC>::v;
    using TeamAcc = TeamAccT<IsTeamAcc<ACC>::rows>;       // (the edge-piece constants; the type itself only when kAcc)
    R *__restrict__ dloc = kAcc ? xch + (NW * 4 + part * 2) * kWave : nullptr;       // this wave's delta slice (kernel: xch[NW * 6 * 64])
    // biases, first and last layer: atomics of the team's first wave -- or, with register accumulators, dealt out over the waves
    R *__restrict__ gedge = (part == 0 && !kAcc) ? g : nullptr;
    // output layer
    R hA = rec[(2 * (L - 1)) * kWave + lane], hB = rec[(2 * (L - 1) + 1) * kWave + lane];
    R dA = R(0), dB = R(0);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        dA = rfma(ew.Wo()[q * H + jA], lq[q], dA);
        dB = rfma(ew.Wo()[q * H + jB], lq[q], dB);
        if (gedge) {
            if (vA) atomic_add(gedge + n.out_off() + q * H + jA, lq[q] * hA);
            if (vB) atomic_add(gedge + n.out_off() + q * H + jB, lq[q] * hB);
        }
    }
    if (gedge && lane < 6) atomic_add(gedge + n.out_off() + 6 * H + lane, kb);
    if constexpr (kAcc) {
        if (g) {
            if (part == TeamAcc::kEdgeOut) {
#pragma unroll
                for (int q = 0; q < 6; ++q) { acc.e[q][0] = rfma(lq[q], hA, acc.e[q][0]); acc.e[q][1] = rfma(lq[q], hB, acc.e[q][1]); }
            } else if (part == TeamAcc::kEdgeIn0) {
                acc.e[5][0] += kb;                             // output bias: lanes 0..5 carry the six components
            }
        }
    }
    dA = vA ? act_bwd(dA, hA, n.act) : R(0);
    dB = vB ? act_bwd(dB, hB, n.act) : R(0);
    // hidden matrices, last to first: matrix l maps h_l (rows 2l, 2l+1 of the record) to h_{l+1}
    // one hidden matrix; `slot` = the matrix's position counted from the LAST one, as a compile-time constant (TeamAcc only)
    auto hidden_bwd = [&](const int l, auto slot) {
        const R inA = rec[(2 * l) * kWave + lane], inB = rec[(2 * l + 1) * kWave + lane];
        if constexpr (kAcc) {
            // delta_{l+1} of this wave into ITS slice of LDS: the sixteen delta_j of the wave's rows then come back as two or four
            // 16-byte broadcast reads instead of sixteen v_readlane from a run-time lane (select + readlane + copy each)
            dloc[lane] = dA;
            dloc[kWave + lane] = dB;
        }
        const R *__restrict__ W = n.Wh(l);
        R *__restrict__ gW = g ? g + n.hid_off(l) : nullptr;
        if (gedge) {
            if (vA) atomic_add(gW + (size_t)H * H + jA, dA);
            if (vB) atomic_add(gW + (size_t)H * H + jB, dB);
        }
        if constexpr (kAcc) {
            if (g && part == TeamAcc::kEdgeBias + l) { acc.e[0][0] += dA; acc.e[0][1] += dB; }    // d is 0 on masked lanes
        }
        // Two passes over the rows.  vmcnt retires in issue order, so a load issued behind an atomic waits for that atomic's
        // round trip to memory: with loads and atomics interleaved row by row every chunk of loads waited ~1 000 cycles for
        // the previous chunk's atomics (29 ms per 32 x 61 adjoint of the 5 x 128 network).  Pass 1 only loads (W^T delta),
        // pass 2 only adds (dW += delta (x) h_in).
        R pA = R(0), pB = R(0);
        int j = j0;
        // (unsigned column offsets on a running wave-uniform row pointer: two scalar adds and two loads per row.  With `int` columns
        //  and row * H recomputed per row hipcc spent six scalar and two 64-bit vector instructions on every row's addresses --
        //  a quarter of the instructions of a stage, and the adjoint is bound by what its waves issue)
        // fp32: BUFFER loads -- descriptor of the matrix in four SGPRs, the lane's column as a 32-bit VGPR byte offset, the row as
        // the instruction's scalar offset: one scalar add and two loads per row.  As flat global loads hipcc spent six scalar and two
        // 64-bit vector instructions on every row's addresses (a quarter of a stage's instructions, and the adjoint is bound by what
        // its waves issue); precomputed per-lane row offsets cost 32 VGPRs and spill.
        const unsigned oA = (unsigned)jA * (unsigned)sizeof(R), oB = (unsigned)jB * (unsigned)sizeof(R);   // byte offsets of the lane's columns
        unsigned roff = (unsigned)j0 * (unsigned)H * (unsigned)sizeof(R);                                   // byte offset of row j in the matrix
        const unsigned rstride = (unsigned)H * (unsigned)sizeof(R);
        auto ld = [&](unsigned o) -> R {
            if constexpr (sizeof(R) == 4) {
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<R *>(W), 0, (int)((unsigned)H * (unsigned)H * 4u), 0x00020000);
                return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)o, (int)roff, 0));
            } else {
                return *reinterpret_cast<const R *>(reinterpret_cast<const char *>(W) + roff + o);
            }
        };
        for (; j + 8 <= j1; j += 8) {                          // eight rows at a time: 16 independent loads in flight
            R wA[8], wB[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                wA[u] = ld(oA);                                 // column jA / jB of row j: 256 contiguous bytes per wave
                wB[u] = ld(oB);
                roff += rstride;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                R dj;
                if constexpr (kAcc) dj = dloc[j + u];          // (own write above: LDS is in order per wave)
                else dj = unit_bcast(dA, dB, j + u);           // a dead unit (ReLU, dj == 0) adds nothing
                pA = rfma(wA[u], dj, pA);
                pB = rfma(wB[u], dj, pB);
            }
        }
        for (; j < j1; ++j) {
            const R dj = unit_bcast(dA, dB, j);
            pA = rfma(ld(oA), dj, pA);
            pB = rfma(ld(oB), dj, pB);
            roff += rstride;
        }
        if constexpr (NW > 1) {
            // partial sums of the team, added in wave order on every wave (same bits everywhere).  Two exchange areas used in turn
            // (xpar: a toggle that runs on across layers AND stages): the writes of the next exchange go to the other area, so ONE
            // barrier per layer is enough -- a wave can be at most one barrier ahead of the slowest reader of the area it writes
            // next but one
            R *__restrict__ xl = xch + xpar * (NW * 2 * kWave);
            xpar ^= 1;
            xl[(part * 2 + 0) * kWave + lane] = pA;
            xl[(part * 2 + 1) * kWave + lane] = pB;
            __syncthreads();
            pA = R(0); pB = R(0);
#pragma unroll
            for (int w = 0; w < NW; ++w) { pA += xl[(w * 2 + 0) * kWave + lane]; pB += xl[(w * 2 + 1) * kWave + lane]; }
        }
        if constexpr (kAcc) {
            if (g) {
                // this wave's rows of dW += delta (x) h_in (masked lanes: h_in = 0), straight into the registers of `slot`
#pragma unroll
                for (int u = 0; u < IsTeamAcc<ACC>::rows; ++u)      // (rows beyond H: delta is zero there -- dloc holds all 128 entries)
                    acc.template fma<decltype(slot)::value>(u, dloc[j0 + u], inA, inB);
            }
        } else if (g) {
            for (j = j0; j < j1; ++j) {
                const R dj = unit_bcast(dA, dB, j);
                if (dj == R(0)) continue;                      // wave-uniform: no atomics for dead units
                if (vA) atomic_add(gW + (size_t)j * H + jA, dj * inA);
                if (vB) atomic_add(gW + (size_t)j * H + jB, dj * inB);
            }
        }
        dA = vA ? act_bwd(pA, inA, n.act) : R(0);
        dB = vB ? act_bwd(pB, inB, n.act) : R(0);
    };
    if constexpr (kAcc) {
        // The accumulators must be addressed by a COMPILE-TIME index.  With the matrix index a run-time value hipcc merges the four
        // "which matrix" branches into one body that addresses the accumulators through a selected offset -- and an array with a
        // dynamic offset lives in scratch: sixteen scratch_load / s_waitcnt vmcnt(0) / v_fmac / scratch_store round trips per
        // matrix and stage (472 B of scratch; 79 % of the wave's time in waits, tools/pmc_generic.sh).  A fixed-trip loop over
        // the POSITION of the matrix from the last one unrolls fully and makes that position a constant.
#pragma unroll
        for (int i = 0; i < kGenAccMats; ++i) {
            if (i > L - 2) break;
            if (i == 0) hidden_bwd(L - 2, std::integral_constant<int, 0>{});
            else if (i == 1) hidden_bwd(L - 3, std::integral_constant<int, 1>{});
            else if (i == 2) hidden_bwd(L - 4, std::integral_constant<int, 2>{});
            else hidden_bwd(L - 5, std::integral_constant<int, 3>{});
        }
    } else {
        for (int l = L - 2; l >= 0; --l) hidden_bwd(l, std::integral_constant<int, 0>{});
    }
    // first layer: input row [t, G, I, Glu, GLP1, GE, FFA, glp1 := GLP1, tvns]
    const R in[9] = {t, G, I, Glu, GLP1, GE, FFA, GLP1, tvns};
    if (gedge) {
        if (vA) atomic_add(gedge + 9 * H + jA, dA);
        if (vB) atomic_add(gedge + 9 * H + jB, dB);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (vA) atomic_add(gedge + jA * 9 + i, dA * in[i]);
            if (vB) atomic_add(gedge + jB * 9 + i, dB * in[i]);
        }
    }
    if constexpr (kAcc) {
        if (g) {
            if (part == TeamAcc::kEdgeIn0) {
#pragma unroll
                for (int i = 0; i < 5; ++i) { acc.e[i][0] = rfma(dA, in[i], acc.e[i][0]); acc.e[i][1] = rfma(dB, in[i], acc.e[i][1]); }
            } else if (part == TeamAcc::kEdgeIn1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc.e[i][0] = rfma(dA, in[5 + i], acc.e[i][0]); acc.e[i][1] = rfma(dB, in[5 + i], acc.e[i][1]); }
                acc.e[4][0] += dA;
                acc.e[4][1] += dB;
            }
        }
    }
    R w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = ew.W1()[jA * 9 + i] * dA + ew.W1()[jB * 9 + i] * dB;       // d is 0 on masked lanes
    const R p[6] = {w[1], w[2], w[3], w[4] + w[7], w[5], w[6]};                                   // GLP1 feeds inputs 4 and 7
    const R nn = wave_reduce6_to_lanes(p, lane);
    if constexpr (GT) *gt_out = wave_allsum(w[0]);
    return (c8 < 6) ? (mech + nn) : R(0);
}

// ------------------------------------------------------------------------------------------ K1 / K5
template <typename R>
__global__ __launch_bounds__(256) void rhs_fwd_generic_kernel(const RhsArgs<R> a, int L)
{
    const int lane = threadIdx.x & 63;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    const StreamNet<R> n{a.nn_p, a.H, L, a.act};
    OdeP<R> o;
    ode_load(o, a.ode_p);
    for (int s = blockIdx.x * 4 + wave; s < a.B; s += gridDim.x * 4) {
        const R Y = (lane < 6) ? a.x[(size_t)s * 6 + lane] : R(0);
        const R gde = a.gd ? gd_effect(o, a.gd[s]) : R(0);
        const R F = rhs_stream<R>(n, EdgeFromParams<R>{n}, o, a.t ? a.t[s] : R(0), Y, a.meal ? a.meal[s] : R(0), a.tvns ? a.tvns[s] : R(0), gde, lane, nullptr);
        if (lane < 6) a.out[(size_t)s * 6 + lane] = F;
    }
}

template <typename R, bool GODE>
__global__ __launch_bounds__(256) void rhs_bwd_generic_kernel(const RhsArgs<R> a, int L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    R *rec = reinterpret_cast<R *>(smem_raw) + (size_t)wave * (2 * L * kWave + 8);      // this wave's record
    const StreamNet<R> n{a.nn_p, a.H, L, a.act};
    OdeP<R> o;
    ode_load(o, a.ode_p);
    R go = R(0);
    for (int s = blockIdx.x * 4 + wave; s < a.B; s += gridDim.x * 4) {
        const R Y = (lane < 6) ? a.x[(size_t)s * 6 + lane] : R(0);
        const R kb = (lane < 6) ? a.gout[(size_t)s * 6 + lane] : R(0);
        const R t = a.t ? a.t[s] : R(0), tvns = a.tvns ? a.tvns[s] : R(0), gdv = a.gd ? a.gd[s] : R(0);
        const R gde = a.gd ? gd_effect(o, gdv) : R(0);
        (void)rhs_stream<R>(n, EdgeFromParams<R>{n}, o, t, Y, a.meal ? a.meal[s] : R(0), tvns, gde, lane, rec);
        __builtin_amdgcn_wave_barrier();
        R gt;
        NoAcc na;
        int xpar = 0;                                            // (one wave per sample: no exchange)
        const R Z = rhs_vjp_stream<R, GODE, true>(n, EdgeFromParams<R>{n}, a.gnn, o, t, tvns, gde, gdv, a.gd != nullptr, lane, rec, kb, go, &gt, 0, (R *)nullptr, na, xpar);
        __builtin_amdgcn_wave_barrier();
        if (lane < 6) a.gx[(size_t)s * 6 + lane] = Z;
        if (a.gt && lane == 0) a.gt[s] = gt;
    }
    if constexpr (GODE) {
        if (a.gode && lane < 17) atomic_add(a.gode + lane, go);
    }
}

template <typename R> int launch_rhs_fwd_generic(hipStream_t s, const RhsArgs<R> &a, int L)
{
    int blocks = (a.B + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) return HODE_OK;
    hipLaunchKernelGGL((rhs_fwd_generic_kernel<R>), dim3(blocks), dim3(256), 0, s, a, L);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}
template <typename R> int launch_rhs_bwd_generic(hipStream_t s, const RhsArgs<R> &a, int L)
{
    int blocks = (a.B + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) return HODE_OK;
    const size_t lds = (size_t)4 * (2 * L * kWave + 8) * sizeof(R);
    if (a.gode) hipLaunchKernelGGL((rhs_bwd_generic_kernel<R, true>), dim3(blocks), dim3(256), lds, s, a, L);
    else hipLaunchKernelGGL((rhs_bwd_generic_kernel<R, false>), dim3(blocks), dim3(256), lds, s, a, L);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

// ------------------------------------------------------------------------------------------ K2 + K3
// A team of NW waves per trajectory.  Every wave runs the integration (solve_one) on identical data -- identical step-size
// decisions, identical control flow -- and the team splits the rows of the hidden matrices inside the right-hand side (rhs_rows).
// The waves of a team write the same y / tape values to the same addresses (benign: identical bits); the stage records are written
// by the first wave only.
// (second launch bound = waves per SIMD the register allocation must leave room for: 4 for the streaming kernels -- at 130 VGPRs, three
//  waves per SIMD, the 1 024 x 61 forward of 5 x 128 took 7.3 ms instead of 5.5 -- and 2 with register-resident weights; fp64 unbounded)
template <typename R, int METHOD, bool TAPE, bool GD, int NW, int CC, int NB>
__global__ __launch_bounds__(64 * NW, (sizeof(R) == 8) ? 1 : (NB > 0 ? 2 : 4)) void solve_fwd_generic_kernel(const SolveArgs<R> a)
{
    __shared__ R rows[8 * kWave];
    __shared__ R cvec[8];
    __shared__ R ybuf[NW * (kWave + 8)];
    __shared__ R xch[(NW * 2 * kWave > 2 * kHbStride) ? NW * 2 * kWave : 2 * kHbStride];   // column split: partial sums; row split: hb[2][128]
    const int lane = threadIdx.x & 63;
    const int part = first_lane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x;
    const int set = b / (a.B / a.n_sets);
    for (int i = threadIdx.x; i < 2 * kHbStride; i += 64 * NW) xch[i] = R(0);      // rhs_rows: zero beyond H, for good
    tableau_rows_store<R>(rows, METHOD, threadIdx.x, 64 * NW);
    if (threadIdx.x < 8) cvec[threadIdx.x] = (R)kTableau[METHOD].c[threadIdx.x];
    OdeP<R> o;
    ode_load(o, a.ode_p + 17 * set);
    __shared__ R edge_img[kEdgeImageMax];                // the edge parameters of this trajectory's set (EdgeImage)
    const StreamNet<R> net{a.nn_p + (size_t)set * a.nn_stride, a.H, a.L, a.act};
    EdgeImage<R>::fill(edge_img, net, threadIdx.x, 64 * NW);
    __syncthreads();
    RhsStream<R, NW, EdgeImage<R>, CC, NB> rhs{net, EdgeImage<R>{edge_img, a.H, a.L}, o, lane, part, xch, {}};
    rows_preload<R, NW, CC, NB>(rhs.rw, net, lane, part);
    solve_one<R, METHOD, TAPE, GD>(a, b, rhs, o, rows, cvec, ybuf + part * (kWave + 8), lane);
}

// TB trajectories per eight-wave workgroup (RhsMulti).  Launched for fp32, at most kGenAccMats hidden matrices, whole parameter sets
// per workgroup (per_set % TB == 0).
template <int METHOD, bool TAPE, bool GD, int TB, int CC, int NB>
__global__ __launch_bounds__(512, 2) void solve_fwd_generic_multi_kernel(const SolveArgs<float> a)
{
    constexpr int NW = 8;
    __shared__ float rows[8 * kWave];
    __shared__ float cvec[8];
    __shared__ float ybuf[NW * (kWave + 8)];
    __shared__ float hbm[TB * 3 * kHbStride];
    __shared__ float edge_img[kEdgeImageMax];
    __shared__ int done_cnt;
    const int lane = threadIdx.x & 63;
    const int part = first_lane((int)(threadIdx.x >> 6));
    const int b = blockIdx.x * TB + part % TB;
    const int set = (blockIdx.x * TB) / (a.B / a.n_sets);
    for (int i = threadIdx.x; i < TB * 3 * kHbStride; i += 64 * NW) hbm[i] = 0.f;
    tableau_rows_store<float>(rows, METHOD, threadIdx.x, 64 * NW);
    if (threadIdx.x < 8) cvec[threadIdx.x] = (float)kTableau[METHOD].c[threadIdx.x];
    if (threadIdx.x == 0) done_cnt = 0;
    OdeP<float> o;
    ode_load(o, a.ode_p + 17 * set);
    const StreamNet<float> net{a.nn_p + (size_t)set * a.nn_stride, a.H, a.L, a.act};
    EdgeImage<float>::fill(edge_img, net, threadIdx.x, 64 * NW);
    __syncthreads();
    RhsMulti<TB, CC, NB> rhs{net, EdgeImage<float>{edge_img, a.H, a.L}, o, lane, part, hbm, {}};
    rows_preload<float, NW, CC, NB>(rhs.rw, net, lane, part);
    if (b < a.B) solve_one<float, METHOD, TAPE, GD>(a, b, rhs, o, rows, cvec, ybuf + part * (kWave + 8), lane);
    // serve the other trajectories' rounds until every wave of the team is here
    if (lane == 0) atomicAdd(&done_cnt, 1);
    for (;;) {
        __syncthreads();                                    // = the first barrier of a round
        if (*(volatile int *)&done_cnt == NW) break;        // (nobody is integrating any more: every wave reads NW in the same round)
        rhs.layers(nullptr);
    }
}

template <int METHOD, int TB, int CC, int NB> static int launch_fwd_generic_multi(hipStream_t s, const SolveArgs<float> &a)
{
    const bool tape = a.tape != nullptr, gd = a.gd_mode != 0;
    const dim3 grid((a.B + TB - 1) / TB), block(512);
    if (tape) {
        if (gd) hipLaunchKernelGGL((solve_fwd_generic_multi_kernel<METHOD, true, true, TB, CC, NB>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((solve_fwd_generic_multi_kernel<METHOD, true, false, TB, CC, NB>), grid, block, 0, s, a);
    } else {
        if (gd) hipLaunchKernelGGL((solve_fwd_generic_multi_kernel<METHOD, false, true, TB, CC, NB>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((solve_fwd_generic_multi_kernel<METHOD, false, false, TB, CC, NB>), grid, block, 0, s, a);
    }
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <typename R, int METHOD, int NW, int CC = 16, int NB = 0> static int launch_fwd_generic_t(hipStream_t s, const SolveArgs<R> &a)
{
    const bool tape = a.tape != nullptr, gd = a.gd_mode != 0;
    const dim3 grid(a.B), block(64 * NW);
    if (tape) {
        if (gd) hipLaunchKernelGGL((solve_fwd_generic_kernel<R, METHOD, true, true, NW, CC, NB>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((solve_fwd_generic_kernel<R, METHOD, true, false, NW, CC, NB>), grid, block, 0, s, a);
    } else {
        if (gd) hipLaunchKernelGGL((solve_fwd_generic_kernel<R, METHOD, false, true, NW, CC, NB>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((solve_fwd_generic_kernel<R, METHOD, false, false, NW, CC, NB>), grid, block, 0, s, a);
    }
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}
template <typename R, int METHOD> static int launch_fwd_generic_m(hipStream_t s, const SolveArgs<R> &a)
{
    // Teams of waves per trajectory, hidden layers split by rows (rhs_rows).  fp32 with at most kGenAccMats hidden matrices and a
    // batch of one or two trajectories per CU: eight waves with their rows of every matrix in registers (one workgroup per CU).
    // Larger batches, fp64, deeper networks: four waves (eight up to 256 trajectories) streaming the weights from L2.
    // Measured, 5 x 128 / 5 x 64, T = 61, forward (ms): B = 32: 1.56 / 0.98 (round-3 column split: 4.7 / 1.9); 256: 1.59 / 1.00;
    // 512: 3.10 resident against 3.57 streamed / 1.97 against 1.71; 1 024: 6.2 against 5.5 / 3.9 against 2.2 (one wave per
    // trajectory, the policy until now: 10.6 / 4.9); 4 096: 24.6 against 22.0 / 15.4 against 7.8.
    const bool narrow = a.H <= 64;
    if constexpr (sizeof(R) == 4) {
        // up to one trajectory per CU's team: one trajectory per team.  Above: several per team, rows still register-resident -- the
        // fewest per team (2, 4, 8) that put the batch on the chip in ONE round of 256 teams (768 trajectories as 384 teams of two: 4.0 +
        // 8.0 ms; as 192 teams of four: 2.9 + 5.1); whole parameter sets per workgroup
        if (a.L >= 2 && a.L - 1 <= kGenAccMats && a.B > 256) {
            const int per_set = a.B / a.n_sets;
            if (a.B <= 512 && per_set % 2 == 0)
                return narrow ? launch_fwd_generic_multi<METHOD, 2, 8, 1>(s, a) : launch_fwd_generic_multi<METHOD, 2, 16, 2>(s, a);
            if (a.B <= 1024 && per_set % 4 == 0)
                return narrow ? launch_fwd_generic_multi<METHOD, 4, 8, 1>(s, a) : launch_fwd_generic_multi<METHOD, 4, 16, 2>(s, a);
            if (per_set % 8 == 0)
                return narrow ? launch_fwd_generic_multi<METHOD, 8, 8, 1>(s, a) : launch_fwd_generic_multi<METHOD, 8, 16, 2>(s, a);
            if (per_set % 4 == 0)
                return narrow ? launch_fwd_generic_multi<METHOD, 4, 8, 1>(s, a) : launch_fwd_generic_multi<METHOD, 4, 16, 2>(s, a);
            if (per_set % 2 == 0)
                return narrow ? launch_fwd_generic_multi<METHOD, 2, 8, 1>(s, a) : launch_fwd_generic_multi<METHOD, 2, 16, 2>(s, a);
        }
        if (a.L >= 2 && a.L - 1 <= kGenAccMats && a.B <= (narrow ? 256 : 512))
            return narrow ? launch_fwd_generic_t<R, METHOD, 8, 8, 1>(s, a) : launch_fwd_generic_t<R, METHOD, 8, 16, 2>(s, a);
    }
    if (a.B <= 256) return narrow ? launch_fwd_generic_t<R, METHOD, 8, 8, 0>(s, a) : launch_fwd_generic_t<R, METHOD, 8, 16, 0>(s, a);
    return narrow ? launch_fwd_generic_t<R, METHOD, 4, 8, 0>(s, a) : launch_fwd_generic_t<R, METHOD, 4, 16, 0>(s, a);
}
template <typename R> int launch_solve_fwd_generic(hipStream_t s, const SolveArgs<R> &a, int method)
{
    return method == HODE_METHOD_DP54 ? launch_fwd_generic_m<R, HODE_METHOD_DP54>(s, a) : launch_fwd_generic_m<R, HODE_METHOD_RK4>(s, a);
}

// ------------------------------------------------------------------------------------------ K4
// Same walk over the tape as solve_bwd_kernel (hode_solve_bwd.hip); records are read from HBM with plain loads (L2 hits: the
// forward has just written them).  Teams of 8 waves (16 for the atomics variant above 64 hidden units); ACCREG: the team's
// register accumulators (TeamAccT), else gradients leave through atomics inside rhs_vjp_stream (fp64, more than four hidden
// matrices, no parameter gradient wanted).
template <typename R, bool GODE, bool GD, int kGenTeam, int ACCREG>      // ACCREG: accumulator rows per wave (8 / 16), 0 = atomics
__global__ __launch_bounds__(64 * kGenTeam) void solve_bwd_generic_kernel(const AdjArgs<R> a, const int method, const int L, const int rows_grid)
{
    using Acc = std::conditional_t<ACCREG != 0, TeamAccT<ACCREG ? ACCREG : 8>, NoAcc>;
    Acc acc;
    int acc_set = -1;                                       // the parameter set the accumulators belong to
    __shared__ R rowsT[8 * kWave];
    __shared__ R xch[kGenTeam * 6 * kWave];                  // partial sums, two areas [2][NW][2][64] | every wave's delta slice [NW][2][64]
    __shared__ R edge_img[kEdgeImageMax];                   // the edge parameters of the current set (EdgeImage)
    int img_set = -1;
    int xpar = 0;                                           // which of the two exchange areas the next layer writes (rhs_vjp_stream)
    const int lane = threadIdx.x & 63;
    const int part = first_lane((int)(threadIdx.x >> 6));      // every wave of the team walks the tape; the matrix rows are split
    const int c8 = lane & 7, grp = lane >> 3;
    const int T = a.T;
    const TableauData &tab = kTableau[method];
    const int S = tab.S;
    const int kSlot = 2 * L * kWave + 8;
    tableau_rowsT_store<R>(rowsT, method, threadIdx.x, 64 * kGenTeam);
    __syncthreads();
    const int per_set = a.B / a.n_sets;
    constexpr bool use_gd = GD;
    const int rows_per_k = (((a.H + kGenTeam - 1) / kGenTeam) + 7) & ~7;
    const int j0_k = (part * rows_per_k < a.H) ? part * rows_per_k : a.H;
    // ROWS mode (a 2-D grid: blockIdx.y = parameter set, register accumulators, a partials area): the workgroup serves trajectories
    // of ONE set and leaves its gradient as ONE row of a.partials with plain stores; adj_reduce_kernel adds the rows of a set in
    // workgroup order -- no floating-point atomics, the same bits run to run (the tuned path's scheme, round 4 for these shapes).
    const bool rows_mode = ACCREG != 0 && rows_grid != 0;
    R go_sum = R(0);
    R *__restrict__ prow = nullptr;
    if (rows_mode) {
        const size_t rowlen = adj_partial_rowlen(a.P);
        prow = a.partials + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * rowlen;
        for (size_t i = threadIdx.x; i < rowlen; i += 64 * kGenTeam) prow[i] = R(0);
        __syncthreads();
    }
    const int b_first = rows_mode ? (int)blockIdx.y * per_set + (int)blockIdx.x : (int)blockIdx.x;
    const int b_end = rows_mode ? ((int)blockIdx.y + 1) * per_set : a.B;
    for (int b = b_first; b < b_end; b += gridDim.x) {
        const int set = b / per_set;
        const StreamNet<R> n{a.nn_p + (size_t)set * a.P, a.H, L, a.act};
        R *__restrict__ g = a.gnn ? a.gnn + (size_t)set * a.P : nullptr;
        if (set != img_set) {                               // (every wave of the workgroup walks the same b: uniform branch)
            __syncthreads();                                // nobody reads the previous set's image any more
            EdgeImage<R>::fill(edge_img, n, threadIdx.x, 64 * kGenTeam);
            __syncthreads();
            img_set = set;
        }
        const EdgeImage<R> ew{edge_img, a.H, L};
        if constexpr (ACCREG != 0) {
            if (set != acc_set) {                           // a workgroup's trajectories come set by set: flush when the set changes
                if (acc_set >= 0) acc.flush(StreamNet<R>{a.nn_p + (size_t)acc_set * a.P, a.H, L, a.act}, a.gnn ? a.gnn + (size_t)acc_set * a.P : nullptr, j0_k, lane, part);
                acc.zero();
                acc_set = set;
            }
        }
        OdeP<R> o;
        ode_load(o, a.ode_p + 17 * set);
        const R *__restrict__ tg = a.t + (a.t_batched ? (size_t)b * T : 0);
        const R *__restrict__ tape = a.tape + (size_t)b * a.max_steps * 8;
        const int *__restrict__ tseg = a.tape_seg + (size_t)b * a.max_steps;
        const R *__restrict__ stg = a.tape_stage + (size_t)b * a.max_steps * 6 * kSlot;
        const R *__restrict__ gyb = a.gy + (size_t)b * T * 6;
        const int nst = a.nsteps[b] < a.max_steps ? a.nsteps[b] : a.max_steps;
        const bool ok = a.status[b] == HODE_ST_OK;
        auto gy_row = [&](int r) -> R { return (c8 < 6) ? gyb[(size_t)r * 6 + c8] : R(0); };
        R lam = R(0), go = R(0);
        int knext = T - 1;
        for (int st = nst - 1; st >= 0; --st) {
            const int kraw = tseg[st];
            const int k = kraw & (kSegClosed - 1);
            int hi = knext;                                   // rows this step produced: see solve_bwd_kernel
            if (st == nst - 1) {
                hi = T - 1;
                if (!ok) {
                    hi = k;
                    if (kraw & kSegClosed) {
                        hi = k + 1;
                        while (hi + 1 < T && !(tg[hi + 1] > tg[hi])) ++hi;
                    }
                }
            }
            for (int r = k + 1; r <= hi; ++r) lam += gy_row(r);
            knext = k;
            const R tc = tape[(size_t)st * 8 + 0], h = tape[(size_t)st * 8 + 1];
            const R t0 = tg[k], t1 = tg[k + 1];
            const R v0 = inp_at(a.tvns, a.tvns_mode, b, T, k), v1 = inp_at(a.tvns, a.tvns_mode, b, T, k + 1);
            const R d0 = inp_at(a.gd, a.gd_mode, b, T, k), d1 = inp_at(a.gd, a.gd_mode, b, T, k + 1);
            const R inv_len = first_lane(R(1) / (t1 - t0));
            const R dv = first_lane(v1 - v0), dd = first_lane(d1 - d0);
            R ZZ = R(0);
            for (int s = S - 1; s >= 0; --s) {
                const R bw_s = rowsT[6 * kWave + s], c_s = rowsT[6 * kWave + 8 + s];
                const R kb = h * rfma(bw_s, lam, group_sum8(rowsT[s * kWave + lane] * ZZ));
                const R ts = rfma(c_s, h, tc);
                const R al = (ts - t0) * inv_len;
                const R gdv = rfma(al, dd, d0);
                R gde = R(0);
                if constexpr (use_gd) gde = gd_effect(o, gdv);
                const R Z = rhs_vjp_stream<R, GODE, false, kGenTeam, Acc, EdgeImage<R>>(n, ew, g, o, ts, rfma(al, dv, v0), gde, gdv, use_gd, lane,
                                                                          stg + ((size_t)st * 6 + s) * kSlot, kb, go, nullptr, part, xch, acc, xpar);
                ZZ = (grp == s) ? Z : ZZ;
            }
            lam += group_sum8(rowsT[7 * kWave + lane] * ZZ);
        }
        int kf = 0;                                           // rows 0..kf are (copies of) x0
        while (kf + 1 < T && !(tg[kf + 1] > tg[kf])) ++kf;
        for (int r = 0; r <= kf; ++r) lam += gy_row(r);
        if (part == 0) {
            if (lane < 6) a.gx0[(size_t)b * 6 + lane] = lam;
            if constexpr (GODE) {
                if (rows_mode) go_sum += go;
                else if (a.gode && lane < 17) atomic_add(a.gode + 17 * set + lane, go);
            }
        }
    }
    if constexpr (ACCREG != 0) {
        if (rows_mode) {
            if (acc_set >= 0) acc.template flush<true>(StreamNet<R>{a.nn_p + (size_t)acc_set * a.P, a.H, L, a.act}, prow, j0_k, lane, part);
            if constexpr (GODE) {
                if (part == 0 && lane < 17) prow[a.P + lane] = go_sum;
            }
        } else if (acc_set >= 0) {
            acc.flush(StreamNet<R>{a.nn_p + (size_t)acc_set * a.P, a.H, L, a.act}, a.gnn ? a.gnn + (size_t)acc_set * a.P : nullptr, j0_k, lane, part);
        }
    }
}

// ------------------------------------------------------------------------------------------ K4, several trajectories per team
// The adjoint of RhsMulti's idea (fp32, at most kGenAccMats hidden matrices, gradient rows).  solve_bwd_generic_kernel gives the whole
// team to ONE trajectory: eight waves walk the same tape, every stage streams every hidden matrix from L2 once PER TRAJECTORY and pays a
// barrier per layer and trajectory.  Here wave w walks the tape of trajectory w % TB of the team's current group -- step headers,
// cotangent injection, the mechanistic and edge-layer VJPs once per trajectory, not once per wave -- and for every hidden matrix the
// owners publish delta_{l+1} and h_l of their trajectories in LDS, every wave multiplies ITS rows (streamed from L2 ONCE per round) with
// all TB cotangents and adds its rows of dW += delta (x) h for all of them, the partial sums go back through LDS to the owners: two
// barriers per matrix and ROUND.  A round = one stage of every trajectory that still has one; a wave whose tape is exhausted zeroes its
// published vectors (its products and outer products are then exact zeros) and serves until the group is done.
constexpr int kMultiEdgeVals = 17;      // eo 0..5 (output matrix rows) | ei 6..14 (first matrix columns) | 15 its bias | 16 output bias
template <int RPW> struct MultiAcc {
    float m0[RPW][2], m1[RPW][2], m2[RPW][2], m3[RPW][2];      // this wave's rows of dW of up to four hidden matrices (slot 0 = the LAST)
    // The edge gradients of the trajectories THIS wave walks (units lane, lane + 64).  Sixteen rows per wave (H > 64) are 128
    // accumulator registers: 34 more do not fit next to them (540 spill instructions, measured) -- those teams keep the edge sums in a
    // wave-private LDS table [value][2][64] and pay 17 independent read-add-write pairs per stage; eight-row teams keep registers.
    static constexpr bool kEdgeLds = RPW > 8;
    float e[kEdgeLds ? 1 : kMultiEdgeVals][2];
    float *eld;
    float eb[2];                 // wave 3 + i: the bias of the hidden matrix in slot i, for ALL trajectories (from the published cotangents)
    __device__ __forceinline__ void zero(float *eld_own, int lane)
    {
#pragma unroll
        for (int u = 0; u < RPW; ++u) m0[u][0] = m0[u][1] = m1[u][0] = m1[u][1] = m2[u][0] = m2[u][1] = m3[u][0] = m3[u][1] = 0.f;
        eld = eld_own;
        if constexpr (kEdgeLds) {
            for (int v = 0; v < kMultiEdgeVals * 2; ++v) eld[v * kWave + lane] = 0.f;
        } else {
#pragma unroll
            for (int v = 0; v < kMultiEdgeVals; ++v) e[v][0] = e[v][1] = 0.f;
        }
        eb[0] = eb[1] = 0.f;
    }
    template <int V> __device__ __forceinline__ void edge_add(float x0, float x1, int lane)
    {
        if constexpr (kEdgeLds) {
            float *__restrict__ q = eld + (V * 2) * kWave + lane;
            q[0] += x0;
            q[kWave] += x1;
        } else {
            e[V][0] += x0;
            e[V][1] += x1;
        }
    }
    // (a sum of products as one FMA per half in the register form -- the same bits either way is not promised between the two forms)
    template <int V> __device__ __forceinline__ void edge_fma(float a0, float b0, float a1, float b1, int lane)
    {
        if constexpr (kEdgeLds) {
            float *__restrict__ q = eld + (V * 2) * kWave + lane;
            q[0] = rfma(a0, b0, q[0]);
            q[kWave] = rfma(a1, b1, q[kWave]);
        } else {
            e[V][0] = rfma(a0, b0, e[V][0]);
            e[V][1] = rfma(a1, b1, e[V][1]);
        }
    }
    __device__ __forceinline__ float edge_get(int v, int half, int lane) const
    {
        if constexpr (kEdgeLds) return eld[(v * 2 + half) * kWave + lane];
        else return e[v][half];
    }
    template <int SLOT> __device__ __forceinline__ void fma(int u, float dj, float inA, float inB)
    {
        float(&m)[RPW][2] = SLOT == 0 ? m0 : SLOT == 1 ? m1 : SLOT == 2 ? m2 : m3;
        m[u][0] = __builtin_fmaf(dj, inA, m[u][0]);
        m[u][1] = __builtin_fmaf(dj, inB, m[u][1]);
    }
    __device__ __forceinline__ void store_one(const float (&m)[RPW][2], float *__restrict__ gW, int H, int j0, int lane) const
    {
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const int j = j0 + u;
            if (j < H) {
                if (lane < H) gW[(size_t)j * H + lane] = m[u][0];
                if (lane + 64 < H) gW[(size_t)j * H + lane + 64] = m[u][1];
            }
        }
    }
};

// one hidden matrix of a round: this wave's rows of W^T delta and of dW for all TB published trajectories.  Between the round's two
// barriers of this matrix (the caller places them).
template <int TB, int RPW, int SLOT>
__device__ __forceinline__ void multi_team_matrix(const StreamNet<float> &n, const int l, MultiAcc<RPW> &acc, const float *__restrict__ dl,
                                                  const float *__restrict__ hin, float *__restrict__ xl, int lane, int part, bool want_g)
{
    constexpr int NW = 8;
    const int H = n.H;
    const int j0 = part * RPW;                                 // this wave's rows [j0, j0 + RPW) (rows beyond H: zero cotangents)
    const float *W = n.Wh(l);
    const bool vA = lane < H, vB = lane + 64 < H;
    const unsigned oA = (unsigned)(vA ? lane : 0) * 4u, oB = (unsigned)(vB ? lane + 64 : 0) * 4u;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W), 0, (int)((unsigned)H * (unsigned)H * 4u), 0x00020000);
    // eight rows at a time stay in registers while a RUN-TIME loop walks the trajectories (unrolled over them, hipcc hoists every load of
    // every trajectory and spills a thousand registers); a trajectory's partial sums of the second chunk are added to the first's in LDS
#pragma unroll
    for (int c = 0; c < RPW / 8; ++c) {
        float wA[8], wB[8];
        // (rows at or beyond H: the buffer load is out of range and returns 0)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned roff = (unsigned)(j0 + 8 * c + u) * (unsigned)H * 4u;
            wA[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)oA, (int)roff, 0));
            wB[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)oB, (int)roff, 0));
        }
#pragma unroll 1
        for (int tt = 0; tt < TB; ++tt) {
            float dj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) dj[u] = dl[tt * 2 * kWave + j0 + 8 * c + u];      // wave-uniform address: a broadcast read
            float *__restrict__ xa = xl + ((tt * NW + part) * 2 + 0) * kWave + lane, *__restrict__ xb = xa + kWave;
            float pA = (c == 0) ? 0.f : *xa, pB = (c == 0) ? 0.f : *xb;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                pA = rfma(wA[u], dj[u], pA);
                pB = rfma(wB[u], dj[u], pB);
            }
            *xa = vA ? pA : 0.f;
            *xb = vB ? pB : 0.f;
            if (want_g) {
                const float inA = hin[tt * 2 * kWave + lane], inB = hin[tt * 2 * kWave + kWave + lane];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc.template fma<SLOT>(8 * c + u, dj[u], inA, inB);
            }
        }
    }
    // the bias of this matrix: its gradient is the sum of the published cotangents -- wave 3 + SLOT adds them for all trajectories
    if (want_g && part == 3 + SLOT) {
#pragma unroll 1
        for (int tt = 0; tt < TB; ++tt) { acc.eb[0] += dl[tt * 2 * kWave + lane]; acc.eb[1] += dl[tt * 2 * kWave + kWave + lane]; }
    }
}

// the hidden matrices of one round as a SERVING wave sees them (its own trajectory is finished): the round's first barrier has been
// passed by the caller
template <int TB, int RPW>
__device__ __forceinline__ void multi_serve_round(const StreamNet<float> &n, MultiAcc<RPW> &acc, const float *__restrict__ dl,
                                                  const float *__restrict__ hin, float *__restrict__ xl, int lane, int part, bool want_g)
{
    const int L = n.L;
#pragma unroll
    for (int i = 0; i < kGenAccMats; ++i) {
        if (i > L - 2) break;
        if (i > 0) __syncthreads();
        if (i == 0) multi_team_matrix<TB, RPW, 0>(n, L - 2, acc, dl, hin, xl, lane, part, want_g);
        else if (i == 1) multi_team_matrix<TB, RPW, 1>(n, L - 3, acc, dl, hin, xl, lane, part, want_g);
        else if (i == 2) multi_team_matrix<TB, RPW, 2>(n, L - 4, acc, dl, hin, xl, lane, part, want_g);
        else multi_team_matrix<TB, RPW, 3>(n, L - 5, acc, dl, hin, xl, lane, part, want_g);
        __syncthreads();
    }
}

// J^T kb of ONE stage of this wave's own trajectory, the hidden matrices as team rounds (see above).  own = this wave accumulates the
// edge gradients of its trajectory (false for the second wave of a trajectory when TB < 8).
template <bool GODE, int TB, int RPW>
__device__ __forceinline__ float rhs_vjp_multi(const StreamNet<float> &n, const EdgeImage<float> &ew, const OdeP<float> &o, float t, float tvns,
                                               float gde, float gd_in, bool use_gd, int lane, const float *__restrict__ rec, float kb, float &go,
                                               int part, MultiAcc<RPW> &acc, float *__restrict__ dl, float *__restrict__ hin,
                                               float *__restrict__ xl, bool want_g, bool own)
{
    constexpr int NW = 8;
    const int H = n.H, L = n.L, ts = part % TB;
    const float *__restrict__ sx = rec + 2 * L * kWave;
    const float G = sx[0], I = sx[1], Glu = sx[2], GLP1 = sx[3], GE = sx[4], FFA = sx[5];
    const float lq[6] = {lane_bcast(kb, 0), lane_bcast(kb, 1), lane_bcast(kb, 2), lane_bcast(kb, 3), lane_bcast(kb, 4), lane_bcast(kb, 5)};
    const int c8 = lane & 7;
    const float mech = mech_vjp<float, GODE>(o, G, I, Glu, GLP1, FFA, lq[0], lq[1], lq[2], lq[3], lq[5], gde, gd_in, use_gd, lane, go);
    const bool vA = lane < H, vB = lane + 64 < H;
    const int jA = vA ? lane : 0, jB = vB ? lane + 64 : 0;
    const bool eg = want_g && own;
    // output layer
    float hA = rec[(2 * (L - 1)) * kWave + lane], hB = rec[(2 * (L - 1) + 1) * kWave + lane];
    float dA = 0.f, dB = 0.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        dA = rfma(ew.Wo()[q * H + jA], lq[q], dA);
        dB = rfma(ew.Wo()[q * H + jB], lq[q], dB);
    }
    if (eg) {
        acc.template edge_fma<0>(lq[0], hA, lq[0], hB, lane); acc.template edge_fma<1>(lq[1], hA, lq[1], hB, lane);
        acc.template edge_fma<2>(lq[2], hA, lq[2], hB, lane); acc.template edge_fma<3>(lq[3], hA, lq[3], hB, lane);
        acc.template edge_fma<4>(lq[4], hA, lq[4], hB, lane); acc.template edge_fma<5>(lq[5], hA, lq[5], hB, lane);
        acc.template edge_add<16>(kb, 0.f, lane);             // lanes 0..5 carry the six components
    }
    dA = vA ? act_bwd(dA, hA, n.act) : 0.f;
    dB = vB ? act_bwd(dB, hB, n.act) : 0.f;
    auto hidden = [&](const int l, auto slot) {
        constexpr int SLOT = decltype(slot)::value;
        const float inA = rec[(2 * l) * kWave + lane], inB = rec[(2 * l + 1) * kWave + lane];
        dl[ts * 2 * kWave + lane] = dA;
        dl[ts * 2 * kWave + kWave + lane] = dB;
        hin[ts * 2 * kWave + lane] = vA ? inA : 0.f;
        hin[ts * 2 * kWave + kWave + lane] = vB ? inB : 0.f;
        __syncthreads();                                       // every trajectory's delta_{l+1} and h_l are published
        multi_team_matrix<TB, RPW, SLOT>(n, l, acc, dl, hin, xl, lane, part, want_g);
        __syncthreads();                                       // every wave's partial sums are in xl
        float pA = 0.f, pB = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { pA += xl[((ts * NW + w) * 2 + 0) * kWave + lane]; pB += xl[((ts * NW + w) * 2 + 1) * kWave + lane]; }
        dA = vA ? act_bwd(pA, inA, n.act) : 0.f;
        dB = vB ? act_bwd(pB, inB, n.act) : 0.f;
    };
#pragma unroll
    for (int i = 0; i < kGenAccMats; ++i) {
        if (i > L - 2) break;
        if (i == 0) hidden(L - 2, std::integral_constant<int, 0>{});
        else if (i == 1) hidden(L - 3, std::integral_constant<int, 1>{});
        else if (i == 2) hidden(L - 4, std::integral_constant<int, 2>{});
        else hidden(L - 5, std::integral_constant<int, 3>{});
    }
    // first layer
    const float in[9] = {t, G, I, Glu, GLP1, GE, FFA, GLP1, tvns};
    if (eg) {
        acc.template edge_fma<6>(dA, in[0], dB, in[0], lane); acc.template edge_fma<7>(dA, in[1], dB, in[1], lane);
        acc.template edge_fma<8>(dA, in[2], dB, in[2], lane); acc.template edge_fma<9>(dA, in[3], dB, in[3], lane);
        acc.template edge_fma<10>(dA, in[4], dB, in[4], lane); acc.template edge_fma<11>(dA, in[5], dB, in[5], lane);
        acc.template edge_fma<12>(dA, in[6], dB, in[6], lane); acc.template edge_fma<13>(dA, in[7], dB, in[7], lane);
        acc.template edge_fma<14>(dA, in[8], dB, in[8], lane);
        acc.template edge_add<15>(dA, dB, lane);
    }
    float w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = ew.W1()[jA * 9 + i] * dA + ew.W1()[jB * 9 + i] * dB;
    const float p[6] = {w[1], w[2], w[3], w[4] + w[7], w[5], w[6]};
    const float nn = wave_reduce6_to_lanes(p, lane);
    return (c8 < 6) ? (mech + nn) : 0.f;
}

template <bool GODE, bool GD, int TB, int RPW>
__global__ __launch_bounds__(512, 2) void solve_bwd_generic_multi_kernel(const AdjArgs<float> a, const int method, const int L)
{
    constexpr int NW = 8;
    MultiAcc<RPW> acc;
    __shared__ float rowsT[8 * kWave];
    __shared__ float eld_all[(MultiAcc<RPW>::kEdgeLds ? TB : 1) * kMultiEdgeVals * 2 * kWave];     // the walking waves' edge sums (H > 64)
    __shared__ float edge_img[kEdgeImageMax];
    __shared__ float dl[TB * 2 * kWave], hin[TB * 2 * kWave];
    constexpr int kXl = TB * NW * 2 * kWave > (kMultiEdgeVals + 1) * 2 * kWave ? TB * NW * 2 * kWave : (kMultiEdgeVals + 1) * 2 * kWave;
    __shared__ float xl[kXl];                                  // partial sums [TB][NW][2][64]; at the end the edge values [18][2][64]
    __shared__ int done_cnt;
    const int lane = threadIdx.x & 63;
    const int part = first_lane((int)(threadIdx.x >> 6));
    const int c8 = lane & 7, grp = lane >> 3;
    const int T = a.T, ts = part % TB;
    const bool own = part < TB;
    acc.zero(eld_all + (MultiAcc<RPW>::kEdgeLds && own ? part : 0) * kMultiEdgeVals * 2 * kWave, lane);      // (only walking waves touch it)
    const TableauData &tab = kTableau[method];
    const int S = tab.S;
    const int kSlot = 2 * L * kWave + 8;
    const int per_set = a.B / a.n_sets, set = blockIdx.y;
    const StreamNet<float> n{a.nn_p + (size_t)set * a.P, a.H, L, a.act};
    const bool want_g = a.gnn != nullptr;
    tableau_rowsT_store<float>(rowsT, method, threadIdx.x, 64 * NW);
    EdgeImage<float>::fill(edge_img, n, threadIdx.x, 64 * NW);
    const size_t rowlen = adj_partial_rowlen(a.P);
    float *__restrict__ prow = a.partials + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * rowlen;
    for (size_t i = threadIdx.x; i < rowlen; i += 64 * NW) prow[i] = 0.f;
    const EdgeImage<float> ew{edge_img, a.H, L};
    OdeP<float> o;
    ode_load(o, a.ode_p + 17 * set);
    constexpr bool use_gd = GD;
    float go_sum = 0.f;
    const int groups = (per_set + TB - 1) / TB;
    for (int g = blockIdx.x; g < groups; g += gridDim.x) {
        for (int i = threadIdx.x; i < TB * 2 * kWave; i += 64 * NW) { dl[i] = 0.f; hin[i] = 0.f; }
        if (threadIdx.x == 0) done_cnt = 0;
        __syncthreads();
        const int bi = g * TB + ts;
        if (bi < per_set) {
            const int b = set * per_set + bi;
            const float *__restrict__ tg = a.t + (a.t_batched ? (size_t)b * T : 0);
            const float *__restrict__ tape = a.tape + (size_t)b * a.max_steps * 8;
            const int *__restrict__ tseg = a.tape_seg + (size_t)b * a.max_steps;
            const float *__restrict__ stg = a.tape_stage + (size_t)b * a.max_steps * 6 * kSlot;
            const float *__restrict__ gyb = a.gy + (size_t)b * T * 6;
            const int nst = a.nsteps[b] < a.max_steps ? a.nsteps[b] : a.max_steps;
            const bool ok = a.status[b] == HODE_ST_OK;
            auto gy_row = [&](int r) -> float { return (c8 < 6) ? gyb[(size_t)r * 6 + c8] : 0.f; };
            float lam = 0.f, go = 0.f;
            int knext = T - 1;
            for (int st = nst - 1; st >= 0; --st) {
                const int kraw = tseg[st];
                const int k = kraw & (kSegClosed - 1);
                int hi = knext;                               // rows this step produced: see solve_bwd_kernel
                if (st == nst - 1) {
                    hi = T - 1;
                    if (!ok) {
                        hi = k;
                        if (kraw & kSegClosed) {
                            hi = k + 1;
                            while (hi + 1 < T && !(tg[hi + 1] > tg[hi])) ++hi;
                        }
                    }
                }
                for (int r = k + 1; r <= hi; ++r) lam += gy_row(r);
                knext = k;
                // the step header as the forward wrote it: {t, h, t0, 1 / (t1 - t0), v0, dv, d0, dd}
                const float tc = tape[(size_t)st * 8 + 0], h = tape[(size_t)st * 8 + 1], t0 = tape[(size_t)st * 8 + 2], inv_len = tape[(size_t)st * 8 + 3];
                const float v0 = tape[(size_t)st * 8 + 4], dv = tape[(size_t)st * 8 + 5], d0 = tape[(size_t)st * 8 + 6], dd = tape[(size_t)st * 8 + 7];
                float ZZ = 0.f;
                for (int s = S - 1; s >= 0; --s) {
                    const float bw_s = rowsT[6 * kWave + s], c_s = rowsT[6 * kWave + 8 + s];
                    const float kb = h * rfma(bw_s, lam, group_sum8(rowsT[s * kWave + lane] * ZZ));
                    const float tst = rfma(c_s, h, tc);
                    const float al = (tst - t0) * inv_len;
                    const float gdv = rfma(al, dd, d0);
                    float gde = 0.f;
                    if constexpr (use_gd) gde = gd_effect(o, gdv);
                    const float Z = rhs_vjp_multi<GODE, TB, RPW>(n, ew, o, tst, rfma(al, dv, v0), gde, gdv, use_gd, lane,
                                                                 stg + ((size_t)st * 6 + s) * kSlot, kb, go, part, acc, dl, hin, xl, want_g, own);
                    ZZ = (grp == s) ? Z : ZZ;
                }
                lam += group_sum8(rowsT[7 * kWave + lane] * ZZ);
            }
            int kf = 0;                                       // rows 0..kf are (copies of) x0
            while (kf + 1 < T && !(tg[kf + 1] > tg[kf])) ++kf;
            for (int r = 0; r <= kf; ++r) lam += gy_row(r);
            if (own) {
                if (lane < 6) a.gx0[(size_t)b * 6 + lane] = lam;
                if constexpr (GODE) go_sum += go;
            }
            // nothing of this trajectory may enter the other waves' sums any more
            dl[ts * 2 * kWave + lane] = 0.f; dl[ts * 2 * kWave + kWave + lane] = 0.f;
            hin[ts * 2 * kWave + lane] = 0.f; hin[ts * 2 * kWave + kWave + lane] = 0.f;
        }
        if (lane == 0) atomicAdd(&done_cnt, 1);
        for (;;) {
            __syncthreads();                                  // = the first barrier of a round
            if (*(volatile int *)&done_cnt == NW) break;      // (every wave is here: they all read NW in the same round)
            multi_serve_round<TB, RPW>(n, acc, dl, hin, xl, lane, part, want_g);
        }
        __syncthreads();                                      // the counter is reset by the next group's prologue
    }
    // ---- this workgroup's gradient row: the hidden matrices' rows straight from their owners, the edge gradients and the ODE-constant
    //      sums added over the walking waves in wave order through LDS (xl is free now)
    if (!want_g && !GODE) return;
    __syncthreads();
    if (want_g) {
        const int nm = L - 1, j0 = part * RPW;
        if (nm > 0) acc.store_one(acc.m0, prow + n.hid_off(nm - 1), a.H, j0, lane);
        if (nm > 1) acc.store_one(acc.m1, prow + n.hid_off(nm - 2), a.H, j0, lane);
        if (nm > 2) acc.store_one(acc.m2, prow + n.hid_off(nm - 3), a.H, j0, lane);
        if (nm > 3) acc.store_one(acc.m3, prow + n.hid_off(nm - 4), a.H, j0, lane);
    }
    // the hidden biases: wave 3 + slot holds the sum over all trajectories (slot sl = hidden matrix L - 2 - sl)
    if (want_g && part >= 3 && part - 3 < L - 1) {
        float *__restrict__ gb = prow + n.hid_off(L - 2 - (part - 3)) + (size_t)a.H * a.H;
        if (lane < a.H) gb[lane] = acc.eb[0];
        if (lane + 64 < a.H) gb[lane + 64] = acc.eb[1];
    }
    // edge values of the walking waves, added in wave order: through xl (free now) when they live in registers, straight from the waves'
    // LDS tables otherwise; value 17 = the ODE-constant sums
    constexpr int kVals = kMultiEdgeVals + 1;
    static_assert(kVals * 2 * kWave <= kXl, "xl holds the edge values");
    for (int w = 0; w < TB; ++w) {
        __syncthreads();
        if (part == w) {
            for (int v = 0; v < kVals; ++v) {
                float x0, x1;
                if (v < kMultiEdgeVals) {
                    if constexpr (MultiAcc<RPW>::kEdgeLds) { x0 = acc.edge_get(v, 0, lane); x1 = acc.edge_get(v, 1, lane); }
                    else {
                        x0 = x1 = 0.f;
#pragma unroll
                        for (int vv = 0; vv < kMultiEdgeVals; ++vv) if (vv == v) { x0 = acc.e[MultiAcc<RPW>::kEdgeLds ? 0 : vv][0]; x1 = acc.e[MultiAcc<RPW>::kEdgeLds ? 0 : vv][1]; }
                    }
                } else {
                    x0 = go_sum; x1 = 0.f;
                }
                float *__restrict__ q = xl + (v * 2) * kWave + lane;
                q[0] = (w == 0 ? 0.f : q[0]) + x0;
                q[kWave] = (w == 0 ? 0.f : q[kWave]) + x1;
            }
        }
    }
    __syncthreads();
    if (part == 0) {
        const int H = a.H;
        const bool vA = lane < H, vB = lane + 64 < H;
        auto get = [&](int v, int half) { return xl[(v * 2 + half) * kWave + lane]; };
        if (want_g) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                if (vA) prow[n.out_off() + q * H + lane] = get(q, 0);
                if (vB) prow[n.out_off() + q * H + lane + 64] = get(q, 1);
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (vA) prow[lane * 9 + i] = get(6 + i, 0);
                if (vB) prow[(lane + 64) * 9 + i] = get(6 + i, 1);
            }
            if (vA) prow[9 * H + lane] = get(15, 0);
            if (vB) prow[9 * H + lane + 64] = get(15, 1);
            if (lane < 6) prow[n.out_off() + 6 * H + lane] = get(16, 0);
        }
        if constexpr (GODE) {
            if (lane < 17) prow[a.P + lane] = get(17, 0);
        }
    }
}

template <int TB, int RPW> static int launch_bwd_generic_multi(hipStream_t s, const AdjArgs<float> &a, int L, int method, int bps)
{
    const dim3 grid(bps, a.n_sets), block(512);
    const bool gd = a.gd_mode != 0;
    if (a.gode) {
        if (gd) hipLaunchKernelGGL((solve_bwd_generic_multi_kernel<true, true, TB, RPW>), grid, block, 0, s, a, method, L);
        else hipLaunchKernelGGL((solve_bwd_generic_multi_kernel<true, false, TB, RPW>), grid, block, 0, s, a, method, L);
    } else {
        if (gd) hipLaunchKernelGGL((solve_bwd_generic_multi_kernel<false, true, TB, RPW>), grid, block, 0, s, a, method, L);
        else hipLaunchKernelGGL((solve_bwd_generic_multi_kernel<false, false, TB, RPW>), grid, block, 0, s, a, method, L);
    }
    launch_adj_reduce(s, a.partials, (int)adj_partial_rowlen(a.P), bps, a.n_sets, a.P, a.gnn, a.gode);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <typename R, int NW, int ACCREG> static int launch_bwd_generic_t(hipStream_t s, const AdjArgs<R> &a, int L, int method)
{
    // with register accumulators a workgroup flushes once: fewer, longer-lived workgroups (a few per CU) beat one per trajectory
    int blocks = a.B < 4096 ? a.B : 4096;
    if (ACCREG && blocks > 1024) blocks = 1024;
    dim3 grid(blocks);
    const dim3 block(64 * NW);
    // gradient rows instead of atomics (fp32 register accumulators, enough rows for one workgroup per set at least)
    const int per_set = a.B / a.n_sets;
    int bps = 0;
    if constexpr (ACCREG != 0 && sizeof(R) == 4) {
        if (a.partials != nullptr && a.gnn != nullptr && a.n_sets <= a.partial_rows && a.n_sets <= 65535) {
            bps = a.partial_rows / a.n_sets;
            if (bps > per_set) bps = per_set;
            if (bps > 1024) bps = 1024;
            if (bps < 1) bps = 1;
            grid = dim3(bps, a.n_sets);
        }
    }
    const bool gd = a.gd_mode != 0;
    if (a.gode) {
        if (gd) hipLaunchKernelGGL((solve_bwd_generic_kernel<R, true, true, NW, ACCREG>), grid, block, 0, s, a, method, L, bps > 0 ? 1 : 0);
        else hipLaunchKernelGGL((solve_bwd_generic_kernel<R, true, false, NW, ACCREG>), grid, block, 0, s, a, method, L, bps > 0 ? 1 : 0);
    } else {
        if (gd) hipLaunchKernelGGL((solve_bwd_generic_kernel<R, false, true, NW, ACCREG>), grid, block, 0, s, a, method, L, bps > 0 ? 1 : 0);
        else hipLaunchKernelGGL((solve_bwd_generic_kernel<R, false, false, NW, ACCREG>), grid, block, 0, s, a, method, L, bps > 0 ? 1 : 0);
    }
    if constexpr (sizeof(R) == 4) {
        if (bps > 0)
            launch_adj_reduce(s, reinterpret_cast<const float *>(a.partials), (int)adj_partial_rowlen(a.P), bps, a.n_sets, a.P,
                              reinterpret_cast<float *>(a.gnn), reinterpret_cast<float *>(a.gode));
    }
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}
template <typename R> int launch_solve_bwd_generic(hipStream_t s, const AdjArgs<R> &a, int L, int method)
{
    if constexpr (sizeof(R) == 4) {
        // large batches: several trajectories per team (solve_bwd_generic_multi_kernel) -- needs gradient rows (one per workgroup)
        if (L >= 2 && L - 1 <= kGenAccMats && a.gnn != nullptr && a.partials != nullptr && a.n_sets <= a.partial_rows && a.n_sets <= 65535) {
            const int per_set = a.B / a.n_sets;
            const int rows_per_set = a.partial_rows / a.n_sets;
            auto bps_for = [&](int tb) { int g = (per_set + tb - 1) / tb; if (g > rows_per_set) g = rows_per_set; if (g > 512) g = 512; return g < 1 ? 1 : g; };
            // (the fewest trajectories per team that put the batch on the chip in one round: see launch_fwd_generic_m)
            if (a.B > 1024)
                return a.H > 64 ? launch_bwd_generic_multi<8, 16>(s, a, L, method, bps_for(8)) : launch_bwd_generic_multi<8, 8>(s, a, L, method, bps_for(8));
            if (a.B > 512)
                return a.H > 64 ? launch_bwd_generic_multi<4, 16>(s, a, L, method, bps_for(4)) : launch_bwd_generic_multi<4, 8>(s, a, L, method, bps_for(4));
            if (a.B > 256)
                return a.H > 64 ? launch_bwd_generic_multi<2, 16>(s, a, L, method, bps_for(2)) : launch_bwd_generic_multi<2, 8>(s, a, L, method, bps_for(2));
        }
        // fp32, at most kGenAccMats hidden matrices: ALL parameter gradients accumulate in the team's registers (the wave's rows of
        // every hidden matrix, the edge pieces dealt out over the waves) and leave once per workgroup.  Teams of EIGHT waves also
        // above 64 hidden units (sixteen rows per wave: 128 accumulator registers): a 16-wave workgroup is capped at 128 VGPRs per
        // wave, which the 76 accumulators + the kernel's own ~60 registers do not fit (484 B of spills inside the stage loop).
        if (L - 1 <= kGenAccMats && a.gnn != nullptr)
            return a.H > 64 ? launch_bwd_generic_t<R, 8, 16>(s, a, L, method) : launch_bwd_generic_t<R, 8, 8>(s, a, L, method);
    }
    return a.H > 64 ? launch_bwd_generic_t<R, 16, 0>(s, a, L, method) : launch_bwd_generic_t<R, 8, 0>(s, a, L, method);
}

template int launch_rhs_fwd_generic<float>(hipStream_t, const RhsArgs<float> &, int);
template int launch_rhs_fwd_generic<double>(hipStream_t, const RhsArgs<double> &, int);
template int launch_rhs_bwd_generic<float>(hipStream_t, const RhsArgs<float> &, int);
template int launch_rhs_bwd_generic<double>(hipStream_t, const RhsArgs<double> &, int);
template int launch_solve_fwd_generic<float>(hipStream_t, const SolveArgs<float> &, int);
template int launch_solve_fwd_generic<double>(hipStream_t, const SolveArgs<double> &, int);
template int launch_solve_bwd_generic<float>(hipStream_t, const AdjArgs<float> &, int, int);
template int launch_solve_bwd_generic<double>(hipStream_t, const AdjArgs<double> &, int, int);

}  // namespace hode
