// hode_solve_bwd_split.hip -- K4 for the tuned fp32 path as TWO kernels: an EXPERIMENT (HODE_BWD=split), not the default.
//
// The fused adjoint (hode_solve_bwd.hip) keeps 192 gradient accumulators in registers, so the transposed matrices it needs
// for delta_{l-1} = W_l^T delta_l have to come from LDS: 48 KB of ds_read_b128 per stage and per wave, at 2 waves per SIMD.
// Its stage time (2 920 cycles per SIMD) is 1 455 cycles of DPP FMAs, ~900 of other VALU and ~550 of waits.  The idea here:
//
//   A  solve_bwd_prop_kernel   reverse sweep proper: lambda, kb, mechanistic J^T, delta propagation.  No parameter-gradient
//                              accumulators, so the transposed matrices live in REGISTERS (192, same rotating-operand
//                              order as the forward): the kernel has the forward's shape and instruction count (333 VALU
//                              per stage, 180 of them DPP FMAs).  It writes what the gradients need, per stage, to the
//                              "delta tape": delta_1 .. delta_L (rows of 64) | kb[6], t, tVNS.
//   B  solve_bwd_accum_kernel  a streaming reduction over all (trajectory, step, stage) records:
//                              dW_l += delta_l (x) h_{l-1} (64 v_fmac_f32_dpp per matrix, accumulators in registers),
//                              first / last layer and bias gradients.  Both tapes arrive by LDS-DMA, three records ahead.
//
// MEASURED (MI355X, 4 096 x 241, fp32): A 5.5 ms + B 3.3 ms = 8.8 ms against 8.0 ms for the fused kernel, identical
// gradients (tests/test_hip_parity.py::test_split_adjoint_matches_the_fused_adjoint).  A runs the forward's instruction
// count 30 % slower than the forward (every stage starts with LDS reads of a DMA'd record); B, with nothing but outer
// products, still needs 1 180 cycles per sample at 2 waves per SIMD.  The LDS reads the split removes were not where the
// fused kernel's time goes, and the layer cotangents cross HBM twice more (+12 GB per step).  Lessons kept in the code:
// vmcnt retires in issue order, so (i) record DMAs must be waited for with a COUNT that leaves the younger delta stores
// in flight, (ii) with stages this short the DMA ring has to run three records ahead.
#include "../hode_device.h"
#include "../hode_kernels.h"
#include <cstdlib>

namespace hode {

namespace {

constexpr int kPropWaves = 4;       // waves per workgroup of kernel A (2 workgroups per CU: 2 waves per SIMD)
constexpr int kAccWaves = 8;        // waves per workgroup of kernel B (1 workgroup per CU)
// records in flight ahead of the one being processed (LDS-DMA rings of kAhead + 1 slots).  A stage of kernel A is ~1 300
// cycles, a sample of kernel B ~800: one record ahead is less lead than the loaded-HBM latency
constexpr int kAhead = 3;
constexpr int kRing = kAhead + 1;

template <typename R> __device__ __forceinline__ R inp_at_s(const R *__restrict__ p, int mode, int b, int T, int k)
{
    if (mode == 0) return R(0);
    return (mode == 1) ? p[b] : p[(size_t)b * T + k];
}

// one (fp32) stage record: `rows` rows of 64 reals + 8 reals, HBM -> LDS without a VGPR destination
__device__ __forceinline__ void record_dma(const float *__restrict__ src, float *dst, int rows, int lane)
{
    for (int l = 0; l < rows; ++l)
        __builtin_amdgcn_global_load_lds(src + l * kWave + lane, (__attribute__((address_space(3))) void *)(dst + l * kWave), 4, 0, 0);
    if (lane < 8)
        __builtin_amdgcn_global_load_lds(src + rows * kWave + lane, (__attribute__((address_space(3))) void *)(dst + rows * kWave), 4, 0, 0);
}

// ------------------------------------------------------------------------------------------ A: propagation
// J^T kb of one stage without any parameter-gradient work; the layer cotangents go to drec (global memory).
template <int NL, bool GODE>
__device__ __forceinline__ float rhs_vjp_prop(const float (&w1)[9], const float (&w5)[6], const WtRegs<NL> &wt, const OdeP<float> &o,
                                              float t, float Y, float tvns, float gde, float gd_in, bool use_gd, int lane,
                                              const MlpActs<float, NL> &acts, float kb, float &go, float *__restrict__ drec)
{
    const float G = lane_bcast(Y, 0), I = lane_bcast(Y, 1), Glu = lane_bcast(Y, 2), GLP1 = lane_bcast(Y, 3),
                FFA = lane_bcast(Y, 5);
    const float lG = lane_bcast(kb, 0), lI = lane_bcast(kb, 1), lGlu = lane_bcast(kb, 2), lGLP = lane_bcast(kb, 3),
                lGE = lane_bcast(kb, 4), lF = lane_bcast(kb, 5);
    const int c8 = lane & 7;
    const float mech = mech_vjp<float, GODE>(o, G, I, Glu, GLP1, FFA, lG, lI, lGlu, lGLP, lF, gde, gd_in, use_gd, lane, go);
    float d = w5[0] * lG;
    d = rfma(w5[1], lI, d);
    d = rfma(w5[2], lGlu, d);
    d = rfma(w5[3], lGLP, d);
    d = rfma(w5[4], lGE, d);
    d = rfma(w5[5], lF, d);
    d = (acts.h[NL - 1] > 0.f) ? d : 0.f;
#ifndef HODE_EXPERIMENT_NO_DSTORE
    drec[(NL - 1) * kWave + lane] = d;                               // delta_NL
#endif
#pragma unroll
    for (int l = NL - 1; l >= 1; --l) {                              // hidden matrix l-1 maps h_l -> h_{l+1}
        const float dp = wt.mul(l - 1, lane, d);
        d = (acts.h[l - 1] > 0.f) ? dp : 0.f;
#ifndef HODE_EXPERIMENT_NO_DSTORE
        drec[(l - 1) * kWave + lane] = d;                            // delta_l
#endif
    }
#ifndef HODE_EXPERIMENT_NO_DSTORE
    if (lane < 8) drec[NL * kWave + lane] = (lane < 6) ? kb : (lane == 6) ? t : tvns;
#endif
    float p[6];
    p[0] = w1[1] * d;
    p[1] = w1[2] * d;
    p[2] = w1[3] * d;
    p[3] = (w1[4] + w1[7]) * d;                                      // GLP1 feeds inputs 4 and 7
    p[4] = w1[5] * d;
    p[5] = w1[6] * d;
    const float nn = wave_reduce6_to_lanes(p, lane);
    return (c8 < 6) ? (mech + nn) : 0.f;
}

template <int NL, bool GODE, bool GD>
__global__ __launch_bounds__(64 * kPropWaves, 2) void solve_bwd_prop_kernel(const AdjArgs<float> a, const int method)
{
    using R = float;
    constexpr int kRows = NL;
    constexpr int kSlot = kRows * kWave + 8;          // stage record (forward) and delta record (this kernel): same shape
    constexpr int kBuf = kRows * kWave + kWave;
    __shared__ R rowsT[8 * kWave];
    __shared__ R recs[kPropWaves * kRing * kBuf];
    const int lane = threadIdx.x & 63;
    const int c8 = lane & 7, grp = lane >> 3;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    const int set = blockIdx.y;
    const int T = a.T;
    const int per_set = a.B / a.n_sets;
    const TableauData &tab = kTableau[method];
    const int S = tab.S;
    const R *__restrict__ nn_set = a.nn_p + (size_t)set * a.P;
    R *rec = recs + (size_t)wave * kRing * kBuf;

    tableau_rowsT_store<R>(rowsT, method, threadIdx.x, 64 * kPropWaves);
    OdeP<R> o;
    ode_load(o, a.ode_p + 17 * set);
    WtRegs<NL> wt;
    wt.load(nn_set, a.H, lane);
    R w1[9], w5[6];
    {
        const R live = (lane < a.H) ? 1.f : 0.f;
        const int j = (lane < a.H) ? lane : a.H - 1;
        const R *pout = nn_set + 9 * a.H + a.H + (size_t)(NL - 1) * ((size_t)a.H * a.H + a.H);
#pragma unroll
        for (int i = 0; i < 9; ++i) w1[i] = live * nn_set[j * 9 + i];
#pragma unroll
        for (int q = 0; q < 6; ++q) w5[q] = live * pout[q * a.H + j];
    }
    __syncthreads();
    constexpr bool use_gd = GD;
    R go = 0.f;

    for (int bi = wave * gridDim.x + blockIdx.x; bi < per_set; bi += gridDim.x * kPropWaves) {
        const int b = set * per_set + bi;
        const R *__restrict__ tg = a.t + (a.t_batched ? (size_t)b * T : 0);
        const R *__restrict__ tape = a.tape + (size_t)b * a.max_steps * 8;
        const int *__restrict__ tseg = a.tape_seg + (size_t)b * a.max_steps;
        const R *__restrict__ stg = a.tape_stage + (size_t)b * a.max_steps * 6 * kSlot;
        R *__restrict__ dtp = a.tape_delta + (size_t)b * a.max_steps * 6 * kSlot;
        const R *__restrict__ gyb = a.gy + (size_t)b * T * 6;
        const int n = a.nsteps[b] < a.max_steps ? a.nsteps[b] : a.max_steps;
        const bool ok = a.status[b] == HODE_ST_OK;
        R lam = 0.f;
        int knext = T - 1, cur = 0;
        // records are consumed in the order (n-1, S-1), (n-1, S-2), ..., (0, 0); (pst, ps) walks kAhead records ahead.
        // vmcnt retires in issue order.  Behind the DMA of record i (issued at the top of stage i - kAhead) come, in steady
        // state, kAhead groups of NL + 1 delta stores and kAhead - 1 younger record DMAs of NL + 1 instructions each:
        // "at most that many outstanding" == record i has landed.  Outside the steady state (head and tail of a trajectory)
        // fewer operations follow and the same count would not prove it: those stages drain.
        constexpr int kYounger = (2 * kAhead - 1) * (NL + 1);
        static_assert(kYounger < 64, "vmcnt field");
        constexpr int kWaitSteady = 0x0f70 | (kYounger & 15) | ((kYounger >> 4) << 14);
        const int Ntot = n * S;
        int pst = n - 1, ps = S - 1, issued = 0, idx = 0;
        auto issue_next = [&]() {
            if (pst < 0) return;
            record_dma(stg + ((size_t)pst * 6 + ps) * kSlot, rec + (issued % kRing) * kBuf, kRows, lane);
            ++issued;
            if (--ps < 0) { ps = S - 1; --pst; }
        };
        for (int j = 0; j < kAhead; ++j) issue_next();
#pragma unroll 1
        for (int st = n - 1; st >= 0; --st) {
            const int kraw = tseg[st];
            const int k = kraw & (kSegClosed - 1);
            int hi = knext;                                   // rows this step produced: see solve_bwd_kernel
            if (st == n - 1) {
                hi = T - 1;
                if (!ok) {
                    hi = k;
                    if (kraw & kSegClosed) {
                        hi = k + 1;
                        while (hi + 1 < T && !(tg[hi + 1] > tg[hi])) ++hi;
                    }
                }
            }
            for (int r = k + 1; r <= hi; ++r) {
                const R *__restrict__ gr = gyb + (size_t)r * 6;
                const R g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3], g4 = gr[4], g5 = gr[5];
                lam += (c8 == 0) ? g0 : (c8 == 1) ? g1 : (c8 == 2) ? g2 : (c8 == 3) ? g3 : (c8 == 4) ? g4 : (c8 == 5) ? g5 : 0.f;
            }
            knext = k;
            const R tc = tape[(size_t)st * 8 + 0], h = tape[(size_t)st * 8 + 1];
            const R t0 = tg[k], t1 = tg[k + 1];
            const R v0 = inp_at_s(a.tvns, a.tvns_mode, b, T, k), v1 = inp_at_s(a.tvns, a.tvns_mode, b, T, k + 1);
            const R d0 = inp_at_s(a.gd, a.gd_mode, b, T, k), d1 = inp_at_s(a.gd, a.gd_mode, b, T, k + 1);
            const R inv_len = first_lane(1.f / (t1 - t0));
            const R dv = first_lane(v1 - v0), dd = first_lane(d1 - d0);
            R ZZ = 0.f;
#pragma unroll 1
            for (int s = S - 1; s >= 0; --s) {
#ifdef HODE_EXPERIMENT_NO_DSTORE
                __builtin_amdgcn_s_waitcnt(0x0f70);
#else
                if (idx >= kAhead && idx + kAhead - 1 < Ntot) __builtin_amdgcn_s_waitcnt(kWaitSteady);
                else __builtin_amdgcn_s_waitcnt(0x0f70);
#endif
                __builtin_amdgcn_wave_barrier();
                issue_next();                                  // record idx + kAhead -> the slot consumed one stage ago
                ++idx;
                MlpActs<R, NL> ac;
#pragma unroll
                for (int l = 0; l < NL; ++l) ac.h[l] = rec[cur * kBuf + l * kWave + lane];
                const R Ys = rec[cur * kBuf + kRows * kWave + c8];
                cur = (cur + 1) % kRing;
                const R bw_s = rowsT[6 * kWave + s], c_s = rowsT[6 * kWave + 8 + s];
                const R kb = h * rfma(bw_s, lam, group_sum8(rowsT[s * kWave + lane] * ZZ));
                const R ts = rfma(c_s, h, tc);
                const R al = (ts - t0) * inv_len;
                const R gdv = rfma(al, dd, d0);
                R gde = 0.f;
                if constexpr (use_gd) gde = gd_effect(o, gdv);
                const R Z = rhs_vjp_prop<NL, GODE>(w1, w5, wt, o, ts, Ys, rfma(al, dv, v0), gde, gdv, use_gd, lane, ac, kb, go,
                                                   dtp + ((size_t)st * 6 + s) * kSlot);
                ZZ = (grp == s) ? Z : ZZ;
            }
            lam += group_sum8(rowsT[7 * kWave + lane] * ZZ);
        }
        int kf = 0;                                           // rows 0..kf are (copies of) x0
        while (kf + 1 < T && !(tg[kf + 1] > tg[kf])) ++kf;
        for (int r = 0; r <= kf; ++r) {
            const R *__restrict__ gr = gyb + (size_t)r * 6;
            const R g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3], g4 = gr[4], g5 = gr[5];
            lam += (c8 == 0) ? g0 : (c8 == 1) ? g1 : (c8 == 2) ? g2 : (c8 == 3) ? g3 : (c8 == 4) ? g4 : (c8 == 5) ? g5 : 0.f;
        }
        if (lane < 6) a.gx0[(size_t)b * 6 + lane] = lam;
        __builtin_amdgcn_s_waitcnt(0x0f70);                   // drain before the next trajectory's first DMA is counted
    }
    if constexpr (GODE) {
        if (a.gode && lane < 17) atomic_add(a.gode + 17 * set + lane, go);
    }
}

// ------------------------------------------------------------------------------------------ B: accumulation
// Every wave keeps the whole gradient of its samples in registers (3 x 64 hidden-matrix accumulators in the rotating-operand
// order of the weights + 16 + NL edge accumulators) and loops over its trajectories / steps / stages; the 8 waves of a
// workgroup reduce through LDS and flush coalesced atomics, as the fused kernel does.
template <int NL>
__global__ __launch_bounds__(64 * kAccWaves, 2) void solve_bwd_accum_kernel(const AdjArgs<float> a, const int method)
{
    using R = float;
    using ES = EdgeSlots<NL>;
    constexpr int kRows = NL;
    constexpr int kSlot = kRows * kWave + 8;
    constexpr int kBuf = kRows * kWave + kWave;
    constexpr int kHid = (NL > 1 ? NL - 1 : 0) * kMaxH * kMaxH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *red = reinterpret_cast<R *>(smem_raw);                            // [(NL-1)][64][64] cross-wave sum of the hidden matrices
    R *redE = red + (kHid > 0 ? kHid : 1);                               // [slots][64] cross-wave sum of the edge parameters
    R *recs = redE + ES::count * kWave;                                  // [waves][2 (h | delta)][kRing][kBuf]
    const int lane = threadIdx.x & 63;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    const int set = blockIdx.y;
    const int per_set = a.B / a.n_sets;
    const int S = kTableau[method].S;
    R *hbuf = recs + (size_t)wave * 2 * kRing * kBuf, *dbuf = hbuf + kRing * kBuf;
    const int p16 = lane & 15;

    R gwh[(NL > 1) ? NL - 1 : 1][kMaxH];
#pragma unroll
    for (int l = 0; l < ((NL > 1) ? NL - 1 : 1); ++l)
#pragma unroll
        for (int k = 0; k < kMaxH; ++k) gwh[l][k] = 0.f;
    R ge[ES::count];
#pragma unroll
    for (int i = 0; i < ES::count; ++i) ge[i] = 0.f;

    for (int bi = wave * gridDim.x + blockIdx.x; bi < per_set; bi += gridDim.x * kAccWaves) {
        const int b = set * per_set + bi;
        const R *__restrict__ stg = a.tape_stage + (size_t)b * a.max_steps * 6 * kSlot;
        const R *__restrict__ dtp = a.tape_delta + (size_t)b * a.max_steps * 6 * kSlot;
        const int n = a.nsteps[b] < a.max_steps ? a.nsteps[b] : a.max_steps;
        const int N = n * S;                                  // records of this trajectory, any order: (i / S, i % S)
        // both records of sample i sit in ring slot i % kRing; the DMAs run kAhead samples ahead.  Only DMAs are in flight
        // here (2 (NL + 1) instructions per sample), in issue order: with the kAhead - 1 younger samples still allowed
        // outstanding, sample i has landed
        constexpr int kYounger = (kAhead - 1) * 2 * (NL + 1);
        static_assert(kYounger < 64, "vmcnt field");
        constexpr int kWaitSteady = 0x0f70 | (kYounger & 15) | ((kYounger >> 4) << 14);
        int cur = 0, pst = 0, ps = 0, issued = 0;
        auto issue_next = [&]() {
            if (issued >= N) return;
            record_dma(stg + ((size_t)pst * 6 + ps) * kSlot, hbuf + (issued % kRing) * kBuf, kRows, lane);
            record_dma(dtp + ((size_t)pst * 6 + ps) * kSlot, dbuf + (issued % kRing) * kBuf, kRows, lane);
            ++issued;
            if (++ps == S) { ps = 0; ++pst; }
        };
        for (int j = 0; j < kAhead; ++j) issue_next();
#pragma unroll 1
        for (int i = 0; i < N; ++i) {
            if (i + kAhead - 1 < N) __builtin_amdgcn_s_waitcnt(kWaitSteady);
            else __builtin_amdgcn_s_waitcnt(0x0f70);          // tail: fewer younger DMAs than the count assumes
            __builtin_amdgcn_wave_barrier();
            issue_next();                                     // sample i + kAhead -> the slot consumed one sample ago
            const R *__restrict__ hr = hbuf + cur * kBuf, *__restrict__ dr = dbuf + cur * kBuf;
            // hidden matrices: dW_l += delta_{l+2} (x) h_{l+1}  (rows l+1 of the delta record, l of the activation record)
#pragma unroll
            for (int l = 0; l < NL - 1; ++l) {
                float Rh[4];
                Rh[0] = hr[l * kWave + p16]; Rh[1] = hr[l * kWave + 16 + p16]; Rh[2] = hr[l * kWave + 32 + p16]; Rh[3] = hr[l * kWave + 48 + p16];
                const R d = dr[(l + 1) * kWave + lane];
                asm volatile("" : "+v"(Rh[0]), "+v"(Rh[1]), "+v"(Rh[2]), "+v"(Rh[3]));
                mlp_outer_step<0>(gwh[l], d, Rh);
                ge[ES::b + l + 1] += d;                       // bias of layer l + 2
            }
            // first layer: input row [t, G, I, Glu, GLP1, GE, FFA, glp1 := GLP1, tvns] (wave-uniform values)
            const R d1 = dr[lane];
            ge[ES::b + 0] += d1;
            const R *__restrict__ xs = hr + kRows * kWave, *__restrict__ ts = dr + kRows * kWave;
            ge[ES::w1 + 0] = rfma(d1, ts[6], ge[ES::w1 + 0]);
            ge[ES::w1 + 1] = rfma(d1, xs[0], ge[ES::w1 + 1]);
            ge[ES::w1 + 2] = rfma(d1, xs[1], ge[ES::w1 + 2]);
            ge[ES::w1 + 3] = rfma(d1, xs[2], ge[ES::w1 + 3]);
            ge[ES::w1 + 4] = rfma(d1, xs[3], ge[ES::w1 + 4]);
            ge[ES::w1 + 5] = rfma(d1, xs[4], ge[ES::w1 + 5]);
            ge[ES::w1 + 6] = rfma(d1, xs[5], ge[ES::w1 + 6]);
            ge[ES::w1 + 7] = rfma(d1, xs[3], ge[ES::w1 + 7]);
            ge[ES::w1 + 8] = rfma(d1, ts[7], ge[ES::w1 + 8]);
            // output layer: dWout[q][j] += kb_q h_NL[j], dbout[q] += kb_q
            const R hl = hr[(NL - 1) * kWave + lane];
#pragma unroll
            for (int q = 0; q < 6; ++q) ge[ES::w5 + q] = rfma(ts[q], hl, ge[ES::w5 + q]);
            ge[ES::b5] += ts[lane & 7];                        // lanes 0..5 hold dbout (slots 6, 7 carry t / tVNS: never flushed)
            cur = (cur + 1) % kRing;
        }
    }

    if (a.gnn) {
        R *__restrict__ gp = a.gnn + (size_t)set * a.P;
        const int nthreads = 64 * kAccWaves;
        const int H = a.H;
        for (int i = threadIdx.x; i < kHid + ES::count * kWave; i += nthreads) red[i] = 0.f;    // red and redE are contiguous
        __syncthreads();
        for (int w = 0; w < kAccWaves; ++w) {
            if (wave == w) {
#pragma unroll
                for (int l = 0; l < NL - 1; ++l)
#pragma unroll
                    for (int r = 0; r < kMaxH; ++r) red[l * kMaxH * kMaxH + lane * kMaxH + wcol<R>(r, lane)] += gwh[l][r];
#pragma unroll
                for (int i = 0; i < ES::count; ++i) redE[i * kWave + lane] += ge[i];
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < kHid; i += nthreads) {
            const int l = i >> 12, row = (i >> 6) & 63, col = i & 63;
            if (row < H && col < H)
                atomic_add(gp + 9 * H + H + (size_t)l * ((size_t)H * H + H) + (size_t)row * H + col, red[i]);
        }
        const size_t off_out = (size_t)9 * H + H + (size_t)(NL - 1) * ((size_t)H * H + H);
        for (int i = threadIdx.x; i < ES::count * kWave; i += nthreads) {
            const int slot = i >> 6, j = i & 63;
            const R v = redE[i];
            if (slot < ES::b) { if (j < H) atomic_add(gp + j * 9 + slot, v); }
            else if (slot < ES::w5) {
                const int l = slot - ES::b;
                if (j < H) atomic_add(gp + 9 * H + (l == 0 ? 0 : H + (size_t)(l - 1) * ((size_t)H * H + H) + (size_t)H * H) + j, v);
            } else if (slot < ES::b5) { if (j < H) atomic_add(gp + off_out + (slot - ES::w5) * H + j, v); }
            else if (j < 6) atomic_add(gp + off_out + 6 * H + j, v);
        }
    }
}

template <int NL> constexpr size_t accum_lds_bytes()
{
    constexpr size_t hid = (NL > 1 ? NL - 1 : 0) * (size_t)kMaxH * kMaxH;
    return ((hid > 0 ? hid : 1) + EdgeSlots<NL>::count * kWave + (size_t)kAccWaves * 2 * kRing * (NL * kWave + kWave)) * sizeof(float);
}

template <int NL, bool GODE, bool GD> int launch_split_g(hipStream_t s, const AdjArgs<float> &a, int method)
{
    const int per_set = a.B / a.n_sets;
    // A: 2 workgroups of 4 waves per CU, waves loop over trajectories
    int blocksA = (per_set + kPropWaves - 1) / kPropWaves;
    const int capA = (512 / a.n_sets) > 0 ? 512 / a.n_sets : 1;
    if (blocksA > capA) blocksA = capA;
    hipLaunchKernelGGL((solve_bwd_prop_kernel<NL, GODE, GD>), dim3(blocksA, a.n_sets), dim3(64 * kPropWaves), 0, s, a, method);
    if (hipGetLastError() != hipSuccess) return HODE_ELAUNCH;
    if (!a.gnn) return HODE_OK;
    // B: one workgroup of 8 waves per CU
    int blocksB = per_set < 256 ? per_set : 256;
    if (a.n_sets > 1 && blocksB * a.n_sets > 256) blocksB = 256 / a.n_sets;
    if (blocksB < 1) blocksB = 1;
    const size_t lds = accum_lds_bytes<NL>();
    auto kern = solve_bwd_accum_kernel<NL>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return HODE_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(blocksB, a.n_sets), dim3(64 * kAccWaves), lds, s, a, method);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <int NL> int launch_split_nl(hipStream_t s, const AdjArgs<float> &a, int method)
{
    const bool gd = a.gd_mode != 0;
    if (a.gode) return gd ? launch_split_g<NL, true, true>(s, a, method) : launch_split_g<NL, true, false>(s, a, method);
    return gd ? launch_split_g<NL, false, true>(s, a, method) : launch_split_g<NL, false, false>(s, a, method);
}

}  // namespace

int launch_solve_bwd_split(hipStream_t s, const AdjArgs<float> &a, int L, int method)
{
    if (!a.tape_delta) return HODE_EINVAL;
    switch (L) {
    case 2: return launch_split_nl<2>(s, a, method);
    case 3: return launch_split_nl<3>(s, a, method);
    case 4: return launch_split_nl<4>(s, a, method);
    }
    return HODE_EUNSUPPORTED;
}

}  // namespace hode
