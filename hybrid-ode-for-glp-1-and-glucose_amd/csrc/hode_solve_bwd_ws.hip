// hode_solve_bwd_ws.hip -- K4 for the tuned fp32 path: the reverse-time adjoint with WAVE-SPECIALISED workgroups.
//
// No reference counterpart (the reference detaches the solve: models/hybrid_ode_nn.py:186,234-237,248).
// Same mathematics, same tape, same results contract as solve_bwd_kernel (hode_solve_bwd.hip, which stays the fp64 / NL = 1
// kernel); CPU restatement: oracle/hode_oracle_impl.h (hode_oracle_solve_bwd).
//
// Why.  The one-role kernel keeps 192 gradient accumulators in every wave: 256 VGPRs, 2 waves per SIMD, where a DPP FMA costs
// 3.8 cycles of SIMD time (2.9 at 4 waves), and PMC shows its waves parked in s_waitcnt 36 % of the time with no third wave to
// fill in.  The accumulators are sums over ALL stages of ALL trajectories -- they do not have to live where the cotangent is
// propagated.  So a workgroup of 9 + 2 (NL - 1) waves splits the work by ROLE:
//
//   P  8 propagation waves   one trajectory each.  Per stage: record (h_1..h_NL, stage state) by LDS-DMA, kb, mechanistic
//                            J^T, delta_NL .. delta_1 through the transposed matrices (LDS image, rotating-operand order).
//                            No gradient accumulators: ~100 VGPRs.  Publishes delta_1..delta_NL, kb, t, tVNS in an LDS
//                            hand-off slot (1.5 KB, double buffered).
//   A  2 (NL - 1) waves      A(m, g) owns dW of hidden matrix m -- 64 accumulators -- for the P-waves 4 g .. 4 g + 3:
//                            dW_m += delta_{m+1} (x) h_m, 64 v_fmac_f32_dpp per P-wave and stage, operands straight from
//                            LDS (the P-wave's record and hand-off slot).
//   E  1 wave                first / last layer and bias gradients of all 8 P-waves (16 + NL accumulators).
//
// Everything is in lock step: one s_barrier per stage.  In iteration i the P-waves process their stage i while A / E consume
// what was published in iteration i - 1 (hand-off double buffered, record ring of three slots), so nobody polls and every
// wave reaches every barrier: the iteration count is the maximum over the P-waves of their total stage count, + 1.
// All 15 waves fit one CU at 128 VGPRs (4 waves per SIMD); LDS 104 KB.
//
// Determinism: A(m, g) adds its four P-waves' products in a fixed order, the two groups and the eight go-registers are summed
// in a fixed order, the workgroup writes ONE gradient row to a.partials and adj_reduce_kernel adds the rows in workgroup
// order: no floating-point atomics anywhere, the same inputs give the same bits.
#include "hode_device.h"
#include "hode_kernels.h"

namespace hode {

namespace {

constexpr int kWsP = 8;                                    // propagation waves per workgroup
constexpr int kWsRing = 3;                                 // record ring: being DMA'd | being propagated | being accumulated
template <int NL> constexpr int ws_waves() { return kWsP + 2 * (NL - 1) + 1; }
template <int NL> constexpr int ws_rec_elems() { return (NL + 1) * kWave; }          // NL rows + the stage state (8 of 64 used)
template <int NL> constexpr int ws_hand_elems() { return (NL + 1) * kWave; }         // delta_1..delta_NL + {kb[6], t, tvns | valid, slot}
template <int NL> constexpr size_t ws_lds_elems()
{
    return (size_t)(NL - 1) * kMaxH * kMaxH + 8 * kWave + (size_t)kWsP * kWsRing * ws_rec_elems<NL>() +
           (size_t)kWsP * 2 * ws_hand_elems<NL>() + 64;
}

__device__ __forceinline__ float inp_at_w(const float *__restrict__ p, int mode, int b, int T, int k)
{
    if (mode == 0) return 0.f;
    return (mode == 1) ? p[b] : p[(size_t)b * T + k];
}

// delta_prev = W^T delta from the LDS image (hode_device.h: wt_rot_store), reads issued two groups of four ahead of their use:
// the propagation wave has no outer-product FMAs to put between a read and its first use, but it has the registers
template <int G> __device__ __forceinline__ void ws_wt_group(const Vec4<float> (&w)[4], const float (&Rd)[4], float (&acc)[4])
{
    static_assert(G >= 0 && G < 4, "four groups of sixteen rotations");
#define HODE_WS_FM(OP0, a0s)                                                                                                   \
    asm(OP0 " %[a0], %[r], %[w0]" a0s "\n\t"                                                                                   \
        "v_fmac_f32_dpp %[a1], %[r], %[w1] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"                                           \
        "v_fmac_f32_dpp %[a2], %[r], %[w2] row_ror:2 row_mask:0xf bank_mask:0xf\n\t"                                           \
        "v_fmac_f32_dpp %[a3], %[r], %[w3] row_ror:3 row_mask:0xf bank_mask:0xf\n\t"                                           \
        "v_fmac_f32_dpp %[a0], %[r], %[w4] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"                                           \
        "v_fmac_f32_dpp %[a1], %[r], %[w5] row_ror:5 row_mask:0xf bank_mask:0xf\n\t"                                           \
        "v_fmac_f32_dpp %[a2], %[r], %[w6] row_ror:6 row_mask:0xf bank_mask:0xf\n\t"                                           \
        "v_fmac_f32_dpp %[a3], %[r], %[w7] row_ror:7 row_mask:0xf bank_mask:0xf\n\t"                                           \
        "v_fmac_f32_dpp %[a0], %[r], %[w8] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"                                           \
        "v_fmac_f32_dpp %[a1], %[r], %[w9] row_ror:9 row_mask:0xf bank_mask:0xf\n\t"                                           \
        "v_fmac_f32_dpp %[a2], %[r], %[w10] row_ror:10 row_mask:0xf bank_mask:0xf\n\t"                                         \
        "v_fmac_f32_dpp %[a3], %[r], %[w11] row_ror:11 row_mask:0xf bank_mask:0xf\n\t"                                         \
        "v_fmac_f32_dpp %[a0], %[r], %[w12] row_ror:12 row_mask:0xf bank_mask:0xf\n\t"                                         \
        "v_fmac_f32_dpp %[a1], %[r], %[w13] row_ror:13 row_mask:0xf bank_mask:0xf\n\t"                                         \
        "v_fmac_f32_dpp %[a2], %[r], %[w14] row_ror:14 row_mask:0xf bank_mask:0xf\n\t"                                         \
        "v_fmac_f32_dpp %[a3], %[r], %[w15] row_ror:15 row_mask:0xf bank_mask:0xf"                                             \
        : [a0] "+v"(acc[0]), [a1] "+v"(acc[1]), [a2] "+v"(acc[2]), [a3] "+v"(acc[3])                                           \
        : [r] "v"(Rd[G]), [w0] "v"(w[0].v[0]), [w1] "v"(w[0].v[1]), [w2] "v"(w[0].v[2]), [w3] "v"(w[0].v[3]), [w4] "v"(w[1].v[0]), \
          [w5] "v"(w[1].v[1]), [w6] "v"(w[1].v[2]), [w7] "v"(w[1].v[3]), [w8] "v"(w[2].v[0]), [w9] "v"(w[2].v[1]),              \
          [w10] "v"(w[2].v[2]), [w11] "v"(w[2].v[3]), [w12] "v"(w[3].v[0]), [w13] "v"(w[3].v[1]), [w14] "v"(w[3].v[2]),         \
          [w15] "v"(w[3].v[3]))
    HODE_WS_FM("v_fmac_f32", "");
#undef HODE_WS_FM
}
__device__ __forceinline__ float ws_wt_mul(const float *__restrict__ wt, int lane, float d)
{
    const Vec4<float> *wt4 = reinterpret_cast<const Vec4<float> *>(wt);
    float Rd[4];
    rows_replicate(d, Rd);
    // rows 4 G .. 4 G + 3 of the image feed group G (r = 16 G + 4 i + c  ->  q = G, n = 4 i + c)
    Vec4<float> w0[4], w1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[i] = wt4[(0 + i) * kMaxH + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) w1[i] = wt4[(4 + i) * kMaxH + lane];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    ws_wt_group<0>(w0, Rd, acc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[i] = wt4[(8 + i) * kMaxH + lane];
    __builtin_amdgcn_sched_barrier(0);
    ws_wt_group<1>(w1, Rd, acc);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w1[i] = wt4[(12 + i) * kMaxH + lane];
    __builtin_amdgcn_sched_barrier(0);
    ws_wt_group<2>(w0, Rd, acc);
    __builtin_amdgcn_sched_barrier(0);
    ws_wt_group<3>(w1, Rd, acc);
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

}  // namespace

template <int NL, bool GODE, bool GD>
__global__ __launch_bounds__(64 * ws_waves<NL>()) void solve_bwd_ws_kernel(const AdjArgs<float> a, const int method)
{
    using R = float;
    using ES = EdgeSlots<NL>;
    static_assert(NL >= 2 && NL <= 4, "the specialised adjoint needs at least one hidden matrix");
    constexpr int kWaves = ws_waves<NL>();
    constexpr int kA = 2 * (NL - 1);
    constexpr int kRec = ws_rec_elems<NL>();
    constexpr int kHand = ws_hand_elems<NL>();
    constexpr int kSlot = NL * kWave + 8;                 // stage record on the tape: NL rows + 8 reals of stage state
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *wt = reinterpret_cast<R *>(smem_raw);              // [(NL-1)][64*64] transposed hidden matrices, rotating-operand order
    R *rowsT = wt + (size_t)(NL - 1) * kMaxH * kMaxH;     // [8][64] transposed tableau rows
    R *recs = rowsT + 8 * kWave;                          // [kWsP][kWsRing][kRec]
    R *hands = recs + (size_t)kWsP * kWsRing * kRec;      // [kWsP][2][kHand]
    int *niter = reinterpret_cast<int *>(hands + (size_t)kWsP * 2 * kHand);

    const int lane = threadIdx.x & 63;
    const int c8 = lane & 7, grp = lane >> 3, p16 = lane & 15;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    const int set = blockIdx.y;
    const int T = a.T;
    const int per_set = a.B / a.n_sets;
    const int S = kTableau[method].S;
    const R *__restrict__ nn_set = a.nn_p + (size_t)set * a.P;
    const bool isP = wave < kWsP, isA = wave >= kWsP && wave < kWsP + kA;

    wt_rot_store<R>(wt, nn_set, a.H, NL - 1, threadIdx.x, 64 * kWaves);
    tableau_rowsT_store<R>(rowsT, method, threadIdx.x, 64 * kWaves);
    if (threadIdx.x == 0) *niter = 0;
    // hand-off slots start invalid
    for (int i = threadIdx.x; i < kWsP * 2; i += 64 * kWaves) reinterpret_cast<int *>(hands + (size_t)i * kHand + NL * kWave)[8] = 0;
    __syncthreads();
    // iterations = the longest propagation wave's number of stages (+ 1: accumulation runs one iteration behind)
    if (isP) {
        int tot = 0;
        for (int bi = wave * gridDim.x + blockIdx.x; bi < per_set; bi += gridDim.x * kWsP) {
            const int nb = a.nsteps[set * per_set + bi];
            tot += (nb < a.max_steps ? nb : a.max_steps) * S;
        }
        if (lane == 0) atomicMax(niter, tot);
    }
    __syncthreads();
    const int n_iter = *niter + 1;

    // ---- the three roles: each has its OWN loop (its registers are live in its branch only: 64 accumulators here, the
    //      propagation state there), every loop executes the same n_iter barriers -------------------------------------------
    R gw[kMaxH];                                          // A: dW of one hidden matrix (rotating-operand register order)
    R ge[ES::count];                                      // E: first / last layer and bias gradients
    R go = 0.f;                                           // P: lane p < 17 holds d/d(ode constant p)
    if (isP) {
        // edge weights in registers (no accumulators here: there is room), trajectory / step / stage cursors
        R w1[9], w5[6];
        OdeP<R> o;
        R lam = 0.f, ZZ = 0.f;
        int bi_next = wave * gridDim.x + blockIdx.x, b = 0, n = 0, st = -1, s = 0, knext = 0, k = 0, cur = 0;
        bool active = false, ok = true;
        R tc = 0.f, h = 0.f, t0 = 0.f, inv_len = 0.f, v0 = 0.f, dv = 0.f, d0 = 0.f, dd = 0.f;
        const R *__restrict__ tg = nullptr, *__restrict__ tape = nullptr, *__restrict__ stg = nullptr, *__restrict__ gyb = nullptr;
        const int *__restrict__ tseg = nullptr;
        R *rec = recs + (size_t)wave * kWsRing * kRec;
        R *hand = hands + (size_t)wave * 2 * kHand;
        {
            const R live = (lane < a.H) ? 1.f : 0.f;
            const int j = (lane < a.H) ? lane : a.H - 1;
            const R *pout = nn_set + 9 * a.H + a.H + (size_t)(NL - 1) * ((size_t)a.H * a.H + a.H);
#pragma unroll
            for (int i = 0; i < 9; ++i) w1[i] = live * nn_set[j * 9 + i];
#pragma unroll
            for (int q = 0; q < 6; ++q) w5[q] = live * pout[q * a.H + j];
            ode_load(o, a.ode_p + 17 * set);
        }
        auto rec_dma = [&](const R *__restrict__ src, R *dst) {
#pragma unroll
            for (int l = 0; l < NL; ++l)
                __builtin_amdgcn_global_load_lds(src + l * kWave + lane, (__attribute__((address_space(3))) void *)(dst + l * kWave), 4, 0, 0);
            if (lane < 8)
                __builtin_amdgcn_global_load_lds(src + NL * kWave + lane, (__attribute__((address_space(3))) void *)(dst + NL * kWave), 4, 0, 0);
        };
        auto inject = [&](int r) {                     // lam += dLoss/dy[b, r, :]  (six wave-uniform scalar loads)
            const R *__restrict__ gr = gyb + (size_t)r * 6;
            const R g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3], g4 = gr[4], g5 = gr[5];
            lam += (c8 == 0) ? g0 : (c8 == 1) ? g1 : (c8 == 2) ? g2 : (c8 == 3) ? g3 : (c8 == 4) ? g4 : (c8 == 5) ? g5 : 0.f;
        };
        // rows 0..kf of a trajectory are (copies of) x0: their cotangents close the trajectory (see solve_bwd_kernel)
        auto finish_traj = [&]() {
            int kf = 0;
            while (kf + 1 < T && !(tg[kf + 1] > tg[kf])) ++kf;
            for (int r = 0; r <= kf; ++r) inject(r);
            if (lane < 6) a.gx0[(size_t)b * 6 + lane] = lam;
        };
        // next trajectory of this wave that has steps on its tape; trajectories without any are closed on the spot
        auto start_next = [&]() {
            active = false;
            while (bi_next < per_set) {
                b = set * per_set + bi_next;
                bi_next += gridDim.x * kWsP;
                tg = a.t + (a.t_batched ? (size_t)b * T : 0);
                tape = a.tape + (size_t)b * a.max_steps * 8;
                tseg = a.tape_seg + (size_t)b * a.max_steps;
                stg = a.tape_stage + (size_t)b * a.max_steps * 6 * kSlot;
                gyb = a.gy + (size_t)b * T * 6;
                n = a.nsteps[b] < a.max_steps ? a.nsteps[b] : a.max_steps;      // never walk past the tape
                ok = a.status[b] == HODE_ST_OK;
                lam = 0.f;
                knext = T - 1;
                if (n > 0) {
                    st = n - 1;
                    s = S - 1;
                    rec_dma(stg + ((size_t)st * 6 + s) * kSlot, rec + cur * kRec);      // its first record
                    active = true;
                    return;
                }
                finish_traj();
            }
        };
        start_next();
#pragma unroll 1
        for (int it = 0; it < n_iter; ++it) {
            R *__restrict__ hd = hand + (it & 1) * kHand;
            if (active) {
                if (s == S - 1) {
                    // ---- step header: cotangents of the grid rows this step produced, step and interval constants
                    const int kraw = tseg[st];
                    k = kraw & (kSegClosed - 1);
                    int hi = knext;
                    if (st == n - 1) {
                        hi = T - 1;
                        if (!ok) {                 // the last step of a FAILED trajectory: see solve_bwd_kernel
                            hi = k;
                            if (kraw & kSegClosed) {
                                hi = k + 1;
                                while (hi + 1 < T && !(tg[hi + 1] > tg[hi])) ++hi;
                            }
                        }
                    }
                    for (int r = k + 1; r <= hi; ++r) inject(r);
                    knext = k;
                    tc = tape[(size_t)st * 8 + 0];
                    h = tape[(size_t)st * 8 + 1];
                    t0 = tg[k];
                    const R t1 = tg[k + 1];
                    v0 = inp_at_w(a.tvns, a.tvns_mode, b, T, k);
                    const R v1 = inp_at_w(a.tvns, a.tvns_mode, b, T, k + 1);
                    d0 = inp_at_w(a.gd, a.gd_mode, b, T, k);
                    const R d1 = inp_at_w(a.gd, a.gd_mode, b, T, k + 1);
                    inv_len = first_lane(1.f / (t1 - t0));
                    dv = first_lane(v1 - v0);
                    dd = first_lane(d1 - d0);
                    ZZ = 0.f;
                }
                // record (st, s) was DMA'd into ring slot `cur` one iteration ago (or at the start of the trajectory)
                __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0)
                __builtin_amdgcn_wave_barrier();
                const int slot = cur;
                cur = (cur + 1 == kWsRing) ? 0 : cur + 1;
                {
                    // the next record of this trajectory goes to the slot behind: the one the accumulation waves read in the
                    // PREVIOUS iteration (they are done with it: a barrier lies in between)
                    const int ns_ = (s > 0) ? s - 1 : S - 1, nst = (s > 0) ? st : st - 1;
                    if (nst >= 0) rec_dma(stg + ((size_t)nst * 6 + ns_) * kSlot, rec + cur * kRec);
                }
                const R *__restrict__ rc = rec + slot * kRec;
                R hact[NL];
#pragma unroll
                for (int l = 0; l < NL; ++l) hact[l] = rc[l * kWave + lane];
                const R Ys = rc[NL * kWave + c8];             // stage state, replicated layout
                const R bw_s = rowsT[6 * kWave + s], c_s = rowsT[6 * kWave + 8 + s];
                const R kb = h * rfma(bw_s, lam, group_sum8(rowsT[s * kWave + lane] * ZZ));
                const R ts = rfma(c_s, h, tc);
                const R al = (ts - t0) * inv_len;
                const R gdv = rfma(al, dd, d0);
                const R tv = rfma(al, dv, v0);
                R gde = 0.f;
                if constexpr (GD) gde = gd_effect(o, gdv);
                // ---- J^T kb: mechanistic part, then the cotangent through the layers; every delta goes to the hand-off slot
                const R G = lane_bcast(Ys, 0), I = lane_bcast(Ys, 1), Glu = lane_bcast(Ys, 2), GLP1 = lane_bcast(Ys, 3),
                        FFA = lane_bcast(Ys, 5);
                const R lG = lane_bcast(kb, 0), lI = lane_bcast(kb, 1), lGlu = lane_bcast(kb, 2), lGLP = lane_bcast(kb, 3),
                        lGE = lane_bcast(kb, 4), lF = lane_bcast(kb, 5);
                const R mech = mech_vjp<R, GODE>(o, G, I, Glu, GLP1, FFA, lG, lI, lGlu, lGLP, lF, gde, gdv, GD, lane, go);
                R d = w5[0] * lG;
                d = rfma(w5[1], lI, d);
                d = rfma(w5[2], lGlu, d);
                d = rfma(w5[3], lGLP, d);
                d = rfma(w5[4], lGE, d);
                d = rfma(w5[5], lF, d);
                d = (hact[NL - 1] > 0.f) ? d : 0.f;
                hd[(NL - 1) * kWave + lane] = d;               // delta_NL
#pragma unroll
                for (int l = NL - 1; l >= 1; --l) {            // hidden matrix l-1 maps h_l -> h_{l+1}
                    const R dp = ws_wt_mul(wt + (size_t)(l - 1) * kMaxH * kMaxH, lane, d);
                    d = (hact[l - 1] > 0.f) ? dp : 0.f;
                    hd[(l - 1) * kWave + lane] = d;            // delta_l
                }
                if (lane < 8) hd[NL * kWave + lane] = (lane < 6) ? kb : (lane == 6) ? ts : tv;
                if (lane == 8) reinterpret_cast<int *>(hd + NL * kWave)[8] = 1 + slot;          // valid, and which ring slot
                R p[6];
                p[0] = w1[1] * d;
                p[1] = w1[2] * d;
                p[2] = w1[3] * d;
                p[3] = (w1[4] + w1[7]) * d;                    // GLP1 feeds inputs 4 and 7
                p[4] = w1[5] * d;
                p[5] = w1[6] * d;
                const R nnv = wave_reduce6_to_lanes(p, lane);
                const R Z = (c8 < 6) ? (mech + nnv) : 0.f;
                ZZ = (grp == s) ? Z : ZZ;
                if (s == 0) {
                    lam += group_sum8(rowsT[7 * kWave + lane] * ZZ);
                    s = S - 1;
                    if (--st < 0) {
                        finish_traj();
                        start_next();
                    }
                } else {
                    --s;
                }
            } else {
                if (lane == 8) reinterpret_cast<int *>(hd + NL * kWave)[8] = 0;
            }
            __syncthreads();
        }
    } else if (isA) {
        const int m = (wave - kWsP) >> 1, g = (wave - kWsP) & 1;
#pragma unroll
        for (int r = 0; r < kMaxH; ++r) gw[r] = 0.f;
#pragma unroll 1
        for (int it = 0; it < n_iter; ++it) {
            if (it > 0 && a.gnn != nullptr) {
                const int rp = (it - 1) & 1;                   // what the propagation waves published one iteration ago
#pragma unroll 1
                for (int q = 0; q < 4; ++q) {
                    const int pw = 4 * g + q;
                    const R *__restrict__ hd = hands + ((size_t)pw * 2 + rp) * kHand;
                    const int tag = first_lane(reinterpret_cast<const int *>(hd + NL * kWave)[8]);
                    if (tag == 0) continue;
                    const R *__restrict__ hr = recs + ((size_t)pw * kWsRing + (tag - 1)) * kRec + m * kWave;      // h_m: input of matrix m
                    float Rh[4];
                    Rh[0] = hr[p16]; Rh[1] = hr[16 + p16]; Rh[2] = hr[32 + p16]; Rh[3] = hr[48 + p16];
                    const R d = hd[(m + 1) * kWave + lane];                                                     // delta_{m+1}
                    asm volatile("" : "+v"(Rh[0]), "+v"(Rh[1]), "+v"(Rh[2]), "+v"(Rh[3]));
                    mlp_outer_step<0>(gw, d, Rh);
                }
            }
            __syncthreads();
        }
    } else {
#pragma unroll
        for (int i = 0; i < ES::count; ++i) ge[i] = 0.f;
#pragma unroll 1
        for (int it = 0; it < n_iter; ++it) {
            if (it > 0 && a.gnn != nullptr) {
                const int rp = (it - 1) & 1;
#pragma unroll 1
                for (int pw = 0; pw < kWsP; ++pw) {
                    const R *__restrict__ hd = hands + ((size_t)pw * 2 + rp) * kHand;
                    const int tag = first_lane(reinterpret_cast<const int *>(hd + NL * kWave)[8]);
                    if (tag == 0) continue;
                    const R *__restrict__ rc = recs + ((size_t)pw * kWsRing + (tag - 1)) * kRec;
                    const R *__restrict__ xs = rc + NL * kWave, *__restrict__ tl = hd + NL * kWave;         // state | kb[6], t, tvns
                    const R d1 = hd[lane];
                    ge[ES::b + 0] += d1;
#pragma unroll
                    for (int l = 1; l < NL; ++l) ge[ES::b + l] += hd[l * kWave + lane];
                    // first layer: input row [t, G, I, Glu, GLP1, GE, FFA, glp1 := GLP1, tvns] (broadcast LDS reads)
                    ge[ES::w1 + 0] = rfma(d1, tl[6], ge[ES::w1 + 0]);
                    ge[ES::w1 + 1] = rfma(d1, xs[0], ge[ES::w1 + 1]);
                    ge[ES::w1 + 2] = rfma(d1, xs[1], ge[ES::w1 + 2]);
                    ge[ES::w1 + 3] = rfma(d1, xs[2], ge[ES::w1 + 3]);
                    ge[ES::w1 + 4] = rfma(d1, xs[3], ge[ES::w1 + 4]);
                    ge[ES::w1 + 5] = rfma(d1, xs[4], ge[ES::w1 + 5]);
                    ge[ES::w1 + 6] = rfma(d1, xs[5], ge[ES::w1 + 6]);
                    ge[ES::w1 + 7] = rfma(d1, xs[3], ge[ES::w1 + 7]);
                    ge[ES::w1 + 8] = rfma(d1, tl[7], ge[ES::w1 + 8]);
                    // output layer: dWout[q][j] += kb_q h_NL[j], dbout[q] += kb_q
                    const R hl = rc[(NL - 1) * kWave + lane];
#pragma unroll
                    for (int q = 0; q < 6; ++q) ge[ES::w5 + q] = rfma(tl[q], hl, ge[ES::w5 + q]);
                    ge[ES::b5] += tl[c8];                      // lanes 0..5 hold dbout (slots 6, 7 carry t / tVNS: never stored)
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: ONE gradient row per workgroup (see the header; same row format as solve_bwd_kernel) -----------------------
    const int nthreads = 64 * kWaves;
    const size_t rowlen = adj_partial_rowlen(a.P);
    R *__restrict__ prow = a.partials + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * rowlen;
    const int H = a.H;
    if (a.gnn) {
        constexpr int kHid = (NL - 1) * kMaxH * kMaxH;
        // the image of the transposed matrices is dead: it becomes the [matrix][row][col] sum of the two accumulation groups
        for (int g = 0; g < 2; ++g) {
            if (isA && ((wave - kWsP) & 1) == g) {
                const int m = (wave - kWsP) >> 1;
#pragma unroll
                for (int r = 0; r < kMaxH; ++r) {
                    R *dst = wt + (size_t)m * kMaxH * kMaxH + lane * kMaxH + wcol<R>(r, lane);
                    *dst = (g == 0) ? gw[r] : *dst + gw[r];
                }
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < kHid; i += nthreads) {
            const int l = i >> 12, row = (i >> 6) & 63, col = i & 63;
            if (row < H && col < H) prow[9 * H + H + (size_t)l * ((size_t)H * H + H) + (size_t)row * H + col] = wt[i];
        }
        if (!isP && !isA && lane < H) {
            const size_t off_out = (size_t)9 * H + H + (size_t)(NL - 1) * ((size_t)H * H + H);
#pragma unroll
            for (int i = 0; i < 9; ++i) prow[lane * 9 + i] = ge[ES::w1 + i];
            prow[9 * H + lane] = ge[ES::b + 0];
#pragma unroll
            for (int l = 1; l < NL; ++l) prow[9 * H + H + (size_t)(l - 1) * ((size_t)H * H + H) + (size_t)H * H + lane] = ge[ES::b + l];
#pragma unroll
            for (int q = 0; q < 6; ++q) prow[off_out + q * H + lane] = ge[ES::w5 + q];
            if (lane < 6) prow[off_out + 6 * H + lane] = ge[ES::b5];
        }
    }
    if constexpr (GODE) {
        if (a.gode) {
            __syncthreads();
            if (isP && lane < 17) wt[wave * 32 + lane] = go;
            __syncthreads();
            if (threadIdx.x < 17) {
                R v = 0.f;
                for (int w = 0; w < kWsP; ++w) v += wt[w * 32 + threadIdx.x];
                prow[a.P + threadIdx.x] = v;
            }
        }
    }
}

template <int NL, bool GODE, bool GD> static int launch_ws_g(hipStream_t s, const AdjArgs<float> &a, int method, int cus)
{
    const int per_set = a.B / a.n_sets;
    int blocks = per_set < cus ? per_set : cus;           // one workgroup per CU; its propagation waves loop over trajectories
    if (a.n_sets > 1 && blocks * a.n_sets > cus) blocks = cus / a.n_sets;
    if (blocks < 1) blocks = 1;
    if (a.partials == nullptr || blocks * a.n_sets > a.partial_rows) return HODE_EUNSUPPORTED;     // caller falls back
    const size_t lds = ws_lds_elems<NL>() * sizeof(float);
    auto kern = solve_bwd_ws_kernel<NL, GODE, GD>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return HODE_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(blocks, a.n_sets), dim3(64 * ws_waves<NL>()), lds, s, a, method);
    if (a.gnn || (GODE && a.gode))
        launch_adj_reduce(s, a.partials, (int)adj_partial_rowlen(a.P), blocks, a.n_sets, a.P, a.gnn, GODE ? a.gode : nullptr);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <int NL> static int launch_ws_nl(hipStream_t s, const AdjArgs<float> &a, int method, int cus)
{
    const bool gd = a.gd_mode != 0;
    if (a.gode) return gd ? launch_ws_g<NL, true, true>(s, a, method, cus) : launch_ws_g<NL, true, false>(s, a, method, cus);
    return gd ? launch_ws_g<NL, false, true>(s, a, method, cus) : launch_ws_g<NL, false, false>(s, a, method, cus);
}

// HODE_EUNSUPPORTED: not a shape / launch this kernel takes (NL = 1, no partial rows) -- the caller runs solve_bwd_kernel
int launch_solve_bwd_ws(hipStream_t s, const AdjArgs<float> &a, int L, int method, int cus)
{
    switch (L) {
    case 2: return launch_ws_nl<2>(s, a, method, cus);
    case 3: return launch_ws_nl<3>(s, a, method, cus);
    case 4: return launch_ws_nl<4>(s, a, method, cus);
    }
    return HODE_EUNSUPPORTED;
}

}  // namespace hode
