// hode_solve_bwd_ws.hip -- K4 for the tuned fp32 path: the reverse-time adjoint with WAVE-SPECIALISED workgroups.
//
// No reference counterpart (the reference detaches the solve: models/hybrid_ode_nn.py:186,234-237,248).
// Same mathematics, same tape, same results contract as solve_bwd_kernel (hode_solve_bwd.hip, which stays the fp64 / NL = 1
// kernel); CPU restatement: oracle/hode_oracle_impl.h (hode_oracle_solve_bwd).
//
// Why.  The one-role kernel keeps 192 gradient accumulators in every wave: 256 VGPRs, 2 waves per SIMD, where a DPP FMA costs
// 3.8 cycles of SIMD time (2.9 at 4 waves), and PMC shows its waves parked in s_waitcnt 36 % of the time with no third wave to
// fill in.  The accumulators are sums over ALL stages of ALL trajectories -- they do not have to live where the cotangent is
// propagated.  A workgroup of 16 waves (one per CU, 4 per SIMD, <= 128 VGPRs) splits the work by ROLE:
//
//   P  8 propagation waves   U (1 or 2) trajectories each, interleaved instruction by instruction (two independent chains fill each
//                            other's gaps and share every 16-byte read of the transposed matrices).  Per stage and trajectory: the
//                            stage record (h_1..h_NL, stage state) by LDS-DMA into a ring of three slots; the STEP HEADER -- the
//                            32-byte tape entry {t, h, t0, 1/len, v0, dv, d0, dd}, up to six cotangent rows gy and the interval word,
//                            gathered by ONE LDS-DMA with per-lane global addresses one step ahead (hdr_dma -> hdrs[][2][64]; the
//                            slow `inject` path only for a failed first step or more than six rows at one step); kb; mechanistic
//                            J^T; delta_NL .. delta_1 through the transposed matrices (LDS image in row-block order: one
//                            v_mov_b32_dpp / 16-byte broadcast read of delta + two v_pk_fma_f32 per rotation).  No gradient
//                            accumulators.  Publishes delta_1..delta_NL, kb, t, tVNS in an LDS hand-off slot (1.25 KB per
//                            trajectory, double buffered).
//   A  8 accumulation waves  A_j owns dW of hidden matrix j % (NL-1) -- 64 accumulators in natural column order -- for its share of
//                            the 8 U trajectory slots: dW_m += delta_{m+1} (x) h_m with the 16 values of h_m's row as four 16-byte
//                            LDS broadcast reads and 32 v_pk_fma_f32 per slot and stage (ws_outer_nat; round 3: 15 v_mov_b32_dpp
//                            instead of the reads -- the LDS pipe has room, the vector pipe does not), + the layer's bias; the
//                            first / last layer gradients (17 accumulators) belong to the waves of ws_edge_owner -- the six that
//                            serve the fewest matrix slots.
//
// LDS reads of regions an LDS-DMA may be writing are inline asm (ws_lds_read128 + ws_lds_wait*): in front of a compiler-visible read
// of such a region hipcc waits for EVERY outstanding vector-memory operation (s_waitcnt vmcnt(0)) -- the record DMA issued a moment
// ago included (+1 000 cycles per stage in the mechanistic phase).
//
// Everything is in lock step: one s_barrier per stage.  In iteration i the P-waves process stage i of their trajectories while
// the A-waves consume what was published in iteration i - 1 (hand-off double buffered, record ring of three slots), so nobody
// polls and every wave reaches every barrier: the iteration count is the maximum over the slots of their total stage count, + 1.
// With U = 2 a 4 096-trajectory batch is ONE round of 16 trajectories per CU.  What a stage waits for is the propagation chain
// (P-waves alone: 5.1 of the 5.6-5.8 ms).
//
// Determinism: every A-wave adds its slots' products in slot order, the waves of a matrix are summed in rank order, the workgroup
// writes ONE gradient row to a.partials and adj_reduce_kernel adds the rows in workgroup order: no floating-point atomics
// anywhere, the same inputs give the same bits.
#include "hode_device.h"
#include "hode_kernels.h"
#include <cstdlib>

namespace hode {

// 1: the propagation waves read the lane's 16-lane row of delta back from the hand-off slot (four 16-byte broadcast reads) and take
//    the operands of the packed FMAs from there (ws_wt_group_lb); 0: round 3's form, one v_mov_b32_dpp row_ror:n per rotation
#ifndef HODE_WS_PROP_LB
#define HODE_WS_PROP_LB 1
#endif

namespace {

constexpr int kWsP = 8;                                    // propagation waves per workgroup
constexpr int kWsA = 8;                                    // accumulation waves per workgroup
constexpr int kWsWaves = kWsP + kWsA;
constexpr int kWsRing = 3;                                 // record ring: being DMA'd | being propagated | being accumulated
#ifdef HODE_LAB
// lab library only (HODE_WS_DBG bit 1024): shader-clock stamps of workgroup (0, 0) over 32 iterations, [iteration][wave][point]
constexpr int kWsTraceIt0 = 200, kWsTraceIts = 32, kWsTracePts = 10;
__device__ unsigned long long g_ws_trace[kWsTraceIts * kWsWaves * kWsTracePts];
#define WS_STAMP(i) if (trace_on) tstamp[i] = __builtin_amdgcn_s_memtime()
#define WS_TRACE_DECL unsigned long long tstamp[kWsTracePts] = {}; const bool trace_wg = (dbg & 1024) && blockIdx.x == 0 && blockIdx.y == 0; bool trace_on = false
#define WS_TRACE_BEGIN(it) trace_on = trace_wg && (it) >= kWsTraceIt0 && (it) < kWsTraceIt0 + kWsTraceIts
#define WS_TRACE_FLUSH(it, wave) if (trace_on && lane == 0) { _Pragma("unroll") for (int i_ = 0; i_ < kWsTracePts; ++i_) g_ws_trace[(((it) - kWsTraceIt0) * kWsWaves + (wave)) * kWsTracePts + i_] = tstamp[i_]; }
#else
#define WS_STAMP(i)
#define WS_TRACE_DECL
#define WS_TRACE_BEGIN(it)
#define WS_TRACE_FLUSH(it, wave)
#endif
template <int NL> constexpr int ws_rec_elems() { return (NL + 1) * kWave; }          // NL rows + the stage state (8 of 64 used)
template <int NL> constexpr int ws_hand_elems() { return (NL + 1) * kWave; }         // delta_1..delta_NL + {kb[6], t, tvns} x 8
template <int NL, int U> constexpr size_t ws_lds_elems()
{
    return (size_t)(NL - 1) * kMaxH * kMaxH + 8 * kWave + (size_t)kWsP * U * kWsRing * ws_rec_elems<NL>() +
           (size_t)kWsP * U * 2 * ws_hand_elems<NL>() + 2 * 16 + 16 + (size_t)kWsP * U * 2 * kWave + 32 + 6 * kWave;
}

// J_mech^T kb as mech_vjp (hode_device.h) computes it, with the five terms evaluated on ALL lanes and selected by lane -- the
// propagation waves have the registers for it, and the exec-masked regions hipcc builds out of a nested ?: cost a
// v_cmp / s_and_saveexec / s_cbranch_execz round trip per term on a wave whose time is its instruction count
template <bool GODE>
__device__ __forceinline__ float ws_mech_vjp(const OdeP<float> &o, float G, float I, float Glu, float GLP1, float FFA, float lG, float lI,
                                             float lGlu, float lGLP, float lF, float gde, float gd_in, bool use_gd, int lane, float &go)
{
    if constexpr (GODE) return mech_vjp<float, true>(o, G, I, Glu, GLP1, FFA, lG, lI, lGlu, lGLP, lF, gde, gd_in, use_gd, lane, go);
    const float Pi = 1.f + o.rho * GLP1;
    const float den1 = o.EC_50 + GLP1, den2 = o.K_m + G;
    const float k_GE = o.k_GE0 * (1.f - gde);
    const float r1 = rdiv(1.f, den1), r2 = rdiv(1.f, den2);
    const float oG = -k_GE * lG + Pi * o.a_GI * lI + o.V_max * o.K_m * r2 * r2 * lGLP + o.p_9 * FFA * lF;
    const float oI = -0.01f * lG - o.k_I * lI - o.p_8 * FFA * lF;
    const float oGlu = 0.005f * lG - o.E_max * GLP1 * r1 * lGlu;
    const float oGLP = o.rho * o.a_GI * (G - o.G_b) * lI - o.E_max * o.EC_50 * r1 * r1 * (Glu - o.Glu_b) * lGlu - o.k_L * lGLP;
    const float oF = (-o.p_7 - o.p_8 * I + o.p_9 * G) * lF;
    const int c8 = lane & 7;
    float r = keep_term(c8 == 0, oG, 0.f);
    r = keep_term(c8 == 1, oI, r);
    r = keep_term(c8 == 2, oGlu, r);
    r = keep_term(c8 == 3, oGLP, r);
    r = keep_term(c8 == 5, oF, r);
    return r;
}

// delta_prev = W^T delta in the ROW-BLOCK order of the forward's hidden layer (hode_device.h mlp_hidden_blk), the transposed matrix
// in an LDS image of its own (wt_blk_store below):
//     img[l][n][lane (r, i)] = { W_l[16 r + ((i - n) & 15)][16 w + i] : w = 0..3 }            n = 0..15
// so delta_{l+1} in its NATURAL layout (unit per lane) is the DPP operand as it is -- no row replication (rows_replicate: 9
// instructions per matrix and trajectory) --, the 16-byte word of rotation n holds the two weight PAIRS of the four accumulators, and
// a rotation is one v_mov_b32_dpp + two v_pk_fma_f32 (both halves take the low half of the moved operand).  The four accumulators
// are added over the rows and transposed by the forward's 2 + 1 swaps and 3 adds.  55 vector instructions per matrix and trajectory
// instead of 76 (rotating-operand order of wt_rot_store: 64 v_fmac_f32_dpp, one per weight); the reads are issued two groups of four
// rotations ahead of their use as before: the propagation wave has no outer-product FMAs to put between a read and its first use.
__device__ __forceinline__ void wt_blk_store(float *__restrict__ wt, const float *__restrict__ nn_p, int H, int NLm1, int tid, int nthreads)
{
    const float *Wl = nn_p + 9 * H + H;
    for (int l = 0; l < NLm1; ++l) {
        for (int e = tid; e < kMaxH * kMaxH; e += nthreads) {
            const int w = ((e & 1) << 1) | ((e >> 1) & 1), lane = (e >> 2) & 63, n = e >> 8;      // word = { w0, w2, w1, w3 }: the pairs (a0, a2), (a1, a3)
            const int i = lane & 15, r = lane >> 4;
#if HODE_WS_PROP_LB
            const int row = 16 * r + n, col = 16 * w + i;                                          // natural column order (ws_wt_group_lb)
#else
            const int row = 16 * r + ((i - n) & 15), col = 16 * w + i;
#endif
            wt[(size_t)l * kMaxH * kMaxH + e] = (row < H && col < H) ? Wl[(size_t)row * H + col] : 0.f;
        }
        Wl += (size_t)H * H + H;
    }
}

// rotations 4 G .. 4 G + 3 (the words w[0..3]) of one matrix for one trajectory: d = delta in the natural layout
template <int G> __device__ __forceinline__ void ws_wt_group(const Vec4<float> (&w)[4], const float d, f2_t &a01, f2_t &a23)
{
    static_assert(G >= 0 && G < 4, "four groups of four rotations");
#define HODE_WS_STEP(I, N)                                                                                                      \
    {                                                                                                                           \
        float lo;                                                                                                               \
        asm("v_mov_b32_dpp %0, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "=v"(lo) : "v"(d));                              \
        f2_t hr;                                                                                                                \
        hr.x = lo;                                                                                                              \
        const f2_t w01 = {w[I].v[0], w[I].v[1]}, w23 = {w[I].v[2], w[I].v[3]};                                                  \
        asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]"                   \
            : "+v"(a01), "+v"(a23) : "v"(w01), "v"(w23), "v"(hr));                                                              \
    }
    if constexpr (G == 0) {
        // rotation 0 is the lane's own delta; products start the sums (and are the wait states the first DPP read of d needs)
        const f2_t w01 = {w[0].v[0], w[0].v[1]}, w23 = {w[0].v[2], w[0].v[3]};
        f2_t hh;
        hh.x = d;
        asm("v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %3, %4 op_sel_hi:[1,0]" : "=&v"(a01), "=&v"(a23) : "v"(w01), "v"(w23), "v"(hh));
        HODE_WS_STEP(1, 1) HODE_WS_STEP(2, 2) HODE_WS_STEP(3, 3)
    } else if constexpr (G == 1) {
        HODE_WS_STEP(0, 4) HODE_WS_STEP(1, 5) HODE_WS_STEP(2, 6) HODE_WS_STEP(3, 7)
    } else if constexpr (G == 2) {
        HODE_WS_STEP(0, 8) HODE_WS_STEP(1, 9) HODE_WS_STEP(2, 10) HODE_WS_STEP(3, 11)
    } else {
        HODE_WS_STEP(0, 12) HODE_WS_STEP(1, 13) HODE_WS_STEP(2, 14) HODE_WS_STEP(3, 15)
    }
#undef HODE_WS_STEP
}
// Columns 4 G .. 4 G + 3 (the words w[0..3]: {W_l[16 r + c][16 w + i] : w}, natural column order) of one matrix for one trajectory;
// q = delta[16 r + 4 G .. + 3], the quarter of the lane's row that the wave read back from the hand-off slot: the packed FMAs pick
// the low / high half of a loaded register pair (op_sel) -- no cross-lane instruction.  15 v_mov_b32_dpp fewer per matrix and
// trajectory than ws_wt_group; same products per accumulator in column order instead of rotation order.
typedef float ws_f4_t __attribute__((ext_vector_type(4)));
template <int G> __device__ __forceinline__ void ws_wt_group_lb(const Vec4<float> (&w)[4], const ws_f4_t q, f2_t &a01, f2_t &a23)
{
    const f2_t q01 = {q.x, q.y}, q23 = {q.z, q.w};
#define HODE_WS_LO(I, Q)                                                                                                        \
    {                                                                                                                           \
        const f2_t w01 = {w[I].v[0], w[I].v[1]}, w23 = {w[I].v[2], w[I].v[3]};                                                  \
        asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]"                   \
            : "+v"(a01), "+v"(a23) : "v"(w01), "v"(w23), "v"(Q));                                                               \
    }
#define HODE_WS_HI(I, Q)                                                                                                        \
    {                                                                                                                           \
        const f2_t w01 = {w[I].v[0], w[I].v[1]}, w23 = {w[I].v[2], w[I].v[3]};                                                  \
        asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]" \
            : "+v"(a01), "+v"(a23) : "v"(w01), "v"(w23), "v"(Q));                                                               \
    }
    if constexpr (G == 0) {
        const f2_t w01 = {w[0].v[0], w[0].v[1]}, w23 = {w[0].v[2], w[0].v[3]};
        asm("v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %3, %4 op_sel_hi:[1,0]" : "=&v"(a01), "=&v"(a23) : "v"(w01), "v"(w23), "v"(q01));
    } else {
        HODE_WS_LO(0, q01)
    }
    HODE_WS_HI(1, q01) HODE_WS_LO(2, q23) HODE_WS_HI(3, q23)
#undef HODE_WS_LO
#undef HODE_WS_HI
}
// one 16-byte LDS read as asm: hipcc puts an s_waitcnt vmcnt(0) in front of every LDS read it can see that may alias a pending
// LDS-DMA (the next record, issued at the top of the iteration), and the hand-off slots sit in the same allocation
// (the value stays in the ONE register tuple the instruction writes until ws_lds_wait has run: any copy hipcc makes of it before
//  that -- e.g. to repack it into a struct -- would read registers the LDS has not written yet)
__device__ __forceinline__ ws_f4_t ws_lds_read128(unsigned addr)
{
    ws_f4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void ws_lds_wait(ws_f4_t &a, ws_f4_t &b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void ws_lds_wait(ws_f4_t &a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)); }
// ... with N younger LDS operations allowed to stay in flight (a wave's LDS operations complete in issue order)
template <int N> __device__ __forceinline__ void ws_lds_wait_n(ws_f4_t &a, ws_f4_t &b)
{
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int N> __device__ __forceinline__ void ws_lds_wait_n(ws_f4_t &a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
// `drow[u]`: LDS byte address of delta_{l+1}[16 r] of trajectory u (the lane's row of the vector the wave has just written to its
// hand-off slot: LDS operations of one wave complete in order).  Quarters are read one group ahead, the weight words two, as before.
template <int U>
__device__ __forceinline__ void ws_wt_mul_lb(const float *__restrict__ wt, const float *__restrict__ wt_next, int lane, const unsigned (&drow)[U],
                                             float (&out)[U], Vec4<float> (&w0)[4], Vec4<float> (&w1)[4])
{
    static_assert(U == 1 || U == 2, "");
    const Vec4<float> *wt4 = reinterpret_cast<const Vec4<float> *>(wt), *nx4 = reinterpret_cast<const Vec4<float> *>(wt_next);
    f2_t a01[U], a23[U];
    ws_f4_t qa[U], qb[U];
    auto wait_all = [](ws_f4_t (&q)[U]) { if constexpr (U == 2) ws_lds_wait(q[0], q[1]); else ws_lds_wait(q[0]); };
    // what is issued BEHIND a quarter and may stay in flight when it is used: four weight words (+ the U quarter reads behind them)
    auto wait_behind_w_q = [](ws_f4_t (&q)[U]) { if constexpr (U == 2) ws_lds_wait_n<4 + U>(q[0], q[1]); else ws_lds_wait_n<4 + U>(q[0]); };
    auto wait_behind_w = [](ws_f4_t (&q)[U]) { if constexpr (U == 2) ws_lds_wait_n<4>(q[0], q[1]); else ws_lds_wait_n<4>(q[0]); };
#pragma unroll
    for (int u = 0; u < U; ++u) qa[u] = ws_lds_read128(drow[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) qb[u] = ws_lds_read128(drow[u] + 16);
    wait_all(qa);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) ws_wt_group_lb<0>(w0, qa[u], a01[u], a23[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[i] = wt4[(8 + i) * kMaxH + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) qa[u] = ws_lds_read128(drow[u] + 32);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) ws_wt_group_lb<1>(w1, qb[u], a01[u], a23[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w1[i] = wt4[(12 + i) * kMaxH + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) qb[u] = ws_lds_read128(drow[u] + 48);
    wait_behind_w_q(qa);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) ws_wt_group_lb<2>(w0, qa[u], a01[u], a23[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[i] = nx4[(0 + i) * kMaxH + lane];
    wait_behind_w(qb);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) ws_wt_group_lb<3>(w1, qb[u], a01[u], a23[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w1[i] = nx4[(4 + i) * kMaxH + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) out[u] = blk_rows_finish<false, false>(a01[u], a23[u], 0.f);
}

// the four accumulators -> (W^T delta)[unit of the lane]: hode_device.h blk_rows_finish (pairs (a0, a2), (a1, a3); no bias)
__device__ __forceinline__ float ws_wt_finish(const f2_t a02, const f2_t a13) { return blk_rows_finish<false, false>(a02, a13, 0.f); }
// The sixteen 16-byte reads of a matrix are issued ahead of their use: on entry w0 / w1 hold the words of groups 0 and 1 (loaded
// while the PREVIOUS matrix -- or, for the first matrix of a stage, the previous iteration's tail -- was being worked on), groups
// 2 and 3 follow into the buffer the group before them has freed, and on exit w0 / w1 hold groups 0 and 1 of `wt_next`: the
// propagation wave is one long dependent chain, an LDS round trip per matrix is 3 x ~150 cycles of it.
__device__ __forceinline__ void ws_wt_preload(const float *__restrict__ wt, int lane, Vec4<float> (&w0)[4], Vec4<float> (&w1)[4])
{
    const Vec4<float> *wt4 = reinterpret_cast<const Vec4<float> *>(wt);
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[i] = wt4[(0 + i) * kMaxH + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) w1[i] = wt4[(4 + i) * kMaxH + lane];
}
// U trajectories: the same sixteen 16-byte reads feed all of them (half the LDS traffic per trajectory at U = 2), and their
// accumulator sets are independent instruction chains
template <int U>
__device__ __forceinline__ void ws_wt_mul(const float *__restrict__ wt, const float *__restrict__ wt_next, int lane, const float (&d)[U],
                                          float (&out)[U], Vec4<float> (&w0)[4], Vec4<float> (&w1)[4])
{
    const Vec4<float> *wt4 = reinterpret_cast<const Vec4<float> *>(wt), *nx4 = reinterpret_cast<const Vec4<float> *>(wt_next);
    f2_t a01[U], a23[U];                            // started by group 0
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) ws_wt_group<0>(w0, d[u], a01[u], a23[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[i] = wt4[(8 + i) * kMaxH + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) ws_wt_group<1>(w1, d[u], a01[u], a23[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w1[i] = wt4[(12 + i) * kMaxH + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) ws_wt_group<2>(w0, d[u], a01[u], a23[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w0[i] = nx4[(0 + i) * kMaxH + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) ws_wt_group<3>(w1, d[u], a01[u], a23[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) w1[i] = nx4[(4 + i) * kMaxH + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) out[u] = ws_wt_finish(a01[u], a23[u]);
}

}  // namespace

// gw[2 c], gw[2 c + 1] += (D01, D23) * h[16 r + c], c = 0..15: the sixteen columns of one outer-product update (see ws_acc_slot).
// The lane's 16-lane row of h arrives as four 16-byte LDS reads (q[0..3]: the sixteen lanes of a row read the same 64 bytes --
// a broadcast, conflict-free); each packed FMA takes the low or the high half of one of the loaded register pairs (op_sel), so no
// cross-lane move is left on the vector pipe: 32 v_pk_fma_f32 per update where round 3 issued 15 v_mov_b32_dpp + 32
// (tools/ubench/lb_ubench.hip: 107 against 153 SIMD cycles per update at four waves per SIMD).
__device__ __forceinline__ void ws_outer_nat(f2_t (&gw)[kMaxH / 2], const f2_t D01, const f2_t D23, const Vec4<float> (&q)[4])
{
#define HODE_WS_OPAIR(K, J)                                                                                                    \
    {                                                                                                                           \
        const f2_t hp = {q[K].v[2 * J], q[K].v[2 * J + 1]};                                                                     \
        asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]"                   \
            : "+v"(gw[2 * (4 * K + 2 * J)]), "+v"(gw[2 * (4 * K + 2 * J) + 1]) : "v"(D01), "v"(D23), "v"(hp));                   \
        asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]" \
            : "+v"(gw[2 * (4 * K + 2 * J + 1)]), "+v"(gw[2 * (4 * K + 2 * J + 1) + 1]) : "v"(D01), "v"(D23), "v"(hp));           \
    }
    HODE_WS_OPAIR(0, 0) HODE_WS_OPAIR(0, 1) HODE_WS_OPAIR(1, 0) HODE_WS_OPAIR(1, 1)
    HODE_WS_OPAIR(2, 0) HODE_WS_OPAIR(2, 1) HODE_WS_OPAIR(3, 0) HODE_WS_OPAIR(3, 1)
#undef HODE_WS_OPAIR
}

// which accumulation wave keeps the first / last layer gradients of trajectory slot t: the waves of the hidden matrices that have
// THREE waves (five or six slots each) share the sixteen slots 2 / 2 / 3 / 3 / 3 / 3, the two waves of the last matrix (eight slots
// each) none -- round 3 gave four slots each to waves 4..7, and wave 5 (eight matrix slots + four edge slots) closed every iteration
// 3 000 cycles behind the lightest wave (tools/ws_trace.py)
__host__ __device__ constexpr int ws_edge_owner(int t)
{
    constexpr int tab[16] = {0, 1, 3, 4, 6, 7, 3, 4, 6, 7, 3, 4, 6, 7, 0, 1};
    return tab[t & 15];
}
__host__ __device__ constexpr bool ws_has_edges(int aj) { return aj != 2 && aj != 5; }

// Accumulation wave AJ: matrix AJ % (NL-1), rank AJ / (NL-1) among that matrix's waves; slot t belongs to the wave of rank
// t % (waves of the matrix); the first / last layer gradients of slot t are kept by wave ws_edge_owner(t).
// One slot (compile-time t: every LDS address is a base register + an immediate), then the next one.
// NL = 4 (three hidden matrices): ws_edge_owner; fewer matrices: every matrix has at least four waves of at most four slots, the old
// deal (waves 4..7, a quarter of the slots each) is balanced
template <int NL> __host__ __device__ constexpr int edge_of(int t) { return NL == 4 ? ws_edge_owner(t) : 4 + (t & 3); }
template <int NL> __host__ __device__ constexpr bool has_edges(int aj) { return NL == 4 ? ws_has_edges(aj) : aj >= 4; }
template <int NL, int U, int AJ, int T0>
__device__ __forceinline__ void ws_acc_slot(const float *__restrict__ recs, const float *__restrict__ hbase, const int mytag, const int lane,
                                            f2_t (&gw)[kMaxH / 2], float &gb, float (&ge)[17])
{
    constexpr int NM = NL - 1, NT = kWsP * U;
    constexpr int am = AJ % NM, ar = AJ / NM, an = (kWsA - 1 - am) / NM + 1;
    constexpr int kRec = ws_rec_elems<NL>(), kHand = ws_hand_elems<NL>();
    if constexpr (T0 < NT) {
        constexpr bool mine = (T0 % an) == ar, edge = edge_of<NL>(T0) == AJ;
        if constexpr (mine || edge) {
            const int tag = __builtin_amdgcn_readlane(mytag, T0);
            if (tag != 0) {
                const float *__restrict__ hd = hbase + (size_t)T0 * 2 * kHand;
                const float *__restrict__ rc = recs + ((size_t)T0 * kWsRing + (tag - 1)) * kRec;
                if constexpr (mine) {
                    // dW_m += delta_{m+1} (x) h_m: register pair (16 w + c, 16 (w + 2) + c) of lane (r, i) is dW[16 w + i][16 r + c] and
                    // dW[16 (w + 2) + i][16 r + c] -- the four accumulators of a column share h_m[16 r + c], which every lane of row r
                    // reads from the slot's record (ws_outer_nat), and the multipliers are delta_{m+1} of the units 16 w + i: four
                    // more LDS reads.  32 v_pk_fma_f32 and no cross-lane instruction.
                    const int p16 = lane & 15;
                    const float *__restrict__ dl = hd + (am + 1) * kWave;                  // delta_{m+1}
                    const Vec4<float> *__restrict__ hrow = reinterpret_cast<const Vec4<float> *>(rc + am * kWave + (lane & 48));   // h_m, the lane's row
                    const Vec4<float> q[4] = {hrow[0], hrow[1], hrow[2], hrow[3]};
                    f2_t D01, D23;
                    D01.x = dl[p16]; D01.y = dl[32 + p16]; D23.x = dl[16 + p16]; D23.y = dl[48 + p16];     // pairs (w0, w2), (w1, w3)
                    const float dm = dl[lane];
                    ws_outer_nat(gw, D01, D23, q);
                    gb += dm;                                                            // bias of hidden layer m + 2
                }
                if constexpr (edge) {
                    const float *__restrict__ xs = rc + NL * kWave, *__restrict__ tl = hd + NL * kWave;     // state | kb[6], t, tvns
                    const float d1 = hd[lane];
                    // first layer: input row [t, G, I, Glu, GLP1, GE, FFA, glp1 := GLP1, tvns] (broadcast LDS reads)
                    ge[0] = rfma(d1, tl[6], ge[0]);
                    ge[1] = rfma(d1, xs[0], ge[1]);
                    ge[2] = rfma(d1, xs[1], ge[2]);
                    ge[3] = rfma(d1, xs[2], ge[3]);
                    ge[4] = rfma(d1, xs[3], ge[4]);
                    ge[5] = rfma(d1, xs[4], ge[5]);
                    ge[6] = rfma(d1, xs[5], ge[6]);
                    ge[7] = rfma(d1, xs[3], ge[7]);
                    ge[8] = rfma(d1, tl[7], ge[8]);
                    ge[9] += d1;
                    // output layer: dWout[q][j] += kb_q h_NL[j], dbout[q] += kb_q
                    const float hl = rc[(NL - 1) * kWave + lane];
#pragma unroll
                    for (int q = 0; q < 6; ++q) ge[10 + q] = rfma(tl[q], hl, ge[10 + q]);
                    ge[16] += tl[lane & 7];                    // lanes 0..5 hold dbout (slots 6, 7 carry t / tVNS: never stored)
                }
            }
        }
        ws_acc_slot<NL, U, AJ, T0 + 1>(recs, hbase, mytag, lane, gw, gb, ge);
    }
}
template <int NL, int U, int AJ>
__device__ __forceinline__ void ws_acc_loop(const float *__restrict__ recs, const float *__restrict__ hands, const int *__restrict__ tags,
                                            const int n_iter, const bool work, const int lane, f2_t (&gw)[kMaxH / 2], float &gb, float (&ge)[17],
                                            const int dbg)
{
    constexpr int kHand = ws_hand_elems<NL>();
    (void)dbg;
    WS_TRACE_DECL;
#pragma unroll 1
    for (int it = 0; it < n_iter; ++it) {
        WS_TRACE_BEGIN(it);
        WS_STAMP(0);
        if (it > 0 && work) {
            const int rp = (it - 1) & 1;                       // what the propagation waves published one iteration ago
            const int mytag = tags[rp * 16 + (lane & 15)];    // all sixteen tags in one read
            ws_acc_slot<NL, U, AJ, 0>(recs, hands + rp * kHand, mytag, lane, gw, gb, ge);
        }
        WS_STAMP(8);
        __syncthreads();
        WS_STAMP(9);
        WS_TRACE_FLUSH(it, kWsP + AJ);
    }
}

template <int NL, int U, bool GODE, bool GD>
__global__ __launch_bounds__(64 * kWsWaves) void solve_bwd_ws_kernel(const AdjArgs<float> a, const int method_dbg)
{
    using R = float;
    using ES = EdgeSlots<NL>;
    static_assert(NL >= 2 && NL <= 4, "the specialised adjoint needs at least one hidden matrix");
    static_assert(U == 1 || U == 2, "one or two trajectories per propagation wave");
    constexpr int NM = NL - 1;                            // hidden matrices
    constexpr int NT = kWsP * U;                          // trajectory slots of the workgroup
    constexpr int kRec = ws_rec_elems<NL>();
    constexpr int kHand = ws_hand_elems<NL>();
    constexpr int kSlot = NL * kWave + 8;                 // stage record on the tape: NL rows + 8 reals of stage state
    constexpr int kMaxRank = (kWsA - 1) / NM + 1;         // accumulation waves of matrix 0 (the most any matrix has)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *wt = reinterpret_cast<R *>(smem_raw);              // [NM][64*64] transposed hidden matrices, rotating-operand order
    R *rowsT = wt + (size_t)NM * kMaxH * kMaxH;           // [8][64] transposed tableau rows
    R *recs = rowsT + 8 * kWave;                          // [NT][kWsRing][kRec]
    R *hands = recs + (size_t)NT * kWsRing * kRec;        // [NT][2][kHand]
    int *tags = reinterpret_cast<int *>(hands + (size_t)NT * 2 * kHand);      // [2][16]: 0 = nothing published, else 1 + ring slot
    int *niter = tags + 2 * 16;
    // step headers, [NT][2][64]: everything a slot's step needs from global memory besides its stage records, gathered by ONE LDS-DMA
    // with per-lane addresses one step ahead (two buffers: the current step's constants are read until its last stage):
    //   words 0..7    the step's tape entry {t, h, t0, 1 / (t1 - t0), v0, v1 - v0, d0, d1 - d0} as the forward wrote it (tape_put)
    //   words 8 j + c dLoss/dy[b, kn + 1 - j, c], j = 1..6, c < 6: the grid rows the step can close (kn = the interval of the step
    //                 walked before it, T - 1 for the first)
    //   words 56..63  the step's interval word
    // Rounds 1-3 read all of it with scalar / vector loads in the header itself -- interval index first, then the rows behind it: two
    // dependent HBM round trips, 4 000-6 000 cycles in every sixth iteration with all sixteen waves of the workgroup waiting at the
    // barrier (tools/ws_trace.py) -- and kept the step constants in twelve VGPRs that pushed the loop into scratch.
    R *hdrs = reinterpret_cast<R *>(niter + 16);
    R *odeL = hdrs + (size_t)NT * 2 * kWave;               // [20] the 17 mechanistic constants of this parameter set (+ padding)
    R *w5L = odeL + 32;                                    // [6][64] output-layer weights, Wout[q][unit of the lane] (0 beyond H)

    const int lane = threadIdx.x & 63;
    const int c8 = lane & 7, grp = lane >> 3;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    const int set = blockIdx.y;
    const int T = a.T;
    const int per_set = a.B / a.n_sets;
#ifdef HODE_LAB
    const int dbg = method_dbg >> 8;                      // lab library: timing experiments (HODE_WS_DBG), results are WRONG with any bit set
    const int method = method_dbg & 0xff;
#else
    constexpr int dbg = 0;
    const int method = method_dbg;
#endif
    const int S = kTableau[method].S;
    const R *__restrict__ nn_set = a.nn_p + (size_t)set * a.P;
    const bool isP = wave < kWsP;
    // restrict-qualified views: wave-uniform reads of them are scalar loads (through the struct members hipcc has to assume
    // they alias the gx0 stores and falls back to exec-masked vector loads with a full vmcnt wait each)
    const R *__restrict__ const gy_ = a.gy, *__restrict__ const tgrid_ = a.t, *__restrict__ const tape_ = a.tape,
                        *__restrict__ const stage_ = a.tape_stage;
    const int *__restrict__ const seg_ = a.tape_seg, *__restrict__ const nsteps_ = a.nsteps, *__restrict__ const status_ = a.status;

    wt_blk_store(wt, nn_set, a.H, NM, threadIdx.x, 64 * kWsWaves);
    tableau_rowsT_store<R>(rowsT, method, threadIdx.x, 64 * kWsWaves);
    if (threadIdx.x < 2 * 16) tags[threadIdx.x] = 0;
    if (threadIdx.x == 0) *niter = 0;
    if (threadIdx.x < 20) odeL[threadIdx.x] = threadIdx.x < 17 ? a.ode_p[17 * set + threadIdx.x] : R(0);
    if (threadIdx.x < 6 * kWave) {
        const int q = threadIdx.x >> 6, j = threadIdx.x & 63;
        w5L[threadIdx.x] = j < a.H ? nn_set[9 * a.H + a.H + (size_t)NM * ((size_t)a.H * a.H + a.H) + q * a.H + j] : R(0);
    }
    __syncthreads();
    // iterations = the longest slot's number of stages (+ 1: accumulation runs one iteration behind).  Slot t = U wave + u takes
    // the trajectories t, t + NT, ... of this workgroup's share (slot-major, like the one-role kernel's wave-major deal)
    if (isP) {
        int mx = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int tot = 0;
            for (int bi = (wave * U + u) * gridDim.x + blockIdx.x; bi < per_set; bi += gridDim.x * NT) {
                const int nb = nsteps_[set * per_set + bi];
                tot += (nb < a.max_steps ? nb : a.max_steps) * S;
            }
            mx = tot > mx ? tot : mx;
        }
        if (lane == 0) atomicMax(niter, mx);
    }
    __syncthreads();
    const int n_iter = *niter + 1;

    // ---- the two roles: each has its OWN loop (its registers are live in its branch only: 64 accumulators here, the
    //      propagation state there), both loops execute the same n_iter barriers ---------------------------------------------
    f2_t gw[kMaxH / 2];                                   // A: dW of one hidden matrix (row-block order, accumulator pairs: ws_acc_slot)
    R gb = 0.f;                                           // A: bias of that matrix's layer
    R ge[17];                                             // A_4..7: W1 (9), b_1, Wout (6), bout
    R go = 0.f;                                           // P: lane p < 17 holds d/d(ode constant p)
    const int aj = wave - kWsP;                           // accumulation wave index
    const int am = aj % NM, ar = aj / NM;                 // its matrix and its rank among that matrix's waves
    if (isP) {
        // (no s_setprio: raising the propagation waves' priority was neutral in round 3 and costs 1 % since the accumulation waves got
        //  lighter -- the SIMD's arbiter already prefers the older waves, and these are the workgroup's first eight; lab switch 16 sets it)
        if (dbg & 16) __builtin_amdgcn_s_setprio(2);
        // edge weights in registers (no accumulators here: there is room); per trajectory: step / stage cursors
        R w1r[8];
        R lam[U], ZZ[U];
        int hp[U];                                        // which of the slot's two header buffers holds the CURRENT step
        int bi_next[U], b[U], n[U], st[U], s[U], knext[U], k[U], cur[U];
        bool active[U], ok[U];
        const R *__restrict__ stg[U];
        {
            // first-layer weights of the six state inputs in the rotating order of out_rot (hode_device.h): lane (r, i) keeps, for
            // input o = i & 7, w1r[n] = W1[16 r + ((i - n) & 15)][1 + o] (GLP1, o = 3, feeds inputs 4 and 7; zero for o >= 6), so
            // that the state cotangent W1^T delta_1 is 8 FMAs on delta_1 in its natural layout + a 7-instruction reduction that
            // lands in the replicated layout of the state -- 15 instructions instead of six products + the 30 of
            // wave_reduce6_to_lanes, in a wave whose time is its instruction count
            const int i16 = lane & 15, r16 = lane >> 4, oin = lane & 7;
#pragma unroll
            for (int n_ = 0; n_ < 8; ++n_) {
                const int ku = 16 * r16 + ((i16 - n_) & 15);
                const bool okw = oin < 6 && ku < a.H;
                const int kk = okw ? ku : 0, oo = okw ? oin : 0;
                R v = nn_set[kk * 9 + 1 + oo];
                if (oo == 3) v += nn_set[kk * 9 + 7];
                w1r[n_] = okw ? v : 0.f;
            }
        }
        auto rec_of = [&](int u) -> R * { return recs + (size_t)(wave * U + u) * kWsRing * kRec; };
        auto hand_of = [&](int u, int par) -> R * { return hands + ((size_t)(wave * U + u) * 2 + par) * kHand; };
        auto hdr_of = [&](int u, int par) -> R * { return hdrs + ((size_t)(wave * U + u) * 2 + par) * kWave; };
        // the header of step `stn` of trajectory b[u] -> buffer `par` (see hdrs above); `kn`: the rows it can close end at kn
        auto hdr_dma = [&](int u, int stn, int kn, int par) {
            // three wave-uniform bases, pinned to the scalar unit at the point of use (left to itself hipcc keeps per-lane copies of the
            // three array pointers live across the whole loop -- six VGPRs it does not have: scratch, with a reload and a full vmcnt
            // wait behind the record DMA in front of every gather)
            uint64_t bg = reinterpret_cast<uint64_t>(gy_ + (size_t)b[u] * T * 6);
            uint64_t bt = reinterpret_cast<uint64_t>(tape_ + ((size_t)b[u] * a.max_steps + stn) * 8);
            uint64_t bs = reinterpret_cast<uint64_t>(seg_ + (size_t)b[u] * a.max_steps + stn);
            asm volatile("" : "+s"(bg), "+s"(bt), "+s"(bs));
            int row = kn + 1 - grp;
            row = row < 0 ? 0 : row;
            row = row > T - 1 ? T - 1 : row;
            const uint32_t off = (grp == 0) ? 4u * c8 : (grp == 7) ? 0u : 4u * (uint32_t)(row * 6 + (c8 < 6 ? c8 : 5));
            const uint64_t base = (grp == 0) ? bt : (grp == 7) ? bs : bg;
            const R *p = reinterpret_cast<const R *>(base + off);
            __builtin_amdgcn_global_load_lds(p, (__attribute__((address_space(3))) void *)hdr_of(u, par), 4, 0, 0);
        };
        // one record = NL rows of 64 reals + the stage state: NL + 1 DMA instructions off ONE address pair, the row offset is the
        // instruction's immediate (it advances the global and the LDS address alike).  The last row is loaded whole although
        // only 8 reals of it belong to the record: the other 56 are the head of the next record (or, behind the very last one,
        // of the gradient rows that follow the stage tape): read, never used -- and no exec mask to set up
        auto rec_dma = [&](const R *__restrict__ src, R *dst) {
            const R *gsrc = src + lane;
            auto ldst = (__attribute__((address_space(3))) void *)dst;
            __builtin_amdgcn_global_load_lds(gsrc, ldst, 4, 0, 0);
            if constexpr (NL >= 1) __builtin_amdgcn_global_load_lds(gsrc, ldst, 4, 256, 0);
            if constexpr (NL >= 2) __builtin_amdgcn_global_load_lds(gsrc, ldst, 4, 512, 0);
            if constexpr (NL >= 3) __builtin_amdgcn_global_load_lds(gsrc, ldst, 4, 768, 0);
            if constexpr (NL >= 4) __builtin_amdgcn_global_load_lds(gsrc, ldst, 4, 1024, 0);
        };
        auto inject = [&](int u, int r) {              // lam += dLoss/dy[b, r, :]  (six wave-uniform scalar loads)
            const R *__restrict__ gr = gy_ + ((size_t)b[u] * T + r) * 6;
            const R g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3], g4 = gr[4], g5 = gr[5];
            lam[u] += (c8 == 0) ? g0 : (c8 == 1) ? g1 : (c8 == 2) ? g2 : (c8 == 3) ? g3 : (c8 == 4) ? g4 : (c8 == 5) ? g5 : 0.f;
        };
        // rows 0..kf of a trajectory are (copies of) x0: their cotangents close the trajectory (see solve_bwd_kernel)
        auto finish_traj = [&](int u) {
            const R *__restrict__ tg = tgrid_ + (a.t_batched ? (size_t)b[u] * T : 0);
            int kf = 0;
            while (kf + 1 < T && !(tg[kf + 1] > tg[kf])) ++kf;
            for (int r = 0; r <= kf; ++r) inject(u, r);
            if (lane < 6) a.gx0[(size_t)b[u] * 6 + lane] = lam[u];
        };
        // next trajectory of this slot that has steps on its tape; trajectories without any are closed on the spot
        auto start_next = [&](int u) {
            active[u] = false;
            while (bi_next[u] < per_set) {
                b[u] = set * per_set + bi_next[u];
                bi_next[u] += gridDim.x * NT;
                stg[u] = stage_ + (size_t)b[u] * a.max_steps * 6 * kSlot;
                n[u] = nsteps_[b[u]] < a.max_steps ? nsteps_[b[u]] : a.max_steps;      // never walk past the tape
                ok[u] = status_[b[u]] == HODE_ST_OK;
                lam[u] = 0.f;
                knext[u] = T - 1;
                if (n[u] > 0) {
                    st[u] = n[u] - 1;
                    s[u] = S - 1;
                    rec_dma(stg[u] + ((size_t)st[u] * 6 + s[u]) * kSlot, rec_of(u) + cur[u] * kRec);      // its first record
                    hdr_dma(u, st[u], T - 1, hp[u]);
                    active[u] = true;
                    return;
                }
                finish_traj(u);
            }
        };
        Vec4<float> wq0[4], wq1[4];                        // the two row groups of the transposed matrices that are read ahead
        ws_wt_preload(wt + (size_t)(NM - 1) * kMaxH * kMaxH, lane, wq0, wq1);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bi_next[u] = (wave * U + u) * gridDim.x + blockIdx.x;
            cur[u] = 0; st[u] = -1; s[u] = 0; k[u] = 0; n[u] = 0; b[u] = 0; knext[u] = 0; ok[u] = true;
            lam[u] = ZZ[u] = 0.f;
            hp[u] = 0;
            hdr_of(u, 0)[lane] = 0.f;
            hdr_of(u, 1)[lane] = 0.f;
            stg[u] = stage_;
            start_next(u);
        }
        WS_TRACE_DECL;
#pragma unroll 1
        for (int it = 0; it < n_iter; ++it) {
            const int par = it & 1;
            WS_TRACE_BEGIN(it);
            WS_STAMP(0);
            // the records to process now and the headers were DMA'd one iteration ago (or at the start of their trajectory)
            WS_STAMP(1);
            __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0)
            WS_STAMP(2);
            __builtin_amdgcn_wave_barrier();
            // ---- step headers: cotangents of the grid rows the step produced, rows k + 1 .. knext; row knext + 1 - j sits in words
            //      8 j .. 8 j + 5 of the header (j = 1..6).  One row (the usual case) is added as it is, several are summed first
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (active[u] && s[u] == S - 1) {
                    const R *__restrict__ hh = hdr_of(u, hp[u]);
                    const int kraw = first_lane(f2i(hh[56]));
                    k[u] = kraw & (kSegClosed - 1);
                    if (st[u] == n[u] - 1 && !ok[u]) {
                        // the last step of a FAILED trajectory (the first of the walk): see solve_bwd_kernel
                        const R *__restrict__ tg = tgrid_ + (a.t_batched ? (size_t)b[u] * T : 0);
                        int hi = k[u];
                        if (kraw & kSegClosed) {
                            hi = k[u] + 1;
                            while (hi + 1 < T && !(tg[hi + 1] > tg[hi])) ++hi;
                        }
                        for (int r = k[u] + 1; r <= hi; ++r) inject(u, r);
                    } else {
                        const int nrow = knext[u] - k[u];
                        const R hv = hh[lane];
                        const R v = (grp >= 1 && grp <= nrow && grp < 7 && c8 < 6) ? hv : 0.f;
                        lam[u] += group_sum8(v);
                        for (int r = k[u] + 1; r <= knext[u] - 6; ++r) inject(u, r);      // (more than six rows behind one step: repeated grid times)
                    }
                    knext[u] = k[u];
                    ZZ[u] = 0.f;
                }
            }
            int slot[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                slot[u] = cur[u];
                if (active[u]) {
                    cur[u] = (cur[u] + 1 == kWsRing) ? 0 : cur[u] + 1;
                    // the next record of the trajectory goes to the slot behind: the one the accumulation waves read in the
                    // PREVIOUS iteration (they are done with it: a barrier lies in between)
                    const int ns_ = (s[u] > 0) ? s[u] - 1 : S - 1, nst = (s[u] > 0) ? st[u] : st[u] - 1;
                    if (nst >= 0) rec_dma(stg[u] + ((size_t)nst * 6 + ns_) * kSlot, rec_of(u) + cur[u] * kRec);
                    // one step ahead: the header of the step below this one, into the slot's other buffer
                    if (s[u] == S - 1 && st[u] >= 1) hdr_dma(u, st[u] - 1, k[u], hp[u] ^ 1);
                }
            }
            // ---- J^T kb for all U trajectories in ONE basic block (an idle slot computes on stale data; nothing of it is kept):
            //      mechanistic part, then the cotangent through the layers; every delta goes to the hand-off slot
            R hact[U][NL], kb[U], ts[U], tv[U], mech[U], d[U];
            R *__restrict__ hd[U];
            // the 17 mechanistic constants: five broadcast reads per iteration into VGPRs that live through the mechanistic part only
            // (as VGPRs because 17 wave-uniform SGPRs are more than this loop has and a VALU instruction reads one SGPR only; held for
            // the whole loop they were 17 of the 128 registers the matrices' operands now need -- asm: see ws_lds_read128)
            OdeP<R> o;
            {
                const unsigned oa = (unsigned)(size_t)(__attribute__((address_space(3))) R *)odeL;
                ws_f4_t c0 = ws_lds_read128(oa), c1 = ws_lds_read128(oa + 16), c2 = ws_lds_read128(oa + 32), c3 = ws_lds_read128(oa + 48),
                        c4 = ws_lds_read128(oa + 64);
                ws_lds_wait(c0, c1);
                ws_lds_wait(c2, c3);
                ws_lds_wait(c4);
                o.a_GI = c0.x; o.k_I = c0.y; o.rho = c0.z; o.G_b = c0.w; o.I_b = c1.x; o.E_max = c1.y; o.EC_50 = c1.z;
                o.Glu_b = c1.w; o.V_max = c2.x; o.K_m = c2.y; o.k_L = c2.z; o.k_GE0 = c2.w; o.IGD_50 = c3.x; o.g = c3.y;
                o.p_7 = c3.z; o.p_8 = c3.w; o.p_9 = c4.x;
            }
            // ... and the lane's six output-layer weights (one table for the workgroup, six 4-byte reads per iteration)
            R w5[6];
            {
                const unsigned wa = (unsigned)(size_t)(__attribute__((address_space(3))) R *)(w5L + lane);
                asm volatile("ds_read_b32 %0, %6\n\tds_read_b32 %1, %6 offset:256\n\tds_read_b32 %2, %6 offset:512\n\t"
                             "ds_read_b32 %3, %6 offset:768\n\tds_read_b32 %4, %6 offset:1024\n\tds_read_b32 %5, %6 offset:1280\n\t"
                             "s_waitcnt lgkmcnt(0)"
                             : "=&v"(w5[0]), "=&v"(w5[1]), "=&v"(w5[2]), "=&v"(w5[3]), "=&v"(w5[4]), "=&v"(w5[5]) : "v"(wa));
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                hd[u] = hand_of(u, par);
                const R *__restrict__ rc = rec_of(u) + slot[u] * kRec;
#pragma unroll
                for (int l = 0; l < NL; ++l) hact[u][l] = rc[l * kWave + lane];
                const R Ys = rc[NL * kWave + c8];             // stage state, replicated layout
                // {t, h, t0, 1 / len}, {v0, dv, d0, dd} of the current step: two broadcast reads of the slot's header.  Written as asm:
                // hipcc puts an s_waitcnt vmcnt(0) in front of any LDS read it sees that MAY alias a pending LDS-DMA -- here the next
                // record and the next header, issued a moment ago into OTHER buffers -- and the wave would sit out a whole HBM round
                // trip (~2 000 cycles per iteration, tools/ws_trace.py).  The second statement is the wait these reads need.
                Vec4<R> sc0, sc1;
                {
                    typedef float f4_t __attribute__((ext_vector_type(4)));
                    f4_t q0, q1;
                    const unsigned ha = (unsigned)(size_t)(__attribute__((address_space(3))) R *)hdr_of(u, hp[u]);
                    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(q0), "=&v"(q1) : "v"(ha));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1));
                    sc0 = Vec4<R>{{q0.x, q0.y, q0.z, q0.w}};
                    sc1 = Vec4<R>{{q1.x, q1.y, q1.z, q1.w}};
                }
                const int su = s[u];
                const R bw_s = rowsT[6 * kWave + su], c_s = rowsT[6 * kWave + 8 + su];
                kb[u] = sc0.v[1] * rfma(bw_s, lam[u], group_sum8(rowsT[su * kWave + lane] * ZZ[u]));
                ts[u] = rfma(c_s, sc0.v[1], sc0.v[0]);
                const R al = (ts[u] - sc0.v[2]) * sc0.v[3];
                const R gdv = rfma(al, sc1.v[3], sc1.v[2]);
                tv[u] = rfma(al, sc1.v[1], sc1.v[0]);
                R gde = 0.f;
                if constexpr (GD) gde = gd_effect(o, gdv);
                const R G = lane_bcast(Ys, 0), I = lane_bcast(Ys, 1), Glu = lane_bcast(Ys, 2), GLP1 = lane_bcast(Ys, 3),
                        FFA = lane_bcast(Ys, 5);
                const R lG = lane_bcast(kb[u], 0), lI = lane_bcast(kb[u], 1), lGlu = lane_bcast(kb[u], 2), lGLP = lane_bcast(kb[u], 3),
                        lGE = lane_bcast(kb[u], 4), lF = lane_bcast(kb[u], 5);
                R gou = 0.f;
                mech[u] = (dbg & 4) ? kb[u] : ws_mech_vjp<GODE>(o, G, I, Glu, GLP1, FFA, lG, lI, lGlu, lGLP, lF, gde, gdv, GD, lane, gou);
                if constexpr (GODE) go += active[u] ? gou : 0.f;
                R dl = w5[0] * lG;
                dl = rfma(w5[1], lI, dl);
                dl = rfma(w5[2], lGlu, dl);
                dl = rfma(w5[3], lGLP, dl);
                dl = rfma(w5[4], lGE, dl);
                dl = rfma(w5[5], lF, dl);
                d[u] = (hact[u][NL - 1] > 0.f) ? dl : 0.f;
                hd[u][(NL - 1) * kWave + lane] = d[u];         // delta_NL
            }
            WS_STAMP(3);
#pragma unroll
            for (int l = NL - 1; l >= 1; --l) {                // hidden matrix l-1 maps h_l -> h_{l+1}
                R dp[U];
                if (dbg & 2) {
#pragma unroll
                    for (int u = 0; u < U; ++u) dp[u] = d[u] * 0.5f;
                } else {
                    // (the matrix after this one: l - 2, or -- behind the last -- the first matrix of the next stage)
#if HODE_WS_PROP_LB
                    unsigned drow[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) drow[u] = (unsigned)(size_t)(__attribute__((address_space(3))) R *)(hd[u] + l * kWave + (lane & 48));
                    ws_wt_mul_lb<U>(wt + (size_t)(l - 1) * kMaxH * kMaxH, wt + (size_t)(l >= 2 ? l - 2 : NM - 1) * kMaxH * kMaxH, lane, drow, dp, wq0, wq1);
#else
                    ws_wt_mul<U>(wt + (size_t)(l - 1) * kMaxH * kMaxH, wt + (size_t)(l >= 2 ? l - 2 : NM - 1) * kMaxH * kMaxH, lane, d, dp, wq0, wq1);
#endif
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    d[u] = (hact[u][l - 1] > 0.f) ? dp[u] : 0.f;
                    hd[u][(l - 1) * kWave + lane] = d[u];      // delta_l
                }
                WS_STAMP(3 + NL - l);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // lanes 0..5 kb, 6 t, 7 tVNS (the other groups of eight: copies, never read): no exec mask.  The tag: every lane writes the same word
                hd[u][NL * kWave + lane] = (c8 < 6) ? kb[u] : (c8 == 6) ? ts[u] : tv[u];
                tags[par * 16 + wave * U + u] = active[u] ? 1 + slot[u] : 0;                      // valid, and which ring slot
                // W1^T delta_1 on the six state inputs, replicated layout, exact zeros on slots 6, 7 (as the mechanistic part)
                const R nnv = (dbg & 8) ? d[u] : out_rot(w1r, 0.f, d[u]);
                const R Z = mech[u] + nnv;
                if (active[u]) {
                    ZZ[u] = (grp == s[u]) ? Z : ZZ[u];
                    if (s[u] == 0) {
                        lam[u] += group_sum8(rowsT[7 * kWave + lane] * ZZ[u]);
                        s[u] = S - 1;
                        if (--st[u] < 0) {
                            finish_traj(u);
                            start_next(u);
                        } else {
                            hp[u] ^= 1;            // the next step's header was gathered while this step's stages ran
                        }
                    } else {
                        --s[u];
                    }
                }
            }
            WS_STAMP(8);
            __syncthreads();
            WS_STAMP(9);
            WS_TRACE_FLUSH(it, wave);
        }
    } else {
#pragma unroll
        for (int r = 0; r < kMaxH / 2; ++r) gw[r] = f2_t{0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 17; ++i) ge[i] = 0.f;
        // one instantiation per accumulation wave: which slots it serves is a compile-time table, every LDS address below is a
        // base register + an immediate
        const bool work = a.gnn != nullptr && !(dbg & 1);
        switch (aj) {
        case 0: ws_acc_loop<NL, U, 0>(recs, hands, tags, n_iter, work, lane, gw, gb, ge, dbg); break;
        case 1: ws_acc_loop<NL, U, 1>(recs, hands, tags, n_iter, work, lane, gw, gb, ge, dbg); break;
        case 2: ws_acc_loop<NL, U, 2>(recs, hands, tags, n_iter, work, lane, gw, gb, ge, dbg); break;
        case 3: ws_acc_loop<NL, U, 3>(recs, hands, tags, n_iter, work, lane, gw, gb, ge, dbg); break;
        case 4: ws_acc_loop<NL, U, 4>(recs, hands, tags, n_iter, work, lane, gw, gb, ge, dbg); break;
        case 5: ws_acc_loop<NL, U, 5>(recs, hands, tags, n_iter, work, lane, gw, gb, ge, dbg); break;
        case 6: ws_acc_loop<NL, U, 6>(recs, hands, tags, n_iter, work, lane, gw, gb, ge, dbg); break;
        default: ws_acc_loop<NL, U, 7>(recs, hands, tags, n_iter, work, lane, gw, gb, ge, dbg); break;
        }
    }

    // ---- epilogue: ONE gradient row per workgroup (same row format as solve_bwd_kernel); every sum in a fixed order ------------
    const int nthreads = 64 * kWsWaves;
    const size_t rowlen = adj_partial_rowlen(a.P);
    R *__restrict__ prow = a.partials + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * rowlen;
    const int H = a.H;
    if (a.gnn) {
        constexpr int kHid = NM * kMaxH * kMaxH;
        // the image of the transposed matrices is dead: it becomes the [matrix][row][col] sum over the waves of each matrix
        for (int rr = 0; rr < kMaxRank; ++rr) {
            if (!isP && ar == rr) {
#pragma unroll
                for (int r = 0; r < kMaxH; ++r) {
                    // accumulator 16 w + c of lane (q, i) is dW[16 w + i][16 q + c]; it is half (w >> 1) of the pair 2 c + (w & 1)
                    const int w = r >> 4, c = r & 15, i = lane & 15, q = lane >> 4;
                    const R v = (w >> 1) ? gw[2 * c + (w & 1)].y : gw[2 * c + (w & 1)].x;
                    R *dst = wt + (size_t)am * kMaxH * kMaxH + (16 * w + i) * kMaxH + 16 * q + c;
                    *dst = (rr == 0) ? v : *dst + v;
                }
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < kHid; i += nthreads) {
            const int l = i >> 12, row = (i >> 6) & 63, col = i & 63;
            if (row < H && col < H) prow[9 * H + H + (size_t)l * ((size_t)H * H + H) + (size_t)row * H + col] = wt[i];
        }
        // biases and edge layers: [slot][64] in the (dead) record area, wave after wave
        R *edgeS = recs;
        for (int i = threadIdx.x; i < ES::count * kWave; i += nthreads) edgeS[i] = 0.f;
        __syncthreads();
        for (int jj = 0; jj < kWsA; ++jj) {
            if (aj == jj) {
                edgeS[(ES::b + am + 1) * kWave + lane] += gb;
                if (has_edges<NL>(jj)) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) edgeS[(ES::w1 + i) * kWave + lane] += ge[i];
                    edgeS[(ES::b + 0) * kWave + lane] += ge[9];
#pragma unroll
                    for (int q = 0; q < 6; ++q) edgeS[(ES::w5 + q) * kWave + lane] += ge[10 + q];
                    edgeS[ES::b5 * kWave + lane] += ge[16];
                }
            }
            __syncthreads();
        }
        const size_t off_out = (size_t)9 * H + H + (size_t)NM * ((size_t)H * H + H);
        for (int i = threadIdx.x; i < ES::count * kWave; i += nthreads) {
            const int slot = i >> 6, j = i & 63;
            const R v = edgeS[i];
            if (slot < ES::b) { if (j < H) prow[j * 9 + slot] = v; }
            else if (slot < ES::w5) {
                const int l = slot - ES::b;
                if (j < H) prow[9 * H + (l == 0 ? 0 : H + (size_t)(l - 1) * ((size_t)H * H + H) + (size_t)H * H) + j] = v;
            } else if (slot < ES::b5) { if (j < H) prow[off_out + (slot - ES::w5) * H + j] = v; }
            else if (j < 6) prow[off_out + 6 * H + j] = v;
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

template <int NL, int U, bool GODE, bool GD> static int launch_ws_u(hipStream_t s, const AdjArgs<float> &a, int method, int blocks)
{
    const size_t lds = ws_lds_elems<NL, U>() * sizeof(float);
    auto kern = solve_bwd_ws_kernel<NL, U, GODE, GD>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return HODE_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(blocks, a.n_sets), dim3(64 * kWsWaves), lds, s, a, method);
    if (a.gnn || (GODE && a.gode))
        launch_adj_reduce(s, a.partials, (int)adj_partial_rowlen(a.P), blocks, a.n_sets, a.P, a.gnn, GODE ? a.gode : nullptr);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <int NL, bool GODE, bool GD> static int launch_ws_g(hipStream_t s, const AdjArgs<float> &a, int method, int cus)
{
    const int per_set = a.B / a.n_sets;
    int blocks = per_set < cus ? per_set : cus;           // one workgroup per CU; its trajectory slots loop over trajectories
    if (a.n_sets > 1 && blocks * a.n_sets > cus) blocks = cus / a.n_sets;
    if (blocks < 1) blocks = 1;
    if (a.partials == nullptr || blocks * a.n_sets > a.partial_rows) return HODE_EUNSUPPORTED;     // caller falls back
    // two trajectories per propagation wave as soon as a workgroup has more than eight to process
    int two = per_set > blocks * kWsP && !GODE;          // (with the 17 ODE-constant partials the two-trajectory instantiation spills)
#ifdef HODE_LAB
    static const int dbg = [] { const char *e = getenv("HODE_WS_DBG"); return e ? atoi(e) : 0; }();
    method |= (dbg & 0xcff) << 8;                          // (256 / 512 are the host's switches below)
    if (dbg & 256) two = 0;
    if (dbg & 512) two = 1;
#endif
    return two ? launch_ws_u<NL, 2, GODE, GD>(s, a, method, blocks) : launch_ws_u<NL, 1, GODE, GD>(s, a, method, blocks);
}

template <int NL> static int launch_ws_nl(hipStream_t s, const AdjArgs<float> &a, int method, int cus)
{
    const bool gd = a.gd_mode != 0;
    if (a.gode) return gd ? launch_ws_g<NL, true, true>(s, a, method, cus) : launch_ws_g<NL, true, false>(s, a, method, cus);
    return gd ? launch_ws_g<NL, false, true>(s, a, method, cus) : launch_ws_g<NL, false, false>(s, a, method, cus);
}

#ifdef HODE_LAB
}  // namespace hode
// lab library only: the stamps of HODE_WS_DBG bit 1024 (tools/ws_trace.py)
extern "C" int hode_lab_ws_trace(unsigned long long *dst, int n)
{
    using namespace hode;
    const int total = kWsTraceIts * kWsWaves * kWsTracePts;
    if (n < total) return -total;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ws_trace), sizeof(unsigned long long) * total) == hipSuccess ? total : -1;
}
namespace hode {
#endif

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
