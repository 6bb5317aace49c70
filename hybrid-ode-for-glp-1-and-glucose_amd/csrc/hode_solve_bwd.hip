// hode_solve_bwd.hip -- K4: reverse-time discrete adjoint of the forward solve; K5: RHS backward.
//
// No reference counterpart: the reference detaches the solve (models/hybrid_ode_nn.py:186,
// 234-237, 248).  north_star asks for adjoint backprop, so this kernel
// differentiates the discrete scheme the forward kernel ran ("discretise-then-differentiate"):
// it walks the tape of accepted steps backwards and pulls the cotangent through the Runge-Kutta stages
// of each step.  Step sizes are treated as constants.
// CPU restatement: oracle/hode_oracle_impl.h (hode_oracle_solve_bwd).
//
// Mapping: one trajectory per wavefront, one hidden unit per lane (see hode_device.h).
//   * the forward recorded, for every stage of every accepted step, the layer activations and the (compact)
//     stage state on the "stage tape" in HBM (6.3 KB per step in fp32): the adjoint streams them back
//     by LDS-DMA (global_load ... lds, one stage ahead, double buffered) instead of recomputing the
//     forward -- a memory-for-compute trade that 288 GB / 8 TB/s of HBM3E make cheap;
//   * lane j keeps row j of the three hidden-matrix gradient accumulators in VGPRs (192 registers in
//     fp32): 2 waves per SIMD.  First/last-layer weights and their accumulators live in LDS tables;
//   * dW_l[j][:] += delta_l[j] * h_{l-1}[:]  is 64 v_fmac_f32_dpp (rotating-operand form);
//   * delta_{l-1} = W_l^T delta_l reads the transposed matrices from an LDS image shared by the
//     8 waves of the workgroup (one 16-byte read per 4 FMAs, again with DPP row_ror operands); the
//     outer-product FMAs sit between the issue and the use of every group of reads;
//   * a wave loops over several trajectories and keeps accumulating; at the end the 8 waves of a
//     workgroup reduce through LDS and flush coalesced atomics (12 k per workgroup).
#include "hode_device.h"
#include "hode_kernels.h"
#include <type_traits>
#include <cstdlib>

namespace hode {

template <typename R> __device__ __forceinline__ R inp_at_b(const R *__restrict__ p, int mode, int b, int T, int k)
{
    if (mode == 0) return R(0);
    return (mode == 1) ? p[b] : p[(size_t)b * T + k];
}

// LDS layout of the adjoint workgroup (kBwdWaves waves sharing one parameter set):
//   wt    [(NL-1)][64*64]   transposed hidden matrices, rotating-operand order (hode_device.h: wt_rot_store)
//   rowsT [8][64]           transposed tableau rows A[lane>>3][s]; row 7 = 1 for the solution stages
constexpr int kBwdWaves = 8;
// stage records in flight AHEAD of the one being processed (LDS-DMA ring of kBwdAhead<R> + 1 slots per wave).  One record
// ahead is ~2 900 cycles of lead; two (-DHODE_BWD_AHEAD=2, counted vmcnt wait) were measured at the same 8.1 ms: the
// record DMA is not what the stage loop waits for
#ifndef HODE_BWD_AHEAD
#define HODE_BWD_AHEAD 1
#endif
// fp64 (parity builds): the 96 KB transposed-matrix image leaves room for one record ahead only
template <typename R> constexpr int kBwdAhead = (sizeof(R) == 4) ? HODE_BWD_AHEAD : 1;
template <typename R> constexpr int kBwdRing = kBwdAhead<R> + 1;
//   edgeW [16+NL][64]       first/last layer weights (fp32 build; shared)
//   edgeG [waves][16+NL][64] first/last layer gradient accumulators (fp32 build; per wave)
//   rec   [waves][ring][NL+1][64] stage records (h_1..h_NL, state) arriving by LDS-DMA (global_load ... lds)
template <typename R> constexpr bool kEdgeLds = (sizeof(R) == 4);
template <typename R, int NL> __host__ __device__ constexpr size_t bwd_lds_elems()
{
    return (size_t)(NL > 1 ? NL - 1 : 1) * kMaxH * kMaxH + 8 * kWave +
           (kEdgeLds<R> ? (size_t)(1 + kBwdWaves) * EdgeSlots<NL>::count * kWave : 0) +
           (size_t)kBwdWaves * kBwdRing<R> * (NL + 1) * kWave;      // rec: [waves][ring][NL rows + state][64] stage-record ring
}

// The adjoint reads, for every stage of every accepted step, what the forward recorded on the stage
// tape (layer activations + stage state): it never recomputes the forward.
// WTREG = false: kBwdWaves (8) waves per workgroup, transposed matrices in LDS, 2 waves/SIMD.
// WTREG = true (fp32): 4 waves per workgroup, transposed matrices in registers, 1 wave/SIMD, no LDS wait
//         inside the 64-FMA loops.
template <typename R, int NL, bool GODE, bool WTREG, bool GD>
__global__ __launch_bounds__(WTREG ? 256 : 64 * kBwdWaves, (sizeof(R) == 4 && !WTREG ? 2 : 1)) void solve_bwd_kernel(const AdjArgs<R> a, const int method)
{
    constexpr int kWaves = WTREG ? 4 : kBwdWaves;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *wt = reinterpret_cast<R *>(smem_raw);
    R *rowsT = wt + (size_t)(NL > 1 ? NL - 1 : 1) * kMaxH * kMaxH;

    const int lane = threadIdx.x & 63;
    const int c8 = lane & 7, grp = lane >> 3;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    const int set = blockIdx.y;
    const int T = a.T;
    const int per_set = a.B / a.n_sets;
    const TableauData &tab = kTableau[method];
    const int S = tab.S;
    constexpr int kRows = NL;                         // rows of 64 in a stage record (h_1 .. h_NL) ...
    constexpr int kSlot = kRows * kWave + 8;          // ... followed by the stage state in 8 reals
    constexpr int kBuf = kRows * kWave + kWave;       // one slot of a wave's record ring: the rows + the state (8 of 64 used)
    constexpr int kRing = kBwdRing<R>;

    const R *__restrict__ nn_set = a.nn_p + (size_t)set * a.P;
    using ES = EdgeSlots<NL>;
    using Edge = std::conditional_t<kEdgeLds<R>, EdgeLds<R, NL>, EdgeRegs<R, NL>>;
    Edge E;
    if constexpr (kEdgeLds<R>) {
        R *edgeW = rowsT + 8 * kWave;
        R *edgeG = edgeW + ES::count * kWave + (size_t)wave * ES::count * kWave;
        edge_table_store<R, NL>(edgeW, nn_set, a.H, threadIdx.x, 64 * kWaves);
        for (int i = lane; i < ES::count * kWave; i += kWave) edgeG[i] = R(0);
        E.w = edgeW; E.gacc = edgeG; E.lane = lane;
    } else {
        const R livej = (lane < a.H) ? R(1) : R(0);
        const int j = (lane < a.H) ? lane : a.H - 1;
        const R *pout = nn_set + 9 * a.H + a.H + (size_t)(NL - 1) * ((size_t)a.H * a.H + a.H);
#pragma unroll
        for (int i = 0; i < ES::count; ++i) { E.w[i] = R(0); E.gacc[i] = R(0); }
#pragma unroll
        for (int i = 0; i < 9; ++i) E.w[ES::w1 + i] = livej * nn_set[j * 9 + i];
#pragma unroll
        for (int q = 0; q < 6; ++q) E.w[ES::w5 + q] = livej * pout[q * a.H + j];
    }
    OdeP<R> o;
    ode_load(o, a.ode_p + 17 * set);
    using WtPol = std::conditional_t<WTREG, WtRegs<NL>, WtLds<R>>;
    WtPol wtp;
    if constexpr (WTREG) wtp.load(nn_set, a.H, lane);
    else { wt_rot_store<R>(wt, nn_set, a.H, NL - 1, threadIdx.x, 64 * kWaves); wtp.wt = wt; }
    tableau_rowsT_store<R>(rowsT, method, threadIdx.x, 64 * kWaves);
    __syncthreads();

    R gwh[(NL > 1) ? NL - 1 : 1][kMaxH];
#pragma unroll
    for (int l = 0; l < ((NL > 1) ? NL - 1 : 1); ++l)
#pragma unroll
        for (int k = 0; k < kMaxH; ++k) gwh[l][k] = R(0);
    constexpr bool use_gd = GD;                   // Hill-term code (pow, log) only in the GD instantiation
    R go = R(0);                                  // lane p < 17: d/d(ode constant p), summed over this wave's trajectories
    R *rec = rowsT + 8 * kWave + (kEdgeLds<R> ? (size_t)(1 + kBwdWaves) * EdgeSlots<NL>::count * kWave : 0) +
             (size_t)wave * kRing * kBuf;
    // the DMA'd part of a record = NL rows of 64 reals; each row is one (fp32) or two (fp64) 4-byte-per-lane DMA instructions
    auto rec_dma = [&](const R *__restrict__ src, R *dst) {
#pragma unroll
        for (int l = 0; l < kRows; ++l) {
            if constexpr (sizeof(R) == 4) {
                __builtin_amdgcn_global_load_lds(src + l * kWave + lane, (__attribute__((address_space(3))) void *)(dst + l * kWave), 4, 0, 0);
            } else {
                const float *s32 = reinterpret_cast<const float *>(src + l * kWave);
                float *d32 = reinterpret_cast<float *>(dst + l * kWave);
                __builtin_amdgcn_global_load_lds(s32 + lane, (__attribute__((address_space(3))) void *)d32, 4, 0, 0);
                __builtin_amdgcn_global_load_lds(s32 + kWave + lane, (__attribute__((address_space(3))) void *)(d32 + kWave), 4, 0, 0);
            }
        }
        // the compact stage state (8 reals) rides along, so that it is prefetched one stage ahead like the rows: read
        // with scalar loads at the point of use it cost an exposed HBM miss per stage (measured: adjoint 8.1 -> 12.7 ms)
        const float *s32 = reinterpret_cast<const float *>(src + kRows * kWave);
        float *d32 = reinterpret_cast<float *>(dst + kRows * kWave);
        if (lane < 8 * (int)(sizeof(R) / 4))
            __builtin_amdgcn_global_load_lds(s32 + lane, (__attribute__((address_space(3))) void *)d32, 4, 0, 0);
    };

    // wave-major distribution: a batch smaller than 8 x the grid keeps every CU busy with fewer active waves each
    for (int bi = wave * gridDim.x + blockIdx.x; bi < per_set; bi += gridDim.x * kWaves) {
        const int b = set * per_set + bi;
        const R *__restrict__ tg = a.t + (a.t_batched ? (size_t)b * T : 0);
        const R *__restrict__ tape = a.tape + (size_t)b * a.max_steps * 8;
        const int *__restrict__ tseg = a.tape_seg + (size_t)b * a.max_steps;
        const R *__restrict__ stg = a.tape_stage + (size_t)b * a.max_steps * 6 * kSlot;
        const R *__restrict__ gyb = a.gy + (size_t)b * T * 6;
        const int n = a.nsteps[b] < a.max_steps ? a.nsteps[b] : a.max_steps;     // never walk past the tape, whatever the caller passes
        const bool ok = a.status[b] == HODE_ST_OK;
        R lam = R(0);                              // cotangent of the state, replicated per 8-lane group
        int knext = T - 1;                         // grid interval of the step after the current one
        // Stage records stream HBM -> LDS by DMA (no VGPR destination): while stage s is processed from one
        // half of the wave's double buffer, the record of the next stage (also across step boundaries) lands
        // in the other half.
        // records are consumed in the order (n-1, S-1), (n-1, S-2), ..., (0, 0); (pst, ps) walks kBwdAhead records ahead
        constexpr int kAhead = kBwdAhead<R>;
        constexpr int kDmaOps = kRows * (int)(sizeof(R) / 4) + 1;                 // DMA instructions per record
        constexpr int kWaitYounger = 0x0f70 | ((kDmaOps * (kAhead - 1)) & 15) | (((kDmaOps * (kAhead - 1)) >> 4) << 14);
        static_assert(kDmaOps * (kAhead - 1) < 64, "vmcnt field");
        int cur = 0, pst = n - 1, ps = S - 1, ahead = 0;                         // ahead = records issued and not yet consumed
        auto issue_next = [&]() {
            if (pst < 0) return;
            rec_dma(stg + ((size_t)pst * 6 + ps) * kSlot, rec + ((cur + ahead) % kRing) * kBuf);
            ++ahead;
            if (--ps < 0) { ps = S - 1; --pst; }
        };
        for (int j = 0; j < kAhead; ++j) issue_next();
#pragma unroll 1
        for (int st = n - 1; st >= 0; --st) {
            const int kraw = tseg[st];
            const int k = kraw & (kSegClosed - 1);
            // cotangents of the grid rows this step produced (row k+1 and any repeated rows that follow it).  The last step
            // of a FAILED trajectory: if it closed its interval, row k+1 and the copies behind it (zero-length intervals up
            // to the interval that failed) were still written; if it did not, nothing after row k was.
            int hi = knext;
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
                // six wave-uniform (scalar) loads + selects: vector-memory traffic stays reserved for the DMAs
                const R *__restrict__ gr = gyb + (size_t)r * 6;
                const R g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3], g4 = gr[4], g5 = gr[5];
                lam += (c8 == 0) ? g0 : (c8 == 1) ? g1 : (c8 == 2) ? g2 : (c8 == 3) ? g3 : (c8 == 4) ? g4 : (c8 == 5) ? g5 : R(0);
            }
            knext = k;
            const R tc = tape[(size_t)st * 8 + 0], h = tape[(size_t)st * 8 + 1];
            const R t0 = tg[k], t1 = tg[k + 1];
            const R v0 = inp_at_b(a.tvns, a.tvns_mode, b, T, k), v1 = inp_at_b(a.tvns, a.tvns_mode, b, T, k + 1);
            const R d0 = inp_at_b(a.gd, a.gd_mode, b, T, k), d1 = inp_at_b(a.gd, a.gd_mode, b, T, k + 1);
            const R inv_len = first_lane(R(1) / (t1 - t0));
            const R dv = first_lane(v1 - v0), dd = first_lane(d1 - d0);

            // reverse sweep over the stages.  kb_s = h (b_s lam + sum_{j>s} a_js Z_j),  Z_s = J_s^T kb_s
            R ZZ = R(0);
#pragma unroll 1
            for (int s = S - 1; s >= 0; --s) {
                // record (st, s) was DMA'd into ring slot `cur` kBwdAhead stages ago.  The only outstanding VMEM ops are our
                // DMAs and they complete in issue order: wait until at most the YOUNGER records are still in flight
                if (ahead == kAhead && kAhead > 1) __builtin_amdgcn_s_waitcnt(kWaitYounger);
                else __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0): tail of the trajectory
                __builtin_amdgcn_wave_barrier();
                --ahead;                                       // slot `cur` is being consumed; the slot behind the ring frees up
                {
                    const int keep = cur;
                    cur = (cur + 1) % kRing;                   // issue_next() addresses slots relative to the NEXT record
                    issue_next();
                    cur = keep;
                }
                MlpActs<R, NL> ac;
#pragma unroll
                for (int l = 0; l < NL; ++l) ac.h[l] = rec[cur * kBuf + l * kWave + lane];
                // the stage state, back in the replicated layout: lane l reads slot l & 7 of the compact tail of the record
                // (six uniform values kept as six VGPRs instead cost the kernel its last registers: 108 B of scratch, 8.1 -> 12.5 ms)
                const R Ys = rec[cur * kBuf + kRows * kWave + c8];
                const R *__restrict__ hrows = rec + cur * kBuf;      // this stage's rows stay valid until the ring comes round
                cur = (cur + 1) % kRing;
                // tableau scalars come from LDS with the record (one wait), not from constant memory (an s_load + wait per stage)
                const R bw_s = rowsT[6 * kWave + s], c_s = rowsT[6 * kWave + 8 + s];
                const R kb = h * rfma(bw_s, lam, group_sum8(rowsT[s * kWave + lane] * ZZ));
                const R ts = rfma(c_s, h, tc);
                const R al = (ts - t0) * inv_len;
                const R gdv = rfma(al, dd, d0);
                R gde = R(0);
                if constexpr (use_gd) gde = gd_effect(o, gdv);
                const R Z = rhs_vjp<R, NL, GODE, false>(E, gwh, wtp, o, ts, Ys, rfma(al, dv, v0), gde, gdv, use_gd, lane, ac, kb,
                                                        go, nullptr, hrows);
                ZZ = (grp == s) ? Z : ZZ;
            }
            lam += group_sum8(rowsT[7 * kWave + lane] * ZZ);
        }
        // rows 1..kf are copies of x0 (the grid starts with repeated times; kf = first interval of positive length, which is
        // also the interval of step 0 and the interval a trajectory without any accepted step failed in): their
        // cotangents go straight to gx0, next to row 0's
        int kf = 0;
        while (kf + 1 < T && !(tg[kf + 1] > tg[kf])) ++kf;
        for (int r = 0; r <= kf; ++r) {
            const R *__restrict__ gr = gyb + (size_t)r * 6;
            const R g0 = gr[0], g1 = gr[1], g2 = gr[2], g3 = gr[3], g4 = gr[4], g5 = gr[5];
            lam += (c8 == 0) ? g0 : (c8 == 1) ? g1 : (c8 == 2) ? g2 : (c8 == 3) ? g3 : (c8 == 4) ? g4 : (c8 == 5) ? g5 : R(0);
        }
        if (lane < 6) a.gx0[(size_t)b * 6 + lane] = lam;
    }

    // ---- epilogue: ONE gradient row per workgroup, no floating-point atomics ------------------------------------------------
    // The 8 waves add their accumulators in a FIXED order into the (now dead) LDS image of the transposed matrices --
    // [layer][row][col], natural order -- and the workgroup stores the sums, coalesced, to its row of a.partials; a second
    // kernel (adj_reduce_kernel) adds the rows of a parameter set in workgroup order into gnn / gode.  Every summation order is
    // fixed by the launch geometry: the same inputs give the same bits (the reference's CPU training is deterministic; the
    // 256 x 12 k coalesced atomics this replaces were not).  Without a partials area (a caller with more parameter sets than
    // rows) the sums leave through atomics as before.
    const int nthreads = 64 * kWaves;
    const size_t rowlen = adj_partial_rowlen(a.P);
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    R *__restrict__ prow = a.partials ? a.partials + (size_t)wg * rowlen : nullptr;
    auto emit = [&](R *__restrict__ gp, size_t off, R v) {          // gp = the set's slice of gnn (atomic route)
        if (prow) prow[off] = v; else atomic_add(gp + off, v);
    };
    if (a.gnn) {
        R *__restrict__ gp = a.gnn + (size_t)set * a.P;
        constexpr int kHid = (NL > 1 ? NL - 1 : 0) * kMaxH * kMaxH;
        __syncthreads();
        for (int i = threadIdx.x; i < kHid; i += nthreads) wt[i] = R(0);
        __syncthreads();
        for (int w = 0; w < kWaves; ++w) {
            if (wave == w) {
#pragma unroll
                for (int l = 0; l < NL - 1; ++l)
#pragma unroll
                    for (int r = 0; r < kMaxH; ++r) wt[l * kMaxH * kMaxH + lane * kMaxH + wcol<R>(r, lane)] += gwh[l][r];
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < kHid; i += nthreads) {
            const int l = i >> 12, row = (i >> 6) & 63, col = i & 63;
            if (row < a.H && col < a.H)
                emit(gp, 9 * a.H + a.H + (size_t)l * ((size_t)a.H * a.H + a.H) + (size_t)row * a.H + col, wt[i]);
        }
        const int H = a.H;
        const size_t off_out = (size_t)9 * H + H + (size_t)(NL - 1) * ((size_t)H * H + H);
        auto emit_edge = [&](int slot, int j, R v) {
            if (slot < ES::b) { if (j < H) emit(gp, j * 9 + slot, v); }
            else if (slot < ES::w5) {
                const int l = slot - ES::b;
                if (j < H) emit(gp, 9 * H + (l == 0 ? 0 : H + (size_t)(l - 1) * ((size_t)H * H + H) + (size_t)H * H) + j, v);
            } else if (slot < ES::b5) { if (j < H) emit(gp, off_out + (slot - ES::w5) * H + j, v); }
            else if (j < 6) emit(gp, off_out + 6 * H + j, v);
        };
        if constexpr (kEdgeLds<R>) {
            // edge accumulators: the per-wave LDS tables summed in wave order
            const R *edgeG0 = rowsT + 8 * kWave + ES::count * kWave;
            for (int i = threadIdx.x; i < ES::count * kWave; i += nthreads) {
                R v = R(0);
                for (int w = 0; w < kWaves; ++w) v += edgeG0[(size_t)w * ES::count * kWave + i];
                emit_edge(i >> 6, i & 63, v);
            }
        } else {
            // edge accumulators live in registers (fp64 parity build): through the dead image, wave by wave
            __syncthreads();
            for (int i = threadIdx.x; i < ES::count * kWave; i += nthreads) wt[i] = R(0);
            __syncthreads();
            for (int w = 0; w < kWaves; ++w) {
                if (wave == w) {
#pragma unroll
                    for (int sl = 0; sl < ES::count; ++sl) wt[sl * kWave + lane] += E.G(sl);
                }
                __syncthreads();
            }
            for (int i = threadIdx.x; i < ES::count * kWave; i += nthreads) emit_edge(i >> 6, i & 63, wt[i]);
        }
    }
    if constexpr (GODE) {
        if (a.gode) {
            // lane p < 17 of every wave holds its share of d/d(ode constant p): summed in wave order
            __syncthreads();
            if (lane < 17) wt[wave * 32 + lane] = go;
            __syncthreads();
            if (threadIdx.x < 17) {
                R v = R(0);
                for (int w = 0; w < kWaves; ++w) v += wt[w * 32 + threadIdx.x];
                if (prow) prow[a.P + threadIdx.x] = v; else atomic_add(a.gode + 17 * set + threadIdx.x, v);
            }
        }
    }
}

// second pass of the adjoint's gradient reduction: row sums in a FIXED order, added to gnn / gode.  A workgroup takes 32
// columns; its eight row groups add the rows w = r, r + 8, ... (four loads in flight each) and are then combined in group order --
// the same order for the same launch geometry, so the result is reproducible bit for bit.  (One thread per column walking all
// rows took 95 us for 256 rows of 54 KB: a chain of 256 dependent-issue loads; this form is bandwidth-bound, ~10 us.)
template <typename R>
__global__ __launch_bounds__(256) void adj_reduce_kernel(const R *__restrict__ partials, const int rowlen, const int blocks_per_set, const int P,
                                                         R *__restrict__ gnn, R *__restrict__ gode)
{
    __shared__ R part[8][32];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5, i = blockIdx.x * 32 + c, set = blockIdx.y;
    R v = R(0);
    if (i < P + 17) {
        const R *__restrict__ p = partials + (size_t)set * blocks_per_set * rowlen + i;
        int w = r;
        for (; w + 24 < blocks_per_set; w += 32) {
            const R a0 = p[(size_t)w * rowlen], a1 = p[(size_t)(w + 8) * rowlen], a2 = p[(size_t)(w + 16) * rowlen], a3 = p[(size_t)(w + 24) * rowlen];
            v += a0; v += a1; v += a2; v += a3;
        }
        for (; w < blocks_per_set; w += 8) v += p[(size_t)w * rowlen];
    }
    part[r][c] = v;
    __syncthreads();
    if (r == 0 && i < P + 17 && !(i < P ? gnn == nullptr : gode == nullptr)) {
        R t = part[0][c];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += part[q][c];
        if (i < P) gnn[(size_t)set * P + i] += t;
        else gode[(size_t)set * 17 + (i - P)] += t;
    }
}

void launch_adj_reduce(hipStream_t s, const float *partials, int rowlen, int blocks_per_set, int n_sets, int P, float *gnn, float *gode)
{
    hipLaunchKernelGGL(adj_reduce_kernel<float>, dim3((P + 17 + 31) / 32, n_sets), dim3(256), 0, s, partials, rowlen, blocks_per_set, P, gnn, gode);
}

// compute units of the current device (one adjoint workgroup per CU); queried once per device, never assumed
static int device_cu_count()
{
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

template <typename R, int NL, bool GODE, bool WTREG, bool GD> static int launch_bwd_g(hipStream_t s, const AdjArgs<R> &a, int method)
{
    constexpr int kW = WTREG ? 4 : kBwdWaves;
    const int per_set = a.B / a.n_sets;
    const int cus = device_cu_count();
    int blocks = per_set < cus ? per_set : cus;   // one workgroup per CU, its waves loop over trajectories (wave-major)
    // never more workgroups than CUs when that is avoidable: 86 x 3 = 258 would leave two workgroups for a second round
    // that doubles the kernel time
    if (a.n_sets > 1 && blocks * a.n_sets > cus) blocks = cus / a.n_sets;
    if (blocks < 1) blocks = 1;
    const size_t lds = bwd_lds_elems<R, NL>() * sizeof(R);
    dim3 grid(blocks, a.n_sets), block(64 * kW);
    auto kern = solve_bwd_kernel<R, NL, GODE, WTREG, GD>;
    if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return HODE_ELAUNCH;
    AdjArgs<R> a2 = a;
    if (blocks * a.n_sets > a.partial_rows) a2.partials = nullptr;       // more parameter sets than rows: atomics
    hipLaunchKernelGGL(kern, grid, block, lds, s, a2, method);
    if (a2.partials && (a.gnn || (GODE && a.gode))) {
        const int rowlen = (int)adj_partial_rowlen(a.P);
        hipLaunchKernelGGL(adj_reduce_kernel<R>, dim3((a.P + 17 + 31) / 32, a.n_sets), dim3(256), 0, s, a2.partials, rowlen, blocks, a.P,
                           a.gnn, GODE ? a.gode : (R *)nullptr);
    }
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <typename R, int NL, bool GODE, bool WTREG> static int launch_bwd_k(hipStream_t s, const AdjArgs<R> &a, int method)
{
    return a.gd_mode != 0 ? launch_bwd_g<R, NL, GODE, WTREG, true>(s, a, method) : launch_bwd_g<R, NL, GODE, WTREG, false>(s, a, method);
}

#ifdef HODE_LAB
// Lab library only.  HODE_BWD_WT=regs: transposed matrices in registers (1 wave/SIMD, 13.9 ms against 8.2 ms);
// HODE_BWD=split: the two-kernel adjoint of lab/hode_solve_bwd_split.hip (fp32, L >= 2; 8.8 ms against 8.0 ms at 4 096 x 241,
// DESIGN.md section 6.2).  Both kept as the reproducible record of those experiments.
static bool bwd_wt_in_regs()
{
    static const bool v = [] { const char *e = getenv("HODE_BWD_WT"); return e && e[0] == 'r'; }();
    return v;
}
bool split_adjoint_enabled()
{
    static const bool v = [] { const char *e = getenv("HODE_BWD"); return e && e[0] == 's'; }();
    return v;
}
#endif

template <typename R, int NL> static int launch_bwd_nl(hipStream_t s, const AdjArgs<R> &a, int method)
{
#ifdef HODE_LAB
    if constexpr (sizeof(R) == 4) {
        if (bwd_wt_in_regs())
            return a.gode ? launch_bwd_k<R, NL, true, true>(s, a, method) : launch_bwd_k<R, NL, false, true>(s, a, method);
    }
#endif
    return a.gode ? launch_bwd_k<R, NL, true, false>(s, a, method) : launch_bwd_k<R, NL, false, false>(s, a, method);
}

#ifdef HODE_LAB
// Lab library only.  HODE_BWD=fused: the one-role kernel also for the fp32 shapes the wave-specialised kernel takes -- A/B timing
// (tools/ws_dbg.sh) and the parity test that compares the two
static bool bwd_force_fused()
{
    static const bool v = [] { const char *e = getenv("HODE_BWD"); return e && e[0] == 'f'; }();
    return v;
}
#else
constexpr bool bwd_force_fused() { return false; }
#endif

template <typename R> int launch_solve_bwd(hipStream_t s, const AdjArgs<R> &a, int L, int method)
{
    if constexpr (sizeof(R) == 4) {
        if (L >= 2 && !bwd_force_fused()) {
            const int rc = launch_solve_bwd_ws(s, a, L, method, device_cu_count());
            if (rc != HODE_EUNSUPPORTED) return rc;
        }
    }
#ifdef HODE_LAB
    if constexpr (sizeof(R) == 4) {
        if (L >= 2 && a.tape_delta && split_adjoint_enabled() && !bwd_wt_in_regs()) return launch_solve_bwd_split(s, a, L, method);
    }
#endif
    switch (L) {
    case 1: return launch_bwd_nl<R, 1>(s, a, method);
    case 2: return launch_bwd_nl<R, 2>(s, a, method);
    case 3: return launch_bwd_nl<R, 3>(s, a, method);
    case 4: return launch_bwd_nl<R, 4>(s, a, method);
    }
    return HODE_EUNSUPPORTED;
}
template int launch_solve_bwd<float>(hipStream_t, const AdjArgs<float> &, int, int);
template int launch_solve_bwd<double>(hipStream_t, const AdjArgs<double> &, int, int);

// ------------------------------------------------------------------------------------------
// K5: RHS backward.  Replaces torch autograd over ode_residual in the physics loss
// (reference models/hybrid_ode_nn.py:318-330).  Same mapping; one sample per wave.
template <typename R, int NL, bool GODE>
__global__ __launch_bounds__(256, 1) void rhs_bwd_kernel(const RhsArgs<R> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    R *wt = reinterpret_cast<R *>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wave = first_lane((int)(threadIdx.x >> 6));
    MlpRegs<R, NL> W;
    mlp_load<R, NL>(W, a.nn_p, a.H, lane, wt + (size_t)wave * kStageElems);   // the image area doubles as staging
    OdeP<R> o;
    ode_load(o, a.ode_p);
    __syncthreads();
    wt_rot_store<R>(wt, a.nn_p, a.H, NL - 1, threadIdx.x, 256);
    __syncthreads();
    using ES = EdgeSlots<NL>;
    EdgeRegs<R, NL> E;
#pragma unroll
    for (int i = 0; i < ES::count; ++i) { E.w[i] = R(0); E.gacc[i] = R(0); }
#pragma unroll
    for (int i = 0; i < 9; ++i) E.w[ES::w1 + i] = W.w1[i];
#pragma unroll
    for (int q = 0; q < 6; ++q) E.w[ES::w5 + q] = W.w5[q];
    R gwh[(NL > 1) ? NL - 1 : 1][kMaxH];
#pragma unroll
    for (int l = 0; l < ((NL > 1) ? NL - 1 : 1); ++l)
#pragma unroll
        for (int k = 0; k < kMaxH; ++k) gwh[l][k] = R(0);
    R go = R(0);
    for (int s = blockIdx.x * 4 + wave; s < a.B; s += gridDim.x * 4) {
        const R Y = ((lane & 7) < 6) ? a.x[(size_t)s * 6 + (lane & 7)] : R(0);     // replicated layout (rhs_eval)
        const R kb = (lane < 6) ? a.gout[(size_t)s * 6 + lane] : R(0);
        const R t = a.t ? a.t[s] : R(0);
        const R meal = a.meal ? a.meal[s] : R(0);
        const R tvns = a.tvns ? a.tvns[s] : R(0);
        const R gdv = a.gd ? a.gd[s] : R(0);
        const R gde = a.gd ? gd_effect(o, gdv) : R(0);
        MlpActs<R, NL> ac;
        (void)rhs_eval<R, NL, true>(W, o, t, Y, meal, tvns, gde, lane, &ac);
        R gt;
        const R Z = rhs_vjp<R, NL, GODE, true>(E, gwh, WtLds<R>{wt}, o, t, Y, tvns, gde, gdv, a.gd != nullptr, lane, ac, kb, go, &gt);
        if (lane < 6) a.gx[(size_t)s * 6 + lane] = Z;
        if (a.gt && lane == 0) a.gt[s] = gt;
    }
    // ---- the four waves' accumulators are summed in LDS first (the image of the transposed matrices is dead now) and ONE wave
    //      flushes: the flush is P atomics per wave on the same 54 KB, and it -- not the arithmetic -- is what the kernel costs.  Round
    //      3 launched one wave per sample, each flushing its own full gradient: 640 samples of the physics term at the reference's
    //      batch (32 windows x 20 indices) = 8.6 M same-address atomics, 0.97 ms -- the largest kernel of the class-path step.
    R *red = wt;                                                     // [(NL-1) * 64 + ES::count + 1][64]
    constexpr int kHid = ((NL > 1) ? NL - 1 : 0) * kMaxH;
    __syncthreads();
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int l = 0; l < ((NL > 1) ? NL - 1 : 0); ++l)
#pragma unroll
                for (int k = 0; k < kMaxH; ++k) red[(l * kMaxH + k) * kWave + lane] = gwh[l][k];
#pragma unroll
            for (int i = 0; i < ES::count; ++i) red[(kHid + i) * kWave + lane] = E.gacc[i];
            red[(kHid + ES::count) * kWave + lane] = go;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int l = 0; l < ((NL > 1) ? NL - 1 : 0); ++l)
#pragma unroll
                for (int k = 0; k < kMaxH; ++k) gwh[l][k] += red[(l * kMaxH + k) * kWave + lane];
#pragma unroll
            for (int i = 0; i < ES::count; ++i) E.gacc[i] += red[(kHid + i) * kWave + lane];
            go += red[(kHid + ES::count) * kWave + lane];
        }
        __syncthreads();
    }
    if (wave != 0) return;
    if (a.gnn) {
        hidden_flush<R, NL>(gwh, a.gnn, a.H, lane);
        edge_flush<R, NL>(E, a.gnn, a.H, lane);
    }
    if constexpr (GODE) {
        if (a.gode && lane < 17) atomic_add(a.gode + lane, go);
    }
}

template <typename R, int NL> static int launch_rhs_bwd_nl(hipStream_t s, const RhsArgs<R> &a)
{
    // sixteen samples per wave before a workgroup is added (a sample is ~3 us of a wave, a workgroup's flush ~1.5 us of contended
    // atomics for everybody): 640 samples -> 10 workgroups, 81 920 (4 096 windows x 20 indices) -> 256 with 80 samples per wave
    int blocks = (a.B + 63) / 64;
    if (blocks > 256) blocks = 256;
    if (blocks < 1) return HODE_OK;
    size_t lds = (size_t)(NL > 1 ? NL - 1 : 1) * kMaxH * kMaxH * sizeof(R);
    if (lds < 4 * (size_t)kStageElems * sizeof(R)) lds = 4 * (size_t)kStageElems * sizeof(R);
    {
        const size_t red = ((size_t)(NL > 1 ? NL - 1 : 0) * kMaxH + EdgeSlots<NL>::count + 1) * kWave * sizeof(R);   // the workgroup reduction
        if (lds < red) lds = red;
    }
    if (a.gode) {
        auto kern = rhs_bwd_kernel<R, NL, true>;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return HODE_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a);
    } else {
        auto kern = rhs_bwd_kernel<R, NL, false>;
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return HODE_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a);
    }
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

template <typename R> int launch_rhs_bwd(hipStream_t s, const RhsArgs<R> &a, int L)
{
    switch (L) {
    case 1: return launch_rhs_bwd_nl<R, 1>(s, a);
    case 2: return launch_rhs_bwd_nl<R, 2>(s, a);
    case 3: return launch_rhs_bwd_nl<R, 3>(s, a);
    case 4: return launch_rhs_bwd_nl<R, 4>(s, a);
    }
    return HODE_EUNSUPPORTED;
}
template int launch_rhs_bwd<float>(hipStream_t, const RhsArgs<float> &, int);
template int launch_rhs_bwd<double>(hipStream_t, const RhsArgs<double> &, int);

}  // namespace hode
