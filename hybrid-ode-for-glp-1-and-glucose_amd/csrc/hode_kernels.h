// hode_kernels.h -- host-visible argument blocks and launchers shared by the .hip files and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hode.h"

namespace hode {

template <typename R> struct SolveArgs {
    int B, T, t_batched, meal_mode, tvns_mode, gd_mode, n_sets, H, P, max_steps;
    const R *x0, *t, *meal, *tvns, *gd, *ode_p, *nn_p;
    R rtol, atol;
    R *y;
    int32_t *status, *nsteps, *nfev;
    R *tape;            // [B][max_steps][8] = {t, h, t0, 1/(t1-t0), v0, v1-v0, d0, d1-d0}: the step + the constants of its grid interval
    int32_t *tape_seg;  // [B][max_steps] grid interval of each accepted step (| kSegClosed)
    R *tape_stage;      // [B][max_steps][6 stages][tape_slot_elems]: layer activations + stage state of every accepted step
    int L;              // hidden layers (plain count)
    int act;            // HODE_ACT_* (generic kernels; the tuned ones are ReLU)
    int nn_stride;      // reals between the networks of consecutive parameter sets: P, or 0 for one shared network (HODE_LAYERS_NN_SHARED)
};

template <typename R> struct AdjArgs {
    int B, T, t_batched, meal_mode, tvns_mode, gd_mode, n_sets, H, P, max_steps;
    const R *t, *meal, *tvns, *gd, *ode_p, *nn_p;
    const int32_t *nsteps, *status;
    const R *tape;
    const int32_t *tape_seg;
    const R *tape_stage;
    const R *gy;
    R *gx0, *gnn, *gode;
    R *tape_delta;      // fp32 tuned shapes: [B][max_steps][6][delta_slot_elems] scratch of the split adjoint (part of the tape)
    R *partials;        // tuned shapes: [adj_partial_rows(B)][adj_partial_rowlen(P)] per-workgroup gradient sums (tail of the tape)
    int partial_rows;
    int act;            // HODE_ACT_* (generic kernels)
};

template <typename R> struct RhsArgs {
    int B, H, P;
    const R *x, *t, *meal, *tvns, *gd, *ode_p, *nn_p;
    R *out;                 // fwd
    const R *gout;          // bwd
    R *gx, *gt, *gnn, *gode;
    int act;                // HODE_ACT_* (generic kernels)
};

template <typename R> int launch_solve_fwd(hipStream_t s, const SolveArgs<R> &a, int L, int method);
#ifdef HODE_LAB
// Experiment kernels (csrc/lab/): compiled into hode/lab/libhode_lab.so only (make lab).  The product
// library libhode.so is built without HODE_LAB: none of these kernels, no environment-variable dispatch on the call path.
int launch_solve_fwd_wg(hipStream_t s, const SolveArgs<float> &a, int L, int method);   // lab/hode_solve_fwd_wg.hip (fp32, L = 2..4)
int launch_solve_fwd_quad(hipStream_t s, const SolveArgs<float> &a, int L, int method); // lab/hode_solve_fwd_quad.hip (fp32, L = 2..4)
int launch_solve_fwd_rows(hipStream_t s, const SolveArgs<float> &a, int L, int method); // lab/hode_solve_fwd_rows.hip (fp32, L = 2..4)
#endif
template <typename R> int launch_solve_bwd(hipStream_t s, const AdjArgs<R> &a, int L, int method);
// wave-specialised fp32 adjoint (hode_solve_bwd_ws.hip; L = 2..4, needs the partial rows); HODE_EUNSUPPORTED -> solve_bwd_kernel
int launch_solve_bwd_ws(hipStream_t s, const AdjArgs<float> &a, int L, int method, int cus);
// second pass of the gradient reduction: rows of a.partials added in workgroup order (hode_solve_bwd.hip)
void launch_adj_reduce(hipStream_t s, const float *partials, int rowlen, int blocks_per_set, int n_sets, int P, float *gnn, float *gode);
template <typename R> int launch_rhs_fwd(hipStream_t s, const RhsArgs<R> &a, int L);
template <typename R> int launch_rhs_bwd(hipStream_t s, const RhsArgs<R> &a, int L);
// generic network path (hode_generic.hip): H <= 128, L <= 8, weights streamed from L2
template <typename R> int launch_solve_fwd_generic(hipStream_t s, const SolveArgs<R> &a, int method);
template <typename R> int launch_solve_bwd_generic(hipStream_t s, const AdjArgs<R> &a, int L, int method);
template <typename R> int launch_rhs_fwd_generic(hipStream_t s, const RhsArgs<R> &a, int L);
template <typename R> int launch_rhs_bwd_generic(hipStream_t s, const RhsArgs<R> &a, int L);
// shapes the tuned (register-resident) kernels are compiled for; everything else up to HODE_MAX_* takes the generic path
// The `L` argument of the C ABI carries the number of hidden layers in bits 0..7 and the activation (HODE_ACT_*) in bits 8..15.
inline int layers_of(int L) { return L & 0xff; }
inline int act_of(int L) { return (L >> 8) & 0xff; }
// (ReLU only: every other activation takes the generic kernels whatever the shape)
inline bool tuned_shape(int H, int L) { return act_of(L) == HODE_ACT_RELU && H <= 64 && layers_of(L) <= 4; }
int launch_adam(hipStream_t s, int64_t n, float *p, const float *g, float *m, float *v, float lr, float b1,
                float b2, float eps, int step, float max_norm, float grad_scale, float wd, void *scratch);
int launch_mse(hipStream_t s, int64_t n, const float *y, const float *obs, float scale, double *loss, float *gy);
int launch_selftest(hipStream_t s, int32_t *out);

// ---- data side (hode_datagen.hip) ----------------------------------------------------------------------------
// 4GI model parameters, in the order of include/hode.h (HODE_4GI_NPAR)
struct FourGIPar {
    double CLglc, CLglci, Qglc, VCglc, VPglc, CLins, VCins, Ke0ins, VCglp, VM_GLP, KM_GLP, CLglg, VCglg, CLgip, VCgip,
        Qgip, VPgip, GLCINS_S, EMAX_1, EC50_1, HILL_1, EMAX_4, EC50_4, FDGLP, FDGIP, FDGLG;
};
static_assert(sizeof(FourGIPar) == HODE_4GI_NPAR * sizeof(double), "parameter block layout");

struct GenArgs {
    int B, T, hv, n_meals, meals_per_subject, max_steps;
    int64_t subject0;
    double interval_min, rtol, atol, noise_cv;
    const double *bsl, *meal_time, *meal_size, *z;
    double *table;      // [B*T][9]
    int32_t *status;
    FourGIPar par;
};

struct WinArgs {
    const double *table;
    int ncols, col_time, col_meal, col_tvns;
    int col_state[6];   // glucose, insulin, glucagon, glp1, ge, ffa  (-1 = absent)
    double time_div;
    const int64_t *row0;  // [N] first table row of every window
    int64_t N, S;
    float *states, *meal, *tvns, *time;
};

int launch_4gi_generate(hipStream_t s, const GenArgs &a);
int launch_4gi_rhs(hipStream_t s, int B, int hv, const FourGIPar &p, const double *bsl, const double *y, const double *meal,
                   double *d);
int launch_4gi_windows(hipStream_t s, const WinArgs &a, int normalize, double *mean_std, void *scratch);
int launch_4gi_window_moments(hipStream_t s, const WinArgs &a, double *moments, void *scratch);

// interval index of an accepted step on the tape: bit 30 set = the step ended exactly on the grid point closing its interval
constexpr int kSegClosed = 1 << 30;

// tape = entries | interval indices | (256-byte aligned) stage tape
inline size_t tape_seg_offset(int B, int max_steps, size_t elem) { return (size_t)B * max_steps * 8 * elem; }
inline size_t tape_stage_offset(int B, int max_steps, size_t elem)
{
    size_t o = tape_seg_offset(B, max_steps, elem) + (size_t)B * max_steps * sizeof(int32_t);
    return (o + 255) & ~(size_t)255;
}
// reals per stage record: h_1..h_L (tuned path: L rows of 64; generic path: 2 L rows, two hidden units per lane) + 8 for the
// stage state
inline size_t tape_slot_elems(int H, int L) { return (size_t)(tuned_shape(H, L) ? layers_of(L) : 2 * layers_of(L)) * 64 + 8; }
// The split adjoint of the tuned fp32 path (lab/hode_solve_bwd_split.hip; lab library only, HODE_BWD=split) hands the layer
// cotangents of every stage from its propagation kernel to its accumulation kernel through HBM: delta_1..delta_L (L rows of
// 64) + {kb[6], t, tVNS} in 8 reals per stage, in a region of the tape behind the stage tape (the tape is the adjoint's
// workspace: nothing is allocated inside the library).  In the product library, and for other dtypes / shapes, there is no
// such region.
inline size_t delta_slot_elems(int L) { return (size_t)L * 64 + 8; }
#ifdef HODE_LAB
bool split_adjoint_enabled();       // HODE_BWD=split (hode_solve_bwd.hip); the default is the fused one-kernel adjoint
#else
constexpr bool split_adjoint_enabled() { return false; }
#endif
inline bool has_delta_tape(size_t elem, int H, int L) { return elem == 4 && tuned_shape(H, L) && layers_of(L) >= 2 && split_adjoint_enabled(); }
inline size_t tape_delta_offset(int B, int max_steps, size_t elem, int H, int L)
{
    size_t o = tape_stage_offset(B, max_steps, elem) + (size_t)B * max_steps * 6 * tape_slot_elems(H, L) * elem;
    return (o + 255) & ~(size_t)255;
}
// Gradient partials of the tuned adjoint: every workgroup writes ONE row [ dL/d nn_p (P) | dL/d ode_p (17) | pad ] and a second,
// fixed-order pass adds the rows into gnn / gode -- no floating-point atomics, so a training step is bit-reproducible run to
// run (the reference's CPU training is deterministic; fp32 atomics from 256 workgroups are not).  The rows live in the tail of
// the tape, the adjoint's workspace (nothing is allocated inside the library): at most kAdjPartialRows of them, never more
// than there are trajectories (a launch has at most one workgroup per trajectory).
constexpr int kAdjPartialRows = 1024;
__host__ __device__ inline int adj_partial_rows(int B) { return B < kAdjPartialRows ? B : kAdjPartialRows; }
__host__ __device__ inline size_t adj_partial_rowlen(int P) { return (size_t)(P + 17 + 63) / 64 * 64; }
inline size_t tape_partials_offset(int B, int max_steps, size_t elem, int H, int L)
{
    size_t o = tape_delta_offset(B, max_steps, elem, H, L);
    if (has_delta_tape(elem, H, L)) o += (size_t)B * max_steps * 6 * delta_slot_elems(L) * elem;
    return (o + 255) & ~(size_t)255;
}
inline size_t tape_total_bytes(int B, int max_steps, size_t elem, int H, int L)
{
    const size_t o = tape_partials_offset(B, max_steps, elem, H, L);
    // (generic shapes: the fp32 team kernels write gradient rows as well since round 4; fp64 and networks with more than four hidden
    //  matrices keep the coalesced atomics of hode_generic.hip)
    if (!tuned_shape(H, L) && elem != 4) return o;
    const int P = 9 * H + H + (layers_of(L) - 1) * (H * H + H) + 6 * H + 6;
    return o + (size_t)adj_partial_rows(B) * adj_partial_rowlen(P) * elem;
}
#ifdef HODE_LAB
int launch_solve_bwd_split(hipStream_t s, const AdjArgs<float> &a, int L, int method);   // lab/hode_solve_bwd_split.hip (fp32, tuned shapes)
#endif

}  // namespace hode
