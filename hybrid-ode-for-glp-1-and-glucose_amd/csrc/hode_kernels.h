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
    R *tape;            // [B][max_steps][8] = {t, h, y0..y5}
    int32_t *tape_seg;  // [B][max_steps] grid interval of each accepted step
    R *tape_stage;      // [B][max_steps][6 stages][L+1][64]: layer activations + stage state of every accepted step
    int L;
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
};

template <typename R> struct RhsArgs {
    int B, H, P;
    const R *x, *t, *meal, *tvns, *gd, *ode_p, *nn_p;
    R *out;                 // fwd
    const R *gout;          // bwd
    R *gx, *gt, *gnn, *gode;
};

template <typename R> int launch_solve_fwd(hipStream_t s, const SolveArgs<R> &a, int L, int method);
template <typename R> int launch_solve_bwd(hipStream_t s, const AdjArgs<R> &a, int L, int method);
template <typename R> int launch_rhs_fwd(hipStream_t s, const RhsArgs<R> &a, int L);
template <typename R> int launch_rhs_bwd(hipStream_t s, const RhsArgs<R> &a, int L);
int launch_adam(hipStream_t s, int64_t n, float *p, const float *g, float *m, float *v, float lr, float b1,
                float b2, float eps, int step, float max_norm, float grad_scale, float wd, void *scratch);
int launch_mse(hipStream_t s, int64_t n, const float *y, const float *obs, float scale, double *loss, float *gy);
int launch_selftest(hipStream_t s, int32_t *out);

// tape = entries | interval indices | (256-byte aligned) stage tape
inline size_t tape_seg_offset(int B, int max_steps, size_t elem) { return (size_t)B * max_steps * 8 * elem; }
inline size_t tape_stage_offset(int B, int max_steps, size_t elem)
{
    size_t o = tape_seg_offset(B, max_steps, elem) + (size_t)B * max_steps * sizeof(int32_t);
    return (o + 255) & ~(size_t)255;
}
inline size_t tape_total_bytes(int B, int max_steps, size_t elem, int L)
{
    return tape_stage_offset(B, max_steps, elem) + (size_t)B * max_steps * 6 * (L + 1) * 64 * elem;
}

}  // namespace hode
