// hode_optim.hip -- K6: fused global-norm clip + Adam, and the fused MSE loss / cotangent pass.
//
// K6 replaces torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0) + torch.optim.Adam.step()
// (reference train/train_hybrid.py:255-261, 438-441).  The parameter vector is tiny (13 510
// floats = 54 KB): two launches on one stream -- squared-norm reduction (one workgroup, fixed summation order), then the update that
// reads the norm from device memory -- no host synchronisation, graph-capturable.
// The MSE pass replaces F.mse_loss(predictions, observations) and its autograd
// (models/hybrid_ode_nn.py:294): one read of y and obs, one write of dLoss/dy; HBM-bound,
// 16-byte vectorised, grid-stride.
#include "hode_device.h"
#include "hode_kernels.h"

namespace hode {

// Squared gradient norm, DETERMINISTIC: one workgroup, every thread sums its strided elements in index order, a fixed
// butterfly inside each wave, then thread 0 adds the 16 wave sums in wave order.  No atomics, so the clip coefficient is
// the same bits on every rank and in every run (data-parallel replicas stay identical without a parameter broadcast).
// 13 510 parameters = 13 elements per thread: the launch is latency, not bandwidth.
__global__ __launch_bounds__(1024) void sqnorm_kernel(int64_t n, const float *__restrict__ g, float scale, float *out)
{
    __shared__ float part[16];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float v = g[i] * scale;
        acc = rfma(v, v, acc);
    }
    acc = wave_allsum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += part[w];
        *out = t;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(int64_t n, float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, float lr, float b1,
                                                   float b2, float eps, float bc1, float bc2, float max_norm,
                                                   float grad_scale, float wd, const float *__restrict__ sqnorm)
{
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    float coef = grad_scale;
    if (max_norm > 0.f) {
        const float tn = sqrtf(*sqnorm);
        const float c = max_norm / (tn + 1e-6f);
        coef *= (c < 1.f) ? c : 1.f;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * coef;
        const float pi = p[i];
        if (wd != 0.f) gi = rfma(wd, pi, gi);
        const float mi = rfma(b1, m[i], (1.f - b1) * gi);
        const float vi = rfma(b2, v[i], (1.f - b2) * gi * gi);
        m[i] = mi;
        v[i] = vi;
        // torch.optim.Adam (no amsgrad): p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
        const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

int launch_adam(hipStream_t s, int64_t n, float *p, const float *g, float *m, float *v, float lr, float b1, float b2,
                float eps, int step, float max_norm, float grad_scale, float wd, void *scratch)
{
    if (n <= 0) return HODE_OK;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    float *sq = (float *)scratch;
    if (hipMemsetAsync(sq, 0, 8, s) != hipSuccess) return HODE_ELAUNCH;
    if (max_norm > 0.f) hipLaunchKernelGGL(sqnorm_kernel, dim3(1), dim3(1024), 0, s, n, g, grad_scale, sq);
    const float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, n, p, g, m, v, lr, b1, b2, eps, bc1, bc2, max_norm,
                       grad_scale, wd, sq);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

__global__ __launch_bounds__(256) void mse_kernel(int64_t n4, int64_t n, const float *__restrict__ y,
                                                  const float *__restrict__ obs, float scale, double *loss,
                                                  float *__restrict__ gy)
{
    double acc = 0.0;
    const float4 *y4 = reinterpret_cast<const float4 *>(y), *o4 = reinterpret_cast<const float4 *>(obs);
    float4 *g4 = reinterpret_cast<float4 *>(gy);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (int64_t i = tid; i < n4; i += stride) {
        const float4 a = y4[i], b = o4[i];
        const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, dw = a.w - b.w;
        // squares and sums in fp64: the value then does not depend on how a launch groups the elements into 16-byte words -- a batch
        // summed piece by piece (tape-budget chunks of any size, also odd ones) gives the sum of the whole to 1e-16, not 1e-9
        acc += ((double)dx * dx + (double)dy * dy) + ((double)dz * dz + (double)dw * dw);
        if (gy) g4[i] = make_float4(2.f * scale * dx, 2.f * scale * dy, 2.f * scale * dz, 2.f * scale * dw);
    }
    for (int64_t i = n4 * 4 + tid; i < n; i += stride) {   // tail
        const float d = y[i] - obs[i];
        acc += (double)d * d;
        if (gy) gy[i] = 2.f * scale * d;
    }
    // one fp64 atomic per WORKGROUP: 8 192 same-address atomics (one per wave of a 2 048-block grid) serialise at the memory
    // side (~12 ns each = 0.1 ms, 6x the time the 71 MB of traffic take)
    __shared__ double part[4];
    acc = wave_allsum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomic_add(loss, (part[0] + part[1]) + (part[2] + part[3]));
}

int launch_mse(hipStream_t s, int64_t n, const float *y, const float *obs, float scale, double *loss, float *gy)
{
    if (n <= 0) return HODE_OK;
    // the 16-byte path needs 16-byte aligned pointers; anything else takes the scalar tail loop
    const bool aligned = (((uintptr_t)y | (uintptr_t)obs | (uintptr_t)gy) & 15) == 0;
    const int64_t n4 = aligned ? n / 4 : 0;
    int blocks = (int)(((aligned ? n4 : n) + 255) / 256);
    if (blocks > 1024) blocks = 1024;               // 4 workgroups per CU: enough loads in flight for HBM, 1 024 atomics
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mse_kernel, dim3(blocks), dim3(256), 0, s, n4, n, y, obs, scale, loss, gy);
    return hipGetLastError() == hipSuccess ? HODE_OK : HODE_ELAUNCH;
}

}  // namespace hode
