// Micro-benchmark of ONE backward hidden layer per wave (outer product into 64 register accumulators + W^T product fed
// by 16 ds_read_b128), two operand-delivery schemes:
//   MODE 0  rotating operand: 2 row replications + 128 v_fmac_f32_dpp row_ror
//   MODE 1  shared scalar:    64 v_readlane + 128 v_fmac_f32 with an SGPR source
//   MODE 2  scalar reads only (64 v_readlane + 64 trivially dependent adds)       MODE 3  128 plain v_fmac (ceiling)
// Build: hipcc -O3 --offload-arch=gfx950 bwd_ubench.hip -o bwd_ubench ; prints cycles per layer per wave at 2 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N> __device__ __forceinline__ float fmac_ror(float acc, float a, float b)
{
    if constexpr (N == 0) { asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }
    else { asm("v_fmac_f32_dpp %0, %1, %2 row_ror:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(N)); }
    return acc;
}
__device__ __forceinline__ float fmac_s(float acc, float s, float v)
{
    asm("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "s"(s), "v"(v));
    return acc;
}
__device__ __forceinline__ float lane_scalar(float v, int j) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j)); }
__device__ __forceinline__ void rows_replicate(float h, float (&R)[4])
{
    float a = h, b = h;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    float a2 = a, b2 = b;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(a2));
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(b), "+v"(b2));
    R[0] = a; R[1] = b; R[2] = a2; R[3] = b2;
    asm volatile("s_nop 1" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]));   // DPP read hazard after the swaps
}
struct alignas(16) V4 { float v[4]; };

template <int G, int MODE> __device__ __forceinline__ void group(float (&gw)[64], const V4 *wt4, int lane, float d, float hin,
                                                                  const float (&Rh)[4], const float (&Rd)[4], float (&acc)[4])
{
    const V4 w0 = wt4[(4 * G + 0) * 64 + lane], w1 = wt4[(4 * G + 1) * 64 + lane], w2 = wt4[(4 * G + 2) * 64 + lane], w3 = wt4[(4 * G + 3) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MODE == 0) {
        constexpr int n = 4 * G;
#define O(q, i) gw[q * 16 + n + i] = fmac_ror<n + i>(gw[q * 16 + n + i], Rh[q], d);
        O(0,0) O(1,0) O(2,0) O(3,0) O(0,1) O(1,1) O(2,1) O(3,1) O(0,2) O(1,2) O(2,2) O(3,2) O(0,3) O(1,3) O(2,3) O(3,3)
#undef O
        __builtin_amdgcn_sched_barrier(0);
#define W(i, w, base) acc[i] = fmac_ror<base + i>(acc[i], Rd[G], w.v[i]);
        W(0,w0,0) W(1,w0,0) W(2,w0,0) W(3,w0,0) W(0,w1,4) W(1,w1,4) W(2,w1,4) W(3,w1,4)
        W(0,w2,8) W(1,w2,8) W(2,w2,8) W(3,w2,8) W(0,w3,12) W(1,w3,12) W(2,w3,12) W(3,w3,12)
#undef W
    } else if constexpr (MODE == 1) {
        float s[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = lane_scalar(d, 16 * G + i);
#pragma unroll
        for (int i = 0; i < 16; ++i) gw[16 * G + i] = fmac_s(gw[16 * G + i], s[i], hin);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmac_s(acc[c], s[c], w0.v[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmac_s(acc[c], s[4 + c], w1.v[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmac_s(acc[c], s[8 + c], w2.v[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmac_s(acc[c], s[12 + c], w3.v[c]);
    } else if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i & 3] += lane_scalar(d, 16 * G + i);
        acc[0] += w0.v[0] + w1.v[0] + w2.v[0] + w3.v[0];
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) gw[16 * G + i] = fmac_ror<0>(gw[16 * G + i], d, hin);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) { acc[c] = fmac_ror<0>(acc[c], d, w0.v[c]); acc[c] = fmac_ror<0>(acc[c], d, w1.v[c]);
                                      acc[c] = fmac_ror<0>(acc[c], d, w2.v[c]); acc[c] = fmac_ror<0>(acc[c], d, w3.v[c]); }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int MODE> __global__ __launch_bounds__(512, 2) void k(const float *__restrict__ Wg, float *__restrict__ out, int iters, long long *cyc)
{
    __shared__ V4 wt4[16 * 64];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 64; i += 512) wt4[i] = reinterpret_cast<const V4 *>(Wg)[i];
    __syncthreads();
    float gw[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) gw[i] = 0.f;
    float d = 0.001f * (lane + 1), hin = 0.002f * (lane + 3);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        float Rh[4] = {0, 0, 0, 0}, Rd[4] = {0, 0, 0, 0}, acc[4] = {0, 0, 0, 0};
        if constexpr (MODE == 0) { rows_replicate(hin, Rh); rows_replicate(d, Rd); }
        __builtin_amdgcn_sched_barrier(0);
        group<0, MODE>(gw, wt4, lane, d, hin, Rh, Rd, acc);
        group<1, MODE>(gw, wt4, lane, d, hin, Rh, Rd, acc);
        group<2, MODE>(gw, wt4, lane, d, hin, Rh, Rd, acc);
        group<3, MODE>(gw, wt4, lane, d, hin, Rh, Rd, acc);
        d = fmaxf((acc[0] + acc[1]) + (acc[2] + acc[3]), 0.f) * 1e-3f + 1e-3f;
    }
    const long long t1 = clock64();
    float r = d;
#pragma unroll
    for (int i = 0; i < 64; ++i) r += gw[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE> void run(const char *name, const float *W, float *out, long long *cyc)
{
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, W, out, 100, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, W, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // 256 workgroups x 8 waves = 2 waves per SIMD; the two waves of a SIMD share its VALU
    printf("%-40s %8.3f ms   %7.1f shader cycles (2.4 GHz) per layer per wave\n", name, ms, ms * 1e6 / iters * 2.4 / 2);
}

int main()
{
    float *W, *out; long long *cyc;
    hipMalloc(&W, 64 * 64 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    hipMemset(W, 0, 64 * 64 * 4);
    run<0>("rotating operand (128 DPP FMAs)", W, out, cyc);
    run<1>("shared scalar (64 readlane + 128 FMA)", W, out, cyc);
    run<2>("64 readlane only", W, out, cyc);
    run<3>("128 plain FMAs", W, out, cyc);
    return 0;
}
