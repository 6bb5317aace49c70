// Hidden-layer micro-benchmark: how should the 64 activations reach the 64 lanes?
//   NLDS of the four 16-lane blocks go through LDS (ds_write_b32 + broadcast ds_read_b128, plain v_fma_f32),
//   the remaining blocks use the rotating DPP operand (v_fmac_f32_dpp row_ror:n).
// Reports true shader cycles per layer per SIMD (s_memtime), 2 waves/SIMD like the solve kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ int f2i(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ float i2f(int v) { return __builtin_bit_cast(float, v); }
template <int N> __device__ __forceinline__ float fmac_ror(float acc, float x, float w);
#define F(N) template <> __device__ __forceinline__ float fmac_ror<N>(float acc, float x, float w) { asm("v_fmac_f32_dpp %0, %1, %2 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(w)); return acc; }
template <> __device__ __forceinline__ float fmac_ror<0>(float acc, float x, float w) { return __builtin_fmaf(x, w, acc); }
F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
template <int Q, int N> __device__ __forceinline__ void dpp_block(const float (&w)[64], float Rq, float (&acc)[4])
{
    acc[N & 3] = fmac_ror<N>(acc[N & 3], Rq, w[Q * 16 + N]);
    if constexpr (N < 15) dpp_block<Q, N + 1>(w, Rq, acc);
}
template <int NLDS>
__global__ __launch_bounds__(64, 2) void layer(const float *__restrict__ Wg, float *__restrict__ out, unsigned long long *st, int iters)
{
    __shared__ __attribute__((aligned(16))) float sh[64];
    const int lane = threadIdx.x;
    float w[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) w[k] = Wg[lane * 64 + k];
    float pad[150];                            // occupy registers so that exactly 2 waves fit per SIMD, like the real kernel
#pragma unroll
    for (int k = 0; k < 150; ++k) pad[k] = Wg[(lane * 7 + k) & 4095];
    float h = (float)(lane + 1) * 0.01f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (NLDS > 0) {
            sh[lane] = h;
            __builtin_amdgcn_wave_barrier();
        }
        float R[4];
        if constexpr (NLDS < 4) {
            auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)f2i(h), (unsigned)f2i(h), false, false);
            auto a = __builtin_amdgcn_permlane32_swap(s16[0], s16[0], false, false);
            auto b = __builtin_amdgcn_permlane32_swap(s16[1], s16[1], false, false);
            R[0] = i2f((int)a[0]); R[2] = i2f((int)a[1]); R[1] = i2f((int)b[0]); R[3] = i2f((int)b[1]);
            asm volatile("s_nop 1" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]));
        }
        const float4 *s4 = reinterpret_cast<const float4 *>(sh);
#pragma unroll
        for (int q = 0; q < NLDS; ++q) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 v = s4[q * 4 + k];
                acc[0] = __builtin_fmaf(w[q * 16 + 4 * k + 0], v.x, acc[0]);
                acc[1] = __builtin_fmaf(w[q * 16 + 4 * k + 1], v.y, acc[1]);
                acc[2] = __builtin_fmaf(w[q * 16 + 4 * k + 2], v.z, acc[2]);
                acc[3] = __builtin_fmaf(w[q * 16 + 4 * k + 3], v.w, acc[3]);
            }
        }
        if constexpr (NLDS <= 0) dpp_block<0, 0>(w, R[0], acc);
        if constexpr (NLDS <= 1) dpp_block<1, 0>(w, R[1], acc);
        if constexpr (NLDS <= 2) dpp_block<2, 0>(w, R[2], acc);
        if constexpr (NLDS <= 3) dpp_block<3, 0>(w, R[3], acc);
        if constexpr (NLDS > 0) __builtin_amdgcn_wave_barrier();
        h = fmaxf((acc[0] + acc[1]) + (acc[2] + acc[3]), 0.f) + 1e-3f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = h;
#pragma unroll
    for (int k = 0; k < 150; ++k) s += pad[k] * 1e-9f;
    out[blockIdx.x * 64 + lane] = s;
    if (lane == 0) st[blockIdx.x] = t1 - t0;
}
template <int NLDS> void run(const float *W, float *out, unsigned long long *st, int iters)
{
    const int nwaves = 2048;
    layer<NLDS><<<nwaves, 64>>>(W, out, st, 10);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    layer<NLDS><<<nwaves, 64>>>(W, out, st, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nwaves);
    (void)hipMemcpy(h.data(), st, nwaves * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= nwaves;
    printf("blocks via LDS = %d, via DPP = %d : %7.3f ms, %7.1f shader cycles per layer per wave (2 waves/SIMD -> %6.1f per layer per SIMD)\n",
           NLDS, 4 - NLDS, ms, avg / iters, avg / iters / 2);
}
int main()
{
    float *W, *out; unsigned long long *st;
    std::vector<float> hw(4096);
    for (int i = 0; i < 4096; ++i) hw[i] = 0.01f * ((i * 7919) % 13 - 6);
    (void)hipMalloc(&W, 4096 * 4); (void)hipMalloc(&out, 2048 * 64 * 4); (void)hipMalloc(&st, 2048 * 8);
    (void)hipMemcpy(W, hw.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0>(W, out, st, 20000); run<1>(W, out, st, 20000); run<2>(W, out, st, 20000); run<3>(W, out, st, 20000); run<4>(W, out, st, 20000);
    return 0;
}
