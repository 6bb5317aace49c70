// What clock does the chip hold under a dense fp32 VALU loop, and how many cycles does a wave64 v_fma_f32 /
// v_fmac_f32_dpp / v_pk_fma_f32 really take?  (s_memtime = shader cycles, s_memrealtime = 100 MHz)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *stamps, int iters)
{
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
    float x = 1.0001f, y = 0.9999f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (MODE == 0) a[i] = __builtin_fmaf(a[i], x, y);
                else if constexpr (MODE == 1) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x), "v"(y));
                else if constexpr (MODE == 2) { if (i % 2 == 0) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {a[i], a[i + 1]}; f2 xx = {x, x}, yy = {y, y}; v = __builtin_elementwise_fma(v, xx, yy); a[i] = v.x; a[i + 1] = v.y; } }
                else if constexpr (MODE == 3) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a[i]), 5)) + y;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}
template <int MODE> void run(const char *name, int blocks, int iters, float *out, unsigned long long *st)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, st, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, st, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double clk_ghz = (double)h[0] / (double)h[1] * 0.1;
    double waves_per_simd = blocks * 4.0 / 1024.0;
    double instr_per_wave = (double)iters * 64 * (MODE == 2 ? 0.5 : 1.0);
    double cyc = ms * 1e-3 * clk_ghz * 1e9 / (instr_per_wave * waves_per_simd);
    printf("%-22s blocks=%5d (%.0f waves/SIMD): %8.3f ms  clock %.3f GHz  -> %.2f cycles per wave-instruction per SIMD\n", name, blocks, waves_per_simd, ms, clk_ghz, cyc);
}
int main()
{
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 4 * 256 * 4096); (void)hipMalloc(&st, 16 * 4096);
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("v_fma_f32", blocks, 20000, out, st);
        run<1>("v_fmac_f32_dpp", blocks, 20000, out, st);
        run<2>("v_pk_fma_f32", blocks, 20000, out, st);
        run<3>("v_readlane+v_add", blocks, 20000, out, st);
    }
    return 0;
}
