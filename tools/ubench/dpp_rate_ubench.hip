// Issue rate of v_fmac_f32_dpp by DPP control: is any DPP pattern cheaper than row_ror?  (No.)
#include <hip/hip_runtime.h>
#include <cstdio>
#define KERNEL(NAME, CTRL)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *st, int iters)                      \
    {                                                                                                               \
        float a[16];                                                                                                \
        for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;                                                \
        float x = 1.0001f + threadIdx.x * 1e-6f, y = 0.9999f;                                                       \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                       \
        for (int it = 0; it < iters; ++it) {                                                                        \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                         \
                _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                      \
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 " CTRL " row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x), "v"(y)); \
            }                                                                                                       \
        }                                                                                                           \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                       \
        float s = 0;                                                                                                \
        for (int i = 0; i < 16; ++i) s += a[i];                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                             \
        if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;                                                             \
    }
KERNEL(k_quad, "quad_perm:[1,0,3,2]")
KERNEL(k_shr1, "row_shr:1")
KERNEL(k_ror3, "row_ror:3")
KERNEL(k_ror8, "row_ror:8")
KERNEL(k_mirror, "row_mirror")
KERNEL(k_hmirror, "row_half_mirror")
KERNEL(k_bcast15, "row_bcast:15")
template <typename K> void run(const char *name, K kern, float *out, unsigned long long *st)
{
    const int blocks = 1024, iters = 20000;        // 4 waves per SIMD
    kern<<<blocks, 256>>>(out, st, 100); (void)hipDeviceSynchronize();
    kern<<<blocks, 256>>>(out, st, iters); (void)hipDeviceSynchronize();
    unsigned long long h[64]; (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 64; ++i) avg += (double)h[i]; avg /= 64;
    // relative comparison between DPP controls (result on MI355X: identical for every control -- there is no
    // "cheap" DPP pattern, all of them issue at half the plain v_fma_f32 rate, see clock_ubench.hip)
    printf("%-22s %.2f shader cycles per instruction as seen by one wave\n", name, avg / ((double)iters * 64));
}
int main()
{
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 4 * 256 * 1024); (void)hipMalloc(&st, 8 * 1024);
    run("quad_perm", k_quad, out, st); run("row_shr:1", k_shr1, out, st); run("row_ror:3", k_ror3, out, st);
    run("row_ror:8", k_ror8, out, st); run("row_mirror", k_mirror, out, st); run("row_half_mirror", k_hmirror, out, st);
    run("row_bcast:15", k_bcast15, out, st);
    return 0;
}
