// Does v_cndmask_b32 cost more when its mask is VCC (VOP2 form) than when it is an SGPR pair (VOP3 form)?
// Shader cycles per instruction per SIMD at 1 / 2 / 4 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 vcc_ubench.hip -o vcc_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","vcc","s20","s21"
#define R16(X) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
#define CND_VCC(n) "v_cndmask_b32 v" #n ", v32, v33, vcc\n\t"
#define CND_SGPR(n) "v_cndmask_b32_e64 v" #n ", v32, v33, s[20:21]\n\t"
#define CMP_CND_VCC(n) "v_cmp_gt_f32 vcc, v34, v" #n "\n\tv_cndmask_b32 v" #n ", v32, v33, vcc\n\t"
#define CMP_CND_SGPR(n) "v_cmp_gt_f32_e64 s[20:21], v34, v" #n "\n\tv_cndmask_b32_e64 v" #n ", v32, v33, s[20:21]\n\t"
#define KERNEL(NAME, BODY, PRE)                                                                       \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *st, int iters)        \
    {                                                                                                 \
        asm volatile(PRE ::: CLOB);                                                                   \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
        for (int it = 0; it < iters; ++it) asm volatile(R16(BODY) R16(BODY) ::: CLOB);                \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
        float s; asm volatile("v_mov_b32 %0, v16" : "=v"(s));                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                               \
        if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;                                               \
    }
#define PRE0 "v_mov_b32 v32, 1.0\n\tv_mov_b32 v33, 2.0\n\tv_mov_b32 v34, 0.5\n\ts_mov_b64 vcc, 0x55\n\ts_mov_b64 s[20:21], 0x55"
KERNEL(k_cnd_vcc, CND_VCC, PRE0)
KERNEL(k_cnd_sgpr, CND_SGPR, PRE0)
KERNEL(k_cmp_cnd_vcc, CMP_CND_VCC, PRE0)
KERNEL(k_cmp_cnd_sgpr, CMP_CND_SGPR, PRE0)
template <typename K> void run(const char *name, K kern, int ninst, float *out, unsigned long long *st)
{
    const int iters = 4000;
    printf("%-16s", name);
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        kern<<<blocks, 256>>>(out, st, 50); (void)hipDeviceSynchronize();
        kern<<<blocks, 256>>>(out, st, iters); (void)hipDeviceSynchronize();
        static unsigned long long h[1024];
        (void)hipMemcpy(h, st, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
        printf("  %dw: %5.2f", wps, avg / ((double)iters * ninst) / wps);
    }
    printf("   cycles/inst/SIMD\n");
}
int main()
{
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 4 * 256 * 1024); (void)hipMalloc(&st, 8 * 1024);
    run("cnd vcc", k_cnd_vcc, 32, out, st);
    run("cnd sgpr", k_cnd_sgpr, 32, out, st);
    run("cmp+cnd vcc", k_cmp_cnd_vcc, 64, out, st);
    run("cmp+cnd sgpr", k_cmp_cnd_sgpr, 64, out, st);
    return 0;
}
