// Issue cost of the instructions that make up the per-RHS "fixed part" of the solve kernels, as seen by the SIMD:
// shader cycles per instruction per SIMD at 1 / 2 / 4 waves per SIMD (independent streams: throughput, not latency).
// Build: hipcc -O3 --offload-arch=gfx950 inst_cost_ubench.hip -o inst_cost_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL(NAME, BODY, NINST)                                                                         \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *st, int iters)            \
    {                                                                                                     \
        float a[16], b[16];                                                                               \
        for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 1.0f + i * 1e-3f; }         \
        float x = 1.0001f + threadIdx.x * 1e-6f, y = 0.9999f;                                             \
        unsigned long long m = 0x5555555555555555ull;                                                     \
        asm volatile("" : "+s"(m));                                                                        \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
        for (int it = 0; it < iters; ++it) {                                                              \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) { REP16(BODY) }                                 \
        }                                                                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
        float s = x + y;                                                                                  \
        for (int i = 0; i < 16; ++i) s += a[i] + b[i];                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                   \
        if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;                                                   \
    }                                                                                                     \
    static const int NAME##_n = NINST;

#define B_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define B_FMAC_DPP(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_ror:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x), "v"(y));
#define B_ADD_DPP(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
#define B_MOV_DPP(i) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b[i]));
#define B_SWAP16(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
#define B_SWAP32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
#define B_READLANE_USE(i) { int sg; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(b[i])); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(sg), "v"(y)); }
#define B_READLANE6_USE(i) if (i < 2) { int s0, s1, s2, s3, s4, s5; \
        asm volatile("v_readlane_b32 %0, %6, 0\n v_readlane_b32 %1, %6, 1\n v_readlane_b32 %2, %6, 2\n v_readlane_b32 %3, %6, 3\n v_readlane_b32 %4, %6, 4\n v_readlane_b32 %5, %6, 5" \
                     : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3), "=s"(s4), "=s"(s5) : "v"(b[i])); \
        asm volatile("v_fmac_f32 %0, %1, %7\n v_fmac_f32 %0, %2, %7\n v_fmac_f32 %0, %3, %7\n v_fmac_f32 %0, %4, %7\n v_fmac_f32 %0, %5, %7\n v_fmac_f32 %0, %6, %7" \
                     : "+v"(a[i]) : "s"(s0), "s"(s1), "s"(s2), "s"(s3), "s"(s4), "s"(s5), "v"(y)); }
#define B_BCAST_DPP6(i) if (i < 2) { \
        asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n" \
                     "v_fmac_f32_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n" \
                     "v_fmac_f32_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" \
                     : "+v"(a[i]) : "v"(b[i]), "v"(y)); }
#define B_CNDMASK(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "s"(m));
#define B_RCP(i) asm volatile("v_rcp_f32_e32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define B_PKMUL(i) if (i < 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&a[2 * i]) : "v"(*(double *)&b[2 * i]));
#define B_FMA_NOP(i) asm volatile("v_fma_f32 %0, %1, %2, %0\n s_nop 0" : "+v"(a[i]) : "v"(x), "v"(y));
#define B_FMA_NOP1(i) asm volatile("v_fma_f32 %0, %1, %2, %0\n s_nop 1" : "+v"(a[i]) : "v"(x), "v"(y));
#define B_MAX(i) asm volatile("v_max_f32_e32 %0, 0, %0" : "+v"(a[i]));
#define B_MOV(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(a[i]) : "v"(b[i]));
#define B_FMA_DEP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));
#define B_DPP_DEP(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a[0]));
#define B_SWAP_DEP(i) asm volatile("v_permlane32_swap_b32 %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a[0]), "+v"(b[0]));
#define B_FMA_SGPR(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "s"(it), "v"(y));

// one "rotation step" of a hidden layer (four accumulators share the rotated activation): four DPP FMAs, or one DPP move + two
// packed FMAs on accumulator / weight PAIRS (v_pk_fma_f32: both halves take the low half of the moved operand)
#define B_STEP_DPP4(i) if (i < 4) asm volatile("v_fmac_f32_dpp %0, %4, %5 row_ror:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %4, %6 row_ror:3 row_mask:0xf bank_mask:0xf\n" \
        "v_fmac_f32_dpp %2, %4, %7 row_ror:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %4, %8 row_ror:3 row_mask:0xf bank_mask:0xf" \
        : "+v"(a[4 * i]), "+v"(a[4 * i + 1]), "+v"(a[4 * i + 2]), "+v"(a[4 * i + 3]) : "v"(x), "v"(b[4 * i]), "v"(b[4 * i + 1]), "v"(b[4 * i + 2]), "v"(b[4 * i + 3]));
typedef float f2_t __attribute__((ext_vector_type(2)));
#define B_STEP_PK2(i) if (i < 4) { f2_t hr; float lo; asm volatile("v_mov_b32_dpp %0, %1 row_ror:3 row_mask:0xf bank_mask:0xf" : "=v"(lo) : "v"(x)); hr.x = lo; \
        f2_t a01 = {a[4 * i], a[4 * i + 1]}, a23 = {a[4 * i + 2], a[4 * i + 3]}, w01 = {b[4 * i], b[4 * i + 1]}, w23 = {b[4 * i + 2], b[4 * i + 3]}; \
        asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]" \
        : "+v"(a01), "+v"(a23) : "v"(w01), "v"(w23), "v"(hr)); a[4 * i] = a01.x; a[4 * i + 1] = a01.y; a[4 * i + 2] = a23.x; a[4 * i + 3] = a23.y; }
KERNEL(k_step_dpp4, B_STEP_DPP4, 8)
KERNEL(k_step_pk2, B_STEP_PK2, 8)
KERNEL(k_fma, B_FMA, 32)
KERNEL(k_fmac_dpp, B_FMAC_DPP, 32)
KERNEL(k_add_dpp, B_ADD_DPP, 32)
KERNEL(k_mov_dpp, B_MOV_DPP, 32)
KERNEL(k_swap16, B_SWAP16, 32)
KERNEL(k_swap32, B_SWAP32, 32)
KERNEL(k_readlane_use, B_READLANE_USE, 64)
KERNEL(k_readlane6_use, B_READLANE6_USE, 48)
KERNEL(k_bcast_dpp6, B_BCAST_DPP6, 24)
KERNEL(k_cndmask, B_CNDMASK, 32)
KERNEL(k_rcp, B_RCP, 32)
KERNEL(k_pkmul, B_PKMUL, 16)
KERNEL(k_fma_nop0, B_FMA_NOP, 32)
KERNEL(k_fma_nop1, B_FMA_NOP1, 32)
KERNEL(k_max, B_MAX, 32)
KERNEL(k_mov, B_MOV, 32)
KERNEL(k_fma_dep, B_FMA_DEP, 32)
KERNEL(k_dpp_dep, B_DPP_DEP, 32)
KERNEL(k_swap_add_dep, B_SWAP_DEP, 64)
KERNEL(k_fma_sgpr, B_FMA_SGPR, 32)

template <typename K> void run(const char *name, K kern, int ninst, float *out, unsigned long long *st)
{
    const int iters = 4000;
    printf("%-18s", name);
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;                   // one 4-wave block per CU per requested wave-per-SIMD
        kern<<<blocks, 256>>>(out, st, 50); (void)hipDeviceSynchronize();
        kern<<<blocks, 256>>>(out, st, iters); (void)hipDeviceSynchronize();
        static unsigned long long h[1024];
        (void)hipMemcpy(h, st, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
        // a wave saw avg cycles for iters*ninst instructions; wps waves share the SIMD
        printf("  %dw/SIMD: %6.2f cyc/inst/wave = %5.2f cyc/inst/SIMD", wps, avg / ((double)iters * ninst), avg / ((double)iters * ninst) / wps);
    }
    printf("\n");
}
#define RUN(K) run(#K, K, K##_n, out, st)
int main()
{
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 4 * 256 * 1024); (void)hipMalloc(&st, 8 * 1024);
    RUN(k_step_dpp4); RUN(k_step_pk2);
    RUN(k_fma); RUN(k_fmac_dpp); RUN(k_add_dpp); RUN(k_mov_dpp); RUN(k_swap16); RUN(k_swap32); RUN(k_readlane_use);
    RUN(k_readlane6_use); RUN(k_bcast_dpp6); RUN(k_cndmask); RUN(k_rcp); RUN(k_pkmul); RUN(k_fma_nop0); RUN(k_fma_nop1);
    RUN(k_max); RUN(k_mov); RUN(k_fma_dep); RUN(k_dpp_dep); RUN(k_swap_add_dep); RUN(k_fma_sgpr);
    return 0;
}
