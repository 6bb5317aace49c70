// Can the LDS crossbar take the operand rotations of a row-block layer off the VALU?
// A hidden layer of the solve kernels is 16 "rotations" per lane: the activation vector (natural layout) permuted inside its
// 16-lane row, then two v_pk_fma_f32 on the four accumulators that share the permuted operand.  Today the permutation is a
// v_mov_b32_dpp row_ror:n (a VALU slot: 3.6 cycles at 2 waves/SIMD).  ds_swizzle_b32 (BITMASK_PERM: lane <- lane ^ n, no LDS memory
// touched) and ds_bpermute_b32 run on the LDS pipe instead.  One "layer" below = 15 permutations + 32 packed FMAs.
//   k_layer_dpp   15 v_mov_b32_dpp + 32 v_pk_fma_f32                       (what the kernels do)
//   k_layer_swz   15 ds_swizzle_b32 (four in flight) + 32 v_pk_fma_f32
//   k_layer_mix   8 ds_swizzle_b32 + 7 v_mov_b32_dpp + 32 v_pk_fma_f32
//   k_layer_bpm   15 ds_bpermute_b32 + 32 v_pk_fma_f32
//   k_layer_pk    32 v_pk_fma_f32 alone                                    (the floor)
//   k_swz_only    ds_swizzle_b32 alone                                     (LDS-pipe rate of the permutation)
// Build: hipcc -O3 --offload-arch=gfx950 swz_ubench.hip -o swz_ubench ; prints shader cycles per layer per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2_t __attribute__((ext_vector_type(2)));

#define PK2(A01, A23, W01, W23, HR)                                                                                   \
    asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]"     \
                 : "+v"(A01), "+v"(A23) : "v"(W01), "v"(W23), "v"(HR))
#define MOVDPP(DST, SRC, N) asm volatile("v_mov_b32_dpp %0, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "=v"(DST) : "v"(SRC))
// BITMASK_PERM: and_mask 0x1f, or_mask 0, xor_mask N  ->  offset = 0x1f | N << 10
#define SWZ(DST, SRC, N) asm volatile("ds_swizzle_b32 %0, %1 offset:%2" : "=v"(DST) : "v"(SRC), "n"(0x1f | ((N) << 10)))
#define BPM(DST, ADDR, SRC) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(DST) : "v"(ADDR), "v"(SRC))
#define WAITL(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")

#define PROLOGUE                                                                                                      \
    f2_t a01 = {threadIdx.x * 1e-3f, 1.f}, a23 = {2.f, 3.f};                                                          \
    f2_t w01[16], w23[16];                                                                                            \
    for (int i = 0; i < 16; ++i) { w01[i] = f2_t{1.f + i * 1e-3f, 1.f - i * 1e-3f}; w23[i] = f2_t{0.5f + i * 1e-3f, 0.25f}; } \
    float h = 1.0001f + threadIdx.x * 1e-6f;                                                                          \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define EPILOGUE                                                                                                      \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a01.x + a01.y + a23.x + a23.y + h;                                   \
    if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;

__global__ __launch_bounds__(256) void k_layer_dpp(float *out, unsigned long long *st, int iters)
{
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
        f2_t hr; float lo;
        hr.x = h; PK2(a01, a23, w01[0], w23[0], hr);
#define R(N) MOVDPP(lo, h, N); hr.x = lo; PK2(a01, a23, w01[N], w23[N], hr);
        R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)
#undef R
        h = a01.x * 1e-30f + 1.f;
    }
    EPILOGUE
}

__global__ __launch_bounds__(256) void k_layer_pk(float *out, unsigned long long *st, int iters)
{
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
        f2_t hr; hr.x = h;
#define R(N) PK2(a01, a23, w01[N], w23[N], hr);
        R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)
#undef R
        h = a01.x * 1e-30f + 1.f;
    }
    EPILOGUE
}

__global__ __launch_bounds__(256) void k_layer_swz(float *out, unsigned long long *st, int iters)
{
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
        float p[16];
        f2_t hr;
        SWZ(p[1], h, 1); SWZ(p[2], h, 2); SWZ(p[3], h, 3); SWZ(p[4], h, 4);
        hr.x = h; PK2(a01, a23, w01[0], w23[0], hr);
        SWZ(p[5], h, 5); SWZ(p[6], h, 6); SWZ(p[7], h, 7); SWZ(p[8], h, 8);
        WAITL(4);
#define U(N) hr.x = p[N]; PK2(a01, a23, w01[N], w23[N], hr);
        U(1) U(2) U(3) U(4)
        SWZ(p[9], h, 9); SWZ(p[10], h, 10); SWZ(p[11], h, 11); SWZ(p[12], h, 12);
        WAITL(4);
        U(5) U(6) U(7) U(8)
        SWZ(p[13], h, 13); SWZ(p[14], h, 14); SWZ(p[15], h, 15);
        WAITL(3);
        U(9) U(10) U(11) U(12)
        WAITL(0);
        U(13) U(14) U(15)
#undef U
        h = a01.x * 1e-30f + 1.f;
    }
    EPILOGUE
}

__global__ __launch_bounds__(256) void k_layer_mix(float *out, unsigned long long *st, int iters)
{
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
        float p[16], lo;
        f2_t hr;
        SWZ(p[4], h, 4); SWZ(p[5], h, 5); SWZ(p[6], h, 6); SWZ(p[9], h, 9);
        SWZ(p[10], h, 10); SWZ(p[11], h, 11); SWZ(p[12], h, 12); SWZ(p[13], h, 13);
        hr.x = h; PK2(a01, a23, w01[0], w23[0], hr);
#define D(N) MOVDPP(lo, h, N); hr.x = lo; PK2(a01, a23, w01[N], w23[N], hr);
#define U(N) hr.x = p[N]; PK2(a01, a23, w01[N], w23[N], hr);
        D(1) D(2) D(3) D(7)
        WAITL(4);
        U(4) U(5) U(6) U(9)
        D(8) D(15) D(14)
        WAITL(0);
        U(10) U(11) U(12) U(13)
#undef D
#undef U
        h = a01.x * 1e-30f + 1.f;
    }
    EPILOGUE
}

__global__ __launch_bounds__(256) void k_layer_bpm(float *out, unsigned long long *st, int iters)
{
    PROLOGUE
    int ad[16];
    for (int n = 0; n < 16; ++n) ad[n] = 4 * ((threadIdx.x & 48) | ((threadIdx.x - n) & 15));
    for (int it = 0; it < iters; ++it) {
        float p[16];
        f2_t hr;
        BPM(p[1], ad[1], h); BPM(p[2], ad[2], h); BPM(p[3], ad[3], h); BPM(p[4], ad[4], h);
        hr.x = h; PK2(a01, a23, w01[0], w23[0], hr);
        BPM(p[5], ad[5], h); BPM(p[6], ad[6], h); BPM(p[7], ad[7], h); BPM(p[8], ad[8], h);
        WAITL(4);
#define U(N) hr.x = p[N]; PK2(a01, a23, w01[N], w23[N], hr);
        U(1) U(2) U(3) U(4)
        BPM(p[9], ad[9], h); BPM(p[10], ad[10], h); BPM(p[11], ad[11], h); BPM(p[12], ad[12], h);
        WAITL(4);
        U(5) U(6) U(7) U(8)
        BPM(p[13], ad[13], h); BPM(p[14], ad[14], h); BPM(p[15], ad[15], h);
        WAITL(3);
        U(9) U(10) U(11) U(12)
        WAITL(0);
        U(13) U(14) U(15)
#undef U
        h = a01.x * 1e-30f + 1.f;
    }
    EPILOGUE
}

__global__ __launch_bounds__(256) void k_swz_only(float *out, unsigned long long *st, int iters)
{
    PROLOGUE
    for (int it = 0; it < iters; ++it) {
        float p[16];
        SWZ(p[1], h, 1); SWZ(p[2], h, 2); SWZ(p[3], h, 3); SWZ(p[4], h, 4); SWZ(p[5], h, 5); SWZ(p[6], h, 6); SWZ(p[7], h, 7);
        SWZ(p[8], h, 8); SWZ(p[9], h, 9); SWZ(p[10], h, 10); SWZ(p[11], h, 11); SWZ(p[12], h, 12); SWZ(p[13], h, 13);
        SWZ(p[14], h, 14); SWZ(p[15], h, 15);
        WAITL(0);
        float s = 0.f;
        for (int n = 1; n < 16; ++n) asm volatile("v_max_f32 %0, %0, %1" : "+v"(s) : "v"(p[n]));
        a01.x += s * 1e-30f;
    }
    EPILOGUE
}

template <typename K> void run(const char *name, K kern, float *out, unsigned long long *st)
{
    const int iters = 4000;
    printf("%-14s", name);
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;                   // one 4-wave block per CU per requested wave-per-SIMD
        kern<<<blocks, 256>>>(out, st, 50); (void)hipDeviceSynchronize();
        kern<<<blocks, 256>>>(out, st, iters); (void)hipDeviceSynchronize();
        static unsigned long long h[1024];
        (void)hipMemcpy(h, st, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
        printf("  %dw/SIMD: %7.1f cyc/layer/wave = %6.1f cyc/layer/SIMD", wps, avg / iters, avg / iters / wps);
    }
    printf("\n");
}
#define RUN(K) run(#K, K, out, st)
int main()
{
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 4 * 256 * 1024); (void)hipMalloc(&st, 8 * 1024);
    RUN(k_layer_pk); RUN(k_layer_dpp); RUN(k_layer_swz); RUN(k_layer_mix); RUN(k_layer_bpm); RUN(k_swz_only);
    return 0;
}
