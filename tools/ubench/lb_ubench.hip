// Row-broadcast through LDS instead of DPP rotations: does it pay?
// A row-block hidden layer needs, on lane (r, i), the 16 activations of the lane's 16-lane row r.  Today: 15 v_mov_b32_dpp row_ror:n
// (VALU slots).  Alternative: the wave drops the activation vector into LDS (one ds_write_b32) and every lane reads its row back
// with four ds_read_b128 (the 16 lanes of a row read the same 64 bytes: broadcast, conflict-free, 4 LDS cycles each); the packed
// FMAs then pick the low / high half of the loaded pairs with op_sel -- no DPP move at all, weights in natural column order.
//   k_chain_dpp   one layer = 2 pk_mul + 15 (mov_dpp + 2 pk_fma) + finish (2 swap16, pk_add, swap32, add, add, max), result feeds the next
//   k_chain_lb    one layer = ds_write_b32 + 4 ds_read_b128 + 32 pk_fma + the same finish
//   k_tput_dpp    throughput form (the adjoint's accumulation waves): 15 mov_dpp + 32 pk_fma on 32 accumulator pairs, operand from LDS
//   k_tput_lb     the same with 4 ds_read_b128 instead of the moves
// Build: hipcc -O3 --offload-arch=gfx950 lb_ubench.hip -o lb_ubench ; prints shader cycles per layer per SIMD at 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f4_t __attribute__((ext_vector_type(4)));

#define PK2LO(A01, A23, W01, W23, HP)                                                                                 \
    asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel_hi:[1,0,1]"     \
                 : "+v"(A01), "+v"(A23) : "v"(W01), "v"(W23), "v"(HP))
#define PK2HI(A01, A23, W01, W23, HP)                                                                                 \
    asm volatile("v_pk_fma_f32 %0, %2, %4, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\tv_pk_fma_f32 %1, %3, %4, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]" \
                 : "+v"(A01), "+v"(A23) : "v"(W01), "v"(W23), "v"(HP))
#define MOVDPP(DST, SRC, N) asm volatile("v_mov_b32_dpp %0, %1 row_ror:" #N " row_mask:0xf bank_mask:0xf" : "=v"(DST) : "v"(SRC))

__device__ __forceinline__ float finish(f2_t a02, f2_t a13, float bias)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %[a2], %[a3]\n\ts_nop 0\n\tv_permlane16_swap_b32 %[a0], %[a1]"
                 : [a0] "+v"(a02.x), [a2] "+v"(a02.y), [a1] "+v"(a13.x), [a3] "+v"(a13.y));
    asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %0, %1" : "+v"(a02) : "v"(a13));
    float a0 = a02.x, a2 = a02.y;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %[a0], %[a2]\n\tv_add_f32 %[a0], %[a0], %[a2]\n\tv_add_f32 %[a0], %[a0], %[bias]\n\tv_max_f32 %[a0], 0, %[a0]"
                 : [a0] "+v"(a0), [a2] "+v"(a2) : [bias] "v"(bias));
    return a0;
}

#define WEIGHTS                                                                                                       \
    f2_t w01[16], w23[16];                                                                                            \
    for (int i = 0; i < 16; ++i) { w01[i] = f2_t{1e-2f + i * 1e-3f, 1e-2f - i * 1e-3f}; w23[i] = f2_t{-5e-3f + i * 1e-3f, 2.5e-3f}; } \
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(w01[i]), "+v"(w23[i]));

__global__ __launch_bounds__(256) void k_chain_dpp(float *out, unsigned long long *st, int iters)
{
    WEIGHTS
    float h = 1.0001f + threadIdx.x * 1e-6f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f2_t a02, a13, hr; float lo;
        { f2_t hh; hh.x = h; asm volatile("v_pk_mul_f32 %0, %2, %4 op_sel_hi:[1,0]\n\tv_pk_mul_f32 %1, %3, %4 op_sel_hi:[1,0]" : "=&v"(a02), "=&v"(a13) : "v"(w01[0]), "v"(w23[0]), "v"(hh)); }
#define R(N) MOVDPP(lo, h, N); hr.x = lo; PK2LO(a02, a13, w01[N], w23[N], hr);
        R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)
#undef R
        h = finish(a02, a13, 0.5f);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
    if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_chain_lb(float *out, unsigned long long *st, int iters)
{
    __shared__ __attribute__((aligned(16))) float hb[4 * 64];
    WEIGHTS
    float h = 1.0001f + threadIdx.x * 1e-6f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mine = hb + wave * 64;
    const f4_t *row = reinterpret_cast<const f4_t *>(mine + (lane & 48));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        mine[lane] = h;
        const f4_t q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
        f2_t a02 = {0.f, 0.f}, a13 = {0.f, 0.f};
#define Q(K, V) { f2_t lo_ = {V.x, V.y}, hi_ = {V.z, V.w}; \
        PK2LO(a02, a13, w01[4 * K], w23[4 * K], lo_); PK2HI(a02, a13, w01[4 * K + 1], w23[4 * K + 1], lo_); \
        PK2LO(a02, a13, w01[4 * K + 2], w23[4 * K + 2], hi_); PK2HI(a02, a13, w01[4 * K + 3], w23[4 * K + 3], hi_); }
        Q(0, q0) Q(1, q1) Q(2, q2) Q(3, q3)
#undef Q
        h = finish(a02, a13, 0.5f);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
    if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}

// throughput form: 32 accumulator pairs (an accumulation wave's dW), multipliers D01 / D23 fixed, operand vector from an LDS record
__global__ __launch_bounds__(256) void k_tput_dpp(float *out, unsigned long long *st, int iters)
{
    __shared__ __attribute__((aligned(16))) float rec[4 * 8 * 64];
    f2_t gw[32];
    for (int i = 0; i < 32; ++i) gw[i] = f2_t{0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 8 * 64; i += 64) rec[wave * 512 + i] = 1.f + i * 1e-3f;
    f2_t D01 = {1e-3f * lane, 2e-3f}, D23 = {3e-3f, 4e-3f};
    asm volatile("" : "+v"(D01), "+v"(D23));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const float hn = rec[wave * 512 + (it & 7) * 64 + lane];
        f2_t hr; float lo;
        hr.x = hn; PK2LO(gw[0], gw[1], D01, D23, hr);
#define R(N) MOVDPP(lo, hn, N); hr.x = lo; PK2LO(gw[2 * N], gw[2 * N + 1], D01, D23, hr);
        R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)
#undef R
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += gw[i].x + gw[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(256) void k_tput_lb(float *out, unsigned long long *st, int iters)
{
    __shared__ __attribute__((aligned(16))) float rec[4 * 8 * 64];
    f2_t gw[32];
    for (int i = 0; i < 32; ++i) gw[i] = f2_t{0.f, 0.f};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 8 * 64; i += 64) rec[wave * 512 + i] = 1.f + i * 1e-3f;
    f2_t D01 = {1e-3f * lane, 2e-3f}, D23 = {3e-3f, 4e-3f};
    asm volatile("" : "+v"(D01), "+v"(D23));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const f4_t *row = reinterpret_cast<const f4_t *>(rec + wave * 512 + (it & 7) * 64 + (lane & 48));
        const f4_t q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
#define Q(K, V) { f2_t lo_ = {V.x, V.y}, hi_ = {V.z, V.w}; \
        PK2LO(gw[8 * K], gw[8 * K + 1], D01, D23, lo_); PK2HI(gw[8 * K + 2], gw[8 * K + 3], D01, D23, lo_); \
        PK2LO(gw[8 * K + 4], gw[8 * K + 5], D01, D23, hi_); PK2HI(gw[8 * K + 6], gw[8 * K + 7], D01, D23, hi_); }
        Q(0, q0) Q(1, q1) Q(2, q2) Q(3, q3)
#undef Q
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += gw[i].x + gw[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}

template <typename K> void run(const char *name, K kern, float *out, unsigned long long *st)
{
    const int iters = 4000;
    printf("%-12s", name);
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;                   // one 4-wave block per CU per requested wave-per-SIMD
        kern<<<blocks, 256>>>(out, st, 50); (void)hipDeviceSynchronize();
        kern<<<blocks, 256>>>(out, st, iters); (void)hipDeviceSynchronize();
        static unsigned long long h[1024];
        (void)hipMemcpy(h, st, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
        printf("  %dw/SIMD: %7.1f cyc/layer/wave = %6.1f cyc/layer/SIMD", wps, avg / iters, avg / iters / wps);
    }
    float v; (void)hipMemcpy(&v, out, 4, hipMemcpyDeviceToHost);
    printf("   [out %g]\n", v);
}
#define RUN(K) run(#K, K, out, st)
int main()
{
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 4 * 256 * 1024); (void)hipMalloc(&st, 8 * 1024);
    RUN(k_chain_dpp); RUN(k_chain_lb); RUN(k_tput_dpp); RUN(k_tput_lb);
    return 0;
}
