// Micro-benchmark: one 64x64 matrix-vector layer per wave, weights register-resident, different ways
// of broadcasting the activation vector.  Prints ns per layer per wave and cycles per layer (2.4 GHz).
// Build: hipcc -O3 --offload-arch=gfx950 layer_ubench.hip -o layer_ubench ; run on MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ float bcast(float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); }

template <int MODE, int WPB>
__global__ __launch_bounds__(64 * WPB) void layer_kernel(const float *__restrict__ Wg, float *__restrict__ out, int iters)
{
    __shared__ __attribute__((aligned(16))) float sh[WPB][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float w[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) w[k] = Wg[lane * 64 + k];
    float h = (float)(lane + 1) * 0.01f;
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {            // readlane, 2 accumulators (what the solver does now)
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int k = 0; k < 64; k += 2) { a0 = __builtin_fmaf(w[k], bcast(h, k), a0); a1 = __builtin_fmaf(w[k + 1], bcast(h, k + 1), a1); }
            h = fmaxf(a0 + a1, 0.f) + 1e-3f;
        } else if constexpr (MODE == 1) {     // readlane, 8 accumulators
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 64; ++k) a[k & 7] = __builtin_fmaf(w[k], bcast(h, k), a[k & 7]);
            h = fmaxf(((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])), 0.f) + 1e-3f;
        } else if constexpr (MODE == 2) {     // LDS broadcast, ds_read_b128, 4 accumulators
            sh[wave][lane] = h;
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            float a[4] = {0, 0, 0, 0};
            const float4 *s4 = reinterpret_cast<const float4 *>(sh[wave]);
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float4 v = s4[k];
                a[0] = __builtin_fmaf(w[4 * k + 0], v.x, a[0]);
                a[1] = __builtin_fmaf(w[4 * k + 1], v.y, a[1]);
                a[2] = __builtin_fmaf(w[4 * k + 2], v.z, a[2]);
                a[3] = __builtin_fmaf(w[4 * k + 3], v.w, a[3]);
            }
            __builtin_amdgcn_wave_barrier();
            h = fmaxf((a[0] + a[1]) + (a[2] + a[3]), 0.f) + 1e-3f;
        } else if constexpr (MODE == 3) {     // half LDS broadcast, half readlane
            sh[wave][lane] = h;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            float a[4] = {0, 0, 0, 0};
            const float4 *s4 = reinterpret_cast<const float4 *>(sh[wave]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float4 v = s4[k];
                a[0] = __builtin_fmaf(w[4 * k + 0], v.x, a[0]);
                a[1] = __builtin_fmaf(w[4 * k + 1], v.y, a[1]);
                a[2] = __builtin_fmaf(w[4 * k + 2], v.z, a[2]);
                a[3] = __builtin_fmaf(w[4 * k + 3], v.w, a[3]);
                a[0] = __builtin_fmaf(w[32 + 4 * k + 0], bcast(h, 32 + 4 * k + 0), a[0]);
                a[1] = __builtin_fmaf(w[32 + 4 * k + 1], bcast(h, 32 + 4 * k + 1), a[1]);
                a[2] = __builtin_fmaf(w[32 + 4 * k + 2], bcast(h, 32 + 4 * k + 2), a[2]);
                a[3] = __builtin_fmaf(w[32 + 4 * k + 3], bcast(h, 32 + 4 * k + 3), a[3]);
            }
            __builtin_amdgcn_wave_barrier();
            h = fmaxf((a[0] + a[1]) + (a[2] + a[3]), 0.f) + 1e-3f;
        } else if constexpr (MODE == 4) {     // pure FMA ceiling: operands all VGPR, no broadcast (wrong math, timing only)
            float a[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 64; ++k) a[k & 3] = __builtin_fmaf(w[k], h, a[k & 3]);
            h = fmaxf((a[0] + a[1]) + (a[2] + a[3]), 0.f) + 1e-3f;
        } else if constexpr (MODE == 5) {     // readlane only (no FMA): cost of 64 v_readlane + 64 s->v adds
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 64; ++k) a += bcast(h, k);
            h = fmaxf(a * 1e-3f, 0.f) + w[it & 63] * 1e-6f;
        } else if constexpr (MODE == 6) {     // DPP row broadcast inside 16-lane rows + 4 readlanes?  (ds_bpermute broadcast)
            float a[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                float hk = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(k * 4, __builtin_bit_cast(int, h)));
                a[k & 3] = __builtin_fmaf(w[k], hk, a[k & 3]);
            }
            h = fmaxf((a[0] + a[1]) + (a[2] + a[3]), 0.f) + 1e-3f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = h;
}

template <int MODE, int WPB> void run(const char *name, const float *W, float *out, int nwaves, int iters)
{
    dim3 grid(nwaves / WPB), block(64 * WPB);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    layer_kernel<MODE, WPB><<<grid, block>>>(W, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    layer_kernel<MODE, WPB><<<grid, block>>>(W, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves resident per SIMD = nwaves / 1024 (256 CUs x 4 SIMDs); cycles per layer per SIMD-slot
    double ns_layer = ms * 1e6 / iters;
    double waves_per_simd = nwaves / 1024.0;
    printf("%-34s waves=%5d (%.0f/SIMD) WPB=%d  %8.1f ns/layer/wave  -> %7.1f cycles per layer per SIMD @2.4GHz  (%.2f cyc/MAC-instr)\n", name,
           nwaves, waves_per_simd, WPB, ns_layer, ns_layer * 2.4 / waves_per_simd, ns_layer * 2.4 / waves_per_simd / 64);
}

int main()
{
    float *W, *out;
    std::vector<float> hw(64 * 64);
    for (int i = 0; i < 64 * 64; ++i) hw[i] = 0.01f * ((i * 7919) % 13 - 6);
    hipMalloc(&W, sizeof(float) * 64 * 64);
    hipMalloc(&out, sizeof(float) * 64 * 8192);
    hipMemcpy(W, hw.data(), sizeof(float) * 64 * 64, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int nw : {1024, 2048, 4096}) {
        run<0, 1>("readlane 2acc", W, out, nw, iters);
        run<1, 1>("readlane 8acc", W, out, nw, iters);
        run<2, 1>("lds b128 bcast", W, out, nw, iters);
        run<2, 4>("lds b128 bcast (4 waves/WG)", W, out, nw, iters);
        run<3, 1>("half lds half readlane", W, out, nw, iters);
        run<4, 1>("pure fma (ceiling)", W, out, nw, iters);
        run<5, 1>("readlane+add only", W, out, nw, iters);
        run<6, 1>("ds_bpermute bcast", W, out, nw, iters);
    }
    return 0;
}
