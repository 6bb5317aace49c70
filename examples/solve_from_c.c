/*
 * solve_from_c.c -- the C ABI of libhode.so used from plain C: no Python, no torch.
 *
 * Integrates a small cohort (B patients, T grid points, 5-minute grid) with a fully populated 4x64 MLP residual (every
 * weight and bias drawn from a 32-bit LCG that tests/test_c_example.py reproduces, so all hidden layers multiply non-zero
 * numbers), once with DP5(4) in fp32 and once in fp64, and prints the final states.  tests/test_c_example.py builds it, runs it
 * on the GPU box and checks the printed numbers against the oracle.
 *
 * Build:  hipcc -x c -I include examples/solve_from_c.c -L hybrid-ode-for-glp-1-and-glucose_amd/hode -lhode \
 *               -Wl,-rpath,$PWD/hybrid-ode-for-glp-1-and-glucose_amd/hode -o examples/solve_from_c
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hode.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

enum { B = 5, T = 25, H = 64, L = 4 };

static void *to_device(const void *src, size_t n)
{
    void *d = NULL;
    if (hipMalloc(&d, n) != hipSuccess || hipMemcpy(d, src, n, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(void)
{
    const int P = hode_nn_param_count(H, L);
    printf("%s; P = %d\n", hode_version(), P);
    /* ODECore constants in registration order (models/ode_core.py:44-71) */
    const double ode[17] = {0.0104, 0.025, 0.003, 5.0, 60.0, 0.1, 50.0, 80.0, 9.0, 7.0, 0.02, 0.01, 1000.0, 2.0, 0.05, 0.001, 0.01};
    double *nn = calloc((size_t)P, sizeof(double));
    /* flat layout = PyTorch parameters() order: W1[H,9] b1[H] (W[H,H] b[H]) x (L-1) Wout[6,H] bout[6].  Numerical-Recipes LCG,
     * u = top 24 bits / 2^24 - 0.5; hidden tensors +-0.15, the output layer +-0.01 (a residual of the size training produces) */
    {
        uint32_t s = 12345u;
        const int n_out = 6 * H + 6;
        for (int i = 0; i < P; ++i) {
            s = s * 1664525u + 1013904223u;
            nn[i] = ((double)(s >> 8) / 16777216.0 - 0.5) * (i >= P - n_out ? 0.02 : 0.3);
        }
    }
    double x0[B * 6], t[T], meal[B * T];
    const double base[6] = {5.0, 60.0, 80.0, 10.0, 0.0, 1.0};
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < 6; ++c) x0[b * 6 + c] = base[c] * (1.0 + 0.02 * b);
    for (int k = 0; k < T; ++k) t[k] = k * (5.0 / 60.0);
    memset(meal, 0, sizeof meal);
    for (int b = 0; b < B; ++b) meal[b * T + 3 + b] = 1.0;                      /* one unit pulse per patient */

    for (int pass = 0; pass < 2; ++pass) {
        const size_t es = pass ? sizeof(double) : sizeof(float);
        /* convert host data to the pass' element type */
        void *hx = malloc(B * 6 * es), *ht = malloc(T * es), *hm = malloc(B * T * es), *ho = malloc(17 * es), *hn = malloc((size_t)P * es);
        for (int i = 0; i < B * 6; ++i) { if (pass) ((double *)hx)[i] = x0[i]; else ((float *)hx)[i] = (float)x0[i]; }
        for (int i = 0; i < T; ++i) { if (pass) ((double *)ht)[i] = t[i]; else ((float *)ht)[i] = (float)t[i]; }
        for (int i = 0; i < B * T; ++i) { if (pass) ((double *)hm)[i] = meal[i]; else ((float *)hm)[i] = (float)meal[i]; }
        for (int i = 0; i < 17; ++i) { if (pass) ((double *)ho)[i] = (double)(float)ode[i]; else ((float *)ho)[i] = (float)ode[i]; }
        for (int i = 0; i < P; ++i) { if (pass) ((double *)hn)[i] = (double)(float)nn[i]; else ((float *)hn)[i] = (float)nn[i]; }
        void *dx = to_device(hx, B * 6 * es), *dt = to_device(ht, T * es), *dm = to_device(hm, B * T * es),
             *dode = to_device(ho, 17 * es), *dnn = to_device(hn, (size_t)P * es), *dy = NULL;
        int32_t *dst = NULL, st[B];
        if (!dx || !dt || !dm || !dode || !dnn) return 2;
        CHECK(hipMalloc(&dy, B * T * 6 * es));
        CHECK(hipMalloc((void **)&dst, B * sizeof(int32_t)));
        hipStream_t stream;
        CHECK(hipStreamCreate(&stream));
        int rc;
        if (pass)
            rc = hode_solve_fwd_f64(stream, B, T, dx, dt, 0, dm, 2, NULL, 0, NULL, 0, dode, dnn, 1, H, L, HODE_METHOD_DP54, 1e-10,
                                    1e-12, 4000, dy, dst, NULL, NULL, NULL);
        else
            rc = hode_solve_fwd_f32(stream, B, T, dx, dt, 0, dm, 2, NULL, 0, NULL, 0, dode, dnn, 1, H, L, HODE_METHOD_DP54, 1e-6,
                                    1e-8, 4000, dy, dst, NULL, NULL, NULL);
        if (rc != HODE_OK) { fprintf(stderr, "hode_solve_fwd -> %d\n", rc); return 3; }
        CHECK(hipStreamSynchronize(stream));
        void *hy = malloc(B * T * 6 * es);
        CHECK(hipMemcpy(hy, dy, B * T * 6 * es, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(st, dst, sizeof st, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {
            printf("%s b=%d status=%d y_end=", pass ? "f64" : "f32", b, st[b]);
            for (int c = 0; c < 6; ++c)
                printf(" %.9e", pass ? ((double *)hy)[(b * T + T - 1) * 6 + c] : (double)((float *)hy)[(b * T + T - 1) * 6 + c]);
            printf("\n");
        }
        /* an invalid call is refused before any launch */
        if (pass == 0 && hode_solve_fwd_f32(stream, B, T, dx, dt, 0, dm, 2, NULL, 0, NULL, 0, dode, dnn, 1, 1024, L, HODE_METHOD_DP54,
                                            1e-6, 1e-8, 4000, dy, dst, NULL, NULL, NULL) != HODE_EUNSUPPORTED) return 4;
        hipFree(dx); hipFree(dt); hipFree(dm); hipFree(dode); hipFree(dnn); hipFree(dy); hipFree(dst);
        free(hx); free(ht); free(hm); free(ho); free(hn); free(hy);
        CHECK(hipStreamDestroy(stream));
    }
    free(nn);
    return 0;
}
