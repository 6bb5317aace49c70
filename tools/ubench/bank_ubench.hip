// Does the VGPR bank (register index mod 4) of the operands of v_fmac_f32_dpp matter?  Hard-coded registers, 64 FMAs per block.
// dst = 4 x 16 accumulators; src0 (DPP) and src1 chosen so that their banks collide with dst's or not.
// Build: hipcc -O3 --offload-arch=gfx950 bank_ubench.hip -o bank_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>

// accumulators v100..v163, src0 candidates v170..v173 (banks 2,3,0,1), src1 candidates v180..v243
#define CLOBBERS "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115", \
  "v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131", \
  "v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147", \
  "v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163", \
  "v170","v171","v172","v173"

#define I(D, S0, S1) "v_fmac_f32_dpp v" #D ", v" #S0 ", v" #S1 " row_ror:3 row_mask:0xf bank_mask:0xf\n"
// 16 instructions with dst D0..D0+15; S0 pattern a,b,c,d repeated; S1 = T0 + i
#define BLK16(D0,D1,D2,D3,D4,D5,D6,D7,D8,D9,D10,D11,D12,D13,D14,D15, A,B,C,Dd, T0,T1,T2,T3,T4,T5,T6,T7,T8,T9,T10,T11,T12,T13,T14,T15) \
  I(D0,A,T0) I(D1,B,T1) I(D2,C,T2) I(D3,Dd,T3) I(D4,A,T4) I(D5,B,T5) I(D6,C,T6) I(D7,Dd,T7) I(D8,A,T8) I(D9,B,T9) I(D10,C,T10) I(D11,Dd,T11) I(D12,A,T12) I(D13,B,T13) I(D14,C,T14) I(D15,Dd,T15)

// dst v100+i has bank i%4.  src0 v170..173 = banks 2,3,0,1.
// NOCONF: dst bank b, src0 bank b+2, src1 bank b+1  -> src0 = 170+(i%4) [banks 2,3,0,1 for i%4=0..3: = b+2 ok], src1 = 181+i -> bank (1+i)%4 = b+1 ok
#define NOCONF BLK16(100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115, 170,171,172,173, 181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196)
// S0CONF: src0 bank == dst bank: src0 for i%4=0 must be bank 0 -> v172, 1->v173, 2->v170, 3->v171
#define S0CONF BLK16(100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115, 172,173,170,171, 181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196)
// S1CONF: src1 bank == dst bank: src1 = 180+i (bank i%4)
#define S1CONF BLK16(100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115, 170,171,172,173, 180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195)
// S01CONF: src0 bank == src1 bank != dst: src0 bank b+2 (170+(i%4)), src1 bank b+2 -> 182+i
#define S01CONF BLK16(100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115, 170,171,172,173, 182,183,184,185,186,187,188,189,190,191,192,193,194,195,196,197)
// ALLCONF: all three in dst's bank
#define ALLCONF BLK16(100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115, 172,173,170,171, 180,181,182,183,184,185,186,187,188,189,190,191,192,193,194,195)
// SAMEACC: the kernels' real pattern -- 4 accumulators (consecutive registers) x 16 different weights: dst v100..103 cycling
#define SAMEACC4 BLK16(100,101,102,103,100,101,102,103,100,101,102,103,100,101,102,103, 170,171,172,173, 181,182,183,184,185,186,187,188,189,190,191,192,193,194,195,196)

#define KERNEL(NAME, BODY)                                                                                      \
    __global__ __launch_bounds__(256) void NAME(float *out, unsigned long long *st, int iters)                  \
    {                                                                                                           \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                   \
        for (int it = 0; it < iters; ++it) {                                                                    \
            asm volatile(BODY BODY BODY BODY ::: CLOBBERS);                                                      \
        }                                                                                                       \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                   \
        float r;                                                                                                \
        asm volatile("v_add_f32 %0, v100, v101" : "=v"(r));                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                         \
        if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;                                                         \
    }
KERNEL(k_noconf, NOCONF)
KERNEL(k_s0conf, S0CONF)
KERNEL(k_s1conf, S1CONF)
KERNEL(k_s01conf, S01CONF)
KERNEL(k_allconf, ALLCONF)
KERNEL(k_sameacc4, SAMEACC4)

template <typename K> void run(const char *name, K kern, float *out, unsigned long long *st)
{
    const int iters = 2000;
    printf("%-12s", name);
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;
        kern<<<blocks, 256>>>(out, st, 20); (void)hipDeviceSynchronize();
        kern<<<blocks, 256>>>(out, st, iters); (void)hipDeviceSynchronize();
        static unsigned long long h[1024];
        (void)hipMemcpy(h, st, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
        printf("  %dw/SIMD: %5.2f cyc/inst/SIMD", wps, avg / ((double)iters * 64) / wps);
    }
    printf("\n");
}
int main()
{
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 4 * 256 * 1024); (void)hipMalloc(&st, 8 * 1024);
    run("no conflict", k_noconf, out, st); run("src0==dst", k_s0conf, out, st); run("src1==dst", k_s1conf, out, st);
    run("src0==src1", k_s01conf, out, st); run("all same", k_allconf, out, st); run("4 acc cycle", k_sameacc4, out, st);
    return 0;
}
