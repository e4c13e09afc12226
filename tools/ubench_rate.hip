// ubench_rate.hip -- STEADY-STATE VALU issue rate of instruction patterns on gfx950.
// Every wave repeats a pattern until a fixed wall time has elapsed (s_memrealtime), so all waves stay
// co-resident for the whole measurement and there is no tail; the result is
//   SIMD cycles per unit = (SIMDs x mean elapsed shader cycles per wave) / (total units executed).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_rate.hip -o tools/ubench_rate
// Development tool (numbers quoted in DESIGN.md); not product code.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15", \
             "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
             "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"
#define REP2(b) b b
#define REP4(b) b b b b
#define REP8(b) b b b b b b b b

#define INIT_REGS                                                                                                  \
    asm volatile(                                                                                                  \
        "v_mov_b32 v0, 0x3f7fbe77\n v_mov_b32 v1, 0x3a83126f\n v_mov_b32 v2, 0x3f7fbe77\n v_mov_b32 v3, 0x3a83126f\n"  \
        "v_mov_b32 v4, 0x3f7fbe77\n v_mov_b32 v5, 0x3a83126f\n v_mov_b32 v6, 0x3f7fbe77\n v_mov_b32 v7, 0x3a83126f\n"  \
        "v_mov_b32 v8, 0x3f7fbe77\n v_mov_b32 v9, 0x3a83126f\n v_mov_b32 v10, 0x3f7fbe77\n v_mov_b32 v11, 0x3a83126f\n" \
        "v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n"                          \
        "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n"                          \
        "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n"                          \
        "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n"                          \
        "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n"                          \
        "v_mov_b32 v32, 0.5\n v_mov_b32 v33, 0.5\n v_mov_b32 v34, 0.5\n v_mov_b32 v35, 0.5\n"                          \
        "v_mov_b32 v36, 1.0\n v_mov_b32 v37, 1.0\n v_mov_b32 v38, 1.0\n v_mov_b32 v39, 1.0\n"                          \
        "v_mov_b32 v40, 1.0\n v_mov_b32 v41, 1.0\n v_mov_b32 v42, 1.0\n v_mov_b32 v43, 1.0\n"                          \
        "v_mov_b32 v44, 1.0\n v_mov_b32 v45, 1.0\n v_mov_b32 v46, 1.0\n v_mov_b32 v47, 1.0\n"                          \
        "v_mov_b32 v48, 1.0\n v_mov_b32 v49, 1.0\n v_mov_b32 v50, 1.0\n v_mov_b32 v51, 1.0\n"                          \
        "v_mov_b32 v52, 1.0\n v_mov_b32 v53, 1.0\n v_mov_b32 v54, 1.0\n v_mov_b32 v55, 1.0\n"                          \
        "v_mov_b32 v56, 1.0\n v_mov_b32 v57, 1.0\n v_mov_b32 v58, 1.0\n v_mov_b32 v59, 1.0\n"                          \
        "v_mov_b32 v60, 1.0\n v_mov_b32 v61, 1.0\n v_mov_b32 v62, 1.0\n v_mov_b32 v63, 1.0\n" ::: CLOB)

// One kernel per pattern: BODY is an asm string, executed until `ticks` of the 100 MHz clock have passed.
#define DEF_KERNEL(ID, BODY)                                                                                       \
    __global__ __launch_bounds__(256) void rate_##ID(unsigned long long *out, unsigned ticks, float sarg)          \
    {                                                                                                              \
        INIT_REGS;                                                                                                 \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                \
        unsigned long long bodies = 0;                                                                             \
        while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) {                                                    \
            asm volatile(BODY ::"s"(sarg) : CLOB);                                                                 \
            ++bodies;                                                                                              \
        }                                                                                                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                            \
        if ((threadIdx.x & 63) == 0) {                                                                             \
            const size_t w = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                 \
            out[4 * w + 0] = bodies;                                                                               \
            out[4 * w + 1] = t1 - t0;                                                                              \
            out[4 * w + 2] = r1 - r0;                                                                              \
            out[4 * w + 3] = r0;                                                                                   \
        }                                                                                                          \
    }

// ---------------------------------------------------------------------------------------------------------
// single-instruction patterns, 8 instructions each
#define P_FMAC   "v_fmac_f32_e32 v18, v0, v1\n v_fmac_f32_e32 v19, v0, v1\n v_fmac_f32_e32 v22, v0, v1\n v_fmac_f32_e32 v23, v0, v1\n" \
                 "v_fmac_f32_e32 v26, v0, v1\n v_fmac_f32_e32 v27, v0, v1\n v_fmac_f32_e32 v30, v0, v1\n v_fmac_f32_e32 v31, v0, v1\n"
#define P_FMAC7  "v_fmac_f32_e32 v18, v0, v1\n v_fmac_f32_e32 v19, v0, v1\n v_fmac_f32_e32 v22, v0, v1\n v_fmac_f32_e32 v23, v0, v1\n" \
                 "v_fmac_f32_e32 v26, v0, v1\n v_fmac_f32_e32 v27, v0, v1\n v_fmac_f32_e32 v30, v0, v1\n"
#define P_FMAC3  "v_fmac_f32_e32 v18, v0, v1\n v_fmac_f32_e32 v19, v0, v1\n v_fmac_f32_e32 v22, v0, v1\n"
#define P_MUL7   "v_mul_f32_e32 v18, v0, v18\n v_mul_f32_e32 v19, v0, v19\n v_mul_f32_e32 v22, v0, v22\n v_mul_f32_e32 v23, v0, v23\n" \
                 "v_mul_f32_e32 v26, v0, v26\n v_mul_f32_e32 v27, v0, v27\n v_mul_f32_e32 v30, v0, v30\n"
#define P_SRC01_SAME "v_fmac_f32_e32 v17, v0, v4\n v_fmac_f32_e32 v18, v0, v4\n v_fmac_f32_e32 v19, v0, v4\n v_fmac_f32_e32 v21, v0, v4\n" \
                     "v_fmac_f32_e32 v22, v0, v4\n v_fmac_f32_e32 v23, v0, v4\n v_fmac_f32_e32 v25, v0, v4\n v_fmac_f32_e32 v26, v0, v4\n"
#define P_SRC02_SAME "v_fmac_f32_e32 v16, v0, v1\n v_fmac_f32_e32 v20, v0, v1\n v_fmac_f32_e32 v24, v0, v1\n v_fmac_f32_e32 v28, v0, v1\n" \
                     "v_fmac_f32_e32 v12, v0, v1\n v_fmac_f32_e32 v8, v0, v1\n v_fmac_f32_e32 v16, v0, v1\n v_fmac_f32_e32 v20, v0, v1\n"
#define P_SRC12_SAME "v_fmac_f32_e32 v17, v0, v1\n v_fmac_f32_e32 v21, v0, v1\n v_fmac_f32_e32 v25, v0, v1\n v_fmac_f32_e32 v29, v0, v1\n" \
                     "v_fmac_f32_e32 v13, v0, v1\n v_fmac_f32_e32 v9, v0, v1\n v_fmac_f32_e32 v17, v0, v1\n v_fmac_f32_e32 v21, v0, v1\n"
#define P_SAMEREG "v_fmac_f32_e32 v17, v0, v0\n v_fmac_f32_e32 v21, v0, v0\n v_fmac_f32_e32 v25, v0, v0\n v_fmac_f32_e32 v29, v0, v0\n" \
                  "v_fmac_f32_e32 v13, v0, v0\n v_fmac_f32_e32 v9, v0, v0\n v_fmac_f32_e32 v18, v0, v0\n v_fmac_f32_e32 v22, v0, v0\n"
#define P_FMA_SGPR "v_fma_f32 v18, v0, %0, v18\n v_fma_f32 v19, v0, %0, v19\n v_fma_f32 v22, v0, %0, v22\n v_fma_f32 v23, v0, %0, v23\n" \
                   "v_fma_f32 v26, v0, %0, v26\n v_fma_f32 v27, v0, %0, v27\n v_fma_f32 v30, v0, %0, v30\n v_fma_f32 v31, v0, %0, v31\n"
#define P_RSQ    "v_rsq_f32_e32 v16, v16\n v_rsq_f32_e32 v17, v17\n v_rsq_f32_e32 v18, v18\n v_rsq_f32_e32 v19, v19\n" \
                 "v_rsq_f32_e32 v20, v20\n v_rsq_f32_e32 v21, v21\n v_rsq_f32_e32 v22, v22\n v_rsq_f32_e32 v23, v23\n"
#define RSQ1 "v_rsq_f32_e32 v16, v16\n"
#define RSQ2 "v_rsq_f32_e32 v16, v16\n v_rsq_f32_e32 v17, v17\n"
#define RSQ4 "v_rsq_f32_e32 v16, v16\n v_rsq_f32_e32 v17, v17\n v_rsq_f32_e32 v20, v20\n v_rsq_f32_e32 v21, v21\n"

DEF_KERNEL(fmac, REP8(P_FMAC))
DEF_KERNEL(src01, REP8(P_SRC01_SAME))
DEF_KERNEL(src02, REP8(P_SRC02_SAME))
DEF_KERNEL(src12, REP8(P_SRC12_SAME))
DEF_KERNEL(samereg, REP8(P_SAMEREG))
DEF_KERNEL(sgpr, REP8(P_FMA_SGPR))
DEF_KERNEL(rsq, REP8(P_RSQ))
DEF_KERNEL(r1f3, REP8(RSQ1 P_FMAC3))
DEF_KERNEL(r1f7, REP8(RSQ1 P_FMAC7))
DEF_KERNEL(r1f15, REP8(RSQ1 P_FMAC7 P_FMAC))
DEF_KERNEL(r1f31, REP8(RSQ1 P_FMAC7 P_FMAC P_FMAC P_FMAC))
DEF_KERNEL(r2f14, REP8(RSQ2 P_FMAC7 P_FMAC7))
DEF_KERNEL(r4f28, REP8(RSQ4 P_FMAC7 P_FMAC7 P_FMAC7 P_FMAC7))
DEF_KERNEL(r1m7, REP8(RSQ1 P_MUL7))
DEF_KERNEL(r1nop, REP8(RSQ1 "s_nop 7\n" P_FMAC7))

// ---------------------------------------------------------------------------------------------------------
// interactions.  pj = v0..v3 (x,y,z,m in banks 0,1,2,3), eps2 = v4.
// Conflict-free allocation (bank = index mod 4; src0/src1 of an instruction never share a bank):
//   row k coords X,Y,Z = v(4k+5), v(4k+6), v(4k+7)   (banks 1,2,3);  accumulators v20..v31
//   temp set t: R = v(32+8t) bank 0, Q = v(33+8t) bank 1, D0,D1,D2 = v(37+8t), v(38+8t), v(39+8t) banks 1,2,3
#define SUBS(X, Y, Z, D0, D1, D2) "v_sub_f32_e32 " D0 ", v0, " X "\n v_sub_f32_e32 " D1 ", v1, " Y "\n v_sub_f32_e32 " D2 ", v2, " Z "\n"
#define R2(D0, D1, D2, R) "v_fma_f32 " R ", " D0 ", " D0 ", v4\n v_fmac_f32_e32 " R ", " D1 ", " D1 "\n v_fmac_f32_e32 " R ", " D2 ", " D2 "\n"
#define RSQ(R) "v_rsq_f32_e32 " R ", " R "\n"
#define SCALE(R, Q) "v_mul_f32_e32 " Q ", " R ", " R "\n v_mul_f32_e32 " R ", v3, " R "\n v_mul_f32_e32 " R ", " R ", " Q "\n"
#define ACC(AX, AY, AZ, D0, D1, D2, R) "v_fmac_f32_e32 " AX ", " D0 ", " R "\n v_fmac_f32_e32 " AY ", " D1 ", " R "\n v_fmac_f32_e32 " AZ ", " D2 ", " R "\n"

#define ROW0 "v5", "v6", "v7"
#define ROW1 "v9", "v10", "v11"
#define ROW2 "v13", "v14", "v15"
#define ROW3 "v17", "v18", "v19"
#define ACC0 "v20", "v21", "v22"
#define ACC1 "v23", "v24", "v25"
#define ACC2 "v26", "v27", "v28"
#define ACC3 "v29", "v30", "v31"
#define DA "v37", "v38", "v39"
#define DB "v45", "v46", "v47"
#define DC "v53", "v54", "v55"
#define DD "v61", "v62", "v63"
#define RA "v32"
#define RB "v40"
#define RC "v48"
#define RD "v56"
#define QA "v33"
#define QB "v41"
#define QC "v49"
#define QD "v57"
#define XSUBS(...) SUBS(__VA_ARGS__)
#define XR2(...) R2(__VA_ARGS__)
#define XACC(...) ACC(__VA_ARGS__)

// S1: row after row, dependent chain, s_nop after the rsq (hipcc's shape), conflict-free registers
#define ROWSEQ(row, acc) XSUBS(row, DA) XR2(DA, RA) RSQ(RA) "s_nop 0\n" SCALE(RA, QA) XACC(acc, DA, RA)
DEF_KERNEL(seq4, REP4(ROWSEQ(ROW0, ACC0) ROWSEQ(ROW1, ACC1) ROWSEQ(ROW2, ACC2) ROWSEQ(ROW3, ACC3)))
// S2: the next row's subs fill the slot after the rsq (2 temp sets alternate)
#define S2BODY                                                                                                     \
    XSUBS(ROW0, DA) XR2(DA, RA) RSQ(RA) XSUBS(ROW1, DB) SCALE(RA, QA) XACC(ACC0, DA, RA)                             \
    XR2(DB, RB) RSQ(RB) XSUBS(ROW2, DA) SCALE(RB, QB) XACC(ACC1, DB, RB)                                             \
    XR2(DA, RA) RSQ(RA) XSUBS(ROW3, DB) SCALE(RA, QA) XACC(ACC2, DA, RA)                                             \
    XR2(DB, RB) RSQ(RB) "s_nop 0\n" SCALE(RB, QB) XACC(ACC3, DB, RB)
DEF_KERNEL(s2, REP4(S2BODY))
// S3: all four rows stage by stage, the four rsq back to back
#define S3BODY                                                                                                     \
    XSUBS(ROW0, DA) XSUBS(ROW1, DB) XSUBS(ROW2, DC) XSUBS(ROW3, DD)                                                  \
    XR2(DA, RA) XR2(DB, RB) XR2(DC, RC) XR2(DD, RD)                                                                  \
    RSQ(RA) RSQ(RB) RSQ(RC) RSQ(RD)                                                                                  \
    SCALE(RA, QA) SCALE(RB, QB) SCALE(RC, QC) SCALE(RD, QD)                                                          \
    XACC(ACC0, DA, RA) XACC(ACC1, DB, RB) XACC(ACC2, DC, RC) XACC(ACC3, DD, RD)
DEF_KERNEL(s3, REP4(S3BODY))
// S4: two rows at a time (two rsq back to back)
#define S4PAIR(rowa, acca, rowb, accb)                                                                             \
    XSUBS(rowa, DA) XSUBS(rowb, DB) XR2(DA, RA) XR2(DB, RB) RSQ(RA) RSQ(RB) SCALE(RA, QA) SCALE(RB, QB)              \
    XACC(acca, DA, RA) XACC(accb, DB, RB)
DEF_KERNEL(s4, REP4(S4PAIR(ROW0, ACC0, ROW1, ACC1) S4PAIR(ROW2, ACC2, ROW3, ACC3)))
// S5: like S3 but each row's chain kept contiguous except that all rsq are batched:
//     [subs+r2 row0][subs+r2 row1][subs+r2 row2][subs+r2 row3][rsq x4][scale+acc row0]...[scale+acc row3]
#define S5BODY                                                                                                     \
    XSUBS(ROW0, DA) XR2(DA, RA) XSUBS(ROW1, DB) XR2(DB, RB) XSUBS(ROW2, DC) XR2(DC, RC) XSUBS(ROW3, DD) XR2(DD, RD)  \
    RSQ(RA) RSQ(RB) RSQ(RC) RSQ(RD)                                                                                  \
    SCALE(RA, QA) XACC(ACC0, DA, RA) SCALE(RB, QB) XACC(ACC1, DB, RB) SCALE(RC, QC) XACC(ACC2, DC, RC)               \
    SCALE(RD, QD) XACC(ACC3, DD, RD)
DEF_KERNEL(s5, REP4(S5BODY))
// S6: S1 without the s_nop but with the rsq result consumed one row later (rsq of row k+1 issued before the
//     scale of row k): chain contiguous, one rsq in flight
#define S6BODY                                                                                                     \
    XSUBS(ROW0, DA) XR2(DA, RA) RSQ(RA)                                                                              \
    XSUBS(ROW1, DB) XR2(DB, RB) SCALE(RA, QA) XACC(ACC0, DA, RA) RSQ(RB)                                             \
    XSUBS(ROW2, DA) XR2(DA, RA) SCALE(RB, QB) XACC(ACC1, DB, RB) RSQ(RA)                                             \
    XSUBS(ROW3, DB) XR2(DB, RB) SCALE(RA, QA) XACC(ACC2, DA, RA) RSQ(RB)                                             \
    "s_nop 0\n" SCALE(RB, QB) XACC(ACC3, DB, RB)
DEF_KERNEL(s6, REP4(S6BODY))
// the 12 non-transcendental instructions of an interaction alone (what the rsq and its neighbours cost on top)
#define ROWNORSQ(row, acc) XSUBS(row, DA) XR2(DA, RA) SCALE(RA, QA) XACC(acc, DA, RA)
DEF_KERNEL(norsq4, REP4(ROWNORSQ(ROW0, ACC0) ROWNORSQ(ROW1, ACC1) ROWNORSQ(ROW2, ACC2) ROWNORSQ(ROW3, ACC3)))

struct Result { double cyc_per_unit, clk, skew_us; };

template <typename Kern>
static Result run(Kern kern, int cus, int blocks_per_cu, double units_per_body, unsigned long long *dev,
                  std::vector<unsigned long long> &h)
{
    const int blocks = cus * blocks_per_cu, nw = blocks * 4;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dev, 2000u, 0.999f);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dev, 30000u, 0.999f);  // 300 us
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), dev, sizeof(unsigned long long) * nw * 4, hipMemcpyDeviceToHost);
    double bodies = 0, cyc = 0, ns = 0;
    unsigned long long first = ~0ull, last = 0;
    for (int i = 0; i < nw; ++i) {
        bodies += (double)h[4 * i];
        cyc += (double)h[4 * i + 1];
        ns += (double)h[4 * i + 2] * 10.0;
        first = first < h[4 * i + 3] ? first : h[4 * i + 3];
        last = last > h[4 * i + 3] ? last : h[4 * i + 3];
    }
    Result r;
    r.cyc_per_unit = cus * 4.0 * (cyc / nw) / (bodies * units_per_body);
    r.clk = cyc / ns;
    r.skew_us = (double)(last - first) * 0.01;
    return r;
}

#define RUN(ID, NAME, UNITS)                                                                                       \
    do {                                                                                                           \
        printf("%-64s", NAME);                                                                                     \
        for (int b : {2, 4, 7}) {                                                                                  \
            Result r = run(rate_##ID, cus, b, UNITS, dev, h);                                                      \
            printf("  w%d: %6.2f%s", b, r.cyc_per_unit, r.skew_us > 5 ? "(!)" : "   ");                           \
        }                                                                                                          \
        printf("\n");                                                                                              \
    } while (0)

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess)
        return 1;
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs=%d; SIMD cycles per unit at 2, 4, 7 waves/SIMD; (!) = waves were not all co-resident\n",
           prop.gcnArchName, cus);
    unsigned long long *dev;
    (void)hipMalloc((void **)&dev, sizeof(unsigned long long) * cus * 8 * 4 * 4);
    std::vector<unsigned long long> h((size_t)cus * 8 * 4 * 4);
    printf("-- per instruction --\n");
    RUN(fmac, "v_fmac, operands in 3 banks", 64);
    RUN(src01, "v_fmac, src0+src1 same bank", 64);
    RUN(src02, "v_fmac, src0+acc same bank", 64);
    RUN(src12, "v_fmac, src1+acc same bank", 64);
    RUN(samereg, "v_fmac, src0 == src1 (same register)", 64);
    RUN(sgpr, "v_fma, SGPR operand", 64);
    RUN(rsq, "v_rsq", 64);
    printf("-- per group --\n");
    RUN(r1f3, "1 rsq + 3 fmac    (additive model: 8 + 3 x 2.1 = 14.3)", 8);
    RUN(r1f7, "1 rsq + 7 fmac    (additive model: 22.7)", 8);
    RUN(r1f15, "1 rsq + 15 fmac   (additive model: 39.5)", 8);
    RUN(r1f31, "1 rsq + 31 fmac   (additive model: 73.1)", 8);
    RUN(r2f14, "2 rsq + 14 fmac   (additive model: 45.4)", 8);
    RUN(r4f28, "4 rsq + 28 fmac   (additive model: 90.8)", 8);
    RUN(r1m7, "1 rsq + 7 mul", 8);
    RUN(r1nop, "1 rsq + s_nop 7 + 7 fmac", 8);
    printf("-- per interaction (13 VALU; additive model 12 x 2.1 + 8 = 33.2) --\n");
    RUN(norsq4, "12 non-transcendental instructions only (model 25.2)", 16);
    RUN(seq4, "S1 row after row + s_nop, conflict-free registers", 16);
    RUN(s2, "S2 next row's subs after the rsq", 16);
    RUN(s6, "S6 rsq consumed one row later", 16);
    RUN(s4, "S4 two rows stage by stage (2 rsq back to back)", 16);
    RUN(s3, "S3 four rows stage by stage (4 rsq back to back)", 16);
    RUN(s5, "S5 four chains contiguous, 4 rsq batched", 16);
    return 0;
}
