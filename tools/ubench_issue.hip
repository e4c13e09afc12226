// ubench_issue.hip -- what does one VALU wave-instruction cost on a gfx950 SIMD, as a function of
// operand register banks, encoding, dependency distance and residency?  Hard-coded registers.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_issue.hip -o tools/ubench_issue
// Development tool (feeds DESIGN.md's VALU cost table); not product code.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15", \
             "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"

// 8 instructions per body, bodies repeated 8x straight-line inside the loop
#define R8(s0,s1,s2,s3,s4,s5,s6,s7) s0 "\n" s1 "\n" s2 "\n" s3 "\n" s4 "\n" s5 "\n" s6 "\n" s7 "\n"
#define REP8(b) b b b b b b b b

enum { T_FMA_DISTINCT, T_FMA_SAMEBANK, T_FMAC_E32, T_FMA_DEP1, T_FMA_DEP2, T_FMA_DEP4, T_MUL_DISTINCT, T_MUL_SAMEBANK,
       T_SUB_E32, T_RSQ, T_RSQ_THEN_INDEP, T_INTERACT_SEQ, T_INTERACT_ILV, T_FMA_SGPR, T_PKFMA, T_COUNT };
static const char *kName[] = {"fma  srcs in 3 banks, 8 chains", "fma  srcs in 1 bank, 8 chains", "fmac_e32 (VOP2), 8 chains",
                              "fmac dependent distance 1", "fmac dependent distance 2", "fmac dependent distance 4",
                              "mul  srcs in 2 banks", "mul  srcs in 1 bank", "sub_e32", "rsq 8 chains",
                              "1 rsq + 7 indep fmac", "interaction x4 rows, row after row (52+4)",
                              "interaction x4 rows, interleaved (52+4)", "fma with SGPR src", "pk_fma 8 chains"};
static const int kInstr[] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 56, 52, 8, 8};

template <int T>
__global__ __launch_bounds__(256) void bench(float *out, unsigned long long *stamps, int iters, float sarg)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // registers: v0..v15 "constants" (finite, ~1), v16..v31 accumulators
    asm volatile(
        "v_mov_b32 v0, 0x3f7fbe77\n v_mov_b32 v1, 0x3a83126f\n v_mov_b32 v2, 0x3f7fbe77\n v_mov_b32 v3, 0x3a83126f\n"
        "v_mov_b32 v4, 0x3f7fbe77\n v_mov_b32 v5, 0x3a83126f\n v_mov_b32 v6, 0x3f7fbe77\n v_mov_b32 v7, 0x3a83126f\n"
        "v_mov_b32 v8, 0x3f7fbe77\n v_mov_b32 v9, 0x3a83126f\n v_mov_b32 v10, 0x3f7fbe77\n v_mov_b32 v11, 0x3a83126f\n"
        "v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n"
        "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n"
        "v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n"
        "v_mov_b32 v24, 1.0\n v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n"
        "v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n" ::: CLOB);
    for (int it = 0; it < iters; ++it) {
        if (T == T_FMA_DISTINCT)  // srcs v0 (bank 0), v1 (bank 1), acc v16+i
            asm volatile(REP8(R8("v_fma_f32 v18, v0, v1, v18", "v_fma_f32 v19, v0, v1, v19", "v_fma_f32 v22, v0, v1, v22",
                                 "v_fma_f32 v23, v0, v1, v23", "v_fma_f32 v26, v0, v1, v26", "v_fma_f32 v27, v0, v1, v27",
                                 "v_fma_f32 v30, v0, v1, v30", "v_fma_f32 v31, v0, v1, v31")) ::: CLOB);
        else if (T == T_FMA_SAMEBANK)  // srcs v0, v4 and acc v16+4k: all bank 0 (index mod 4)
            asm volatile(REP8(R8("v_fma_f32 v16, v0, v4, v16", "v_fma_f32 v20, v0, v4, v20", "v_fma_f32 v24, v0, v4, v24",
                                 "v_fma_f32 v28, v0, v4, v28", "v_fma_f32 v12, v0, v4, v12", "v_fma_f32 v8, v0, v4, v8",
                                 "v_fma_f32 v16, v0, v4, v16", "v_fma_f32 v20, v0, v4, v20")) ::: CLOB);
        else if (T == T_FMAC_E32)
            asm volatile(REP8(R8("v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v19, v0, v1", "v_fmac_f32_e32 v22, v0, v1",
                                 "v_fmac_f32_e32 v23, v0, v1", "v_fmac_f32_e32 v26, v0, v1", "v_fmac_f32_e32 v27, v0, v1",
                                 "v_fmac_f32_e32 v30, v0, v1", "v_fmac_f32_e32 v31, v0, v1")) ::: CLOB);
        else if (T == T_FMA_DEP1)
            asm volatile(REP8(R8("v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v18, v0, v1",
                                 "v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v18, v0, v1",
                                 "v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v18, v0, v1")) ::: CLOB);
        else if (T == T_FMA_DEP2)
            asm volatile(REP8(R8("v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v19, v0, v1", "v_fmac_f32_e32 v18, v0, v1",
                                 "v_fmac_f32_e32 v19, v0, v1", "v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v19, v0, v1",
                                 "v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v19, v0, v1")) ::: CLOB);
        else if (T == T_FMA_DEP4)
            asm volatile(REP8(R8("v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v19, v0, v1", "v_fmac_f32_e32 v22, v0, v1",
                                 "v_fmac_f32_e32 v23, v0, v1", "v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v19, v0, v1",
                                 "v_fmac_f32_e32 v22, v0, v1", "v_fmac_f32_e32 v23, v0, v1")) ::: CLOB);
        else if (T == T_MUL_DISTINCT)
            asm volatile(REP8(R8("v_mul_f32_e32 v17, v0, v17", "v_mul_f32_e32 v18, v0, v18", "v_mul_f32_e32 v19, v0, v19",
                                 "v_mul_f32_e32 v21, v0, v21", "v_mul_f32_e32 v22, v0, v22", "v_mul_f32_e32 v23, v0, v23",
                                 "v_mul_f32_e32 v25, v0, v25", "v_mul_f32_e32 v26, v0, v26")) ::: CLOB);
        else if (T == T_MUL_SAMEBANK)
            asm volatile(REP8(R8("v_mul_f32_e32 v16, v0, v16", "v_mul_f32_e32 v20, v0, v20", "v_mul_f32_e32 v24, v0, v24",
                                 "v_mul_f32_e32 v28, v0, v28", "v_mul_f32_e32 v12, v0, v12", "v_mul_f32_e32 v8, v0, v8",
                                 "v_mul_f32_e32 v16, v0, v16", "v_mul_f32_e32 v20, v0, v20")) ::: CLOB);
        else if (T == T_SUB_E32)
            asm volatile(REP8(R8("v_sub_f32_e32 v17, v1, v17", "v_sub_f32_e32 v18, v1, v18", "v_sub_f32_e32 v19, v1, v19",
                                 "v_sub_f32_e32 v21, v1, v21", "v_sub_f32_e32 v22, v1, v22", "v_sub_f32_e32 v23, v1, v23",
                                 "v_sub_f32_e32 v25, v1, v25", "v_sub_f32_e32 v26, v1, v26")) ::: CLOB);
        else if (T == T_RSQ)
            asm volatile(REP8(R8("v_rsq_f32_e32 v16, v16", "v_rsq_f32_e32 v17, v17", "v_rsq_f32_e32 v18, v18",
                                 "v_rsq_f32_e32 v19, v19", "v_rsq_f32_e32 v20, v20", "v_rsq_f32_e32 v21, v21",
                                 "v_rsq_f32_e32 v22, v22", "v_rsq_f32_e32 v23, v23")) ::: CLOB);
        else if (T == T_RSQ_THEN_INDEP)
            asm volatile(REP8(R8("v_rsq_f32_e32 v16, v16", "v_fmac_f32_e32 v18, v0, v1", "v_fmac_f32_e32 v19, v0, v1",
                                 "v_fmac_f32_e32 v22, v0, v1", "v_fmac_f32_e32 v23, v0, v1", "v_fmac_f32_e32 v26, v0, v1",
                                 "v_fmac_f32_e32 v27, v0, v1", "v_fmac_f32_e32 v30, v0, v1")) ::: CLOB);
        else if (T == T_FMA_SGPR)
            asm volatile(REP8(R8("v_fma_f32 v18, v0, %0, v18", "v_fma_f32 v19, v0, %0, v19", "v_fma_f32 v22, v0, %0, v22",
                                 "v_fma_f32 v23, v0, %0, v23", "v_fma_f32 v26, v0, %0, v26", "v_fma_f32 v27, v0, %0, v27",
                                 "v_fma_f32 v30, v0, %0, v30", "v_fma_f32 v31, v0, %0, v31")) ::"s"(sarg) : CLOB);
        else if (T == T_PKFMA)
            asm volatile(REP8(R8("v_pk_fma_f32 v[16:17], v[0:1], v[2:3], v[16:17]", "v_pk_fma_f32 v[18:19], v[0:1], v[2:3], v[18:19]",
                                 "v_pk_fma_f32 v[20:21], v[0:1], v[2:3], v[20:21]", "v_pk_fma_f32 v[22:23], v[0:1], v[2:3], v[22:23]",
                                 "v_pk_fma_f32 v[24:25], v[0:1], v[2:3], v[24:25]", "v_pk_fma_f32 v[26:27], v[0:1], v[2:3], v[26:27]",
                                 "v_pk_fma_f32 v[28:29], v[0:1], v[2:3], v[28:29]", "v_pk_fma_f32 v[30:31], v[0:1], v[2:3], v[30:31]")) ::: CLOB);
        else if (T == T_INTERACT_SEQ) {
            // pj = v0..v3 (x,y,z,m); rows k: xi,yi,zi = v4+3k..; acc = v16+3k..; temps v28..v31; eps2 = v15
            // one row after the other, each a dependent chain (what hipcc emits), s_nop after the rsq
#define ROW_SEQ(X, Y, Z, AX, AY, AZ)                                                                             \
    "v_sub_f32_e32 v28, v0, " X "\n v_sub_f32_e32 v29, v1, " Y "\n v_fma_f32 v31, v28, v28, v15\n"                   \
    "v_sub_f32_e32 v30, v2, " Z "\n v_fmac_f32_e32 v31, v29, v29\n v_fmac_f32_e32 v31, v30, v30\n"                   \
    "v_rsq_f32_e32 v31, v31\n s_nop 0\n v_mul_f32_e32 v14, v31, v31\n v_mul_f32_e32 v31, v3, v31\n"                  \
    "v_mul_f32_e32 v31, v31, v14\n v_fmac_f32_e32 " AX ", v28, v31\n v_fmac_f32_e32 " AY ", v29, v31\n"              \
    "v_fmac_f32_e32 " AZ ", v30, v31\n"
            asm volatile(REP8(ROW_SEQ("v4", "v5", "v6", "v16", "v17", "v18") ROW_SEQ("v7", "v8", "v9", "v19", "v20", "v21")
                              ROW_SEQ("v10", "v11", "v12", "v22", "v23", "v24") ROW_SEQ("v4", "v8", "v12", "v25", "v26", "v27")) ::: CLOB);
        } else if (T == T_INTERACT_ILV) {
            // same 52 VALU + 4 rsq, the four rows interleaved stage by stage (no instruction reads its predecessor)
            // temps: row0 v28,v29,v30,v31 ; rows share? no: use distinct temps per row from v13,v14 + acc area is needed;
            // keep it legal with 4 temps per row would need 16 registers: reuse v0..v3 is not possible, so rows are
            // processed two at a time (2 x 2 interleave), temps row A: v28..v31, row B: v12,v13,v14 + v11.
#define PAIR_ILV(XA, YA, ZA, AXA, AYA, AZA, XB, YB, ZB, AXB, AYB, AZB)                                                 \
    "v_sub_f32_e32 v28, v0, " XA "\n v_sub_f32_e32 v12, v0, " XB "\n v_sub_f32_e32 v29, v1, " YA "\n"                    \
    "v_sub_f32_e32 v13, v1, " YB "\n v_sub_f32_e32 v30, v2, " ZA "\n v_sub_f32_e32 v14, v2, " ZB "\n"                    \
    "v_fma_f32 v31, v28, v28, v15\n v_fma_f32 v11, v12, v12, v15\n v_fmac_f32_e32 v31, v29, v29\n"                       \
    "v_fmac_f32_e32 v11, v13, v13\n v_fmac_f32_e32 v31, v30, v30\n v_fmac_f32_e32 v11, v14, v14\n"                       \
    "v_rsq_f32_e32 v31, v31\n v_rsq_f32_e32 v11, v11\n v_mul_f32_e32 v10, v31, v31\n v_mul_f32_e32 v31, v3, v31\n"      \
    "v_mul_f32_e32 v9, v11, v11\n v_mul_f32_e32 v11, v3, v11\n v_mul_f32_e32 v31, v31, v10\n v_mul_f32_e32 v11, v11, v9\n" \
    "v_fmac_f32_e32 " AXA ", v28, v31\n v_fmac_f32_e32 " AXB ", v12, v11\n v_fmac_f32_e32 " AYA ", v29, v31\n"           \
    "v_fmac_f32_e32 " AYB ", v13, v11\n v_fmac_f32_e32 " AZA ", v30, v31\n v_fmac_f32_e32 " AZB ", v14, v11\n"
            asm volatile(REP8(PAIR_ILV("v4", "v5", "v6", "v16", "v17", "v18", "v7", "v8", "v5", "v19", "v20", "v21")
                              PAIR_ILV("v6", "v7", "v4", "v22", "v23", "v24", "v5", "v8", "v7", "v25", "v26", "v27")) ::: CLOB);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s;
    asm volatile("v_add_f32 %0, v16, v17\n v_add_f32 %0, %0, v18\n v_add_f32 %0, %0, v19\n v_add_f32 %0, %0, v20\n"
                 "v_add_f32 %0, %0, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v23\n v_add_f32 %0, %0, v24\n"
                 "v_add_f32 %0, %0, v25\n v_add_f32 %0, %0, v26\n v_add_f32 %0, %0, v27\n v_add_f32 %0, %0, v28\n"
                 "v_add_f32 %0, %0, v29\n v_add_f32 %0, %0, v30\n v_add_f32 %0, %0, v31\n v_add_f32 %0, %0, v12\n"
                 : "=v"(s)::CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        stamps[4 * w + 0] = t1 - t0;
        stamps[4 * w + 1] = r0;
        stamps[4 * w + 2] = r1;
        stamps[4 * w + 3] = 0;
    }
}

template <int T>
static void run(int cus, int blocks_per_cu, int iters, float *out, unsigned long long *st, std::vector<unsigned long long> &h)
{
    const int blocks = cus * blocks_per_cu;
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bench<T>, 256, 0);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<T>, dim3(blocks), dim3(256), 0, 0, out, st, iters / 8, 0.999f);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(bench<T>, dim3(blocks), dim3(256), 0, 0, out, st, iters, 0.999f);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int nw = blocks * 4;
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * nw * 4, hipMemcpyDeviceToHost);
    std::vector<double> cyc(nw), dur(nw);
    unsigned long long first = ~0ull, last = 0, last_start = 0;
    for (int i = 0; i < nw; ++i) {
        cyc[i] = (double)h[4 * i];
        dur[i] = (double)(h[4 * i + 2] - h[4 * i + 1]) * 10.0;  // ns
        first = std::min(first, h[4 * i + 1]);
        last = std::max(last, h[4 * i + 2]);
        last_start = std::max(last_start, h[4 * i + 1]);
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(dur.begin(), dur.end());
    const double span_ns = (double)(last - first) * 10.0, late_ns = (double)(last_start - first) * 10.0;
    const double instr_wave = (double)iters * 8 * kInstr[T];
    const double instr_simd = instr_wave * blocks_per_cu;  // one wave of each block per SIMD
    printf("%-44s blk/CU=%d occ=%d  wave %6.2f cyc/instr  SIMD %6.3f cyc/instr (by span: %6.3f ns/instr = %5.2f cyc @clk)  "
           "clk %.2f GHz  span %.3f ms  last wave started at %.3f ms  kernel %.3f ms\n",
           kName[T], blocks_per_cu, occ, cyc[nw / 2] / instr_wave, cyc[nw / 2] / instr_simd, span_ns / instr_simd,
           span_ns / instr_simd * (cyc[nw / 2] / dur[nw / 2]), cyc[nw / 2] / dur[nw / 2], span_ns * 1e-6, late_ns * 1e-6, ms);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess)
        return 1;
    const int cus = prop.multiProcessorCount;
    printf("device %s CUs=%d\n", prop.gcnArchName, cus);
    float *out;
    unsigned long long *st;
    (void)hipMalloc((void **)&out, sizeof(float) * cus * 8 * 256);
    (void)hipMalloc((void **)&st, sizeof(unsigned long long) * cus * 8 * 4 * 4);
    std::vector<unsigned long long> h((size_t)cus * 8 * 4 * 4);
    const int iters = 2000;
    for (int b : {1, 2, 3, 4, 6, 8}) {
        run<T_FMA_DISTINCT>(cus, b, iters, out, st, h);
        run<T_FMA_SAMEBANK>(cus, b, iters, out, st, h);
        run<T_FMAC_E32>(cus, b, iters, out, st, h);
        run<T_FMA_DEP1>(cus, b, iters, out, st, h);
        run<T_FMA_DEP2>(cus, b, iters, out, st, h);
        run<T_FMA_DEP4>(cus, b, iters, out, st, h);
        run<T_MUL_DISTINCT>(cus, b, iters, out, st, h);
        run<T_MUL_SAMEBANK>(cus, b, iters, out, st, h);
        run<T_SUB_E32>(cus, b, iters, out, st, h);
        run<T_RSQ>(cus, b, iters, out, st, h);
        run<T_RSQ_THEN_INDEP>(cus, b, iters, out, st, h);
        run<T_FMA_SGPR>(cus, b, iters, out, st, h);
        run<T_PKFMA>(cus, b, iters, out, st, h);
        run<T_INTERACT_SEQ>(cus, b, iters / 4, out, st, h);
        run<T_INTERACT_ILV>(cus, b, iters / 4, out, st, h);
        printf("\n");
    }
    return 0;
}
