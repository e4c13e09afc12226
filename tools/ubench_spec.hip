// ubench_spec.hip -- do transcendental instructions of ONE wave slow the fp32 instructions of ANOTHER wave on the
// same SIMD?  Even workgroups run a pure v_rsq_f32 stream, odd workgroups a pure v_fmac_f32 stream (or both mixed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CLOB "v0","v1","v2","v3","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
#define REP8(b) b b b b b b b b
#define P_FMAC "v_fmac_f32_e32 v16, v0, v1\n v_fmac_f32_e32 v17, v0, v1\n v_fmac_f32_e32 v18, v0, v1\n v_fmac_f32_e32 v19, v0, v1\n" \
               "v_fmac_f32_e32 v20, v0, v1\n v_fmac_f32_e32 v21, v0, v1\n v_fmac_f32_e32 v22, v0, v1\n v_fmac_f32_e32 v23, v0, v1\n"
#define P_RSQ  "v_rsq_f32_e32 v24, v24\n v_rsq_f32_e32 v25, v25\n v_rsq_f32_e32 v26, v26\n v_rsq_f32_e32 v27, v27\n" \
               "v_rsq_f32_e32 v28, v28\n v_rsq_f32_e32 v29, v29\n v_rsq_f32_e32 v30, v30\n v_rsq_f32_e32 v31, v31\n"
// mode 0: every block fmac; 1: every block rsq; 2: even blocks rsq, odd blocks fmac; 3: every block alternates
//         one rsq body / eight fmac bodies (mixed within a wave at coarse granularity)
__global__ __launch_bounds__(256) void k(unsigned long long *out, unsigned ticks, int mode)
{
    asm volatile("v_mov_b32 v0, 0x3f7fbe77\n v_mov_b32 v1, 0x3a83126f\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n"
                 "v_mov_b32 v19, 1.0\n v_mov_b32 v20, 1.0\n v_mov_b32 v21, 1.0\n v_mov_b32 v22, 1.0\n v_mov_b32 v23, 1.0\n v_mov_b32 v24, 1.0\n"
                 "v_mov_b32 v25, 1.0\n v_mov_b32 v26, 1.0\n v_mov_b32 v27, 1.0\n v_mov_b32 v28, 1.0\n v_mov_b32 v29, 1.0\n v_mov_b32 v30, 1.0\n v_mov_b32 v31, 1.0\n" ::: CLOB);
    const bool rsq_role = mode == 1 || (mode == 2 && (blockIdx.x & 1) == 0);
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    unsigned long long bodies = 0;
    while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) {
        if (mode == 3) {
            asm volatile(P_RSQ REP8(P_FMAC) ::: CLOB);
        } else if (mode == 4) {
            asm volatile(P_RSQ "s_sleep 1\n" REP8(P_FMAC) ::: CLOB);
        } else if (mode == 5) {
            asm volatile(P_RSQ "s_sleep 2\n" REP8(P_FMAC) ::: CLOB);
        } else if (mode == 6) {
            asm volatile(P_RSQ "s_sleep 4\n" REP8(P_FMAC) ::: CLOB);
        } else if (mode == 7) {
            asm volatile(P_RSQ "s_nop 15\n s_nop 15\n" REP8(P_FMAC) ::: CLOB);
        } else if (mode == 8) {  // consume every rsq result right after the batch
            asm volatile(P_RSQ "v_add_f32 v2, v24, v25\n v_add_f32 v2, v26, v27\n v_add_f32 v2, v28, v29\n v_add_f32 v2, v30, v31\n" REP8(P_FMAC) ::: CLOB, "v2");
        } else if (mode == 9) {  // one rsq spread between fmac groups
            asm volatile("v_rsq_f32_e32 v24, v24\n" P_FMAC "v_rsq_f32_e32 v25, v25\n" P_FMAC "v_rsq_f32_e32 v26, v26\n" P_FMAC "v_rsq_f32_e32 v27, v27\n" P_FMAC
                         "v_rsq_f32_e32 v28, v28\n" P_FMAC "v_rsq_f32_e32 v29, v29\n" P_FMAC "v_rsq_f32_e32 v30, v30\n" P_FMAC "v_rsq_f32_e32 v31, v31\n" P_FMAC ::: CLOB);
        } else if (rsq_role) {
            asm volatile(REP8(P_RSQ) ::: CLOB);
        } else {
            asm volatile(REP8(P_FMAC) ::: CLOB);
        }
        ++bodies;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[2 * w] = bodies;
        out[2 * w + 1] = t1 - t0;
    }
}
int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    unsigned long long *dev;
    (void)hipMalloc((void **)&dev, sizeof(unsigned long long) * cus * 8 * 4 * 2);
    std::vector<unsigned long long> h((size_t)cus * 8 * 4 * 2);
    for (int bpc : {4, 8})
        for (int mode : {0, 3, 4, 5, 6, 7, 8, 9}) {
            const int blocks = cus * bpc, nw = blocks * 4;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dev, 2000u, mode);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, dev, 30000u, mode);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), dev, sizeof(unsigned long long) * nw * 2, hipMemcpyDeviceToHost);
            double b[2] = {0, 0}, c[2] = {0, 0};
            int cnt[2] = {0, 0};
            for (int i = 0; i < nw; ++i) {
                const int role = (mode == 2) ? (((i / 4) & 1) == 0 ? 1 : 0) : (mode == 1 ? 1 : 0);
                b[role] += (double)h[2 * i]; c[role] += (double)h[2 * i + 1]; cnt[role]++;
            }
            const double simds = cus * 4.0;
            printf("blk/CU=%d mode=%d:", bpc, mode);
            for (int role = 0; role < 2; ++role)
                if (cnt[role]) {
                    const double instr = b[role] * (mode >= 3 ? 72.0 : 64.0);
                    // instructions of this role per SIMD cycle (all SIMDs, mean elapsed cycles)
                    printf("  %s: %.3f instr/SIMD-cycle (%.2f cyc/instr)", role ? "rsq " : (mode >= 3 ? "8 rsq + 64 fmac, cycles per body/72" : "fmac"),
                           instr / (simds * c[role] / cnt[role]), simds * c[role] / cnt[role] / instr);
                }
            printf("\n");
        }
    return 0;
}
