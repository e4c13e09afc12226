// ubench_valu.hip -- instruction-throughput microbenchmark for the VALU mix of the pair kernel on gfx950.
// Measures, per primitive and per occupancy (waves/SIMD), the SIMD cycles one wave-instruction costs:
//   cycles/instr = elapsed shader cycles (s_memtime) * waves_per_simd / instructions per wave.
// Build:  hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
// Run on the GPU box; prints one line per (primitive, occupancy).  Development tool, not product code.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

enum Op { FMA, PK_FMA, MUL, PK_MUL, SUB, PK_ADD, RSQ, FMA_SGPR, MIX12_1, MIX_PK6_2, LDS_B128, MIX_LDS, NOPS };
static const char *kNames[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_sub_f32", "v_pk_add_f32",
                               "v_rsq_f32", "v_fma_f32(sgpr)", "mix 12 valu + 1 rsq", "mix 6 pk + 2 rsq",
                               "ds_read_b128 bcast", "mix 52 valu + 4 rsq + 1 ds_read", ""};
// wave-instructions issued per loop iteration, and interactions they stand for
static const int kInstr[] = {8, 8, 8, 8, 8, 8, 8, 8, 13, 8, 8, 57, 0};
constexpr int kRep = 8;

template <int OP>
__global__ __launch_bounds__(256) void bench(float *out, unsigned long long *cycles, int iters, float seed, float sarg)
{
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(seed, seed * 2, seed * 3, seed * 4);
    __syncthreads();
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = seed + 0.001f * (threadIdx.x + i);
        p[i] = f2{a[i], a[i] * 0.5f};
    }
    const float b = 0.999f, c = 0.001f;
    const f2 pb = f2{0.999f, 0.998f}, pc = f2{0.001f, 0.002f};
    float4 q = make_float4(0, 0, 0, 0);
    unsigned laddr = 0;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < kRep; ++rep) {  // straight-line repeats: loop overhead < 1 %
        if (OP == FMA) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        } else if (OP == PK_FMA) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
        } else if (OP == MUL) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        } else if (OP == PK_MUL) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
        } else if (OP == SUB) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (OP == PK_ADD) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
        } else if (OP == RSQ) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
        } else if (OP == FMA_SGPR) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sarg), "v"(c));
        } else if (OP == MIX12_1) {
            // the instruction multiset of one interaction, on independent registers
            asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
            asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[1]) : "v"(c));
            asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[2]) : "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4]) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[5]) : "v"(b), "v"(c));
            asm volatile("v_rsq_f32 %0, %0" : "+v"(a[6]));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[7]) : "v"(b));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[1]) : "v"(b));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(b), "v"(c));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4]) : "v"(b), "v"(c));
        } else if (OP == MIX_PK6_2) {
            // two interactions: 3 pk_add + 3 pk_fma ... here the packed part only (6 pk) + 2 rsq
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[0]) : "v"(pc));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[1]) : "v"(pc));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[2]) : "v"(pc));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[3]) : "v"(pb), "v"(pc));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[4]) : "v"(pb), "v"(pc));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[5]) : "v"(pb), "v"(pc));
            asm volatile("v_rsq_f32 %0, %0" : "+v"(a[6]));
            asm volatile("v_rsq_f32 %0, %0" : "+v"(a[7]));
        } else if (OP == LDS_B128) {
            float4 r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(laddr), "i"(i * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(q.x) : "v"(r[i].x));
        } else if (OP == MIX_LDS) {
            // 4 rows x (12 valu + 1 rsq) per broadcast LDS read: the RPL=4 inner loop's issue pattern
            float4 r;
            asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(laddr));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[1]) : "v"(c));
                asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[2]) : "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4]) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[5]) : "v"(b), "v"(c));
                asm volatile("v_rsq_f32 %0, %0" : "+v"(a[6]));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[7]) : "v"(b));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[1]) : "v"(b));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(b), "v"(c));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4]) : "v"(b), "v"(c));
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n v_add_f32 %0, %0, %1" : "+v"(q.x) : "v"(r.x));
        }
      }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = q.x;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0)
    {
        cycles[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6)] = t1 - t0;
        cycles[2 * ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) + 1] = r1 - r0;  // 100 MHz ticks
    }
}

template <int OP>
static void run(int cus, int waves_per_simd, int iters, float *out, unsigned long long *cyc, std::vector<unsigned long long> &h)
{
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one wave per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters / 10, 1.5f, 0.999f);  // warm-up
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.5f, 0.999f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int nw = blocks * 4;
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * nw * 2, hipMemcpyDeviceToHost);
    std::vector<double> cy(nw), rt(nw);
    for (int i = 0; i < nw; ++i) { cy[i] = (double)h[2 * i]; rt[i] = (double)h[2 * i + 1]; }
    std::sort(cy.begin(), cy.end());
    std::sort(rt.begin(), rt.end());
    const double med = cy[nw / 2], med_ns = rt[nw / 2] * 10.0;  // s_memrealtime: 100 MHz
    const double instr_per_wave = (double)iters * kRep * kInstr[OP];
    const double ghz = med / med_ns;
    printf("%-34s w/SIMD=%d  wave: %6.2f cyc/instr  SIMD: %6.3f cyc/instr %6.3f ns/instr  wave-time %7.3f ms kernel %7.3f ms  shader clk %.2f GHz\n",
           kNames[OP], waves_per_simd, med / instr_per_wave, med / (instr_per_wave * waves_per_simd),
           med_ns / (instr_per_wave * waves_per_simd), med_ns * 1e-6, ms, ghz);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) {
        printf("no device\n");
        return 1;
    }
    const int cus = prop.multiProcessorCount;
    printf("device %s %s CUs=%d clock=%d MHz\n", prop.name, prop.gcnArchName, cus, prop.clockRate / 1000);
    const int max_blocks = cus * 8;
    float *out;
    unsigned long long *cyc;
    hipMalloc((void **)&out, sizeof(float) * max_blocks * 256);
    hipMalloc((void **)&cyc, sizeof(unsigned long long) * max_blocks * 8);
    std::vector<unsigned long long> h(max_blocks * 8);
    const int iters = 4000;
    for (int w : {1, 2, 4, 8}) {
        run<FMA>(cus, w, iters, out, cyc, h);
        run<PK_FMA>(cus, w, iters, out, cyc, h);
        run<MUL>(cus, w, iters, out, cyc, h);
        run<PK_MUL>(cus, w, iters, out, cyc, h);
        run<SUB>(cus, w, iters, out, cyc, h);
        run<PK_ADD>(cus, w, iters, out, cyc, h);
        run<RSQ>(cus, w, iters, out, cyc, h);
        run<FMA_SGPR>(cus, w, iters, out, cyc, h);
        run<MIX12_1>(cus, w, iters, out, cyc, h);
        run<MIX_PK6_2>(cus, w, iters, out, cyc, h);
        run<LDS_B128>(cus, w, iters, out, cyc, h);
        run<MIX_LDS>(cus, w, iters / 4, out, cyc, h);
        printf("\n");
    }
    hipFree(out);
    hipFree(cyc);
    return 0;
}
