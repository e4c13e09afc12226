#!/usr/bin/env python3
"""Generates tools/ubench_sched2.hip: steady-state cost of batched-rsq schedules of the pair interaction with a
tight register allocation (so that the occupancy is the real kernel's) and an idle gap after the rsq batch.

Finding that motivates it (tools/ubench_spec.hip): fp32 instructions a wave issues shortly after its own
transcendental instructions run at about half rate; if the wave idles for a while after a BATCH of v_rsq_f32
(s_nop / s_sleep) and other waves use the SIMD meanwhile, the cost approaches the additive floor.

Registers (bank = index mod 4; src0/src1 of an instruction never share a bank):
  v0..v(4U-1)   column bodies {x,y,z,m} (banks 0,1,2,3)         eps2 = v(4U) ... rounded up to bank 0
  rows: X,Y,Z in banks 1,2,3 (one spare bank-0 register per row holds nothing: packed as 4 per row)
  acc : 3 per row, any bank
  temp set t: R = base (bank 0), D0,D1,D2 = base+1..base+3 (banks 1,2,3); two shared Q temporaries in bank 1
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))


class Alloc:
    def __init__(self, R, U):
        self.R, self.U = R, U
        n = 0
        self.pj = [[f"v{4*u+c}" for c in range(4)] for u in range(U)]
        n = 4 * U
        self.eps = f"v{n}"          # bank 0, only ever src2
        n += 4
        self.row = []
        for k in range(R):          # X,Y,Z at banks 1,2,3 of a group of 4; the bank-0 slot of the group hosts an accumulator
            self.row.append((f"v{n+1}", f"v{n+2}", f"v{n+3}"))
            n += 4
        spare = [f"v{4*U+4+4*k}" for k in range(R)]   # the bank-0 slots
        self.acc = []
        extra = []
        for k in range(R):
            extra += [f"v{n}", f"v{n+1}"]
            self.acc.append((spare[k], f"v{n}", f"v{n+1}"))
            n += 2
        n = (n + 3) // 4 * 4
        self.tmp = []
        for t in range(R * U):
            self.tmp.append((f"v{n}", f"v{n+1}", f"v{n+2}", f"v{n+3}"))   # R, D0, D1, D2
            n += 4
        self.q = [f"v{n+1}", f"v{n+5}"]   # bank 1
        n += 8
        self.nreg = n


def ops(al, k, u, t, qi):
    px, py, pz, pm = al.pj[u]
    X, Y, Z = al.row[k]
    AX, AY, AZ = al.acc[k]
    Rr, D0, D1, D2 = al.tmp[t]
    Q = al.q[qi]
    pre = [f"v_sub_f32_e32 {D0}, {px}, {X}", f"v_sub_f32_e32 {D1}, {py}, {Y}", f"v_sub_f32_e32 {D2}, {pz}, {Z}",
           f"v_fma_f32 {Rr}, {D0}, {D0}, {al.eps}", f"v_fmac_f32_e32 {Rr}, {D1}, {D1}", f"v_fmac_f32_e32 {Rr}, {D2}, {D2}"]
    rsq = [f"v_rsq_f32_e32 {Rr}, {Rr}"]
    post = [f"v_mul_f32_e32 {Q}, {Rr}, {Rr}", f"v_mul_f32_e32 {Rr}, {pm}, {Rr}", f"v_mul_f32_e32 {Rr}, {Rr}, {Q}",
            f"v_fmac_f32_e32 {AX}, {D0}, {Rr}", f"v_fmac_f32_e32 {AY}, {D1}, {Rr}", f"v_fmac_f32_e32 {AZ}, {D2}, {Rr}"]
    return pre, rsq, post


def batched(al, wait, order="chain"):
    """[pre of all] [rsq of all] [wait] [post chain by chain]"""
    pairs = [(k, u) for u in range(al.U) for k in range(al.R)]
    body = []
    for t, (k, u) in enumerate(pairs):
        body += ops(al, k, u, t, t & 1)[0]
    for t, (k, u) in enumerate(pairs):
        body += ops(al, k, u, t, t & 1)[1]
    body += wait
    for t, (k, u) in enumerate(pairs):
        body += ops(al, k, u, t, t & 1)[2]
    return body


def pipelined(al, wait):
    """software pipeline: [rsq batch of group n] [post of group n-1 -- covers the gap] [pre of group n+1] ...
    expressed for the steady state as: post(prev set) ; pre(next set) between rsq batches; needs 2x temp sets."""
    return None


def seq(al):
    body = []
    for u in range(al.U):
        for k in range(al.R):
            pre, rsq, post = ops(al, k, u, 0, 0)
            body += pre + rsq + ["s_nop 0"] + post
    return body


PATTERNS = []


def add(pid, desc, al, body, rep):
    PATTERNS.append((pid, desc, al, body, rep, al.R * al.U * rep))


WAITS = {"none": [], "nop16": ["s_nop 15"], "nop32": ["s_nop 15"] * 2, "nop64": ["s_nop 15"] * 4, "nop128": ["s_nop 15"] * 8,
         "sleep1": ["s_sleep 1"], "sleep2": ["s_sleep 2"], "sleep4": ["s_sleep 4"]}

def nops(total):
    out = []
    while total > 0:
        k = min(total, 16)
        out.append(f"s_nop {k-1}")
        total -= k
    return out


a41 = Alloc(4, 1)
add("seq4", "row after row (reference point)", a41, seq(a41), 4)
for t in (8, 12, 16, 20, 24, 28, 32, 36, 40, 48):
    add(f"b4_n{t}", f"4 rsq batched (4 rows), {t} wait states", a41, batched(a41, nops(t)), 4)
a21 = Alloc(2, 1)
for t in (4, 8, 12, 16, 20, 24, 28):
    add(f"b2_n{t}", f"2 rsq batched (2 rows), {t} wait states", a21, batched(a21, nops(t)), 8)
a31 = Alloc(3, 1)
for t in (8, 16, 24, 32):
    add(f"b3_n{t}", f"3 rsq batched (3 rows), {t} wait states", a31, batched(a31, nops(t)), 4)
a61 = Alloc(6, 1)
for t in (16, 24, 32, 40):
    add(f"b6_n{t}", f"6 rsq batched (6 rows), {t} wait states", a61, batched(a61, nops(t)), 2)
a42 = Alloc(4, 2)
for t in (20, 24, 28, 32, 36, 40, 48):
    add(f"b8_n{t}", f"8 rsq batched (4 rows x 2 cols), {t} wait states", a42, batched(a42, nops(t)), 2)


TEMPLATE = r'''// GENERATED by tools/gen_sched2.py -- do not edit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
%(kernels)s
template <typename Kern>
static void run(const char *name, Kern kern, int cus, double units_per_body, int nreg, unsigned long long *dev, std::vector<unsigned long long> &h)
{
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0);
    if (occ > 8) occ = 8;
    const int blocks = cus * occ, nw = blocks * 4;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dev, 2000u);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dev, 30000u);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), dev, sizeof(unsigned long long) * nw * 4, hipMemcpyDeviceToHost);
    double bodies = 0, cyc = 0;
    unsigned long long first = ~0ull, last = 0;
    for (int i = 0; i < nw; ++i) {
        bodies += (double)h[4 * i]; cyc += (double)h[4 * i + 1];
        first = first < h[4 * i + 3] ? first : h[4 * i + 3]; last = last > h[4 * i + 3] ? last : h[4 * i + 3];
    }
    printf("%%-58s vgpr %%3d waves/SIMD %%d  %%7.2f SIMD cycles per interaction%%s\n", name, nreg, occ,
           cus * 4.0 * (cyc / nw) / (bodies * units_per_body), (last - first) > 500 ? " (!)" : "");
}
int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    unsigned long long *dev;
    (void)hipMalloc((void **)&dev, sizeof(unsigned long long) * cus * 8 * 4 * 4);
    std::vector<unsigned long long> h((size_t)cus * 8 * 4 * 4);
%(runs)s
    return 0;
}
'''

KERNEL = r'''__global__ __launch_bounds__(256) void rate_%(pid)s(unsigned long long *out, unsigned ticks)
{
    asm volatile(%(init)s ::: %(clob)s);
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long bodies = 0;
    while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) {
        asm volatile(%(body)s ::: %(clob)s);
        ++bodies;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        out[4 * w + 0] = bodies; out[4 * w + 1] = t1 - t0; out[4 * w + 2] = 0; out[4 * w + 3] = r0;
    }
}
'''


def cstr(lines):
    return "\n        ".join('"' + l + '\\n"' for l in lines)


def main():
    kernels, runs = [], []
    for pid, desc, al, body, rep, units in PATTERNS:
        clob = ", ".join(f'"v{i}"' for i in range(al.nreg))
        init = cstr([f"v_mov_b32 v{i}, {'0x3f7fbe77' if i % 2 == 0 else '0x3f810000'}" for i in range(al.nreg)])
        kernels.append(KERNEL % dict(pid=pid, init=init, clob=clob, body=cstr(body * rep)))
        runs.append(f'    run("{desc}", rate_{pid}, cus, {units}, {al.nreg}, dev, h);')
    out = os.path.join(HERE, "ubench_sched2.hip")
    open(out, "w").write(TEMPLATE % dict(kernels="\n".join(kernels), runs="\n".join(runs)))
    print("wrote", out, len(PATTERNS), "patterns")


if __name__ == "__main__":
    main()
