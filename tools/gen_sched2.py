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
for t in (20, 24, 28):
    add(f"b4_n{t}", f"4 rsq batched (4 rows), {t} wait states, 8 waves", a41, batched(a41, nops(t)), 4)
for nreg, tag in ((72, "7 waves"), (80, "6 waves"), (96, "5 waves"), (128, "4 waves")):
    al = Alloc(4, 1)
    al.nreg = nreg
    for t in (24,):
        add(f"b4_n{t}_r{nreg}", f"4 rsq batched, {t} wait states, {tag}", al, batched(al, nops(t)), 4)


# ---- the pair-once step (nbody_symmetric.hip): 16 fp32 + rsq per pair, 4 rows per lane, per-step extras ------------
class SymAlloc:
    """The register map of nbody_symmetric.hip's SY_STEP (banks: see the comment there)."""
    nreg = 72
    R, U = 4, 1


def sym_step(gap, rotate=True, addr=True, post_order="row", halves=False, prio=None):
    """rotate: True/"wave_rol" = 3 x v_mov_b32_dpp wave_rol:1; "row_ror" = 3 x v_mov_b32_dpp row_ror:1 (cost probe);
    "bpermute" = 3 x ds_bpermute_b32 through the LDS crossbar, waited for just before the POST phase."""
    px, py, pz, pm = "v2", "v3", "v4", "v5"
    pre, rsq, post = [], [], []
    for k in range(4):
        X, Y, Z, M = (f"v{12+4*k+c}" for c in range(4))
        D0, D1, D2, Rr = (f"v{28+4*k+c}" for c in range(4))
        AZ, AX, AY = (f"v{52+4*k+c}" for c in range(3))
        Q, T, SC = ("v44", "v46", "v47") if k % 2 == 0 else ("v48", "v50", "v51")
        pre.append([f"v_sub_f32_e32 {D0}, {px}, {X}", f"v_sub_f32_e32 {D1}, {py}, {Y}", f"v_sub_f32_e32 {D2}, {pz}, {Z}",
                    f"v_fma_f32 {Rr}, {D0}, {D0}, v11", f"v_fmac_f32_e32 {Rr}, {D1}, {D1}", f"v_fmac_f32_e32 {Rr}, {D2}, {D2}"])
        rsq.append([f"v_rsq_f32_e32 {Rr}, {Rr}"])
        post.append([f"v_mul_f32_e32 {Q}, {Rr}, {Rr}", f"v_mul_f32_e32 {T}, {Rr}, {Q}", f"v_mul_f32_e32 {SC}, {M}, {T}",
                     f"v_mul_f32_e32 {Rr}, {pm}, {T}", f"v_fmac_f32_e32 {AX}, {D0}, {Rr}", f"v_fmac_f32_e32 {AY}, {D1}, {Rr}",
                     f"v_fmac_f32_e32 {AZ}, {D2}, {Rr}", f"v_fmac_f32_e32 v45, {D0}, {SC}", f"v_fmac_f32_e32 v68, {D1}, {SC}",
                     f"v_fmac_f32_e32 v49, {D2}, {SC}"])
    body = []
    if addr:
        body += ["v_add_u32_e32 v1, 16, v1", "v_and_or_b32 v0, v1, v55, v10"]
    if rotate == "bpermute":
        body += ["s_waitcnt lgkmcnt(3)"]
    groups = [(0, 1), (2, 3)] if halves else [(0, 1, 2, 3)]
    for grp in groups:
        for k in grp:
            body += pre[k]
        for k in grp:
            body += rsq[k]
        body += nops(gap)
        if rotate == "bpermute":
            body += ["s_waitcnt lgkmcnt(0)"]
        if prio:
            body += [f"s_setprio {prio[1]}"]
        if post_order == "row":
            for k in grp:
                body += post[k]
        else:  # stage by stage across the rows
            for i in range(10):
                for k in grp:
                    body.append(post[k][i])
    if prio:
        body += [f"s_setprio {prio[0]}"]
    if rotate == "bpermute":
        body += [f"ds_bpermute_b32 {r}, v59, {r}" for r in ("v45", "v68", "v49")]
    elif rotate:
        ctrl = "row_ror:1" if rotate == "row_ror" else "wave_rol:1"
        body += ["s_nop 1"] + [f"v_mov_b32_dpp {r}, {r} {ctrl} row_mask:0xf bank_mask:0xf" for r in ("v45", "v68", "v49")]
    return body


def sym_two_steps(gap, gap2=0):
    """Two steps' eight pairs in one v_rsq_f32 batch (arithmetic only): second column in v6..v9, second temp set v72..v87."""
    pre, rsq, post = [], [], []
    for u, (px, py, pz, pm) in enumerate((("v2", "v3", "v4", "v5"), ("v6", "v7", "v8", "v9"))):
        for k in range(4):
            X, Y, Z, M = (f"v{12+4*k+c}" for c in range(4))
            base = 28 + 4 * k if u == 0 else 72 + 4 * k
            D0, D1, D2, Rr = (f"v{base+c}" for c in range(4))
            AZ, AX, AY = (f"v{52+4*k+c}" for c in range(3))
            Q, T, SC = ("v44", "v46", "v47") if k % 2 == 0 else ("v48", "v50", "v51")
            pre += [f"v_sub_f32_e32 {D0}, {px}, {X}", f"v_sub_f32_e32 {D1}, {py}, {Y}", f"v_sub_f32_e32 {D2}, {pz}, {Z}",
                    f"v_fma_f32 {Rr}, {D0}, {D0}, v11", f"v_fmac_f32_e32 {Rr}, {D1}, {D1}", f"v_fmac_f32_e32 {Rr}, {D2}, {D2}"]
            rsq += [f"v_rsq_f32_e32 {Rr}, {Rr}"]
            post += [f"v_mul_f32_e32 {Q}, {Rr}, {Rr}", f"v_mul_f32_e32 {T}, {Rr}, {Q}", f"v_mul_f32_e32 {SC}, {M}, {T}",
                     f"v_mul_f32_e32 {Rr}, {pm}, {T}", f"v_fmac_f32_e32 {AX}, {D0}, {Rr}", f"v_fmac_f32_e32 {AY}, {D1}, {Rr}",
                     f"v_fmac_f32_e32 {AZ}, {D2}, {Rr}", f"v_fmac_f32_e32 v45, {D0}, {SC}", f"v_fmac_f32_e32 v68, {D1}, {SC}",
                     f"v_fmac_f32_e32 v49, {D2}, {SC}"]
            if u == 0 and k == 3 and gap2:
                post += nops(gap2)
    return pre + rsq + nops(gap) + post


def sym_step8(gap, rotate="bpermute", addr=True):
    """8 rows per lane: two 4-row sub-batches share the column body, the address update and the three permutes.
    Rows 4..7 at v72..v87, their sums at v88..v103 (same bank pattern as rows 0..3)."""
    px, py, pz, pm = "v2", "v3", "v4", "v5"
    body = []
    if addr:
        body += ["v_add_u32_e32 v1, 16, v1", "v_and_or_b32 v0, v1, v55, v10"]
    if rotate == "bpermute":
        body += ["s_waitcnt lgkmcnt(3)"]
    for half in range(2):
        pre, rsq, post = [], [], []
        for k in range(4):
            rb = 12 + 4 * k if half == 0 else 72 + 4 * k
            sb = 52 + 4 * k if half == 0 else 88 + 4 * k
            X, Y, Z, M = (f"v{rb+c}" for c in range(4))
            D0, D1, D2, Rr = (f"v{28+4*k+c}" for c in range(4))
            AZ, AX, AY = (f"v{sb+c}" for c in range(3))
            Q, T, SC = ("v44", "v46", "v47") if k % 2 == 0 else ("v48", "v50", "v51")
            pre += [f"v_sub_f32_e32 {D0}, {px}, {X}", f"v_sub_f32_e32 {D1}, {py}, {Y}", f"v_sub_f32_e32 {D2}, {pz}, {Z}",
                    f"v_fma_f32 {Rr}, {D0}, {D0}, v11", f"v_fmac_f32_e32 {Rr}, {D1}, {D1}", f"v_fmac_f32_e32 {Rr}, {D2}, {D2}"]
            rsq += [f"v_rsq_f32_e32 {Rr}, {Rr}"]
            post += [f"v_mul_f32_e32 {Q}, {Rr}, {Rr}", f"v_mul_f32_e32 {T}, {Rr}, {Q}", f"v_mul_f32_e32 {SC}, {M}, {T}",
                     f"v_mul_f32_e32 {Rr}, {pm}, {T}", f"v_fmac_f32_e32 {AX}, {D0}, {Rr}", f"v_fmac_f32_e32 {AY}, {D1}, {Rr}",
                     f"v_fmac_f32_e32 {AZ}, {D2}, {Rr}", f"v_fmac_f32_e32 v45, {D0}, {SC}", f"v_fmac_f32_e32 v68, {D1}, {SC}",
                     f"v_fmac_f32_e32 v49, {D2}, {SC}"]
        body += pre + rsq + nops(gap)
        if rotate == "bpermute" and half == 0:
            body += ["s_waitcnt lgkmcnt(0)"]
        body += post
    if rotate == "bpermute":
        body += [f"ds_bpermute_b32 {r}, v59, {r}" for r in ("v45", "v68", "v49")]
    return body


class SymAlloc8(SymAlloc):
    nreg = 112
    R, U = 8, 1


for nreg in (104, 112, 120):
    for gap in (12, 24):
        a8 = SymAlloc8(); a8.nreg = nreg
        add(f"sym8_g{gap}_r{nreg}", f"pair-once, 8 rows per lane (2 sub-batches) + 3 ds_bpermute + 2 address ops, gap {gap}, {nreg} regs",
            a8, sym_step8(gap), 1)
a8 = SymAlloc8(); a8.nreg = 112
add("sym8_arith", "pair-once, 8 rows per lane, arithmetic only, gap 24, 112 regs", a8, sym_step8(24, rotate=False, addr=False), 1)


class SymAlloc2(SymAlloc):
    nreg = 112
    R, U = 4, 2


for gap in (0, 12, 24, 40):
    add(f"sym2_g{gap}", f"pair-once, TWO steps per rsq batch (8 pairs), arithmetic only, gap {gap}, 112 regs", SymAlloc2(),
        sym_two_steps(gap), 1)
al2 = SymAlloc2(); al2.nreg = 96
add("sym2_g24_r96", "pair-once, two steps per rsq batch, gap 24, 96 regs (hypothetical)", al2, sym_two_steps(24), 1)

for nreg, tag in ((96, "5 waves"), (128, "4 waves"), (64, "8 waves")):
    for gap in (0, 12, 24):
        al = SymAlloc()
        al.nreg = max(nreg, 72)
        if nreg == 64 and gap != 12:
            continue
        add(f"sym_g{gap}_r{nreg}", f"pair-once step, arithmetic only, gap {gap}, {tag} (floor 40.8/pair)", al,
            sym_step(gap, rotate=False, addr=False), 2)
al = SymAlloc(); al.nreg = 96
add("sym_full", "pair-once step + 3 DPP + 2 address ops, gap 12, 5 waves", al, sym_step(12), 2)
add("sym_dpp", "pair-once step + 3 DPP, gap 12, 5 waves", al, sym_step(12, addr=False), 2)
add("sym_rowror", "pair-once step + 3 DPP row_ror:1 (cost probe), gap 12", al, sym_step(12, rotate="row_ror", addr=False), 2)
add("sym_bperm", "pair-once step + 3 ds_bpermute_b32, gap 12", al, sym_step(12, rotate="bpermute", addr=False), 2)
add("sym_bperm_full", "pair-once step + 3 ds_bpermute_b32 + 2 address ops, gap 12", al, sym_step(12, rotate="bpermute"), 2)
add("sym_bperm_full24", "pair-once step + 3 ds_bpermute_b32 + 2 address ops, gap 24", al, sym_step(24, rotate="bpermute"), 2)
al80 = SymAlloc(); al80.nreg = 80
add("sym_bperm_full_r80", "pair-once step + 3 ds_bpermute_b32 + 2 address ops, gap 12, 80 regs", al80, sym_step(12, rotate="bpermute"), 2)
add("sym_full_r80", "pair-once step + 3 DPP wave_rol + 2 address ops, gap 12, 80 regs", al80, sym_step(12), 2)
for nreg_p in (80, 96):
    alp = SymAlloc(); alp.nreg = nreg_p
    for gap_p in (0, 6, 12):
        add(f"sym_prio20_g{gap_p}_r{nreg_p}", f"pair-once full step, s_setprio 2 (PRE,rsq) / 0 (POST), gap {gap_p}, {nreg_p} regs", alp,
            sym_step(gap_p, rotate="bpermute", prio=(2, 0)), 2)
    add(f"sym_prio02_r{nreg_p}", f"pair-once full step, s_setprio 0 (PRE,rsq) / 2 (POST), gap 12, {nreg_p} regs", alp,
        sym_step(12, rotate="bpermute", prio=(0, 2)), 2)
    add(f"sym_arith_prio_r{nreg_p}", f"pair-once arithmetic only, s_setprio 2/0, gap 0, {nreg_p} regs", alp,
        sym_step(0, rotate=False, addr=False, prio=(2, 0)), 2)
add("sym_stage", "pair-once step, POST stage by stage, gap 12, 5 waves", al, sym_step(12, rotate=False, addr=False, post_order="stage"), 2)
add("sym_halves", "pair-once step, two 2-row half batches, gap 12 each, 5 waves", al, sym_step(12, rotate=False, addr=False, halves=True), 2)
add("sym_halves0", "pair-once step, two 2-row half batches, no gap, 5 waves", al, sym_step(0, rotate=False, addr=False, halves=True), 2)


TEMPLATE = r'''// GENERATED by tools/gen_sched2.py -- do not edit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
%(kernels)s
template <typename Kern>
static void run(const char *name, Kern kern, int cus, double units_per_body, int nreg, unsigned long long *dev, std::vector<unsigned long long> &h)
{
    int occ = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0);
    if (occ > 8) occ = 8;
    const int blocks = cus * occ, nw = blocks * 4;
    // UBENCH_TICKS (100 MHz real-time ticks per measurement, default 30000 = 0.3 ms): long runs (3000000 = 30 ms) let the
    // power management settle, and the clock column then says what each schedule really sustains
    static const unsigned ticks = getenv("UBENCH_TICKS") ? (unsigned)atol(getenv("UBENCH_TICKS")) : 30000u;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dev, 2000u);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, dev, ticks);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), dev, sizeof(unsigned long long) * nw * 4, hipMemcpyDeviceToHost);
    double bodies = 0, cyc = 0;
    unsigned long long first = ~0ull, last = 0;
    for (int i = 0; i < nw; ++i) {
        bodies += (double)h[4 * i]; cyc += (double)h[4 * i + 1];
        first = first < h[4 * i + 3] ? first : h[4 * i + 3]; last = last > h[4 * i + 3] ? last : h[4 * i + 3];
    }
    const double cycles_per_unit = cus * 4.0 * (cyc / nw) / (bodies * units_per_body);
    const double clock_ghz = (cyc / nw) / ((double)ticks * 10.0);  // shader cycles per 10 ns tick
    printf("%%-58s vgpr %%3d waves/SIMD %%d  %%7.2f SIMD cycles per interaction  %%5.3f GHz  %%6.2f ns%%s\n", name, nreg, occ,
           cycles_per_unit, clock_ghz, cycles_per_unit / clock_ghz, (last - first) > 500 ? " (!)" : "");
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
