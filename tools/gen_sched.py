#!/usr/bin/env python3
"""Generates tools/ubench_sched.hip: steady-state cost of candidate instruction schedules for the
pair interaction on gfx950 (the same harness as ubench_rate.hip).  Development tool.

An interaction (row k, column u) is 13 VALU instructions:
    s0 s1 s2      d = p_j - x_i                     (v_sub_f32 x3)
    f0 f1 f2      r2 = fma(dx,dx,eps2); r2 += dy*dy; r2 += dz*dz
    q             inv = rsq(r2)                     (transcendental)
    m0 m1 m2      q2 = inv*inv; s = m*inv; s = s*q2
    a0 a1 a2      acc += d*s                        (v_fmac_f32 x3)
A schedule is an ordered list of (op, k, u, tempset).  Registers are allocated so that src0 and src1 of an
instruction never share a VGPR bank (bank = index mod 4):
    p_j(u) = v[4u..4u+3]  eps2 = v8   row k coords = v(13+4k), v(14+4k), v(15+4k)   acc k = v(28+3k)..
    temp set t (base 40+8t): R = base (bank 0), Q = base+1 (bank 1), D0..D2 = base+5..base+7 (banks 1,2,3)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def regs(k, u, t):
    base = 48 + 6 * t  # R, Q, D0..D2 packed in 6 registers
    return dict(px=f"v{4*u}", py=f"v{4*u+1}", pz=f"v{4*u+2}", pm=f"v{4*u+3}", eps="v176",
                X=f"v{17+4*k}", Y=f"v{18+4*k}", Z=f"v{19+4*k}",
                AX=f"v{148+3*k}", AY=f"v{149+3*k}", AZ=f"v{150+3*k}",
                R=f"v{base}", Q=f"v{base+1}", D0=f"v{base+2}", D1=f"v{base+3}", D2=f"v{base+4}")


OPS = {
    "s0": "v_sub_f32_e32 {D0}, {px}, {X}", "s1": "v_sub_f32_e32 {D1}, {py}, {Y}", "s2": "v_sub_f32_e32 {D2}, {pz}, {Z}",
    "f0": "v_fma_f32 {R}, {D0}, {D0}, {eps}", "f1": "v_fmac_f32_e32 {R}, {D1}, {D1}", "f2": "v_fmac_f32_e32 {R}, {D2}, {D2}",
    "q": "v_rsq_f32_e32 {R}, {R}",
    "m0": "v_mul_f32_e32 {Q}, {R}, {R}", "m1": "v_mul_f32_e32 {R}, {pm}, {R}", "m2": "v_mul_f32_e32 {R}, {R}, {Q}",
    "a0": "v_fmac_f32_e32 {AX}, {D0}, {R}", "a1": "v_fmac_f32_e32 {AY}, {D1}, {R}", "a2": "v_fmac_f32_e32 {AZ}, {D2}, {R}",
    "nop": "s_nop 0",
    # variants
    "f1v": "v_fma_f32 {R}, {D1}, {D1}, {R}", "f2v": "v_fma_f32 {R}, {D2}, {D2}, {R}", "m0v": "v_mul_f32_e64 {Q}, {R}, {R}",
    "m0c": "v_mul_f32_e32 {Q}, {R}, {R}",
    "m0b": "v_mul_f32_e32 {Q}, {pm}, {R}",            # s = (m*inv) first: consumes the rsq with distinct registers
    "m1b": "v_mul_f32_e32 {R}, {R}, {R}",             # inv^2 in place
    "m2b": "v_mul_f32_e32 {R}, {R}, {Q}",
}
PRE = ["s0", "s1", "s2", "f0", "f1", "f2"]
POST = ["m0", "m1", "m2", "a0", "a1", "a2"]


def emit(sched):
    return "".join('"' + OPS[op].format(**regs(k, u, t)) + '\\n"\n        ' for (op, k, u, t) in sched)


def rows(n=4, u=0):
    return [(k, u) for k in range(n)]


def sched_seq(pairs, nop=True):
    out = []
    for i, (k, u) in enumerate(pairs):
        out += [(o, k, u, 0) for o in PRE] + [("q", k, u, 0)] + ([("nop", k, u, 0)] if nop else []) + [(o, k, u, 0) for o in POST]
    return out


def sched_batch(pairs, consume_first=True, chain_post=False):
    """[pre of all] [rsq of all] then either stage-by-stage post (consume_first) or chain by chain."""
    out = []
    for t, (k, u) in enumerate(pairs):
        out += [(o, k, u, t) for o in PRE]
    out += [("q", k, u, t) for t, (k, u) in enumerate(pairs)]
    if chain_post:
        for t, (k, u) in enumerate(pairs):
            out += [(o, k, u, t) for o in POST]
    else:
        for o in POST:
            out += [(o, k, u, t) for t, (k, u) in enumerate(pairs)]
    return out


def sched_batch_m0_then_chain(pairs):
    """[pre of all] [rsq of all] [m0 of all: every rsq consumed] then the rest chain by chain."""
    out = []
    for t, (k, u) in enumerate(pairs):
        out += [(o, k, u, t) for o in PRE]
    out += [("q", k, u, t) for t, (k, u) in enumerate(pairs)]
    out += [("m0", k, u, t) for t, (k, u) in enumerate(pairs)]
    for t, (k, u) in enumerate(pairs):
        out += [(o, k, u, t) for o in POST[1:]]
    return out


def sched_groups(pairs, g):
    """batches of g interactions: batch_m0_then_chain on each group."""
    out = []
    for i in range(0, len(pairs), g):
        grp = pairs[i:i + g]
        for t, (k, u) in enumerate(grp):
            out += [(o, k, u, t) for o in PRE]
        out += [("q", k, u, t) for t, (k, u) in enumerate(grp)]
        out += [("m0", k, u, t) for t, (k, u) in enumerate(grp)]
        for t, (k, u) in enumerate(grp):
            out += [(o, k, u, t) for o in POST[1:]]
    return out


def sched_rsq_late(pairs):
    """software pipeline across groups of 4: rsq+m0 of group g issued right after the pre of group g, post of g-1 after."""
    out = []
    return out


def sched_seq_altconsume(pairs):
    """row after row; the rsq is consumed by m*inv (distinct registers) first."""
    out = []
    for (k, u) in pairs:
        out += [(o, k, u, 0) for o in PRE] + [("q", k, u, 0), ("nop", k, u, 0), ("m0b", k, u, 0), ("m1b", k, u, 0),
                                               ("m2b", k, u, 0)] + [(o, k, u, 0) for o in ("a0", "a1", "a2")]
    return out


def raw(lines):
    return [("raw", l) for l in lines]


FM = ["v_fmac_f32_e32 v180, v0, v1", "v_fmac_f32_e32 v181, v0, v1", "v_fmac_f32_e32 v182, v0, v1", "v_fmac_f32_e32 v183, v0, v1",
      "v_fmac_f32_e32 v184, v0, v1", "v_fmac_f32_e32 v185, v0, v1", "v_fmac_f32_e32 v186, v0, v1", "v_fmac_f32_e32 v187, v0, v1"]

PATTERNS = []   # (id, description, units per body, asm body)


def add(pid, desc, units, sched=None, rawlines=None, rep=4):
    if rawlines is not None:
        body = "".join('"' + l + '\\n"\n        ' for l in rawlines)
    else:
        body = emit(sched)
    PATTERNS.append((pid, desc, units * rep, body, rep))


P4 = rows(4, 0)
FMX = [f"v_fmac_f32_e32 v{180 + (i % 16)}, v0, v1" for i in range(64)]
import os as _os
MODE = _os.environ.get("SCHED_MODE", "scan")

def sym_step(pp, R=1, lds=True):
    """one step of the symmetric (pair-once) scheme: column body of lane (l+s) from LDS, R row bodies per lane,
    column accumulators rotate by one lane per step (v_add_f32_dpp wave_ror:1, ping-pong registers)."""
    ca = [f"v{60+3*pp+i}" for i in range(3)]
    cb = [f"v{60+3*(1-pp)+i}" for i in range(3)]
    L = []
    if lds:
        L += ["ds_read_b128 v[0:3], v10", "s_waitcnt lgkmcnt(0)"]
    for k in range(R):
        X, Y, Z, M = f"v{13+4*k}", f"v{14+4*k}", f"v{15+4*k}", f"v{16+4*k}"
        AX, AY, AZ = f"v{100+3*k}", f"v{101+3*k}", f"v{102+3*k}"
        D0, D1, D2, RR, Q, SC = "v45", "v46", "v47", "v40", "v41", "v42"
        L += [f"v_sub_f32_e32 {D0}, v0, {X}", f"v_sub_f32_e32 {D1}, v1, {Y}", f"v_sub_f32_e32 {D2}, v2, {Z}",
              f"v_fma_f32 {RR}, {D0}, {D0}, v176", f"v_fmac_f32_e32 {RR}, {D1}, {D1}", f"v_fmac_f32_e32 {RR}, {D2}, {D2}",
              f"v_rsq_f32_e32 {RR}, {RR}", "s_nop 0",
              f"v_mul_f32_e32 {Q}, {RR}, {RR}", f"v_mul_f32_e32 {RR}, {RR}, {Q}",
              f"v_mul_f32_e32 {SC}, {M}, {RR}", f"v_mul_f32_e32 {RR}, v3, {RR}",
              f"v_fmac_f32_e32 {AX}, {D0}, {RR}", f"v_fmac_f32_e32 {AY}, {D1}, {RR}", f"v_fmac_f32_e32 {AZ}, {D2}, {RR}"]
        if k == 0:
            L += [f"v_mul_f32_e32 {D0}, {D0}, {SC}", f"v_mul_f32_e32 {D1}, {D1}, {SC}", f"v_mul_f32_e32 {D2}, {D2}, {SC}",
                  f"v_add_f32_dpp {cb[0]}, {ca[0]}, {D0} wave_ror:1 row_mask:0xf bank_mask:0xf",
                  f"v_add_f32_dpp {cb[1]}, {ca[1]}, {D1} wave_ror:1 row_mask:0xf bank_mask:0xf",
                  f"v_add_f32_dpp {cb[2]}, {ca[2]}, {D2} wave_ror:1 row_mask:0xf bank_mask:0xf"]
        else:
            L += [f"v_fmac_f32_e32 {cb[0]}, {D0}, {SC}", f"v_fmac_f32_e32 {cb[1]}, {D1}, {SC}", f"v_fmac_f32_e32 {cb[2]}, {D2}, {SC}"]
    return L


add("seq4", "reference point: one-sided, row after row (per ORDERED interaction)", 4, sched_seq(P4))
for R in (1, 2, 4):
    add(f"sym{R}", f"symmetric step, {R} row(s) per lane, column from LDS (per PAIR = 2 ordered interactions)", 2 * R,
        rawlines=sym_step(0, R) + sym_step(1, R), rep=4)
add("sym1_nolds", "symmetric step, 1 row, no LDS read (per pair)", 2, rawlines=sym_step(0, 1, False) + sym_step(1, 1, False), rep=4)
add("dpp8", "8 x v_add_f32_dpp wave_ror:1", 8,
    rawlines=[f"v_add_f32_dpp v{60+i}, v{70+i}, v0 wave_ror:1 row_mask:0xf bank_mask:0xf" for i in range(8)], rep=8)
add("movdpp8", "8 x v_mov_b32_dpp wave_ror:1", 8,
    rawlines=[f"v_mov_b32_dpp v{60+i}, v{70+i} wave_ror:1 row_mask:0xf bank_mask:0xf" for i in range(8)], rep=8)
add("dpprow8", "8 x v_add_f32_dpp row_ror:1 (within 16 lanes)", 8,
    rawlines=[f"v_add_f32_dpp v{60+i}, v{70+i}, v0 row_ror:1 row_mask:0xf bank_mask:0xf" for i in range(8)], rep=8)
TEMPLATE = r'''// GENERATED by tools/gen_sched.py -- do not edit.  Steady-state SIMD cycles of instruction schedules (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CLOB %(clob)s
#define INIT_REGS asm volatile(%(init)s ::: CLOB)
#ifdef DENORM_FLUSH
#define MODE_SETUP asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 4, 2), 0" ::: "memory")
#else
#define MODE_SETUP
#endif
#define DEF_KERNEL(ID, BODY)                                                                                       \
    __global__ __launch_bounds__(256) void rate_##ID(unsigned long long *out, unsigned ticks)                      \
    {                                                                                                              \
        INIT_REGS;                                                                                                 \
        MODE_SETUP;                                                                                                \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                            \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                \
        unsigned long long bodies = 0;                                                                             \
        while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) {                                                    \
            asm volatile(BODY ::: CLOB);                                                                           \
            ++bodies;                                                                                              \
        }                                                                                                          \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                            \
        if ((threadIdx.x & 63) == 0) {                                                                             \
            const size_t w = (size_t)(blockIdx.x * blockDim.x + threadIdx.x) >> 6;                                 \
            out[4 * w + 0] = bodies; out[4 * w + 1] = t1 - t0; out[4 * w + 2] = r1 - r0; out[4 * w + 3] = r0;      \
        }                                                                                                          \
    }
%(kernels)s
template <typename Kern>
static double run(Kern kern, int cus, int blocks_per_cu, double units_per_body, unsigned long long *dev,
                  std::vector<unsigned long long> &h, bool *skewed)
{
    const int blocks = cus * blocks_per_cu, nw = blocks * 4;
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
    *skewed = (last - first) > 500;
    return cus * 4.0 * (cyc / nw) / (bodies * units_per_body);
}
#define RUN(ID, NAME, UNITS)                                                                                       \
    do {                                                                                                           \
        printf("%%-78s", NAME);                                                                                    \
        for (int b : {2, 4}) { bool sk; double c = run(rate_##ID, cus, b, UNITS, dev, h, &sk); printf("  w%%d: %%7.2f%%s", b, c, sk ? "(!)" : "   "); } \
        printf("\n");                                                                                              \
    } while (0)
int main()
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    const int cus = prop.multiProcessorCount;
    printf("SIMD cycles per unit (interaction, or group for the probes) at 2 and 4 waves/SIMD\n");
    unsigned long long *dev;
    (void)hipMalloc((void **)&dev, sizeof(unsigned long long) * cus * 8 * 4 * 4);
    std::vector<unsigned long long> h((size_t)cus * 8 * 4 * 4);
%(runs)s
    return 0;
}
'''


def main():
    nreg = 200
    clob = ",".join(f'"v{i}"' for i in range(nreg))
    init = " \\\n        ".join('"' + " \\n ".join(f"v_mov_b32 v{i}, {'0x3f7fbe77' if i % 2 == 0 else '0x3f810000'}"
                                               for i in range(j, min(j + 8, nreg))) + '\\n"' for j in range(0, nreg, 8))
    kernels, runs = [], []
    for pid, desc, units, body, rep in PATTERNS:
        kernels.append(f"#define BODY_{pid} \\\n        " + body.replace("\n", " \\\n").rstrip(" \\\n") + "\n"
                       + f"DEF_KERNEL({pid}, " + " ".join([f"BODY_{pid}"] * rep) + ")\n")
        runs.append(f'    RUN({pid}, "{desc}", {units});')
    src = TEMPLATE % dict(clob=clob, init=init, kernels="\n".join(kernels), runs="\n".join(runs))
    out = os.path.join(HERE, "ubench_sched.hip")
    open(out, "w").write(src)
    print("wrote", out, len(PATTERNS), "patterns")


if __name__ == "__main__":
    main()
