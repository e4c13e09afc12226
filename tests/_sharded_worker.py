"""Helpers of test_sharded_cpu.py: a CPU stand-in for the HIP kernels (built on the oracle, which only
tests may use) and the per-rank worker run under torch.distributed/gloo."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleKernels:
    """Same interface and the same split/partial-sum semantics as NBodySystem, on CPU tensors."""

    def __init__(self, n_total, row_lo, row_count, split_len):
        import torch
        import oracle
        self.oracle = oracle
        self.n, self.row_lo, self.rows, self.L = n_total, row_lo, row_count, split_len
        self.n_splits = -(-n_total // split_len)
        self.positions = torch.zeros((n_total, 4), dtype=torch.float32)
        self.velocities = torch.zeros((row_count, 4), dtype=torch.float32)
        self.partials = {}
        self.calls = []

    def forces(self, col_lo, col_count, softening, positions=None):
        assert col_lo % self.L == 0 and ((col_lo + col_count) % self.L == 0 or col_lo + col_count == self.n)
        pos = self.positions.numpy()
        self.calls.append(("range", col_lo, col_count))
        for s in range(col_lo // self.L, -(-(col_lo + col_count) // self.L)):
            j0, j1 = s * self.L, min((s + 1) * self.L, self.n)
            assert s not in self.partials, "split computed twice in one step"
            self.partials[s] = self.oracle.accel_f32(pos, self.row_lo, self.row_lo + self.rows, j0, j1, softening,
                                                     threads=1)

    def forces_complement(self, col_lo, col_count, softening, positions=None):
        self.calls.append(("complement", col_lo, col_count))
        if col_lo > 0:
            self.forces(0, col_lo, softening)
        if col_lo + col_count < self.n:
            self.forces(col_lo + col_count, self.n - (col_lo + col_count), softening)

    def update(self, dt, positions=None, velocities=None):
        assert sorted(self.partials) == list(range(self.n_splits)), "a split is missing"
        acc = self.partials[0].copy()
        for s in range(1, self.n_splits):
            acc += self.partials[s]
        self.partials = {}
        rows = self.positions.numpy()[self.row_lo:self.row_lo + self.rows]
        self.oracle.update_f32(rows, self.velocities.numpy(), acc, dt)

    def _reduce(self):
        assert sorted(self.partials) == list(range(self.n_splits)), "a split is missing"
        acc = self.partials[0].copy()
        for s in range(1, self.n_splits):
            acc += self.partials[s]
        self.partials = {}
        return acc

    def kdk_prepare(self):
        self.acc = self._reduce()

    def kdk_kick_drift(self, dt, positions=None, velocities=None):
        rows = self.positions.numpy()[self.row_lo:self.row_lo + self.rows]
        v = self.velocities.numpy()
        hh, h = 0.5 * float(np.float32(dt)), float(np.float32(dt))
        v[:, :3] = (self.acc.astype(np.float64) * hh + v[:, :3].astype(np.float64)).astype(np.float32)
        rows[:, :3] = (v[:, :3].astype(np.float64) * h + rows[:, :3].astype(np.float64)).astype(np.float32)

    def kdk_kick(self, dt, velocities=None):
        self.acc = self._reduce()
        v = self.velocities.numpy()
        hh = 0.5 * float(np.float32(dt))
        v[:, :3] = (self.acc.astype(np.float64) * hh + v[:, :3].astype(np.float64)).astype(np.float32)

    def sync(self):
        pass

    def energy(self, softening):
        # rows of this rank against all columns, as nbody_energy defines it
        p = self.positions.numpy()
        v = np.zeros_like(p)
        v[self.row_lo:self.row_lo + self.rows] = self.velocities.numpy()
        m = p[:, 3].astype(np.float64)
        k = 0.5 * (m[:, None] * v[:, :3].astype(np.float64) ** 2).sum()
        u = 0.0
        for i in range(self.row_lo, self.row_lo + self.rows):
            d = p[:, :3].astype(np.float64) - p[i, :3].astype(np.float64)
            s = (d * d).sum(1) + float(np.float32(softening)) ** 2
            s[i] = np.inf
            with np.errstate(divide="ignore"):
                u -= 0.5 * m[i] * (m / np.sqrt(s)).sum()
        return np.array([k, u, k + u])

    def momentum(self):
        m = self.positions.numpy()[self.row_lo:self.row_lo + self.rows, 3].astype(np.float64)
        v = self.velocities.numpy()[:, :3].astype(np.float64)
        return np.array([*(m[:, None] * v).sum(0), m.sum()])


class PairOnceOracleKernels(OracleKernels):
    """The DATA FLOW of the pair-once mode on CPU tensors (one-sided oracle arithmetic inside each tile): row-side sums
    P_row[C][own b], column-side sums P_col[own R][any c], the per-group reduction the ranks exchange and the fixed
    summation order of nbody::sym_finalize_kernel.  World sizes 1, 2, 4, 8 must give the same bits."""

    def __init__(self, n_total, row_lo, row_count, split_len):
        import torch
        from n_body_problem_amd import system
        super().__init__(n_total, row_lo, row_count, split_len)
        self.groups = system.SYM_GROUPS
        self.gs = -(-self.n_splits // self.groups)
        self.p_col = {}
        self.colparts = torch.zeros((self.groups, n_total, 4), dtype=torch.float32)
        self.reduced = False

    def set_force_mode(self, mode):
        assert mode == "pair_once"
        assert self.row_lo % (self.gs * self.L) == 0 and self.rows % (self.gs * self.L) == 0

    def sym_groups(self):
        return self.row_lo // (self.gs * self.L), self.rows // (self.gs * self.L), self.gs

    def sym_own_slice(self):
        lo, cnt, _ = self.sym_groups()
        return self.colparts[lo:lo + cnt]

    def forces(self, col_lo, col_count, softening, positions=None):
        from sharded_harness import sym_rows_side
        pos = self.positions.numpy()
        self.calls.append(("range", col_lo, col_count))
        own = range(self.row_lo // self.L, (self.row_lo + self.rows) // self.L)
        for C in range(col_lo // self.L, -(-(col_lo + col_count) // self.L)):
            assert C not in self.partials, "split computed twice in one step"
            self.partials[C] = np.zeros((self.rows, 3), dtype=np.float32)
            c0, c1 = C * self.L, min((C + 1) * self.L, self.n)
            for R in own:
                r0, r1 = R * self.L, (R + 1) * self.L
                if C == R or sym_rows_side(R, C, self.n_splits):
                    self.partials[C][r0 - self.row_lo:r1 - self.row_lo] = self.oracle.accel_f32(pos, r0, r1, c0, c1,
                                                                                               softening, threads=1)
                if C != R and sym_rows_side(R, C, self.n_splits):
                    self.p_col[(R, C)] = self.oracle.accel_f32(pos, c0, c1, r0, r1, softening, threads=1)
        self.reduced = False

    def sym_reduce(self):
        from sharded_harness import sym_rows_side
        assert sorted(self.partials) == list(range(self.n_splits)), "a split is missing"
        lo, cnt, gs = self.sym_groups()
        cp = self.colparts.numpy()
        for g in range(lo, lo + cnt):
            cp[g] = 0.0
            for R in range(g * gs, min((g + 1) * gs, self.n_splits)):
                for C in range(self.n_splits):
                    if sym_rows_side(R, C, self.n_splits):
                        cp[g, C * self.L:C * self.L + self.p_col[(R, C)].shape[0], :3] += self.p_col[(R, C)]
        self.p_col = {}
        self.reduced = True

    def _reduce(self):
        from sharded_harness import sym_rows_side
        assert self.reduced, "sym_reduce (and the exchange) must precede the update"
        cp = self.colparts.numpy()
        acc = np.zeros((self.rows, 3), dtype=np.float32)
        for g in range(-(-self.n_splits // self.gs)):
            rp = np.zeros((self.rows, 3), dtype=np.float32)
            for C in range(g * self.gs, min((g + 1) * self.gs, self.n_splits)):
                for R in range(self.row_lo // self.L, (self.row_lo + self.rows) // self.L):
                    if C == R or sym_rows_side(R, C, self.n_splits):
                        sl = slice(R * self.L - self.row_lo, (R + 1) * self.L - self.row_lo)
                        rp[sl] += self.partials[C][sl]
            acc += rp + cp[g, self.row_lo:self.row_lo + self.rows, :3]
        self.partials = {}
        self.reduced = False
        return acc

    def update(self, dt, positions=None, velocities=None):
        acc = self._reduce()
        rows = self.positions.numpy()[self.row_lo:self.row_lo + self.rows]
        self.oracle.update_f32(rows, self.velocities.numpy(), acc, dt)


def run_rank(rank, world_size, port, exchange, n, split_len, steps, out_dir, integrator="kick_drift",
             force_mode="one_sided"):
    import torch.distributed as dist
    from n_body_problem_amd import initial_conditions as ic
    from sharded_harness import ShardedNBodySystem
    if world_size > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world_size)
    try:
        pos, vel = ic.plummer(n, seed=1234)
        s = ShardedNBodySystem(n, exchange=exchange, split_len=split_len, integrator=integrator, force_mode=force_mode,
                               kernels_factory=PairOnceOracleKernels if force_mode == "pair_once" else OracleKernels)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        e0 = s.energy(1e-2)
        s.step_n(steps, 1e-3, 1e-2)
        p, v = s.download()
        e1 = s.energy(1e-2)
        mom = s.momentum()
        tag = exchange if integrator == "kick_drift" else exchange + "_" + integrator
        if force_mode != "one_sided":
            tag += "_" + force_mode
        np.savez(os.path.join(out_dir, f"w{world_size}_{tag}_r{rank}.npz"), p=p, v=v, e0=e0, e1=e1, mom=mom,
                 calls=np.array([c[1:] for c in s.kernels.calls if c[0] == "range"][-2:], dtype=np.int64),
                 kinds=np.array([c[0] for c in s.kernels.calls]), n_padded=s.n_padded, chunk=s.chunk)
        s.close()
    finally:
        if world_size > 1:
            dist.destroy_process_group()


def run_rank_gpu(rank, world_size, port, exchange, n, steps, out_dir, integrator="kick_drift", force_mode="one_sided",
                 split_len=0):
    """One rank of a sharded run with the REAL HIP kernels; all ranks share cuda:0, gloo carries the exchange."""
    import torch
    import torch.distributed as dist
    from n_body_problem_amd import initial_conditions as ic
    from sharded_harness import ShardedNBodySystem
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world_size)
    try:
        pos, vel = ic.plummer(n, seed=4321)
        s = ShardedNBodySystem(n, device=0, exchange=exchange, integrator=integrator, force_mode=force_mode,
                               split_len=split_len)
        s.setParticlesPosition(pos)
        s.setParticlesVelocity(vel)
        s.step_n(steps, 1e-3, 1e-3)
        p, v = s.download()
        e = s.energy(1e-3)
        tag = exchange if integrator == "kick_drift" else exchange + "_" + integrator
        if force_mode != "one_sided":
            tag += "_" + force_mode
        np.savez(os.path.join(out_dir, f"gpu_w{world_size}_{tag}_r{rank}.npz"), p=p, v=v, e=e,
                 n_padded=s.n_padded, chunk=s.chunk, split_len=s.split_len)
        s.close()
    finally:
        dist.destroy_process_group()
