"""Dataset loaders of the reference (SURVEY.md component C3, "next" row N2) and raw snapshots (N3).

The reference fills its two host arrays -- positions ``float4 {x,y,z,mass}`` and velocities
``float4 {vx,vy,vz,eps}`` -- from four file formats (main_project/kernel.cu):

* Tipsy binary   ``readTipsyFile`` :190-282  (structs :103-128)      e.g. data/galaxy_20K.bin
* ``.tab`` text  ``readTabFile``   :305-354  "mass x y z vx vy vz"    (dubinski.tab, tab65536.tab: not shipped)
* ``.dat`` text  ``readDatFile``   :368-418  "z y x vz vy vx", mass 1 e.g. data/stars.dat
* ``.snap`` text ``readSnapFile``  :433-544  n, ndim, time, masses, positions, velocities, eps

Every reader returns ``(pos, vel)`` float32 arrays of the REAL bodies; the reference's zero-mass padding to
``roundup(n,256)+1`` (:260-278) is a separate, optional step (``initial_conditions.pad_reference_style``)
because the MI355X kernels do not need it.  Reference quirks fixed on purpose: the
``while(!eof)`` loops of the text readers append one garbage body after the last line and leave ``vel.w``
uninitialised -- here blank lines are skipped and ``vel.w`` is 0; ``load_data`` choices 4/5 feed ``.snap`` files
to the ``.dat`` parser (:1001-1011) -- here a ``.snap`` file gets the ``.snap`` parser.
"""
from __future__ import annotations

import os
import struct

import numpy as np

# struct Header{double time; int nbodies, ndimension, nsph, ndark, nstar;} -> 28 bytes + 4 of tail padding
TIPSY_HEADER = struct.Struct("<diiiii4x")
# DarkParticle{mass, pos[3], vel[3], eps, int phi} = 36 B; StarParticle{mass, pos[3], vel[3], metals, tform, eps, int phi} = 44 B
TIPSY_DARK = np.dtype([("mass", "<f4"), ("pos", "<f4", 3), ("vel", "<f4", 3), ("eps", "<f4"), ("phi", "<i4")])
TIPSY_STAR = np.dtype([("mass", "<f4"), ("pos", "<f4", 3), ("vel", "<f4", 3), ("metals", "<f4"), ("tform", "<f4"),
                       ("eps", "<f4"), ("phi", "<i4")])
assert TIPSY_HEADER.size == 32 and TIPSY_DARK.itemsize == 36 and TIPSY_STAR.itemsize == 44


def _pack(mass, xyz, vxyz, eps):
    n = len(mass)
    pos = np.empty((n, 4), dtype=np.float32)
    vel = np.empty((n, 4), dtype=np.float32)
    pos[:, :3], pos[:, 3] = xyz, mass
    vel[:, :3], vel[:, 3] = vxyz, eps
    return pos, vel


def read_tipsy(path: str):
    """Tipsy binary as the reference reads it: the first ``ndark`` records are dark particles, the rest star
    particles (gas is ignored, kernel.cu:217-247).  ``vel[:,3]`` carries the per-particle eps."""
    with open(path, "rb") as f:
        raw = f.read()
    time, nbodies, ndim, nsph, ndark, nstar = TIPSY_HEADER.unpack_from(raw, 0)
    nd = min(ndark, nbodies)
    ns = nbodies - nd
    need = TIPSY_HEADER.size + nd * TIPSY_DARK.itemsize + ns * TIPSY_STAR.itemsize
    if len(raw) < need:
        raise ValueError(f"{path}: truncated Tipsy file ({len(raw)} < {need} bytes)")
    dark = np.frombuffer(raw, dtype=TIPSY_DARK, count=nd, offset=TIPSY_HEADER.size)
    star = np.frombuffer(raw, dtype=TIPSY_STAR, count=ns, offset=TIPSY_HEADER.size + nd * TIPSY_DARK.itemsize)
    pos, vel = _pack(np.concatenate([dark["mass"], star["mass"]]), np.concatenate([dark["pos"], star["pos"]]),
                     np.concatenate([dark["vel"], star["vel"]]), np.concatenate([dark["eps"], star["eps"]]))
    return pos, vel


def write_tipsy(path: str, pos, vel, ndark: int, time: float = 0.0) -> None:
    """Inverse of :func:`read_tipsy` (used to build fixtures and to export states)."""
    pos = np.asarray(pos, dtype=np.float32).reshape(-1, 4)
    vel = np.asarray(vel, dtype=np.float32).reshape(-1, 4)
    n = pos.shape[0]
    dark = np.zeros(ndark, dtype=TIPSY_DARK)
    star = np.zeros(n - ndark, dtype=TIPSY_STAR)
    for rec, sl in ((dark, slice(0, ndark)), (star, slice(ndark, n))):
        rec["mass"], rec["pos"], rec["vel"], rec["eps"] = pos[sl, 3], pos[sl, :3], vel[sl, :3], vel[sl, 3]
    with open(path, "wb") as f:
        f.write(TIPSY_HEADER.pack(time, n, 3, 0, ndark, n - ndark))
        f.write(dark.tobytes())
        f.write(star.tobytes())


def _rows(path: str, ncol: int) -> np.ndarray:
    rows = []
    with open(path) as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue  # the reference's while(!eof) would turn the trailing blank line into a garbage body
            if len(parts) < ncol:
                raise ValueError(f"{path}: expected {ncol} columns, got {len(parts)}: {line!r}")
            rows.append([float(x) for x in parts[:ncol]])
    return np.array(rows, dtype=np.float64).reshape(-1, ncol)


def read_tab(path: str):
    """``mass x y z vx vy vz`` per line (kernel.cu:321-323)."""
    a = _rows(path, 7)
    return _pack(a[:, 0], a[:, 1:4], a[:, 4:7], 0.0)


def write_tab(path: str, pos, vel) -> None:
    pos, vel = np.asarray(pos), np.asarray(vel)
    with open(path, "w") as f:
        for p, v in zip(pos, vel):
            f.write(f"{p[3]:.9g} {p[0]:.9g} {p[1]:.9g} {p[2]:.9g} {v[0]:.9g} {v[1]:.9g} {v[2]:.9g}\n")


def read_dat(path: str):
    """``z y x vz vy vx`` per body, every mass 1.0 (kernel.cu:379-387: note the reversed axis order).

    Parsed as a stream of 6 numbers per body: ``data/stars.dat`` wraps 35 of its records over two lines
    (5 + 1 numbers), which the reference's line-based reader turns into 70 corrupted bodies (43837 instead of
    43802) -- fixed here on purpose, like the other loader quirks."""
    with open(path) as f:
        tok = f.read().split()
    if len(tok) % 6:
        raise ValueError(f"{path}: {len(tok)} numbers is not a whole number of 6-column records")
    a = np.array(tok, dtype=np.float64).reshape(-1, 6)
    return _pack(np.ones(len(a)), a[:, [2, 1, 0]], a[:, [5, 4, 3]], 0.0)


def write_dat(path: str, pos, vel) -> None:
    pos, vel = np.asarray(pos), np.asarray(vel)
    with open(path, "w") as f:
        for p, v in zip(pos, vel):
            f.write(f" {p[2]:.9g} {p[1]:.9g} {p[0]:.9g} {v[2]:.9g} {v[1]:.9g} {v[0]:.9g}\n")


def read_snap(path: str):
    """NEMO-style ASCII snapshot: n, ndim, time, then n masses, n positions, n velocities, n eps
    (kernel.cu:445-529)."""
    with open(path) as f:
        tok = f.read().split()
    n, ndim = int(tok[0]), int(tok[1])
    need = 3 + n + 2 * n * ndim + n
    if ndim != 3 or len(tok) < need:
        raise ValueError(f"{path}: not a 3-d snap file with {n} bodies ({len(tok)} tokens, need {need})")
    a = np.array(tok[3:need], dtype=np.float64)
    mass = a[:n]
    xyz = a[n:n + 3 * n].reshape(n, 3)
    vxyz = a[n + 3 * n:n + 6 * n].reshape(n, 3)
    eps = a[n + 6 * n:n + 7 * n]
    return _pack(mass, xyz, vxyz, eps)


def write_snap(path: str, pos, vel, time: float = 0.0) -> None:
    pos, vel = np.asarray(pos), np.asarray(vel)
    n = pos.shape[0]
    with open(path, "w") as f:
        f.write(f"  {n}\n      3\n   {time:.6E}\n")
        for m in pos[:, 3]:
            f.write(f"   {m:.9E}\n")
        for p in pos:
            f.write(f"   {p[0]:.9E}   {p[1]:.9E}   {p[2]:.9E}\n")
        for v in vel:
            f.write(f"   {v[0]:.9E}   {v[1]:.9E}   {v[2]:.9E}\n")
        for e in vel[:, 3]:
            f.write(f"   {e:.9E}\n")


def read_any(path: str):
    """Dispatch on the extension, as ``load_data`` (kernel.cu:975-1013) should have."""
    ext = os.path.splitext(path)[1].lower()
    reader = {".bin": read_tipsy, ".tipsy": read_tipsy, ".tab": read_tab, ".dat": read_dat, ".snap": read_snap,
              ".nbs": lambda p: load_snapshot(p)[:2]}.get(ext)
    if reader is None:
        raise ValueError(f"unknown dataset extension {ext!r}")
    return reader(path)


#: ``load_data(choice)`` of kernel.cu:975-1013: file per dataset id (ids 1 and 2 are not shipped with the reference)
REFERENCE_DATASETS = {0: "galaxy_20K.bin", 1: "dubinski.tab", 2: "tab65536.tab", 3: "stars.dat", 4: "k17c.snap",
                      5: "k17hp.snap"}


def load_reference_dataset(choice: int, data_dir: str):
    return read_any(os.path.join(data_dir, REFERENCE_DATASETS[choice]))


# ---- raw snapshots (the reference never writes its state to disk) -------------------
SNAP_MAGIC = b"NBODYAMD"
SNAP_HEADER = struct.Struct("<8sIIqqd")   # magic, version, reserved, n, step, time


def save_snapshot(path: str, pos, vel, step: int = 0, time: float = 0.0) -> None:
    """Header + the two float4 buffers exactly as they sit in device memory."""
    pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 4)
    vel = np.ascontiguousarray(vel, dtype=np.float32).reshape(-1, 4)
    if pos.shape != vel.shape:
        raise ValueError("positions and velocities differ in shape")
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(SNAP_HEADER.pack(SNAP_MAGIC, 1, 0, pos.shape[0], int(step), float(time)))
        f.write(pos.tobytes())
        f.write(vel.tobytes())
    os.replace(tmp, path)


def load_snapshot(path: str):
    """(pos, vel, step, time) written by :func:`save_snapshot` (or by host/nbody_run)."""
    with open(path, "rb") as f:
        head = f.read(SNAP_HEADER.size)
        magic, version, _, n, step, time = SNAP_HEADER.unpack(head)
        if magic != SNAP_MAGIC or version != 1:
            raise ValueError(f"{path}: not an nbody snapshot")
        body = f.read()
    if len(body) != 2 * n * 16:
        raise ValueError(f"{path}: truncated snapshot")
    a = np.frombuffer(body, dtype=np.float32)
    return a[:4 * n].reshape(n, 4).copy(), a[4 * n:].reshape(n, 4).copy(), step, time
