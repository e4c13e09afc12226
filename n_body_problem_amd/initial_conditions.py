"""Synthetic, seeded initial conditions in the reference's buffer layout.

The reference reads its bodies from files (kernel.cu:190-556); the benchmark and the parity tests
use synthetic input instead.  Everything here produces the two host arrays
the reference hands to ``setParticlesPosition`` / ``setParticlesVelocity`` (kernel.cu:163-188):
``pos`` float32 (n,4) = {x,y,z,mass} and ``vel`` float32 (n,4) = {vx,vy,vz,eps}.

Random numbers come from a counter-based SplitMix64 (``seed``, body index, draw index), so body i
is the same whatever n-range, rank or order it is generated in.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

#: seeds of the BASELINE.json configs (``0x5EED0000 + config#``)
CONFIG_SEED = {k: 0x5EED0000 + k for k in range(1, 6)}


def splitmix64(seed: int, counter: np.ndarray) -> np.ndarray:
    """SplitMix64 output for state ``seed + (counter+1)*golden`` (vectorised, wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + (counter.astype(np.uint64) + np.uint64(1)) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform01(seed: int, body: np.ndarray, draw: int, attempt: int = 0) -> np.ndarray:
    """Uniform double in the open interval (0,1) for (body, draw, attempt)."""
    ctr = (body.astype(np.uint64) << np.uint64(16)) | np.uint64((draw & 0xFF) << 8) | np.uint64(attempt & 0xFF)
    return ((splitmix64(seed, ctr) >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def plummer(n: int, seed: int = CONFIG_SEED[3], r_max: float = 10.0, lo: int = 0, hi: int | None = None,
            recentre: bool = True):
    """Plummer sphere in equilibrium: total mass 1, scale radius 1, G = 1, equal masses 1/n.

    Aarseth-Henon-Wielen sampling; radii truncated at ``r_max`` through the inverse CDF; with
    ``recentre`` the centre of mass position and velocity are subtracted (needs the whole set, so
    it is only applied when the full range ``[0,n)`` is requested).
    Returns ``(pos, vel)`` float32 arrays of shape ``(hi-lo, 4)``; ``vel[:,3]`` (the reference's
    unused per-particle eps, kernel.cu:223) is 0.
    """
    hi = n if hi is None else hi
    body = np.arange(lo, hi, dtype=np.uint64)
    x1_max = (1.0 + 1.0 / (r_max * r_max)) ** -1.5
    x1 = uniform01(seed, body, 0) * x1_max
    r = 1.0 / np.sqrt(x1 ** (-2.0 / 3.0) - 1.0)
    cz = 1.0 - 2.0 * uniform01(seed, body, 1)
    ph = 2.0 * np.pi * uniform01(seed, body, 2)
    sxy = np.sqrt(np.maximum(0.0, 1.0 - cz * cz))
    pos = np.empty((hi - lo, 4), dtype=np.float64)
    pos[:, 0] = r * sxy * np.cos(ph)
    pos[:, 1] = r * sxy * np.sin(ph)
    pos[:, 2] = r * cz
    pos[:, 3] = 1.0 / n

    # speed: q = v / v_esc with density g(q) = q^2 (1-q^2)^(7/2), rejection against 0.1
    q = np.zeros(hi - lo, dtype=np.float64)
    todo = np.ones(hi - lo, dtype=bool)
    for attempt in range(256):
        if not todo.any():
            break
        idx = np.nonzero(todo)[0]
        x4 = uniform01(seed, body[idx], 3, attempt)
        x5 = uniform01(seed, body[idx], 4, attempt)
        ok = 0.1 * x5 < x4 * x4 * (1.0 - x4 * x4) ** 3.5
        q[idx[ok]] = x4[ok]
        todo[idx[ok]] = False
    speed = q * np.sqrt(2.0) * (1.0 + r * r) ** -0.25
    cz = 1.0 - 2.0 * uniform01(seed, body, 5)
    ph = 2.0 * np.pi * uniform01(seed, body, 6)
    sxy = np.sqrt(np.maximum(0.0, 1.0 - cz * cz))
    vel = np.zeros((hi - lo, 4), dtype=np.float64)
    vel[:, 0] = speed * sxy * np.cos(ph)
    vel[:, 1] = speed * sxy * np.sin(ph)
    vel[:, 2] = speed * cz

    if recentre and lo == 0 and hi == n and n > 0:
        pos[:, :3] -= pos[:, :3].mean(axis=0)
        vel[:, :3] -= vel[:, :3].mean(axis=0)
    return pos.astype(np.float32), vel.astype(np.float32)


def uniform_cube(n: int, seed: int = 1, random_masses: bool = True, speed: float = 0.0):
    """Bodies uniform in [-1,1]^3; masses uniform in [0.5,1.5]/n (or 1/n); velocities uniform in
    [-speed, speed]^3.  Used for small unit fixtures only."""
    body = np.arange(n, dtype=np.uint64)
    pos = np.empty((n, 4), dtype=np.float64)
    for c in range(3):
        pos[:, c] = 2.0 * uniform01(seed, body, c) - 1.0
    pos[:, 3] = (0.5 + uniform01(seed, body, 3)) / max(n, 1) if random_masses else 1.0 / max(n, 1)
    vel = np.zeros((n, 4), dtype=np.float64)
    for c in range(3):
        vel[:, c] = speed * (2.0 * uniform01(seed, body, 4 + c) - 1.0)
    return pos.astype(np.float32), vel.astype(np.float32)


def padded_count(n: int, block: int = 256) -> int:
    """The reference's padded body count ``roundup(n, 256) + 1`` (kernel.cu:260-264)."""
    return (n + block - 1) // block * block + 1


def pad_reference_style(pos: np.ndarray, vel: np.ndarray, block: int = 256):
    """Append the reference's zero-mass padding bodies at the origin (kernel.cu:265-277)."""
    n = pos.shape[0]
    npad = padded_count(n, block)
    p = np.zeros((npad, 4), dtype=np.float32)
    v = np.zeros((npad, 4), dtype=np.float32)
    p[:n] = pos
    v[:n] = vel
    return p, v
