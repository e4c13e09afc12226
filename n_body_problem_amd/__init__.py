"""MI355X-native all-pairs N-body step: drop-in for the step path of ctbfl/N_body_problem.

Layout: ``csrc/`` hand-written HIP kernels (gfx950) + the C ABI of ``include/nbody.h``;
``system.py`` the host-side mirror of the reference's step interface; ``multi.py`` rows
sharded over the GPUs of one node (the exchange inside the library); ``initial_conditions.py`` seeded synthetic inputs.
"""
from .initial_conditions import plummer, uniform_cube, pad_reference_style, padded_count, CONFIG_SEED  # noqa: F401
from ._lib import NBodyError  # noqa: F401
from .system import (NBodySystem, initialize, step, default_split_len, TIME_TICK, SOFTENING_VERSION3,  # noqa: F401
                     SOFTENING_VERSION1, BLOCK_SIZE, pair_once_split_len, morton_order)

__all__ = ["NBodySystem", "initialize", "step", "default_split_len", "plummer", "uniform_cube",
           "pad_reference_style", "padded_count", "NBodyError", "TIME_TICK", "SOFTENING_VERSION3",
           "SOFTENING_VERSION1", "BLOCK_SIZE", "CONFIG_SEED", "pair_once_split_len", "morton_order"]
