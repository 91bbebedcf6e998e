"""mpcgpu_amd — MI355X-native PCG solver for MPCGPU's block-tridiagonal Schur system.

Product = libmpcg_hip.so (C ABI in include/mpcg.h, kernels in mpcgpu_amd/csrc/).
This package is the thin host-side mirror of the reference's interface for that path.
"""
from .solver import PcgSolver, Plant, QdldlSolver, pcg_config, pcgSharedMemSize  # noqa: F401

__all__ = ["PcgSolver", "Plant", "QdldlSolver", "pcg_config", "pcgSharedMemSize"]
