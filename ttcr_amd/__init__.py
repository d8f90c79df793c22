"""ttcr_amd -- MI355X-native fast-sweeping eikonal solver behind ttcrpy's
Grid3d(...).raytrace() / Grid2d(...).raytrace() (method='FSM').

Only the FSM hot path of groupeLIAMG/ttcr is provided; the compute
runs in hand-written HIP kernels (ttcr_amd/csrc) through the C ABI of include/ttcr_amd.h.
"""
from .rgrid import Grid2d, Grid2d_d, Grid2d_f, Grid3d, Grid3d_d, Grid3d_f, set_verbose  # noqa: F401

__all__ = ["Grid3d", "Grid3d_d", "Grid3d_f", "Grid2d", "Grid2d_d", "Grid2d_f", "set_verbose"]
