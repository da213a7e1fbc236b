"""consistentnerf_amd — MI355X (gfx950) native NeRF render/train hot path behind the reference's
run_nerf.py / run_nerf_view.py surface.  Kernels: consistentnerf_amd/csrc (HIP), ABI: include/cnerf.h.

Importing the package does not touch the GPU; the first kernel call loads libcnerf_hip.so and raises
CnerfError if it is missing (there is no CPU fallback)."""
from ._lib import CnerfError  # noqa: F401
from .ops import NetSpec  # noqa: F401

__all__ = ["CnerfError", "NetSpec", "run_nerf", "run_nerf_view", "run_nerf_helpers", "ops", "optim", "distributed"]
