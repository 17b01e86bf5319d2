"""lidar4d_amd -- MI355X-native (gfx950) implementation of LiDAR4D's volumetric LiDAR ray-rendering hot path.

Drop-in for the reference's ``model.lidar4d.LiDAR4D`` (same constructor kwargs, ``render`` contract and
state-dict keys); kernels are hand-written HIP behind the C ABI of include/lidar4d_hip.h.  There is no CPU
fallback: the CPU restatement used for parity checks lives in ``oracle/`` (test infrastructure).
"""
from . import tcnn  # noqa: F401  (tinycudann-compatible Encoding / Network)
from .lidar4d import LiDAR4D  # noqa: F401
from .renderer import LiDAR_Renderer  # noqa: F401
from .hash_field import HashGrid4D, HashGridT  # noqa: F401
from .planes_field import Planes4D  # noqa: F401
from .flow_field import FlowField  # noqa: F401
from .activation import trunc_exp  # noqa: F401

__all__ = ["LiDAR4D", "LiDAR_Renderer", "HashGrid4D", "HashGridT", "Planes4D", "FlowField", "trunc_exp", "tcnn"]
