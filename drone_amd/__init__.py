"""drone_amd — MI355X-native vectorised drone RL environment (hot path only).

``binding.DroneVec`` is the C-ABI handle (HIP kernels behind include/drone_vec.h),
``env.Drone`` the PufferLib-shaped env class over it, ``dist`` the env sharding
and host-boundary gather. Importing the package loads no native code; the
first ``binding.load()`` does, and fails loudly if libdrone_hip.so is missing.
"""
from . import abi  # noqa: F401

__all__ = ["abi", "binding", "env", "dist"]
