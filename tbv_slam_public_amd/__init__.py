"""tbv_slam_public_amd -- MI355X-native CFEAR scan-registration hot path of TBV Radar SLAM.

Only what the path needs: csrc/ (HIP kernels + the C-ABI, built to libcfear_hip.so), _lib.py
(ctypes binding), api.py (host-side mirror of the reference's cfear_radarodometry classes),
synth.py (synthetic polar sweeps) and dist.py (candidate-batch sharding over the GPUs of a node).
"""
__all__ = ["api", "synth", "_lib"]
