"""MI355X-native ego-prior renderer: host-side mirror of the reference's scripts/render_vipe_pointcloud.py over the
C ABI in include/egoprior.h (HIP kernels under ../csrc).  There is no CPU execution path."""
__all__ = ["_lib", "engine", "synth"]
