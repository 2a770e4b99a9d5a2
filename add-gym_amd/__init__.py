"""MI355X-native rollout + PPO/ADD engine behind add-gym's plugin surface.

Hot path = libaddhip.so (hand-written HIP for gfx950, C ABI in include/addhip.h) driven from
this Python host on PyTorch-ROCm tensors.  No CPU fallback: `add_gym_amd._lib.load()` raises
if the library is missing."""
__version__ = "0.1.0"
