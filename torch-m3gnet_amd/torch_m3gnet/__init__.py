"""MI355X-native M3GNet energy/force engine behind the `torch_m3gnet` module API.

Drop-in for the forward path of lan496/torch-m3gnet: same `torch_m3gnet.nn` classes, constructor
signatures and `state_dict` keys, same `build_model`, same `MaterialGraph` tensor schema -- but the
compute runs in hand-written HIP kernels for gfx950 (libm3gnet_hip.so, C ABI in include/m3gnet_hip.h).
There is no CPU fallback: calling a module on CPU tensors, or without the built library, raises.
"""
__version__ = "0.1.0"
