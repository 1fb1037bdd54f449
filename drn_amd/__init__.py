"""drn_amd -- MI355X-native (gfx950) forward/backward hot path of the Dense
Regression Network for video grounding.  Host code is Python; all math runs in
hand-written HIP kernels behind the C-ABI in include/drn_hip.h (libdrn_hip.so).
"""
__version__ = "0.1.0"
