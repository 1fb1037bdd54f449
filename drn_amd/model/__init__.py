"""Reference-shaped modules (same class names, constructor/forward signatures and state_dict keys as
/root/reference/model/*) whose math runs in the HIP kernels of libdrn_hip.so."""
from .main_model import mainModel  # noqa: F401
from .fcos import build_fcos  # noqa: F401
