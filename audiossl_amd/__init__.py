"""MI355X-native ATST / ATST-Frame pre-training hot path (HIP kernels behind the audiossl.methods.atst surface)."""
__version__ = "0.1.0"
from .compat import install_as_audiossl  # noqa: F401,E402  (registers the upstream module names; see compat.py)
