"""MI355X-native ATST / ATST-Frame pre-training hot path (HIP kernels behind the audiossl.methods.atst surface)."""
__version__ = "0.1.0"
