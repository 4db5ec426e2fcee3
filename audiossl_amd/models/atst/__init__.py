from .atst import ATST, FrameATST  # noqa: F401
