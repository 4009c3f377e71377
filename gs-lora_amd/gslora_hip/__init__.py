"""gslora_hip — Python side of libgslora_hip.so (the MI355X kernels of the GS-LoRA step)."""
from . import _lib  # noqa: F401
