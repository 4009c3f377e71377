"""Drop-in for the reference package `vit_pytorch_face` (vit_pytorch_face/__init__.py:1-3)."""
from .vit_face import ViT_face, ViT_face_low, ViT_face_up, CosFace  # noqa: F401
from .vit_face import ViTs_face  # noqa: F401
from .modified_VIT import ModifiedViT  # noqa: F401
