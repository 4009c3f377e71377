"""Drop-in for the torchvision-free part of the reference's `image_iter.py`: `CustomSubset` (:124-137). The MXNet / ImageFolder readers
of that file are data plumbing outside the hot path."""
from util.utils import CustomSubset  # noqa: F401
