"""Patch extraction and augmentation for the training path: the
volumes stay resident in HBM, patches are cut, warped and grey-augmented on the GPU and
handed to the training plan as device tensors -- no host round trip per patch."""
from . import transformations          # noqa: F401
from .batch import PatchSampler, RingFeeder   # noqa: F401
from .image import make_affinities    # noqa: F401
