"""Host-side mirror of the pieces of the reference's utils/utils.py that the training tools call."""
from .weights import TIMM_TO_SIT, load_weights_imagenet  # noqa: F401
