"""dino_amd -- MI355X-native DINOSeg hot path (drop-in for ``from dt_segmentation import DINOSeg``)."""
from .dinoseg import DINOSeg, get_transforms  # noqa: F401
from .weights import VIT_B8, VIT_S8, ViTConfig, procedural_state_dict  # noqa: F401

__all__ = ["DINOSeg", "get_transforms", "ViTConfig", "VIT_S8", "VIT_B8", "procedural_state_dict"]
