"""dino_amd -- MI355X-native DINOSeg hot path (drop-in for ``from dt_segmentation import DINOSeg``)."""
from .dinoseg import DINOSeg, get_transforms  # noqa: F401
from .weights import VIT_B8, VIT_S8, ViTConfig, procedural_state_dict  # noqa: F401



def set_option(key: str, value: int) -> None:
    """Process-wide library switches (include/dinoseg.h: dinoseg_set_option), e.g. set_option("streams", 2): batches of >= 16
    frames run as two half-batches on two HIP streams."""
    from . import capi
    capi.check(capi.lib().dinoseg_set_option(key.encode(), int(value)))


__all__ = ["DINOSeg", "get_transforms", "ViTConfig", "VIT_S8", "VIT_B8", "procedural_state_dict", "set_option"]
