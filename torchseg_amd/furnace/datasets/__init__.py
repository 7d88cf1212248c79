from .ade import ADE
from .voc import VOC
from .cityscapes import Cityscapes

__all__ = ["ADE", "VOC", "Cityscapes"]
