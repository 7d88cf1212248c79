from .cityscapes import Cityscapes
