from .voc import VOC
