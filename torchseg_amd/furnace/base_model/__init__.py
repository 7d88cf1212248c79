from .resnet import ResNet, resnet18, resnet34, resnet50, resnet101, resnet152

__all__ = ['ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152']
