"""Weight init and optimizer param grouping (furnace/utils/init_func.py:11-57)."""
import torch.nn as nn

_CONV = (nn.Conv2d, nn.Conv3d, nn.ConvTranspose2d)
_OTHER_NORM = (nn.GroupNorm, nn.InstanceNorm2d, nn.LayerNorm)


def _init_one(feature, conv_init, norm_layer, bn_eps, bn_momentum, **kwargs):
    for m in feature.modules():
        if isinstance(m, _CONV):
            conv_init(m.weight, **kwargs)
        elif isinstance(m, norm_layer):
            m.eps = bn_eps
            m.momentum = bn_momentum
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def init_weight(module_list, conv_init, norm_layer, bn_eps, bn_momentum, **kwargs):
    """conv weights <- conv_init(**kwargs); norm layers <- eps/momentum, weight 1, bias 0
    (init_func.py:11-31; called at train.py:61-63 on model.business_layer)."""
    feats = module_list if isinstance(module_list, list) else [module_list]
    for f in feats:
        _init_one(f, conv_init, norm_layer, bn_eps, bn_momentum, **kwargs)


def group_weight(weight_group, module, norm_layer, lr, no_decay_lr=None):
    """Append {decay} and {no-decay, weight_decay=0} groups for `module`
    (init_func.py:34-57): conv/linear weights decay; biases and norm affine params do not."""
    decay, no_decay = [], []
    for m in module.modules():
        if isinstance(m, (nn.Linear,) + _CONV):
            decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
        elif isinstance(m, norm_layer) or isinstance(m, _OTHER_NORM):
            if m.weight is not None:
                no_decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
    assert len(list(module.parameters())) == len(decay) + len(no_decay)
    weight_group.append(dict(params=decay, lr=lr))
    weight_group.append(dict(params=no_decay, weight_decay=.0, lr=lr if no_decay_lr is None else no_decay_lr))
    return weight_group
