"""norm_act of furnace/seg_opr/seg_oprs.py for package modules that cannot import the furnace tree by its bare name."""


def norm_act(bn, relu, x, residual=None):
    from .syncbn import SyncBatchNorm
    if isinstance(bn, SyncBatchNorm):
        return bn(x, residual=residual, relu=relu is not None)
    x = bn(x)
    if residual is not None:
        x = x + residual
    return relu(x) if relu is not None else x
