"""Distributed / checkpoint helpers with the surface of furnace/utils/pyt_utils.py.

`all_reduce_tensor` is the per-step loss collective of train.py:129-131; on
ROCm `torch.distributed`'s "nccl" backend is RCCL, so it rides xGMI."""
import argparse
import logging
import os
import time
from collections import OrderedDict

import torch
import torch.distributed as dist

logger = logging.getLogger()

model_urls = {
    'resnet18': 'https://download.pytorch.org/models/resnet18-5c106cde.pth',
    'resnet34': 'https://download.pytorch.org/models/resnet34-333f7ec4.pth',
    'resnet50': 'https://download.pytorch.org/models/resnet50-19c8e357.pth',
    'resnet101': 'https://download.pytorch.org/models/resnet101-5d3b4d8f.pth',
    'resnet152': 'https://download.pytorch.org/models/resnet152-b121ed2d.pth',
}


def reduce_tensor(tensor, dst=0, op=dist.ReduceOp.SUM, world_size=1):
    """pyt_utils.py:25-31: reduce a copy to `dst`, which then holds the mean."""
    out = tensor.detach().clone()
    dist.reduce(out, dst, op)
    if dist.get_rank() == dst:
        out.div_(world_size)
    return out


def all_reduce_tensor(tensor, op=dist.ReduceOp.SUM, world_size=1):
    """pyt_utils.py:34-39: mean over ranks of a copy of `tensor`."""
    out = tensor.detach().clone()
    if out.is_cuda and op == dist.ReduceOp.SUM and out.dtype == torch.float32:
        # the SAME RCCL communicator the SyncBN exchanges and the gradient buckets use, on the current stream: only one
        # communicator is ever in flight (two — this one and the ProcessGroup's — issued concurrently from different
        # streams are the classic cross-communicator deadlock)
        from torchseg_amd import comm as _tsg_comm
        c = _tsg_comm.get(None, like=out)
        if c is not None:
            flat = out.reshape(-1) if out.is_contiguous() else None
            if flat is not None and flat.numel() > 0:
                c.all_reduce(flat)
                out.div_(world_size)
                return out
    dist.all_reduce(out, op)
    out.div_(world_size)
    return out


def load_model(model, model_file, is_restore=False):
    """Non-strict load; `is_restore` re-prefixes keys with 'module.' for a wrapped
    model (pyt_utils.py:42-79)."""
    t0 = time.time()
    if isinstance(model_file, str):
        state = torch.load(model_file, map_location=torch.device('cpu'))
        if 'model' in state.keys():
            state = state['model']
    else:
        state = model_file
    t1 = time.time()
    if is_restore:
        state = OrderedDict(('module.' + k, v) for k, v in state.items())
    model.load_state_dict(state, strict=False)
    own = set(model.state_dict().keys())
    ckpt = set(state.keys())
    if own - ckpt:
        logger.warning('Missing key(s) in state_dict: {}'.format(', '.join(sorted(own - ckpt))))
    if ckpt - own:
        logger.warning('Unexpected key(s) in state_dict: {}'.format(', '.join(sorted(ckpt - own))))
    logger.info("Load model, Time usage:\n\tIO: {}, initialize parameters: {}".format(t1 - t0, time.time() - t1))
    return model


def parse_devices(input_devices):
    """'0,1', '0-3', '*' -> device index list (pyt_utils.py:82-106); '' -> every
    visible device (the reference raises ValueError on int(''))."""
    ndev = torch.cuda.device_count()
    spec = (input_devices or '').strip()
    if spec == '' or spec.endswith('*'):
        return list(range(ndev))
    devices = []
    for tok in spec.split(','):
        if '-' in tok:
            lo, hi = tok.split('-')[0], tok.split('-')[1]
            assert lo != '' and hi != ''
            lo, hi = int(lo), int(hi)
            assert lo < hi and hi < ndev
            devices.extend(range(lo, hi + 1))
        else:
            d = int(tok)
            assert d < ndev
            devices.append(d)
    logger.info('using devices {}'.format(', '.join(str(d) for d in devices)))
    return devices


def extant_file(x):
    """argparse type: path must exist (pyt_utils.py:109-117)."""
    if not os.path.exists(x):
        raise argparse.ArgumentTypeError("{0} does not exist".format(x))
    return x


def link_file(src, target):
    """Replace `target` by a symlink to `src` (pyt_utils.py:120-123, without the shell)."""
    if os.path.islink(target) or os.path.isfile(target):
        os.remove(target)
    elif os.path.isdir(target):
        os.rmdir(target)
    os.symlink(src, target)


def ensure_dir(path):
    os.makedirs(path, exist_ok=True)
