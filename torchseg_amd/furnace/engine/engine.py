"""Training engine with the surface of furnace/engine/engine.py:23-163.

Kept: `with Engine(custom_parser=parser) as engine`, the injected CLI flags
(-d/--devices, -c/--continue, --local_rank; engine.py:71-80), `.distributed`,
`.local_rank`, `.world_size`, `.devices`, `.state`, `.continue_state_object`,
register_state / update_iteration / save_checkpoint / save_and_link_checkpoint /
restore_checkpoint, checkpoint dict layout {model, optimizer, epoch, iteration}
with 'module.' stripped (engine.py:93-108).

Changed for MI355X / modern launchers:
  * the process group is RCCL (torch's "nccl" backend on ROCm) when a GPU is
    present, gloo otherwise (CPU plumbing tests);
  * LOCAL_RANK from torchrun is honoured next to --local_rank (engine.py:62);
  * WORLD_SIZE=1 under a launcher still counts as distributed so that the
    reference's train.py (which only binds BatchNorm2d when distributed,
    train.py:54-55) runs on one GPU;
  * HSA_ENABLE_IPC_MODE_LEGACY=0 is exported before RCCL starts (dmabuf IPC).
"""
import argparse
import os
import os.path as osp
import time
from collections import OrderedDict

import torch
import torch.distributed as dist

from .logger import get_logger
from .version import __version__
from utils.pyt_utils import load_model, parse_devices, extant_file, link_file, ensure_dir

logger = get_logger()


class State(object):
    _FIELDS = ('epoch', 'iteration', 'dataloader', 'model', 'optimizer')

    def __init__(self):
        self.epoch = 0
        self.iteration = 0
        self.dataloader = None
        self.model = None
        self.optimizer = None

    def register(self, **kwargs):
        for k, v in kwargs.items():
            assert k in self._FIELDS, k
            setattr(self, k, v)


class Engine(object):
    def __init__(self, custom_parser=None):
        self.version = __version__
        logger.info("PyTorch Version {}, Furnace Version {}".format(torch.__version__, self.version))
        self.state = State()
        self.devices = None
        self.distributed = False
        self.local_rank = 0
        self.world_size = 1

        if custom_parser is None:
            self.parser = argparse.ArgumentParser()
        else:
            assert isinstance(custom_parser, argparse.ArgumentParser)
            self.parser = custom_parser
        self.inject_default_parser()
        self.args = self.parser.parse_args()
        self.continue_state_object = self.args.continue_fpath

        if 'WORLD_SIZE' in os.environ:
            self.distributed = int(os.environ['WORLD_SIZE']) >= 1

        if self.distributed:
            self.local_rank = int(os.environ.get('LOCAL_RANK', self.args.local_rank))
            self.world_size = int(os.environ['WORLD_SIZE'])
            use_gpu = torch.cuda.is_available()
            if use_gpu:
                os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
                from torchseg_amd.tuning import use_shipped_miopen_db
                use_shipped_miopen_db(rank=self.local_rank)
                torch.cuda.set_device(self.local_rank)
            if not dist.is_initialized():
                dist.init_process_group(backend="nccl" if use_gpu else "gloo", init_method='env://')
            self.devices = list(range(self.world_size))
        else:
            self.devices = parse_devices(self.args.devices)

    def inject_default_parser(self):
        p = self.parser
        p.add_argument('-d', '--devices', default='', help='set data parallel training')
        p.add_argument('-c', '--continue', type=extant_file, metavar="FILE", dest="continue_fpath",
                       help='continue from one certain checkpoint')
        p.add_argument('--local_rank', '--local-rank', default=0, type=int, help='process rank on node')

    def register_state(self, **kwargs):
        self.state.register(**kwargs)

    def update_iteration(self, epoch, iteration):
        self.state.epoch = epoch
        self.state.iteration = iteration

    def save_checkpoint(self, path):
        logger.info("Saving checkpoint to file {}".format(path))
        t0 = time.time()
        model_state = OrderedDict()
        for k, v in self.state.model.state_dict().items():
            model_state[k[7:] if k.startswith('module.') else k] = v
        blob = {'model': model_state, 'optimizer': self.state.optimizer.state_dict(),
                'epoch': self.state.epoch, 'iteration': self.state.iteration}
        t1 = time.time()
        torch.save(blob, path)
        logger.info("Save checkpoint to file {}, Time usage:\n\tprepare snapshot: {}, IO: {}".format(
            path, t1 - t0, time.time() - t1))

    def save_and_link_checkpoint(self, snapshot_dir, log_dir, log_dir_link):
        ensure_dir(snapshot_dir)
        if not osp.exists(log_dir_link):
            link_file(log_dir, log_dir_link)
        current = osp.join(snapshot_dir, 'epoch-{}.pth'.format(self.state.epoch))
        self.save_checkpoint(current)
        link_file(current, osp.join(snapshot_dir, 'epoch-last.pth'))

    def restore_checkpoint(self):
        t0 = time.time()
        # CPU first: avoids a device-memory surge on every rank (engine.py:130-137)
        tmp = torch.load(self.continue_state_object, map_location=torch.device('cpu'))
        t1 = time.time()
        wrapped = any(k.startswith('module.') for k in self.state.model.state_dict().keys())
        self.state.model = load_model(self.state.model, tmp['model'], wrapped)
        self.state.optimizer.load_state_dict(tmp['optimizer'])
        self.state.epoch = tmp['epoch'] + 1
        self.state.iteration = tmp['iteration']
        del tmp
        logger.info("Load checkpoint from file {}, Time usage:\n\tIO: {}, restore snapshot: {}".format(
            self.continue_state_object, t1 - t0, time.time() - t1))

    def __enter__(self):
        return self

    def __exit__(self, type, value, tb):
        try:                                    # our RCCL communicators go before the process group and the HIP runtime
            from torchseg_amd import comm as _tsg_comm
            _tsg_comm.shutdown()
        except Exception:                       # noqa: BLE001 - never mask the training error that brought us here
            pass
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        if type is not None:
            logger.warning("A exception occurred during Engine initialization, give up running process")
            return False
