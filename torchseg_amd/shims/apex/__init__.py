"""`apex` import shim: the reference's train.py files do
`from apex.parallel import DistributedDataParallel, SyncBatchNorm`
(model/bisenet/cityscapes.bisenet.R18/train.py:24-25).  Put torchseg_amd/shims
on PYTHONPATH and those names resolve to the MI355X-native implementations."""
from . import parallel  # noqa: F401
