from torchseg_amd.ddp import DistributedDataParallel, Reducer
from torchseg_amd.syncbn import SyncBatchNorm, convert_syncbn_model

__all__ = ['DistributedDataParallel', 'Reducer', 'SyncBatchNorm', 'convert_syncbn_model']
