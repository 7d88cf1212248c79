"""One fused upsample -> OHEM head at the bench shape on trained-like logits (bench.ohem_kth_branch_probe): the run to put
under `rocprofv3 --kernel-trace --stats` to see which selection kernels the k-th-value branch spends its time in."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torchseg_amd.upsample import install_aten_overrides
install_aten_overrides()
out = bench.ohem_kth_branch_probe(torch.device("cuda:0"), 16, 1024, reps=int(os.environ.get("REPS", "5")))
print(json.dumps(out))
