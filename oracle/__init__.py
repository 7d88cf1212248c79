"""CPU restatement of the reference's hot-path arithmetic.

TEST INFRASTRUCTURE ONLY.  Nothing under torchseg_amd/ or furnace/ may import
this package; it is used by tests/, by __graft_entry__.smoke() and by the
`cpu_baseline` leg of bench.py as the checker / timed CPU port, never as the
thing shipped.  Every function cites the reference file:line it restates
(paths relative to the TorchSeg checkout).

Pinning status: the reference ships no tests or golden vectors (SURVEY.md §4,
§8c).  The restatements of loss_opr.py (OHEM, focal), of the networks and of
seg_oprs are pinned against outputs of the reference's own Python imported in
the build container (tests/golden/make_golden.py -> tests/golden/*.npz); the
SyncBN restatement follows the legacy in-tree kernels, which cannot be built
(CUDA-only, torch-1.0 API) and is pinned against torch.nn.BatchNorm2d on the
rank-concatenated batch instead.  apex itself is absent: "parity unpinned" for
anything that only apex defines.  metric_ref.py (confusion matrix, mIoU) is
pinned against the reference's furnace/seg_opr/metric.py through
tests/golden/metric_golden.npz.  conv_ref.py restates torch's Conv2d
definition (the reference's stems and layer1 call cuDNN through it) and is
pinned against torch's CPU convolution and autograd in fp64: no reference
vectors can exist for a library call ("parity unpinned" beyond that).
"""
