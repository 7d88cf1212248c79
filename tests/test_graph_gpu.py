"""hipGraph replay of the training step (bench.py --graph 1 / 2) against the eager step: same seed, same inputs, 12
optimizer steps.  Root cause of the round-1 divergence: state the eager warm-up leaves in the convolution
library is bound to the stream the warm-up ran on; bench.GraphedStep therefore keeps warm-up, capture, replays and the
optimizer on ONE stream, and this test fails (NaN by the second replay) if that discipline is broken."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BATCH, SIZE, WARM, STEPS = 8, 512, 3, 12


def _run(cuda, mode, fused, fork_in_graph=False):
    import bench
    from torchseg_amd import convwrw
    convwrw._WRW_IN_GRAPH = bool(fork_in_graph)          # the weight-gradient side stream captured as graph edges (opt-in)
    from torchseg_amd.ddp import DistributedDataParallel
    from torchseg_amd.losses import ProbOhemCrossEntropy2d
    from torchseg_amd.syncbn import SyncBatchNorm
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    from engine.lr_policy import PolyLR
    model, opt, base_lr = bench.build_model(cuda, BATCH, SIZE, ProbOhemCrossEntropy2d, SyncBatchNorm, fused_sgd=fused)
    model = DistributedDataParallel(model, compute_dtype=torch.bfloat16)
    model.train()
    imgs, gts = bench.synthetic_batch(cuda, BATCH, SIZE)
    pol = PolyLR(base_lr, 0.9, 1000)
    losses = []
    if mode == 0:
        for it in range(WARM + STEPS):
            loss = bench.train_step(model, opt, (imgs, gts), pol, it, 1)
            if it >= WARM:
                losses.append(loss.item())
        return losses
    stream = bench.GraphedStep.capture_stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        for it in range(WARM):
            bench.train_step(model, opt, (imgs, gts), pol, it, 1)
        torch.cuda.synchronize()
        graphed = bench.GraphedStep(model, opt, (imgs, gts), 1, opt_inside=mode == 2)
        for it in range(STEPS):
            bench.set_lr(opt, pol, WARM + it)
            loss = graphed()
            losses.append(loss.item())
    torch.cuda.current_stream().wait_stream(stream)
    torch.cuda.synchronize()
    return losses


@pytest.mark.parametrize("mode,fused,fork", [(1, True, False), (1, False, False), (2, True, False), (2, True, True)],
                         ids=["fwd+bwd", "fwd+bwd torch SGD", "whole step (bench default)", "whole step, side stream forked inside the graph"])
def test_graph_replay_trajectory_equals_eager(cuda, mode, fused, fork):
    """The eager run takes the weight-gradient side stream (the default without a reducer); the captured step is linear
    unless `fork` (TSG_WRW_IN_GRAPH=1: the forks and the join become graph edges — correct, but slower to launch)."""
    from torchseg_amd import convwrw
    assert convwrw._WRW_STREAM
    eager = _run(cuda, 0, fused)
    try:
        graph = _run(cuda, mode, fused, fork)
    finally:
        convwrw._WRW_IN_GRAPH = False
    print("eager", ["%.4f" % v for v in eager], "\ngraph", ["%.4f" % v for v in graph])
    assert np.isfinite(graph).all(), graph
    # two EAGER runs of this 8 x 512^2 bf16 configuration already differ by ~0.6 % (MIOpen's atomic split-K weight
    # gradients through the 1x1-map BN layers); at the bench shape graph and eager agree to 1e-3 over 50 steps
    assert np.allclose(graph, eager, rtol=2e-2, atol=0), (graph, eager)
    assert graph[-1] < graph[0]
