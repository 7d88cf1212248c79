"""Parity at the HEADLINE geometry (BASELINE configs[1]: BiSeNet-R18, 1024 x 1024 crops, SyncBN + OHEM with
min_kept = B*H*W/16, thresh 0.7), against the CPU oracle network run live on the host (reference architecture,
nn.BatchNorm2d, loss_opr.py restatement; bisenet network.py:75-111, loss_opr.py:68-98), plus index-width guards for
the BN kernels at the bench's largest activations and a multi-step trajectory against stock PyTorch ops.

Bars (round 4).  north_star asks for fp32 loss / logits within 1e-4 of the reference CPU path.  Both hold here END TO END:
the loss to 1e-6 and the full-resolution logits of all three heads to 6.5-8.0e-5 ABSOLUTE at a logit scale of 1.6-2.3 —
which is the CPU path's own distance from the float64 evaluation of the network (tools/diag_fp64_truth.py: CPU 6.9-8.3e-5,
ours 1.3-1.9e-5 from the truth).  Rounds 1-3 measured 0.8-1.5e-3 and blamed the vendor library's convolutions; the cause
was the fp32 sum / square-sum formulation of the BatchNorm statistics (the reference's own, syncbn_kernel.cu:12-23),
whose cancellation the batch-2 BatchNorm of the global-context branch amplifies ~300x (fixed in csrc/bn.hip: fp64
accumulators + hi / lo partial rows for fp32 tensors; csrc/pool.hip: fp64 pooling sums; csrc/convf32.hip: exact
convolutions).  Stock PyTorch-ROCm modules on the same GPU are still measured beside ours (`floor`, 7-9e-4).  The test
asserts (a) logits <= 1e-4 of the logit scale (measured: below 1e-4 absolute), (b) loss within 1e-4, (c) the OHEM kept mask equal to the reference's except pixels
whose probability lies within the band the MEASURED logit error of the run implies (2 x that error, relative: 0 pixels),
(d) gradients 1e-2 in relative L2 over all parameters (measured 3.0e-3).  Selection exactness at the real batch (16 images,
min_kept 1 048 576) is checked without any convolution in between by
test_batch16_selection_against_the_oracle_at_the_real_min_kept."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

B, S, C = 2, 1024, 19          # a CPU step of batch 2 at 1024^2 is a few seconds on the GPU box's host


def _nets(cuda, compute_dtype):
    from oracle.ohem_ref import ProbOhemCrossEntropy2d as OracleOhem
    from torchseg_amd.ddp import DistributedDataParallel
    from torchseg_amd.losses import ProbOhemCrossEntropy2d
    from torchseg_amd.syncbn import SyncBatchNorm
    from torchseg_amd.workloads.bisenet import BiSeNet
    min_kept = B * S * S // 16                                          # train.py:48-49
    torch.manual_seed(12345)
    ref = BiSeNet(C, True, OracleOhem(255, thresh=0.7, min_kept=min_kept), None, nn.BatchNorm2d)
    crit = ProbOhemCrossEntropy2d(255, thresh=0.7, min_kept=min_kept)
    net = BiSeNet(C, True, crit, None, SyncBatchNorm)
    net.load_state_dict(ref.state_dict())
    net = DistributedDataParallel(net.to(cuda), compute_dtype=compute_dtype)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, S, S, generator=g)
    y = torch.randint(0, C, (B, S, S), generator=g)
    y[:, :8] = 255
    return ref, net, crit, x, y, min_kept


@pytest.fixture(scope="module")
def oracle_run():
    """One CPU forward+backward of the oracle network, shared by the fp32 and bf16 tests."""
    cache = {}

    def get(cuda):
        if not cache:
            torch.set_num_threads(min(torch.get_num_threads(), 64))
            ref, _, _, x, y, _ = _nets(cuda, torch.float32)
            ref.train()
            with torch.no_grad():
                logits = [t.clone() for t in ref.logits(x)]             # train-mode BN statistics of this batch
            ref2, _, _, _, _, _ = _nets(cuda, torch.float32)             # fresh running stats for the loss pass
            loss = ref2(x, y)
            loss.backward()
            cache.update(logits=logits, loss=loss.item(), grads={n: p.grad.clone() for n, p in ref2.named_parameters()})
        return cache
    return get


def _stock_logits(cuda, oracle_state, x, dtype):
    """The same network on stock PyTorch-ROCm ops (nn.BatchNorm2d, MIOpen, ATen upsample), same weights."""
    from torchseg_amd import workloads
    from torchseg_amd.ddp import apply_channels_last
    from torchseg_amd.workloads.bisenet import BiSeNet
    workloads.NATIVE_FUSIONS = False
    try:
        m = BiSeNet(C, True, None, None, nn.BatchNorm2d)
        m.load_state_dict(oracle_state)
        m = m.to(cuda).train()
        apply_channels_last(m)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            return [t.float().cpu() for t in m.logits(x.to(cuda))]
    finally:
        workloads.NATIVE_FUSIONS = True


def test_fp32_logits_loss_and_kept_mask_at_1024(cuda, oracle_run):
    from oracle.ohem_ref import ohem_select
    from torchseg_amd import kernels as K
    o = oracle_run(cuda)
    ref, net, crit, x, y, min_kept = _nets(cuda, torch.float32)
    net.train()
    xd, yd = x.to(cuda), y.to(cuda)
    with torch.no_grad():
        logits = net.module.logits(xd)
    stock = _stock_logits(cuda, {k: v for k, v in ref.state_dict().items() if "criterion" not in k}, x, torch.float32)
    kp = K.provider()
    for h, (got, want, stk) in enumerate(zip(logits, o["logits"], stock)):
        assert tuple(got.shape) == (B, C, S, S)
        scale = max(1.0, want.abs().max().item())
        err = (got.cpu() - want).abs().max().item()
        floor = (stk - want).abs().max().item()
        print("head %d: logits max |ours - cpu| %.2e, max |stock torch - cpu| %.2e, scale %.2f" % (h, err, floor, scale))
        # round 4: north_star's 1e-4 holds END TO END and in ABSOLUTE terms (measured 6.5-8.0e-5 = the CPU path's own
        # distance from the float64 truth, tools/diag_fp64_truth.py).  What kept rounds 1-3 at 1.1-1.6e-3 was the fp32
        # sum / square-sum formulation of the BatchNorm statistics (csrc/bn.hip RedAcc: fp64 accumulators, hi / lo partial
        # rows for fp32 tensors) amplified ~300x by the batch-2 BatchNorm of the global-context branch; stock
        # PyTorch-ROCm modules (`floor`) still carry it.
        assert err <= 1e-4 * scale, (h, err, floor, scale)      # 1e-4 of the logit scale (as for the loss); measured: < 1e-4 absolute
        # OHEM selection on OUR logits vs the reference selection on the ORACLE's logits
        _, nll, _, sel = kp.ohem_fwd(got.contiguous(), yd, 255, 0.7, min_kept, None)
        sel = sel.cpu()
        thr = sel[0:1].view(torch.float32).item()
        valid = (y != 255)
        kept = valid & (kp.ohem_target_prob(nll, yd, C, 255).cpu().view(B, S, S) <= thr)
        _, info = ohem_select(want, y, 255, 0.7, min_kept)
        assert int(sel[3]) == info["branch"] and int(sel[2]) == info["num_valid"]
        mp = info["mask_prob"].view(B, S, S)
        # The band is the one the MEASURED logit error of this head implies (round 2 used a fixed 8e-3): logits within
        # `err` of the oracle's move log p_t by at most 2 err (target logit and log-sum-exp by err each), i.e. p_t by a
        # relative exp(2 err) - 1; on the k-th-value branch the threshold moves by as much again.  Pixels that close to
        # the threshold may fall on either side; everything else must agree exactly.
        band = (2.0 if info["branch"] == 0 else 4.0) * 1.05 * err + 4e-7
        near = (mp - info["threshold"]).abs() <= band * max(info["threshold"], 1e-30)
        print("head %d: kept-mask exemption band %.2e relative (%d pixels of %d)" % (h, band, int(near.sum()), B * S * S))
        assert torch.equal(kept[~near], info["kept"][~near]), h
        assert abs(int(sel[1]) - info["n_kept"]) <= int(near.sum())
        assert int(near.sum()) <= 0.004 * B * S * S
    loss = net(xd, yd)
    loss.backward()
    assert abs(loss.item() - o["loss"]) <= 1e-4 * max(1.0, abs(o["loss"])), (loss.item(), o["loss"])
    num = den = 0.0
    for n, p in net.module.named_parameters():
        d = p.grad.cpu().double() - o["grads"][n].double()
        num += float((d * d).sum())
        den += float((o["grads"][n].double() ** 2).sum())
    rel = (num / den) ** 0.5
    print("headline fp32: loss %.6f (oracle %.6f), grad rel-L2 %.2e" % (loss.item(), o["loss"], rel))
    assert rel <= 1e-2, rel      # measured 5.7e-3: the same MIOpen-vs-CPU convolution noise as in the logits, through backward


def test_fp32_gradients_per_parameter_against_float64(cuda):
    """ABSOLUTE bars for the fp32 gradients, against the float64 evaluation of the oracle network —
    the fp32 CPU path cannot serve as this reference: it sits 5e-2 .. 9e-2 from float64 at its worst parameter (torch's CPU
    BatchNorm backward accumulates in fp32 and cancels; profiles/r05_smoke_grads_vs_float64.txt), which is what round 4's
    `worst per-param 5.37e-02` in smoke() was.

    * all parameters together: relative L2 <= 5e-5 (measured 6.9e-6; the fp32 CPU path: 3e-3);
    * every parameter: max |d| / max |g| <= 2e-2; the parameters above 1e-4 are at most one branch's worth (<= 12) and all
      sit in ONE branch of the network.  An fp32 evaluation cannot promise more per parameter: ONE ReLU whose argument is
      7e-8 in float64 and 0 in fp32 (the output of spatial_path.conv_1x1 at this seed, found with
      tools/r5/debug_ffm_grad.py: 1 flip among 524 288 activations) moves the 11 SpatialPath parameters behind it by
      1.2e-3 .. 7.8e-3; every other parameter is below 1e-5.  smoke() holds ALL parameters to 1e-4 at a configuration
      without such a flip."""
    from oracle.ohem_ref import ProbOhemCrossEntropy2d as OracleOhem
    from torchseg_amd.ddp import DistributedDataParallel
    from torchseg_amd.losses import ProbOhemCrossEntropy2d
    from torchseg_amd.syncbn import SyncBatchNorm
    from torchseg_amd.workloads.bisenet import BiSeNet
    b, s = 4, 256            # batch 4: the global-context BatchNorm over [B, 128, 1, 1] is ill-conditioned at 2 values per channel
    min_kept = b * s * s // 16
    torch.manual_seed(12345)
    ref32 = BiSeNet(C, True, OracleOhem(255, thresh=0.7, min_kept=min_kept), None, nn.BatchNorm2d)
    ref64 = BiSeNet(C, True, OracleOhem(255, thresh=0.7, min_kept=min_kept), None, nn.BatchNorm2d)
    ref64.load_state_dict(ref32.state_dict())
    ref64 = ref64.double()
    net = BiSeNet(C, True, ProbOhemCrossEntropy2d(255, thresh=0.7, min_kept=min_kept), None, SyncBatchNorm)
    net.load_state_dict(ref32.state_dict())
    net = DistributedDataParallel(net.to(cuda), compute_dtype=torch.float32)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(b, 3, s, s, generator=g)
    y = torch.randint(0, C, (b, s, s), generator=g)
    y[:, :8] = 255
    ref32(x, y).backward()
    l64 = ref64(x.double(), y)
    l64.backward()
    loss = net(x.to(cuda), y.to(cuda))
    loss.backward()
    assert abs(loss.item() - l64.item()) <= 1e-5 * abs(l64.item()), (loss.item(), l64.item())

    def distance(grads):
        rows, num, den = [], 0.0, 0.0
        for (n, p), q in zip(grads, ref64.parameters()):
            d = p.double() - q.grad
            num += float((d * d).sum())
            den += float((q.grad ** 2).sum())
            rows.append((float(d.abs().max() / (q.grad.abs().max() + 1e-30)), n))
        return (num / den) ** 0.5, sorted(rows, reverse=True)
    rel, rows = distance([(n, p.grad.cpu()) for n, p in net.module.named_parameters()])
    rel32, rows32 = distance([(n, p.grad) for n, p in ref32.named_parameters()])
    above = [r for r in rows if r[0] > 1e-4]
    print("fp32 gradients vs float64: ours rel-L2 %.2e, worst %.2e (%s), %d of %d parameters above 1e-4; "
          "the fp32 CPU path rel-L2 %.2e, worst %.2e (%s)" % (rel, rows[0][0], rows[0][1], len(above), len(rows),
                                                              rel32, rows32[0][0], rows32[0][1]))
    assert rel <= 5e-5, rel
    assert rows[0][0] <= 2e-2 and len(above) <= 12, above
    assert len({n.split(".")[0] for _, n in above}) <= 1, above        # one flipped ReLU = one branch


def test_bf16_step_at_1024_tracks_the_oracle(cuda, oracle_run):
    """The dtype the bench runs (bf16 activations, fp32 statistics and loss).  A randomly initialised network's logits
    are small differences of large intermediate values, so in bf16 they sit ~20 % (relative RMS) from the fp32 oracle
    whichever kernels compute them; the bar is therefore stated against the oracle network itself run under CPU bf16
    autocast (ours may be at most 1.5x as far from the fp32 oracle: we also store the BN outputs in bf16), plus absolute
    bar on the loss (2e-2) and the gradients held to the same CPU-bf16 floor."""
    o = oracle_run(cuda)
    ref, net, crit, x, y, _ = _nets(cuda, torch.bfloat16)
    net.train()
    xd, yd = x.to(cuda), y.to(cuda)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        logits = net.module.logits(xd)
    # the bf16 floor from an INDEPENDENT implementation: the same oracle network under CPU bf16 autocast (stock
    # PyTorch-ROCm bf16 modules segfault inside the framework at this shape on the test box, tools/diag_fp32_logits.py)
    ref.train()
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        stock = [t.float() for t in ref.logits(x)]
    ref.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ref(x, y).backward()                                 # the same floor for the gradients
    g_floor = {n: p.grad.clone() for n, p in ref.named_parameters()}

    def rel_rms(a, b):
        return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    for h, (got, want, stk) in enumerate(zip(logits, o["logits"], stock)):
        assert got.dtype == torch.bfloat16
        ours, floor = rel_rms(got.float().cpu(), want), rel_rms(stk, want)
        print("head %d: bf16 logits rel-RMS vs fp32 oracle: ours %.3f, CPU bf16 autocast %.3f" % (h, ours, floor))
        assert ours <= 1.5 * floor + 0.02, (h, ours, floor)
    loss = net(xd, yd)
    loss.backward()
    assert abs(loss.item() - o["loss"]) <= 2e-2 * max(1.0, abs(o["loss"])), (loss.item(), o["loss"])
    num = den = 0.0
    for n, p in net.module.named_parameters():
        d = p.grad.cpu().double() - o["grads"][n].double()
        num += float((d * d).sum())
        den += float((o["grads"][n].double() ** 2).sum())
    fnum = sum(float(((g_floor[n].double() - o["grads"][n].double()) ** 2).sum()) for n in g_floor)
    ours, floor = (num / den) ** 0.5, (fnum / den) ** 0.5
    print("headline bf16: loss %.6f (oracle %.6f), grad rel-L2 vs fp32 oracle: ours %.2e, CPU bf16 autocast %.2e"
          % (loss.item(), o["loss"], ours, floor))
    # batch 2 puts TWO values per channel into the three 1x1-map BN layers (x_hat = +-1: their input gradient is a
    # difference of rounding errors), so in bf16 the gradient of this configuration is dominated by noise whichever
    # kernels compute it; what can be asserted is that ours is no noisier than the CPU's own bf16 arithmetic
    assert ours <= 1.5 * floor + 0.05, (ours, floor)


# ---- index-width guards: the bench's largest BN activations --------------------------------------------------
@pytest.mark.parametrize("shape", [(16, 64, 512, 512), (16, 128, 128, 128), (16, 64, 256, 256)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True)])
def test_bn_kernels_at_bench_sizes_vs_torch_fp32(cuda, shape, relu, res):
    """16 x 64 x 512 x 512 is 2^28 elements: a 32-bit index or grid-limit slip in any BN kernel shows up here and
    nowhere in the small-shape tests.  Reference: torch's fp32 BatchNorm2d (+ add + ReLU) on the same device, fed the
    same bf16-rounded values."""
    from torchseg_amd.syncbn import SyncBatchNorm
    N, Cc, H, W = shape
    g = torch.Generator(device=cuda).manual_seed(N + H)
    x = (torch.randn(shape, generator=g, device=cuda) * 1.5 + 0.3).bfloat16().contiguous(memory_format=torch.channels_last)
    r = torch.randn(shape, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last) if res else None
    dy = torch.randn(shape, generator=g, device=cuda).bfloat16().contiguous(memory_format=torch.channels_last)
    bn = SyncBatchNorm(Cc).to(cuda)
    ref = nn.BatchNorm2d(Cc).to(cuda)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.uniform_(-0.5, 0.5, generator=g)
        ref.weight.copy_(bn.weight); ref.bias.copy_(bn.bias)
    xd = x.clone().requires_grad_(True)
    rd = r.clone().requires_grad_(True) if res else None
    y = bn(xd, residual=rd, relu=relu)
    y.backward(dy)
    xr = x.float().requires_grad_(True)
    rr = r.float().requires_grad_(True) if res else None
    yr = ref(xr)
    if res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy.float())
    torch.testing.assert_close(bn.running_mean, ref.running_mean, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(bn.running_var, ref.running_var, rtol=1e-4, atol=1e-5)
    assert (y.float() - yr).abs().max().item() <= 2 ** -7 * max(1.0, yr.abs().max().item())
    # the bf16 ReLU mask can differ from the fp32 one only where y rounds to 0; compare gradients in relative L2 and the
    # exact channel reductions (dgamma, dbeta) tightly -- an indexing slip moves these by O(1)
    def rel(a, b):
        return ((a.float() - b).pow(2).sum().sqrt() / b.pow(2).sum().sqrt()).item()
    assert rel(xd.grad, xr.grad) <= 1e-2
    if res:
        assert rel(rd.grad, rr.grad) <= 1e-2
    assert rel(bn.weight.grad, ref.weight.grad) <= 2e-3
    assert rel(bn.bias.grad, ref.bias.grad) <= 2e-3
    # spot rows at the far end of the tensor (beyond 2^31 bytes from the base for the 512^2 case)
    assert (y[-1, :, -1, -8:].float() - yr[-1, :, -1, -8:]).abs().max().item() <= 2 ** -7 * max(1.0, yr.abs().max().item())


# ---- multi-step trajectory: is the eager step ordered correctly? ----------------------------------------------
def _trajectory(cuda, hip, fused_sgd, dtype, steps=10, batch=4, size=256):
    import bench
    from oracle.ohem_ref import ProbOhemCrossEntropy2d as OracleOhem
    from torchseg_amd import workloads
    from torchseg_amd.ddp import DistributedDataParallel, apply_channels_last
    from torchseg_amd.losses import ProbOhemCrossEntropy2d
    from torchseg_amd.syncbn import SyncBatchNorm
    workloads.ensure_furnace_on_path()
    from engine.lr_policy import PolyLR
    workloads.NATIVE_FUSIONS = hip
    try:
        if hip:
            model, opt, base_lr = bench.build_model(cuda, batch, size, ProbOhemCrossEntropy2d, SyncBatchNorm,
                                                    fused_sgd=fused_sgd)
            model = DistributedDataParallel(model, compute_dtype=dtype)
        else:
            model, opt, base_lr = bench.build_model(cuda, batch, size, OracleOhem, nn.BatchNorm2d)
            apply_channels_last(model)
        model.train()
        imgs, gts = bench.synthetic_batch(cuda, batch, size)
        pol = PolyLR(base_lr, 0.9, 100)
        losses = []
        for it in range(steps):
            bench.set_lr(opt, pol, it)
            opt.zero_grad()
            if hip:
                loss = model(imgs, gts)
            else:
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
                    loss = model(imgs, gts)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return losses
    finally:
        workloads.NATIVE_FUSIONS = True


def test_ten_step_trajectory_matches_stock_torch(cuda):
    """Same seed, same inputs, 10 optimizer steps with the reference's poly schedule: the HIP path (SyncBN, OHEM,
    upsample, stems, weight-gradient kernels, FusedSGD or torch.optim.SGD) against stock PyTorch-ROCm modules on the same
    GPU (nn.BatchNorm2d, ATen upsample, the loss_opr.py restatement, torch.optim.SGD).  A stream-ordering bug between the
    gradient buffers and the optimizer (the suspected cause of the hipGraph divergence) would make the
    eager trajectories drift apart.  At batch 4 the trajectory amplifies rounding differences by ~5x per step for the first
    steps (the stock run itself is not reproducible: its atomics-based kernels move the step-3 loss by 1e-3 between two
    runs, and our two optimizers differ by 1.2e-3 there), so fp32 must agree to 5e-3 over the 10 steps — an ordering bug
    shows up as 1e-1 — and bf16 within bf16 noise of the fp32 stock run."""
    stock = _trajectory(cuda, False, False, torch.float32)
    for fused in (True, False):
        ours = _trajectory(cuda, True, fused, torch.float32)
        print("fp32 fused=%s" % fused, ["%.4f" % v for v in ours], ["%.4f" % v for v in stock])
        assert np.allclose(ours, stock, rtol=5e-3, atol=0), (fused, ours, stock)
    ours_bf16 = _trajectory(cuda, True, True, torch.bfloat16)
    stock_bf16 = _trajectory(cuda, False, False, torch.bfloat16)
    print("bf16", ["%.4f" % v for v in ours_bf16], ["%.4f" % v for v in stock_bf16])
    assert np.allclose(ours_bf16, stock, rtol=6e-2, atol=0), (ours_bf16, stock)
    # our bf16 run must be no further from the fp32 reference than stock bf16 autocast is (x2 slack)
    d_ours = np.abs(np.array(ours_bf16) - np.array(stock)).max()
    d_stock = np.abs(np.array(stock_bf16) - np.array(stock)).max()
    assert d_ours <= 2.0 * d_stock + 2e-2, (d_ours, d_stock)
    assert ours_bf16[-1] < ours_bf16[0] and stock[-1] < stock[0]                  # and it trains


def test_batch16_selection_against_the_oracle_at_the_real_min_kept(cuda):
    """the headline fp32 test runs batch 2, so its k-th order statistic is taken over 2 M candidates;
    BASELINE config 2 selects among 16.8 M with min_kept = 1 048 576 (train.py:48-49).  Here the ORACLE's selection
    (loss_opr.py:68-98 restated, run on the host) and the HIP kernels see the same 16 x 19 x 1024 x 1024 fp32 logits — no
    convolution in between, so what is compared is the selection alone, in the k-th-value branch:
      * the device threshold is bit-equal to torch.sort(p_device)[k-1] (the reference's own statement on the device's
        probabilities) and within 2 ulp of the oracle's threshold (two devices' expf);
      * the kept INDICES equal the oracle's everywhere except pixels whose probability lies within 2 ulp of the threshold
        (counted and bounded);
      * the loss is within 1e-4."""
    from oracle import ohem_ref
    from torchseg_amd import kernels as K
    kp = K.provider()
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    B, C, S = 16, 19, 1024
    k = B * S * S // 16
    g = torch.Generator().manual_seed(1)
    t = torch.randint(0, C, (B, S, S), generator=g)
    t[:, :8] = 255
    t2 = t.clone()
    t2[t2 == 255] = 0
    flip = torch.rand(t.shape, generator=g) < 0.03          # 3 % wrong labels: fewer than P / 16 pixels have p <= 0.7
    t2[flip] = torch.randint(0, C, (int(flip.sum()),), generator=g)
    pred = torch.randn(B, C, S, S, generator=g)
    pred.scatter_add_(1, t2[:, None], torch.full((B, 1, S, S), 8.0))
    del t2, flip
    with torch.no_grad():
        loss_ref, info = ohem_ref.ohem_cross_entropy(pred, t, 255, 0.7, k, None, return_info=True)
    assert info["branch"] == 1 and info["threshold"] > 0.7
    td = t.to(cuda)
    loss, nll, lse, sel = kp.ohem_fwd(pred.to(cuda), td, 255, 0.7, k, None)
    sel = sel.cpu()
    assert int(sel[3]) == 1 and int(sel[2]) == info["num_valid"]
    p_dev = kp.ohem_target_prob(nll, td, C, 255).cpu()
    thr_dev = sel[0:1].view(torch.float32).item()
    # the reference's statement on the device's own probabilities: bit-exact
    assert int(sel[0]) == torch.sort(p_dev)[0][k - 1].view(torch.int32).item()
    thr = info["threshold"]
    assert abs(thr_dev - thr) <= 4e-7 * thr, (thr_dev, thr)
    valid = t.view(-1) != 255
    kept_dev = valid & (p_dev <= thr_dev)
    assert int(sel[1]) == int(kept_dev.sum())
    mp = info["mask_prob"]
    near = (mp - thr).abs() <= 4e-7 * thr
    kept_ref = info["kept"].view(-1)
    assert torch.equal(kept_dev[~near], kept_ref[~near])
    n_near = int(near.sum())
    print("batch 16: threshold %.9g (oracle %.9g), kept %d (oracle %d), %d pixels within 2 ulp of the threshold"
          % (thr_dev, thr, int(sel[1]), info["n_kept"], n_near))
    assert n_near <= 64 and abs(int(sel[1]) - info["n_kept"]) <= n_near
    assert abs(loss.item() - float(loss_ref)) <= 1e-4 * max(1.0, abs(float(loss_ref)))
