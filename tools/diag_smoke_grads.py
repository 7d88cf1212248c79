"""Which parameter gradients of smoke()'s step are furthest from the CPU path, who is right (float64 evaluation of the same
network on the host), and which of our operator substitutions moves the figure (`smoke()` printed
`worst per-param 5.37e-02 (ffm.conv_1x1.conv.weight)` in the fp32 parity mode).

    python tools/diag_smoke_grads.py [--size 128 --batch 4]

Per configuration (default, then one TSG_* switch off at a time): the five worst parameters by max |d| / max |g_ref| and by
relative L2, against the fp32 CPU oracle AND against the float64 truth."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=4)
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    from oracle.ohem_ref import ProbOhemCrossEntropy2d as OracleOhem
    from torchseg_amd.ddp import DistributedDataParallel
    from torchseg_amd.losses import ProbOhemCrossEntropy2d
    from torchseg_amd.syncbn import SyncBatchNorm
    from torchseg_amd.workloads.bisenet import BiSeNet
    dev = torch.device("cuda:0")
    B, S = args.batch, args.size
    min_kept = B * S * S // 16
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, S, S, generator=g)
    y = torch.randint(0, 19, (B, S, S), generator=g)
    y[:, :8] = 255

    def oracle(dtype):
        torch.manual_seed(12345)
        ref = BiSeNet(19, True, OracleOhem(255, thresh=0.7, min_kept=min_kept), None, nn.BatchNorm2d)
        sd = {k: v.clone() for k, v in ref.state_dict().items()}
        ref = ref.to(dtype)
        loss = ref(x.to(dtype), y)
        loss.backward()
        return sd, float(loss), {n: p.grad.double() for n, p in ref.named_parameters()}

    sd, loss32, g32 = oracle(torch.float32)
    _, loss64, g64 = oracle(torch.float64)
    print("CPU fp32 loss %.7f, float64 loss %.7f" % (loss32, loss64))

    def report(tag, grads, ref, k=5):
        rows = []
        for n, q in ref.items():
            d = grads[n] - q
            rows.append((float(d.abs().max() / (q.abs().max() + 1e-30)), float(d.norm() / (q.norm() + 1e-30)), n,
                         float(q.abs().max())))
        num = sum(float(((grads[n] - q) ** 2).sum()) for n, q in ref.items())
        den = sum(float((q ** 2).sum()) for q in ref.values())
        print("  [%s] global rel-L2 %.3e; worst by max|d|/max|g|:" % (tag, (num / den) ** 0.5))
        for r in sorted(rows, reverse=True)[:k]:
            print("      %.3e  (rel-L2 %.3e, max|g| %.3e)  %s" % (r[0], r[1], r[3], r[2]))

    print("CPU fp32 oracle against the float64 truth:")
    report("cpu32 vs f64", g32, g64)

    switches = [None, "TSG_VEC_CONV", "TSG_CAT", "TSG_CLS_HEAD", "TSG_FUSE_HEAD", "TSG_ADAPTIVE_POOL", "TSG_SPLIT_BIAS",
                "TSG_STEM_CONV", "TSG_FP32_EXACT"]
    for sw in switches:
        for k in switches[1:]:
            os.environ.pop(k, None)
        if sw:
            os.environ[sw] = "0"
        torch.manual_seed(12345)
        crit = ProbOhemCrossEntropy2d(255, thresh=0.7, min_kept=min_kept)
        net = BiSeNet(19, True, crit, None, SyncBatchNorm)
        net.load_state_dict(sd)
        net = DistributedDataParallel(net.to(dev), compute_dtype=torch.float32)
        loss = net(x.to(dev), y.to(dev))
        loss.backward()
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().cpu().double() for n, p in net.module.named_parameters()}
        print("%s: GPU loss %.7f (kept %d)" % ("default" if sw is None else sw + "=0", float(loss),
                                               int(crit.last_selection[1])))
        report("gpu vs cpu32", grads, g32)
        report("gpu vs f64  ", grads, g64)


if __name__ == "__main__":
    main()
