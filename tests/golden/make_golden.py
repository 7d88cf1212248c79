"""Generate golden vectors from the REFERENCE's own Python (run in the build
container only: needs /root/reference).  Output: tests/golden/*.npz (committed).

    python tests/golden/make_golden.py

The only edit applied to reference source (in memory, nothing is copied into
this repo) is `1 - valid_mask` -> `~valid_mask` in loss_opr.py:81,95, which
torch >= 1.2 requires and which does not change results.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_loss_opr():
    sys.path.insert(0, os.path.join(REF, "furnace"))
    import utils.pyt_utils  # noqa: F401  (must precede engine.logger: circular import in the reference)
    src = open(os.path.join(REF, "furnace/seg_opr/loss_opr.py")).read()
    assert src.count("1 - valid_mask") == 2
    src = src.replace("1 - valid_mask", "~valid_mask")
    mod = types.ModuleType("ref_loss_opr")
    exec(compile(src, "ref_loss_opr.py", "exec"), mod.__dict__)
    return mod


def ohem_cases():
    g = torch.Generator().manual_seed(1234)
    B, C, H, W = 2, 19, 24, 24
    P = B * H * W

    def labels(ignore_rows=2, all_ignored=False):
        t = torch.randint(0, C, (B, H, W), generator=g)
        t[:, :ignore_rows] = 255
        if all_ignored:
            t[:] = 255
        return t

    def confident(t, sharp=8.0, flip=0.1):
        t2 = t.clone()
        t2[t2 == 255] = 0
        flipm = torch.rand(t.shape, generator=g) < flip
        t2[flipm] = torch.randint(0, C, (int(flipm.sum()),), generator=g)
        return sharp * torch.nn.functional.one_hot(t2, C).permute(0, 3, 1, 2).float() + torch.randn(B, C, H, W, generator=g)

    cases = {}
    t = labels()
    cases["random_thr"] = dict(pred=torch.randn(B, C, H, W, generator=g), target=t, thresh=0.7, min_kept=P // 16, use_weight=False)
    t = labels()
    cases["confident_kth"] = dict(pred=confident(t), target=t, thresh=0.7, min_kept=P // 2, use_weight=False)
    t = labels()
    cases["confident_thr"] = dict(pred=confident(t, flip=0.4), target=t, thresh=0.7, min_kept=P // 16, use_weight=False)
    t = labels(ignore_rows=20)
    cases["minkept_gt_valid"] = dict(pred=torch.randn(B, C, H, W, generator=g), target=t, thresh=0.7, min_kept=P // 2, use_weight=False)
    t = labels(all_ignored=True)
    cases["all_ignored"] = dict(pred=torch.randn(B, C, H, W, generator=g), target=t, thresh=0.7, min_kept=0, use_weight=False)
    t = labels()
    cases["minkept_zero"] = dict(pred=confident(t), target=t, thresh=0.7, min_kept=0, use_weight=False)
    t = labels()
    cases["weighted_kth"] = dict(pred=confident(t), target=t, thresh=0.6, min_kept=256, use_weight=True)
    t = labels(ignore_rows=0)
    cases["kept_all"] = dict(pred=confident(t), target=t, thresh=0.7, min_kept=P, use_weight=False)
    return cases


def main():
    ref = load_reference_loss_opr()
    out = {}
    for name, c in ohem_cases().items():
        crit = ref.ProbOhemCrossEntropy2d(ignore_label=255, thresh=c["thresh"], min_kept=c["min_kept"],
                                          use_weight=c["use_weight"])
        pred = c["pred"].clone().requires_grad_(True)
        loss = crit(pred, c["target"].clone())
        if torch.isfinite(loss):
            loss.backward()
            grad = pred.grad
        else:
            grad = torch.zeros_like(pred)
        out[name + "/pred"] = c["pred"].numpy()
        out[name + "/target"] = c["target"].numpy().astype(np.uint8)
        out[name + "/cfg"] = np.array([c["thresh"], c["min_kept"], float(c["use_weight"])], dtype=np.float64)
        out[name + "/loss"] = np.array(loss.item(), dtype=np.float64)
        out[name + "/grad"] = grad.numpy()
        print(name, "loss", loss.item(), "kept", int((grad.abs().sum(1) > 0).sum()))
    np.savez_compressed(os.path.join(HERE, "ohem_golden.npz"), **out)

    out = {}
    g = torch.Generator().manual_seed(99)
    for name, (gamma, alpha) in {"dfn_default": (2.0, 0.25), "dfn_city": (2.0, 0.1), "g15": (1.5, 0.4)}.items():
        crit = ref.SigmoidFocalLoss(ignore_label=255, gamma=gamma, alpha=alpha)
        pred = (torch.randn(2, 1, 20, 28, generator=g) * 2).requires_grad_(True)
        tgt = torch.randint(0, 2, (2, 20, 28), generator=g)
        tgt[torch.rand(2, 20, 28, generator=g) < 0.15] = 255
        loss = crit(pred, tgt)
        loss.backward()
        out[name + "/pred"] = pred.detach().numpy()
        out[name + "/target"] = tgt.numpy().astype(np.uint8)
        out[name + "/cfg"] = np.array([gamma, alpha])
        out[name + "/loss"] = np.array(loss.item(), dtype=np.float64)
        out[name + "/grad"] = pred.grad.numpy()
        print(name, "focal", loss.item())
    np.savez_compressed(os.path.join(HERE, "focal_golden.npz"), **out)
    metric_golden()
    syncbn_golden()


def metric_golden():
    """hist_info / compute_score of the reference's furnace/seg_opr/metric.py on seeded label maps."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_metric", os.path.join(REF, "furnace/seg_opr/metric.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.RandomState(7)
    out = {}
    for name, (n_cl, shape, ignore, p_right) in {"city19": (19, (2, 40, 56), 255, 0.7), "ade150": (150, (1, 64, 48), -1, 0.4),
                                                 "tiny2": (2, (1, 5, 7), 255, 0.5), "all_ignored": (19, (1, 8, 8), 255, 0.5)}.items():
        gt = rng.randint(0, n_cl, size=shape).astype(np.int64)
        pred = np.where(rng.rand(*shape) < p_right, gt, rng.randint(0, n_cl, size=shape)).astype(np.int64)
        gt[rng.rand(*shape) < 0.1] = ignore
        if name == "all_ignored":
            gt[:] = ignore
        hist, labeled, correct = ref.hist_info(n_cl, pred, gt)
        iu, miu, miu_nb, acc = ref.compute_score(hist, correct, labeled)
        out[name + "/n_cl"] = np.array(n_cl)
        out[name + "/pred"] = pred.astype(np.int16)
        out[name + "/gt"] = gt.astype(np.int16)
        out[name + "/hist"] = hist.astype(np.int64)
        out[name + "/counts"] = np.array([labeled, correct], dtype=np.int64)
        out[name + "/iu"] = np.asarray(iu, dtype=np.float64)
        out[name + "/scores"] = np.array([miu, miu_nb, acc], dtype=np.float64)
        print(name, "labeled", labeled, "correct", correct, "mIoU", miu)
    np.savez_compressed(os.path.join(HERE, "metric_golden.npz"), **out)


def syncbn_golden():
    """_SyncBatchNorm._compute_mean_std (furnace/legacy/sync_bn/syncbn.py:86-98) executed verbatim: the one piece of the
    reference's SyncBN arithmetic that is plain Python (mean, inv_std, running statistics with the UNBIASED variance from
    the cross-GPU sum / square-sum / count).  The method is cut out of the file by its indentation and bound to a bare
    object, because the module itself cannot be imported (it pulls in the CUDA-only extension, functions.py:14-16)."""
    import textwrap
    lines = open(os.path.join(REF, "furnace/legacy/sync_bn/syncbn.py")).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.strip().startswith("def _compute_mean_std"))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].strip() and not lines[i].startswith("        "))
    ns = {}
    exec(compile(textwrap.dedent("\n".join(lines[start:end])), "ref_syncbn_compute_mean_std.py", "exec"), ns)
    fn = ns["_compute_mean_std"]

    class Bare(object):
        pass

    rng = np.random.RandomState(11)
    out = {}
    for name, (C, ranks, shape, eps, momentum) in {"c64_1rank": (64, 1, (16, 12, 10), 1e-5, 0.1),
                                                    "c128_3ranks": (128, 3, (4, 7, 5), 1e-5, 0.1),
                                                    "c19_2ranks_1x1": (19, 2, (2, 1, 1), 1e-3, 0.01),
                                                    "c8_8ranks": (8, 8, (2, 9, 3), 1e-5, 0.9)}.items():
        xs = [rng.standard_normal((shape[0] + r, C) + shape[1:]) * (1.0 + 0.3 * r) + 0.2 * r for r in range(ranks)]
        ax = (0, 2, 3)
        sum_ = torch.from_numpy(sum(x.sum(ax) for x in xs)).float()          # what sum_square + ReduceAddCoalesced deliver
        ssum = torch.from_numpy(sum((x * x).sum(ax) for x in xs)).float()
        size = int(sum(x.size // C for x in xs))
        mod = Bare()
        mod.momentum, mod.eps = momentum, eps
        mod.running_mean = torch.from_numpy(rng.standard_normal(C)).float()
        mod.running_var = torch.from_numpy(rng.rand(C) + 0.5).float()
        rm0, rv0 = mod.running_mean.clone(), mod.running_var.clone()
        mean, inv_std = fn(mod, sum_, ssum, size)
        out[name + "/cfg"] = np.array([C, ranks, size, eps, momentum], dtype=np.float64)
        for r, x in enumerate(xs):
            out[name + "/x%d" % r] = x.astype(np.float32)
        out[name + "/sum"] = sum_.numpy(); out[name + "/ssum"] = ssum.numpy()
        out[name + "/rm0"] = rm0.numpy(); out[name + "/rv0"] = rv0.numpy()
        out[name + "/mean"] = mean.numpy(); out[name + "/inv_std"] = inv_std.numpy()
        out[name + "/rm1"] = mod.running_mean.numpy(); out[name + "/rv1"] = mod.running_var.numpy()
        print(name, "size", size, "mean[0]", float(mean[0]), "inv_std[0]", float(inv_std[0]))
    np.savez_compressed(os.path.join(HERE, "syncbn_golden.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "syncbn":
        syncbn_golden()
    else:
        main()
