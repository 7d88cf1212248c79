"""CPU plumbing (BASELINE config 1): the reference's UNCHANGED network.py files
import against our furnace/ + shims and run one forward/backward; our BiSeNet
workload builder is the same network (same init under a seed, same loss).
Skipped where /root/reference is absent (the GPU box)."""
import json
import os

import numpy as np
import pytest

from _dropin import ROOT, have_reference, run_in, stage

pytestmark = pytest.mark.skipif(not have_reference(), reason="reference checkout not present")

_BISENET = r'''
import json, sys, torch, torch.nn as nn
from config import config            # unchanged reference config.py (adds furnace/ to sys.path)
from network import BiSeNet          # unchanged reference network.py
from oracle.ohem_ref import ProbOhemCrossEntropy2d
from torchseg_amd.workloads.bisenet import BiSeNet as Ours
def build(cls):
    torch.manual_seed(config.seed)
    crit = ProbOhemCrossEntropy2d(ignore_label=255, thresh=0.7, min_kept=2*96*96//16, use_weight=False)
    return cls(config.num_classes, is_training=True, criterion=crit, pretrained_model=None, norm_layer=nn.BatchNorm2d)
ref, ours = build(BiSeNet), build(Ours)
sd_r, sd_o = ref.state_dict(), ours.state_dict()
assert list(sd_r.keys()) == list(sd_o.keys())
assert all(torch.equal(sd_r[k], sd_o[k]) for k in sd_r), "seeded init differs"
g = torch.Generator().manual_seed(0)
x = torch.randn(2, 3, 96, 96, generator=g); y = torch.randint(0, 19, (2, 96, 96), generator=g); y[:, :8] = 255
lr, lo = ref(x, y), ours(x, y)
lr.backward(); lo.backward()
gr = torch.cat([p.grad.reshape(-1) for p in ref.parameters()]); go = torch.cat([p.grad.reshape(-1) for p in ours.parameters()])
print(json.dumps(dict(loss_ref=lr.item(), loss_ours=lo.item(), gdiff=(gr-go).abs().max().item(), gmax=gr.abs().max().item(),
                      nparam=sum(p.numel() for p in ref.parameters()))))
'''


def test_bisenet_reference_network_runs_and_matches_workload(tmp_path):
    d = stage(tmp_path, "bisenet", "cityscapes.bisenet.R18")
    out = json.loads(run_in(d, _BISENET).strip().splitlines()[-1])
    assert out["nparam"] == 13494777
    assert abs(out["loss_ref"] - out["loss_ours"]) <= 1e-5 * abs(out["loss_ref"])
    assert out["gdiff"] <= 1e-4 * out["gmax"]


_FAMILY = r'''
import json, torch, torch.nn as nn
from config import config
import network
torch.manual_seed(1)
crit = nn.CrossEntropyLoss(reduction='mean', ignore_index=-1)
kind = "%s"
if kind == "dfn":
    from oracle.focal_ref import SigmoidFocalLoss
    model = network.DFN(config.num_classes, criterion=nn.CrossEntropyLoss(ignore_index=255),
                        aux_criterion=SigmoidFocalLoss(ignore_label=255, gamma=2.0, alpha=0.25),
                        alpha=config.aux_loss_alpha, pretrained_model=None, norm_layer=nn.BatchNorm2d)
    S = 64
    x = torch.randn(2, 3, S, S); y = torch.randint(0, config.num_classes, (2, S, S)); e = torch.randint(0, 2, (2, S, S))
    loss = model(x, y, e)
else:
    model = network.PSPNet(config.num_classes, criterion=crit, pretrained_model=None, norm_layer=nn.BatchNorm2d)
    S = 480 if kind == "psanet" else 96
    B = 1 if kind == "psanet" else 2
    x = torch.randn(B, 3, S, S); y = torch.randint(0, config.num_classes, (B, S, S))
    if kind == "psanet":
        model.eval()      # BN over a batch of 1 at 1x1 pooling is undefined in train mode; plumbing only
        for p in model.parameters(): p.requires_grad_(True)
    loss = model(x, y)
loss.backward()
nograd = [n for n, p in model.named_parameters() if p.grad is None]
print(json.dumps(dict(loss=loss.item(), nparam=sum(p.numel() for p in model.parameters()), nograd=len(nograd))))
'''


@pytest.mark.parametrize("family,exp,kind,nparam", [
    ("pspnet", "ade.pspnet.R50_v1c", "pspnet", None),
    ("dfn", "cityscapes.dfn.R101_v1c", "dfn", None),
    ("psanet", "ade.psanet.R50_v1c", "psanet", None),
])
def test_other_families_import_and_step(tmp_path, family, exp, kind, nparam):
    d = stage(tmp_path, family, exp)
    out = json.loads(run_in(d, _FAMILY % kind, timeout=900).strip().splitlines()[-1])
    assert np.isfinite(out["loss"])
    if kind == "dfn":
        assert out["nograd"] == 5      # statically unused params: DDP must tolerate them


_WORKLOAD_EQ = r'''
import json, torch, torch.nn as nn
from config import config
import network
kind = "%s"
def build(ref):
    torch.manual_seed(3)
    crit = nn.CrossEntropyLoss(reduction='mean', ignore_index=-1)
    if kind == "dfn":
        from oracle.focal_ref import SigmoidFocalLoss
        args = dict(criterion=nn.CrossEntropyLoss(ignore_index=255), aux_criterion=SigmoidFocalLoss(255, 2.0, 0.25),
                    alpha=config.aux_loss_alpha, pretrained_model=None, norm_layer=nn.BatchNorm2d)
        if ref:
            return network.DFN(config.num_classes, **args)
        from torchseg_amd.workloads.dfn import DFN
        return DFN(config.num_classes, **args)
    if ref:
        return network.PSPNet(config.num_classes, criterion=crit, pretrained_model=None, norm_layer=nn.BatchNorm2d)
    from torchseg_amd.workloads.pspnet import PSPNet, PSANet
    cls = PSPNet if kind == "pspnet" else PSANet
    return cls(config.num_classes, crit, None, nn.BatchNorm2d, depth=50)
ref, ours = build(True), build(False)
sr, so = ref.state_dict(), ours.state_dict()
assert list(sr.keys()) == list(so.keys()), [k for k in sr if k not in so][:5]
assert all(torch.equal(sr[k], so[k]) for k in sr), "seeded init differs"
ref.eval(); ours.eval()            # dropout off, running stats: deterministic comparison
g = torch.Generator().manual_seed(1)
if kind == "dfn":
    x = torch.randn(1, 3, 64, 64, generator=g); y = torch.randint(0, config.num_classes, (1, 64, 64), generator=g)
    e = torch.randint(0, 2, (1, 64, 64), generator=g)
    lr, lo = ref(x, y, e), ours(x, y, e)
else:
    S = 480 if kind == "psanet" else 96
    x = torch.randn(1, 3, S, S, generator=g); y = torch.randint(0, config.num_classes, (1, S, S), generator=g)
    lr, lo = ref(x, y), ours(x, y)
print(json.dumps(dict(loss_ref=lr.item(), loss_ours=lo.item(), nkeys=len(sr))))
'''


@pytest.mark.parametrize("family,exp,kind", [
    ("pspnet", "ade.pspnet.R50_v1c", "pspnet"),
    ("dfn", "cityscapes.dfn.R101_v1c", "dfn"),
    ("psanet", "ade.psanet.R50_v1c", "psanet"),
])
def test_workload_builders_equal_reference_networks(tmp_path, family, exp, kind):
    """torchseg_amd/workloads/{pspnet,dfn}.py are the reference networks: same state-dict keys,
    identical seeded init, identical loss."""
    d = stage(tmp_path, family, exp)
    out = json.loads(run_in(d, _WORKLOAD_EQ % kind, timeout=900).strip().splitlines()[-1])
    assert abs(out["loss_ref"] - out["loss_ours"]) <= 1e-5 * max(1.0, abs(out["loss_ref"])), out


_CKPT_COMMON = r'''
import sys, types, json, torch, torch.nn as nn
def make():
    torch.manual_seed(5)
    m = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 4, 1))
    return m
def groups(m):
    return [dict(params=[p for p in m.parameters() if p.dim() > 1], lr=0.05),
            dict(params=[p for p in m.parameters() if p.dim() <= 1], lr=0.5, weight_decay=0.0)]
def digest(m, opt):
    sd = opt.state_dict()
    return dict(params=[float(p.double().sum()) for p in m.parameters()],
                bufs=[float(b.double().sum()) for b in m.buffers()],
                mom=[float(sd['state'][k]['momentum_buffer'].double().sum()) for k in sorted(sd['state'])],
                mom_shapes=[list(sd['state'][k]['momentum_buffer'].shape) for k in sorted(sd['state'])],
                lrs=[g['lr'] for g in sd['param_groups']])
class Wrapped(nn.Module):            # what apex / our DDP wrapper looks like to the state dict: 'module.' prefix
    def __init__(self, m):
        super().__init__(); self.module = m
'''

_CKPT_REF_WRITE = _CKPT_COMMON + r'''
sys.path.insert(0, "/root/reference/furnace")
import utils.pyt_utils
from engine.engine import Engine, State
m = make(); opt = torch.optim.SGD(groups(m), lr=0.05, momentum=0.9, weight_decay=5e-4)
for _ in range(2):
    opt.zero_grad(); m(torch.randn(2, 3, 8, 8)).square().mean().backward(); opt.step()
st = State(); st.register(model=Wrapped(m), optimizer=opt); st.epoch = 3; st.iteration = 77
Engine.save_checkpoint(types.SimpleNamespace(state=st), sys.argv[1])
print(json.dumps(digest(m, opt)))
'''

_CKPT_OURS = _CKPT_COMMON + r'''
from engine.engine import Engine, State          # torchseg_amd/furnace (cwd-staged path)
from torchseg_amd.optim import FusedSGD
m = make()
for p in m.parameters():
    p.data.add_(1.0)                             # make sure the restore really overwrites
m[3].weight.data = m[3].weight.data.contiguous(memory_format=torch.channels_last)
opt = FusedSGD(groups(m), lr=0.01, momentum=0.9, weight_decay=5e-4)
st = State(); st.register(model=Wrapped(m), optimizer=opt)
eng = types.SimpleNamespace(state=st, continue_state_object=sys.argv[1], distributed=True)
Engine.restore_checkpoint(eng)
assert st.epoch == 4 and st.iteration == 77, (st.epoch, st.iteration)
Engine.save_checkpoint(eng, sys.argv[2])
print(json.dumps(digest(st.model.module, opt)))
'''

_CKPT_REF_READ = _CKPT_COMMON + r'''
sys.path.insert(0, "/root/reference/furnace")
import utils.pyt_utils
from engine.engine import Engine, State
m = make(); opt = torch.optim.SGD(groups(m), lr=0.01, momentum=0.9, weight_decay=5e-4)
st = State(); st.register(model=Wrapped(m), optimizer=opt)
eng = types.SimpleNamespace(state=st, continue_state_object=sys.argv[1], distributed=True)
Engine.restore_checkpoint(eng)
assert st.epoch == 5 and st.iteration == 77
opt.zero_grad(); m(torch.ones(1, 3, 8, 8)).square().mean().backward(); opt.step()    # the restored state is usable
print(json.dumps(digest(m, opt)))
'''


def test_checkpoints_interchange_with_reference_engine(tmp_path):
    """a checkpoint written by the reference Engine (torch SGD, 'module.'-stripped keys) restores
    into our Engine + FusedSGD, and the checkpoint our Engine writes restores into the reference's."""
    import subprocess
    import sys as _sys
    d = stage(tmp_path, "bisenet", "cityscapes.bisenet.R18")
    ref_ckpt, our_ckpt = str(tmp_path / "ref.pth"), str(tmp_path / "ours.pth")

    def run(script, *args, ours):
        env = dict(os.environ)
        env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "torchseg_amd", "shims")]
                                            + ([os.path.join(ROOT, "torchseg_amd", "furnace")] if ours else []))
        r = subprocess.run([_sys.executable, "-c", script, *args], cwd=d, env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-3000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    a = run(_CKPT_REF_WRITE, ref_ckpt, ours=False)
    b = run(_CKPT_OURS, ref_ckpt, our_ckpt, ours=True)
    for k in ("params", "bufs", "mom", "mom_shapes", "lrs"):
        assert a[k] == b[k], (k, a[k], b[k])
    c = run(_CKPT_REF_READ, our_ckpt, ours=False)
    assert c["mom_shapes"] == a["mom_shapes"] and c["lrs"] == a["lrs"]


_IMPORTS = r'''
import ast, importlib, json, sys
from config import config                         # puts <TorchSeg>/furnace on sys.path, as the scripts do first
missing = []
for script in ("train.py", "eval.py", "dataloader.py"):
    tree = ast.parse(open(script).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.level == 0:
            if node.module.split(".")[0] in ("tools",):      # tools/benchmark of the .speed variants: out of scope
                continue
            try:
                mod = importlib.import_module(node.module)
            except ImportError as e:
                if node.module.startswith("apex"):
                    raise
                missing.append("%s: import %s (%s)" % (script, node.module, e)); continue
            for a in node.names:
                if a.name != "*" and not hasattr(mod, a.name):
                    try:
                        importlib.import_module(node.module + "." + a.name)
                    except ImportError:
                        missing.append("%s: from %s import %s" % (script, node.module, a.name))
        elif isinstance(node, ast.Import):
            for a in node.names:
                try:
                    importlib.import_module(a.name)
                except ImportError as e:
                    missing.append("%s: import %s (%s)" % (script, a.name, e))
print(json.dumps(missing))
'''


@pytest.mark.parametrize("family,exp", [("bisenet", "cityscapes.bisenet.R18"), ("dfn", "cityscapes.dfn.R101_v1c"),
                                        ("pspnet", "ade.pspnet.R50_v1c"), ("psanet", "ade.psanet.R50_v1c")])
def test_every_import_of_the_unchanged_scripts_resolves(tmp_path, family, exp):
    """INTEGRATION.md's drop-in step must not end in an ImportError: every `import` / `from ... import name` of the
    UNCHANGED train.py, eval.py and dataloader.py of each family resolves against our furnace/ (+ the apex / easydict
    shims, + the cv2 stand-in only because this image has no OpenCV)."""
    files = [f for f in ("config.py", "network.py", "train.py", "eval.py", "dataloader.py")
             if os.path.exists(os.path.join("/root/reference/model", family, exp, f))]
    d = stage(tmp_path, family, exp, files=files)
    missing = json.loads(run_in(d, _IMPORTS).strip().splitlines()[-1])
    assert not missing, missing


_TRAINPRE = r'''
import json, random, numpy as np
from config import config
from dataloader import TrainPre                   # the UNCHANGED reference class, running on OUR utils.img_utils
sys_path_ok = True
rng = np.random.RandomState(0)
img = rng.randint(0, 256, size=(96, 160, 3)).astype(np.uint8)
gt = rng.randint(0, 19, size=(96, 160)).astype(np.uint8)
config.image_height, config.image_width = 64, 80
config.train_scale_array = [0.75, 1.0, 1.5]
pre = TrainPre(config.image_mean, config.image_std)
random.seed(3)
p_img, p_gt, extra = pre(img, gt)
from oracle import augment_ref as R
random.seed(3)
par = R.draw_params(img.shape[:2], config.train_scale_array, (64, 80))
w_img, w_gt = R.train_pre(img, gt, par, config.image_mean, config.image_std, (64, 80))
print(json.dumps(dict(shape=list(p_img.shape), gt_equal=bool(np.array_equal(p_gt, w_gt)),
                      frac_off=float((np.abs(p_img - w_img) > 1e-5).mean()), max_off=float(np.abs(p_img - w_img).max()))))
'''


def test_unchanged_trainpre_runs_on_our_img_utils_and_equals_the_oracle(tmp_path):
    """The reference's own TrainPre (bisenet dataloader.py:11-35), unchanged, on our utils.img_utils: same sample as the
    step-by-step oracle the GPU pipeline is tested against (labels equal; image equal up to uint8 .5 ties)."""
    d = stage(tmp_path, "bisenet", "cityscapes.bisenet.R18", files=("config.py", "network.py", "dataloader.py"))
    out = json.loads(run_in(d, _TRAINPRE).strip().splitlines()[-1])
    assert out["shape"] == [3, 64, 80] and out["gt_equal"]
    assert out["frac_off"] <= 1e-3 and out["max_off"] <= (1 / 255) / 0.224 + 1e-5


def test_dataset_tables_match_the_reference(tmp_path):
    d = stage(tmp_path, "bisenet", "cityscapes.bisenet.R18")
    script = r"""
import importlib.util, json, sys
from config import config
from datasets import Cityscapes, VOC
def ref(path, name):
    src = open(path).read().replace("from datasets.BaseDataset import BaseDataset", "BaseDataset = object")
    ns = {}
    exec(compile(src.split("if __name__")[0], path, "exec"), ns)
    return ns[name]
RC = ref("/root/reference/furnace/datasets/cityscapes/cityscapes.py", "Cityscapes")
RV = ref("/root/reference/furnace/datasets/voc/voc.py", "VOC")
print(json.dumps(dict(city=Cityscapes.get_class_colors() == RC.get_class_colors() and Cityscapes.get_class_names() == RC.get_class_names()
                           and Cityscapes.trans_labels == RC.trans_labels,
                      voc=VOC.get_class_colors() == RV.get_class_colors() and VOC.get_class_names() == RV.get_class_names())))
"""
    out = json.loads(run_in(d, script).strip().splitlines()[-1])
    assert out == {"city": True, "voc": True}
