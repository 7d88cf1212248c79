"""bench.py's cpu_baseline of kind "reference": tools/cpu_reference.py runs the REFERENCE'S OWN network.py / seg_oprs.py /
resnet.py / loss_opr.py (staged by tools/stage_reference.py) on the host.  Checked here: it runs from the checkout AND from
the archive the GPU box gets, imports the reference's modules (not ours), and its first loss equals the oracle port's
(same seed, same synthetic batch) — the two CPU legs time the same arithmetic."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import stage_reference  # noqa: E402

pytestmark = pytest.mark.skipif(not stage_reference.available(), reason="neither /root/reference nor the staged archive")


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_reference.py"), "--size", "128", "--batch", "2",
                        "--budget", "0.5", "--max-steps", "1", "--threads", "4", "--check"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_reference_leg_from_archive_and_checkout_equals_the_port():
    if stage_reference.have_reference():
        stage_reference.pack()
    recs = []
    if os.path.exists(stage_reference.ARCHIVE):
        recs.append(_run({"TSG_REFERENCE_DIR": "/nonexistent"}))           # what the GPU box does
        assert recs[-1]["source"].endswith("reference_models.tar.gz")
    if stage_reference.have_reference():
        recs.append(_run({}))
        assert recs[-1]["source"] == "reference checkout"
    assert recs
    for r in recs:
        assert r["kind"] == "reference" and r["params"] == 13494777 and r["value"] > 0     # 13.49 M parameters
    import torch.nn as nn
    sys.path.insert(0, ROOT)
    import bench
    from oracle.ohem_ref import ProbOhemCrossEntropy2d as OracleOhem
    from torchseg_amd.workloads import ensure_furnace_on_path
    ensure_furnace_on_path()
    torch.set_num_threads(4)
    model, opt, _ = bench.build_model(torch.device("cpu"), 2, 128, OracleOhem, nn.BatchNorm2d)
    model.train()
    imgs, gts = bench.synthetic_batch(torch.device("cpu"), 2, 128)
    loss = float(model(imgs, gts).item())
    for r in recs:
        assert abs(r["first_loss"] - loss) <= 1e-5 * abs(loss), (r["first_loss"], loss)
