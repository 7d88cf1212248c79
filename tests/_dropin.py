"""Helper: stage an unchanged reference experiment dir in a temp tree whose path
contains 'TorchSeg' (config.py:23-26 needs that), with OUR furnace/ in place of
the reference's, and run a driver script inside it (subprocess: the reference's
`config` / `network` module names collide across families)."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tools"))
import stage_reference as _sr  # noqa: E402


def have_reference():
    return os.path.isdir(os.path.join(REF, "model"))


def have_staged_reference():
    """The reference checkout, or the archive of its experiment files the build container packed for the GPU box
    (oracle/_ref/reference_models.tar.gz, tools/stage_reference.py)."""
    return _sr.available()


def stage(tmp_path, family, exp, files=("config.py", "network.py")):
    if have_reference():
        base = os.path.join(str(tmp_path), "TorchSeg")
        exp_dir = os.path.join(base, "model", family, exp)
        os.makedirs(exp_dir, exist_ok=True)
        for f in files:
            shutil.copy(os.path.join(REF, "model", family, exp, f), exp_dir)   # test-time copy only, never committed
        link = os.path.join(base, "furnace")
        if not os.path.exists(link):
            os.symlink(os.path.join(ROOT, "torchseg_amd", "furnace"), link)
        return exp_dir
    return _sr.stage(tmp_path, family, exp, files)


def run_in(exp_dir, script, timeout=600):
    env = dict(os.environ)
    paths = [ROOT, os.path.join(ROOT, "torchseg_amd", "shims")]
    import importlib.util
    if importlib.util.find_spec("cv2") is None:                 # the stand-in must never shadow a real OpenCV
        paths.append(os.path.join(ROOT, "torchseg_amd", "shims_optional"))
    env["PYTHONPATH"] = os.pathsep.join(paths)
    r = subprocess.run([sys.executable, "-c", script], cwd=exp_dir, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return r.stdout
