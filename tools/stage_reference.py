"""Packs the reference's UNCHANGED experiment files
`model/<family>/<experiment>/{network,config}.py` into `oracle/_ref/reference_models.tar.gz` so that they can travel to
the GPU box, where /root/reference does not exist (the archive is git-ignored like the built `.so` files: nothing of the
reference enters the history; `.gpurunignore` does not list it, so it ships with the snapshot).

Why: VERDICT r4 item 1 — the path BASELINE.json's `north_star` names is the reference's own `network.py` behind our
furnace / apex surface; `bench.py --network reference`, `tests/test_dropin_gpu.py` and the `reference_network` record of
the default bench line unpack this archive into a scratch directory whose path contains `TorchSeg` (the reference's
config.py:23-26 needs that), put OUR furnace/ beside it and import the files as they are.

    python tools/stage_reference.py           # (re)build the archive; __graft_entry__.build() calls pack() too

The archive lives under oracle/_ref/ (the one place reference-derived build outputs go); nothing under torchseg_amd/
imports this module."""
import io
import os
import shutil
import sys
import tarfile
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("TSG_REFERENCE_DIR", "/root/reference")     # (the override: tests force the archive path)
OUT_DIR = os.path.join(ROOT, "oracle", "_ref")
ARCHIVE = os.path.join(OUT_DIR, "reference_models.tar.gz")

# BASELINE.json configs[1..4] (+ the R50 PSANet the CPU plumbing test uses)
EXPERIMENTS = (
    ("bisenet", "cityscapes.bisenet.R18"),
    ("pspnet", "ade.pspnet.R50_v1c"),
    ("dfn", "cityscapes.dfn.R101_v1c"),
    ("psanet", "ade.psanet.R101_v1c"),
    ("psanet", "ade.psanet.R50_v1c"),
)
FILES = ("network.py", "config.py")
# the reference's OWN furnace files its network.py / loss run on (SURVEY 8(d) "CPU reference timing": the unchanged network.py,
# seg_oprs.py, resnet.py, loss_opr.py, init_func.py + what they import) -> archive members under furnace_ref/; used ONLY by
# tools/cpu_reference.py (bench.py's cpu_baseline leg of kind "reference", a subprocess of its own: their module names are
# the ones our furnace/ answers to as well)
FURNACE_REF = (
    "base_model/__init__.py", "base_model/resnet.py", "base_model/xception.py",
    "seg_opr/__init__.py", "seg_opr/seg_oprs.py", "seg_opr/loss_opr.py",
    "utils/__init__.py", "utils/init_func.py", "utils/pyt_utils.py",
    "engine/__init__.py", "engine/logger.py", "engine/lr_policy.py",
)


def have_reference():
    return os.path.isdir(os.path.join(REF, "model"))


def pack():
    """reference checkout -> archive (deterministic member order and mtimes).  Returns the archive path, or None when
    the reference checkout is not present (the GPU box: the archive made in the build container is used)."""
    if not have_reference():
        return ARCHIVE if os.path.exists(ARCHIVE) else None
    os.makedirs(OUT_DIR, exist_ok=True)
    import gzip
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", mtime=0) as gz, \
            tarfile.open(fileobj=gz, mode="w", format=tarfile.PAX_FORMAT) as tar:
        for family, exp in EXPERIMENTS:
            for f in FILES:
                src = os.path.join(REF, "model", family, exp, f)
                info = tar.gettarinfo(src, arcname=os.path.join("model", family, exp, f))
                info.mtime, info.uid, info.gid, info.uname, info.gname = 0, 0, 0, "", ""
                with open(src, "rb") as fh:
                    tar.addfile(info, fh)
        for f in FURNACE_REF:
            src = os.path.join(REF, "furnace", f)
            info = tar.gettarinfo(src, arcname=os.path.join("furnace_ref", f))
            info.mtime, info.uid, info.gid, info.uname, info.gname = 0, 0, 0, "", ""
            with open(src, "rb") as fh:
                tar.addfile(info, fh)
    data = buf.getvalue()
    if not (os.path.exists(ARCHIVE) and open(ARCHIVE, "rb").read() == data):
        with open(ARCHIVE, "wb") as fh:
            fh.write(data)
    return ARCHIVE


def available():
    return have_reference() or os.path.exists(ARCHIVE)


def stage_reference_furnace(base_dir):
    """`<base_dir>/TorchSeg/furnace/` = the REFERENCE's own furnace files (FURNACE_REF), for the CPU leg that times the
    reference code itself.  Call before stage(): stage() then finds `furnace` present and does not link ours."""
    fdir = os.path.join(str(base_dir), "TorchSeg", "furnace")
    for f in FURNACE_REF:
        dst = os.path.join(fdir, f)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if have_reference():
            shutil.copy(os.path.join(REF, "furnace", f), dst)                     # scratch copy, never committed
        elif os.path.exists(ARCHIVE):
            with tarfile.open(ARCHIVE, "r:gz") as tar:
                with tar.extractfile(tar.getmember(os.path.join("furnace_ref", f))) as src, open(dst, "wb") as out:
                    out.write(src.read())
        else:
            raise FileNotFoundError("neither %s nor %s is present" % (REF, ARCHIVE))
    return fdir


def stage(base_dir, family, exp, files=FILES):
    """Lay out `<base_dir>/TorchSeg/model/<family>/<exp>/{files}` (from the checkout when present, else from the
    archive) with `<base_dir>/TorchSeg/furnace` -> our furnace package.  Returns the experiment directory."""
    base = os.path.join(str(base_dir), "TorchSeg")
    exp_dir = os.path.join(base, "model", family, exp)
    os.makedirs(exp_dir, exist_ok=True)
    if have_reference():
        for f in files:
            shutil.copy(os.path.join(REF, "model", family, exp, f), exp_dir)      # scratch copy, never committed
    elif os.path.exists(ARCHIVE):
        with tarfile.open(ARCHIVE, "r:gz") as tar:
            for f in files:
                member = tar.getmember(os.path.join("model", family, exp, f))
                with tar.extractfile(member) as src, open(os.path.join(exp_dir, f), "wb") as dst:
                    dst.write(src.read())
    else:
        raise FileNotFoundError("neither %s nor %s is present: run `python tools/stage_reference.py` where the "
                                "reference checkout exists" % (REF, ARCHIVE))
    link = os.path.join(base, "furnace")
    if not os.path.exists(link):
        os.symlink(os.path.join(ROOT, "torchseg_amd", "furnace"), link)
    return exp_dir


def import_experiment(family, exp, base_dir=None):
    """Stage an experiment and import its `config` and `network` modules IN THIS PROCESS, the way the reference's
    train.py does (cwd = the experiment directory, config.py:23-26 derives the repository root from it and puts
    <root>/furnace on sys.path).  Returns (network module, config object, experiment dir).  One experiment per process:
    the module names `config` / `network` are the reference's own and collide across families."""
    if "network" in sys.modules or "config" in sys.modules:
        raise RuntimeError("a reference experiment is already imported in this process")
    base_dir = base_dir or tempfile.mkdtemp(prefix="tsg_refstage_")
    exp_dir = stage(base_dir, family, exp)
    for p in (os.path.join(ROOT, "torchseg_amd", "shims"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    cwd = os.getcwd()
    os.chdir(exp_dir)
    sys.path.insert(0, exp_dir)
    try:
        import config as ref_config          # noqa: F401  (unchanged reference config.py)
        import network as ref_network        # unchanged reference network.py
    finally:
        os.chdir(cwd)
    return ref_network, ref_config.config, exp_dir


if __name__ == "__main__":
    print(pack())
