"""MIOpen tuning data for the convolutions of the supported workloads.

The reference trains with `cudnn.benchmark = True` (model/*/train.py:35), i.e.
it lets the vendor library search for the fastest convolution algorithm.  On
ROCm that search (MIOpen "find mode") writes its results to a per-user database;
`torchseg_amd/miopen_db/` ships the entries found on an MI355X for the
BiSeNet-R18 16x1024^2 bf16 shapes, so a fresh machine starts from them instead
of MIOpen's untuned immediate-mode fallback (measured: 25.6 vs 35-40 ms/step).
MIOpen needs the directory writable, so it is copied to a per-process scratch dir.
"""
import os
import shutil
import tempfile

_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def use_shipped_miopen_db(rank=0):
    """Point MIOPEN_USER_DB_PATH at a writable copy of the shipped db (unless the
    user already set one).  Must run before the first convolution."""
    if "MIOPEN_USER_DB_PATH" in os.environ or not os.path.isdir(_DB):
        return os.environ.get("MIOPEN_USER_DB_PATH")
    dst = os.path.join(tempfile.gettempdir(), "tsg_miopen_db_%d_%d" % (os.getuid(), rank))
    os.makedirs(dst, exist_ok=True)
    for f in os.listdir(_DB):
        if not os.path.exists(os.path.join(dst, f)):
            shutil.copy(os.path.join(_DB, f), dst)
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst
