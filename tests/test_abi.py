"""CPU: the C-ABI library loads and exports every symbol include/tsg_hip.h declares,
and the ctypes prototype table covers exactly that set (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "tsg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(tsg_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    from torchseg_amd import _lib, build
    build.build()
    names = _declared()
    assert len(names) >= 20
    h = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(h, n)]
    assert not missing, missing
    assert set(_lib._PROTOS) == names
    assert _lib.lib().tsg_version() >= 100


def test_argument_validation_without_gpu():
    from torchseg_amd import _lib
    lib = _lib.lib()
    assert lib.tsg_bn_num_partials(_lib.NCHW, 16, 64, 512 * 512) >= 1
    assert lib.tsg_bn_num_partials(_lib.NHWC, 0, 64, 4) < 0
    plan = _lib.OhemPlan()
    assert lib.tsg_ohem_make_plan(16, 19, 1024 * 1024, 0.7, ctypes.byref(plan)) == 0
    assert plan.P == 16 * 1024 * 1024 and plan.levels == 2 and plan.bins[0] == 1229 and plan.bins[1] == 4096
    assert plan.ws_bytes > 0
    assert lib.tsg_ohem_make_plan(0, 19, 4, 0.7, ctypes.byref(plan)) < 0
    # NULL pointers are rejected before any launch
    assert lib.tsg_bn_stats(None, 0, 0, 1, 1, 1, None, None, None) < 0
    assert lib.tsg_sgd_step(None, None, None, 4, 0.1, 0.9, 0.0, 1.0, 1, None) < 0
    # host-side helpers of the multi-tensor SGD and the stem convolution
    import numpy as np
    n = np.array([1, 4096, 4097, 10000], dtype=np.int64)
    nb = lib.tsg_sgd_multi_blockmap(n.ctypes.data, 4, None, 0)
    m = np.empty((nb, 2), np.int32)
    assert nb == 7 and lib.tsg_sgd_multi_blockmap(n.ctypes.data, 4, m.ctypes.data, nb) == 7
    assert m.tolist() == [[0, 0], [1, 0], [2, 0], [2, 1], [3, 0], [3, 1], [3, 2]]
    assert lib.tsg_sgd_multi_blockmap(n.ctypes.data, 129, None, 0) < 0
    assert lib.tsg_stem_conv_supported(_lib.BF16, 3, 64, 7, 7, 2, 3, 1, 1, 1024, 1024) == 1
    assert lib.tsg_stem_conv_supported(_lib.F32, 3, 64, 7, 7, 2, 3, 1, 1, 1024, 1024) == 0
    assert lib.tsg_stem_conv_supported(_lib.BF16, 3, 64, 3, 3, 2, 1, 1, 1, 1024, 1024) == 0
    assert lib.tsg_stem_conv_ws_bytes() > 64 * 176 * 2
    assert lib.tsg_stem_conv_fwd(None, None, None, 1, 8, 8, None, 0, None) < 0


def test_product_path_refuses_cpu_tensors():
    import pytest
    import torch
    from torchseg_amd.losses import ProbOhemCrossEntropy2d
    from torchseg_amd.syncbn import SyncBatchNorm
    with pytest.raises(Exception):
        SyncBatchNorm(4)(torch.randn(2, 4, 3, 3))
    with pytest.raises(Exception):
        ProbOhemCrossEntropy2d(255, thresh=0.7, min_kept=1)(torch.randn(1, 3, 4, 4), torch.zeros(1, 4, 4, dtype=torch.long))


def test_comm_entry_points_without_gpu():
    """tsg_comm_*: librccl resolves at run time, argument validation happens before any RCCL call, and RCCL failures
    come back in their own error range with RCCL's message (no GPU here: no communicator can exist)."""
    import torch  # noqa: F401  (puts the framework's librccl.so into the process, the copy the library must reuse)
    from torchseg_amd import _lib
    lib = _lib.lib()
    assert lib.tsg_comm_init_library(None) == 0
    assert lib.tsg_comm_unique_id_bytes() == 128
    assert lib.tsg_comm_xgmi_handle_bytes() == 64
    assert lib.tsg_comm_get_unique_id(None) == -5
    h = ctypes.c_void_p()
    assert lib.tsg_comm_create(None, 0, 0, 0, ctypes.byref(h)) == -3          # world < 1
    assert lib.tsg_comm_create(None, 2, 2, 0, ctypes.byref(h)) == -3          # rank >= world
    assert lib.tsg_comm_allreduce(None, None, 4, 0, None) == -5
    assert lib.tsg_comm_allgather(None, None, None, 4, 0, None) == -5
    assert lib.tsg_comm_broadcast(None, None, 4, 0, 0, None) == -5
    assert lib.tsg_xgmi_small_allreduce(None, None, 4, None) == -5
    assert b"librccl" in lib.tsg_comm_error_string(-7)
    if not torch.cuda.is_available():
        buf = ctypes.create_string_buffer(128)
        rc = lib.tsg_comm_get_unique_id(buf)        # RCCL builds differ: some hand out an id without a device
        assert rc == 0 or rc <= -100
        if rc:
            import pytest
            assert len(lib.tsg_comm_error_string(rc)) > 0
            with pytest.raises(_lib.TsgError, match="RCCL error"):
                _lib.check(rc, "tsg_comm_get_unique_id")
        else:
            assert any(buf.raw)
