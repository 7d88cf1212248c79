"""INTEGRATION.md section 3 shows the ctypes stubs a maintainer would copy.  Round 2's communicator stub had drifted from
include/tsg_hip.h (wrong argument list), so the code blocks of that section are now executed: every `lib.tsg_*(...)`
call is checked against the header's prototype on CPU, and the stubs themselves run on a GPU."""
import ast
import ctypes
import os
import re
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _section3_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3."):text.index("## 4.")]
    blocks = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    assert len(blocks) >= 2
    return blocks


def _header_arity():
    hdr = open(os.path.join(ROOT, "include", "tsg_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(tsg_\w+)\s*\(([^;{]*?)\)\s*;", hdr):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_section3_calls_match_the_header_prototypes():
    arity = _header_arity()
    seen = set()
    for block in _section3_blocks():
        tree = ast.parse(block)
        for node in ast.walk(tree):
            if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith("tsg_")
                    and isinstance(node.func.value, ast.Name) and node.func.value.id == "lib"):
                name = node.func.attr
                assert name in arity, f"{name} is not declared in include/tsg_hip.h"
                assert len(node.args) == arity[name], (name, len(node.args), arity[name])
                seen.add(name)
    assert {"tsg_comm_get_unique_id", "tsg_comm_create", "tsg_comm_allreduce", "tsg_ohem_fwd"} <= seen


def _namespace():
    import torch
    ns = {"ctypes": ctypes, "torch": torch}
    for block in _section3_blocks():
        block = block.replace("/path/to/repo", ROOT)
        exec(compile(block, "INTEGRATION.md#3", "exec"), ns)
    return ns


def test_section3_blocks_execute_and_bind_the_library():
    ns = _namespace()                                      # ctypes.CDLL of the built .so + the function definitions
    for fn in ("ohem_forward", "make_comm", "allreduce_stats", "destroy_comm", "cls_head_forward"):
        assert callable(ns[fn])
    assert ns["lib"].tsg_comm_unique_id_bytes() == 128


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_section3_stubs_run_on_the_gpu(cuda):
    import torch
    import torch.distributed as dist
    ns = _namespace()
    lib = ns["lib"]
    # the OHEM stub against the oracle
    from oracle.ohem_ref import ohem_cross_entropy
    g = torch.Generator().manual_seed(3)
    pred = torch.randn(2, 19, 32, 48, generator=g)
    target = torch.randint(0, 19, (2, 32, 48), generator=g)
    target[:, :3] = 255
    loss, _ = ns["ohem_forward"](pred.cuda(), target.cuda(), 255, 0.7, 2 * 32 * 48 // 16)
    ref = ohem_cross_entropy(pred, target, ignore_label=255, thresh=0.7, min_kept=2 * 32 * 48 // 16)
    assert abs(loss.item() - float(ref)) < 1e-4
    # the classifier-head stub against torch's own 1x1 convolution on the bf16-rounded operands
    x = torch.randn(2, 64, 16, 24, generator=g).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(19, 64, 1, 1, generator=g) * 0.1).cuda()
    b = torch.randn(19, generator=g).cuda()
    z = ns["cls_head_forward"](x, w, b)
    want = torch.nn.functional.conv2d(x.float(), w.bfloat16().float(), b)
    assert z.is_contiguous() and (z.float() - want).abs().max().item() <= 2.0 ** -7 * want.abs().max().item()
    # the communicator stub on a 1-rank group, id carried by a TCPStore as the text says
    store = dist.TCPStore("127.0.0.1", _free_port(), 1, True)
    comm = ns["make_comm"](lib, store, 0, 1, 0)
    msg = torch.arange(130, dtype=torch.float32, device="cuda")
    ns["allreduce_stats"](lib, comm, msg)
    torch.cuda.synchronize()
    assert torch.equal(msg.cpu(), torch.arange(130, dtype=torch.float32))
    ns["destroy_comm"](lib, comm)
