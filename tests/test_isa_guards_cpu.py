"""Build-time guards on the generated gfx950 ISA (no GPU needed: hipcc cross-compiles).

psa.hip's kernels with untracked prefetch (`psa_mm<..., UT = true>`: the fragment-order AF kernels and the opt-in dA variant) prefetch with global loads the compiler does
not track (`TSG_ASM_LD16` + a hand-placed `s_waitcnt`).  That is only sound while the registers those loads
write are never spilled or copied before the wait: a spill stores a register the load has not written yet (the
af256x64x3 instantiation failed its parity test exactly so).  So: no scratch, no VGPR spills in any UT instantiation, and
every untracked load is followed by a wait before the kernel ends."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _isa(tmp_path_factory, src):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / (src + ".s")
    cmd = [HIPCC, "-x", "hip", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=off",
           "--cuda-device-only", "-S", os.path.join(ROOT, "torchseg_amd", "csrc", src),
           "-I", os.path.join(ROOT, "include"), "-o", str(out)]
    subprocess.run(cmd, check=True, cwd=str(out.parent), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out.read_text()


@pytest.fixture(scope="module")
def psa_isa(tmp_path_factory):
    return _isa(tmp_path_factory, "psa.hip")


@pytest.fixture(scope="module")
def wrw_isa(tmp_path_factory):
    return _isa(tmp_path_factory, "conv3wrw.hip")


def _kernel_meta(isa):
    """name -> {vgpr_spill_count, private_segment_fixed_size} from the amdhsa.kernels metadata."""
    meta = {}
    for block in isa.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", block)
        scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
        if name and spill and scratch:
            meta[name.group(1)] = (int(spill.group(1)), int(scratch.group(1)))
    return meta


def test_fragment_order_psa_kernels_do_not_spill(psa_isa):
    meta = _kernel_meta(psa_isa)
    af = {k: v for k, v in meta.items() if "psa_mm" in k and k.endswith("Lb1EEEvNS_6MmArgsE")}
    assert len(af) >= 5, sorted(meta)            # forward + dX for 128x64 and 256x64, dA with two sets
    for name, (spill, scratch) in af.items():
        assert spill == 0 and scratch == 0, (name, spill, scratch)


def test_untracked_loads_are_waited_for(psa_isa):
    """Every AF kernel holds the inline-asm loads, the per-set inline-asm waits with a non-zero count (prefetch in
    flight behind the set that is consumed) and the drain `s_waitcnt vmcnt(0)` in front of the epilogue."""
    bodies = re.split(r"\n(_ZN3tsg6psa_mmI\w+):", psa_isa)
    checked = 0
    for name, body in zip(bodies[1::2], bodies[2::2]):
        if not name.endswith("Lb1EEEvNS_6MmArgsE"):
            continue
        body = body.split("s_endpgm")[0]
        asm = re.findall(r";;#ASMSTART\n(.*?)\n\s*;;#ASMEND", body, flags=re.S)
        stmts = [a.strip() for a in asm if a.strip()]
        assert any(s.startswith("global_load_dwordx4") for s in stmts), name
        waits = [s for s in stmts if s.startswith("s_waitcnt vmcnt(")]
        assert "s_waitcnt vmcnt(0)" in waits, (name, waits)                     # prologue wait / drain
        assert any(w != "s_waitcnt vmcnt(0)" for w in waits), (name, waits)     # counted waits of the K loop
        checked += 1
    assert checked >= 5


def test_two_tiles_ahead_weight_gradient_does_not_spill(wrw_isa):
    """conv3_wrw_gen_k<S, PF = 2, AFF>: two register sets of untracked loads (44 VGPRs each at stride 1).  The stride-2
    instantiation spilled (23 loads per set) and is therefore not built at all."""
    meta = _kernel_meta(wrw_isa)
    pf2 = {k: v for k, v in meta.items() if "conv3_wrw_gen_kILi" in k and "ELi2ELb" in k}
    assert len(pf2) == 2, sorted(meta)           # stride 1, with / without BN-on-load
    for name, (spill, scratch) in pf2.items():
        assert name.startswith("_ZN3tsg15conv3_wrw_gen_kILi1ELi2E"), name
        assert spill == 0 and scratch == 0, (name, spill, scratch)
    bodies = re.split(r"\n(_ZN3tsg15conv3_wrw_gen_kI\w+):", wrw_isa)
    for name, body in zip(bodies[1::2], bodies[2::2]):
        if name not in pf2:
            continue
        asm = [a.strip() for a in re.findall(r";;#ASMSTART\n(.*?)\n\s*;;#ASMEND", body.split("s_endpgm")[0], flags=re.S)]
        assert sum(a.startswith("global_load_dwordx4") for a in asm) == 4 * 11     # prologue 2 sets + one per loop half
        assert "s_waitcnt vmcnt(11)" in asm and "s_waitcnt vmcnt(0)" in asm, name


@pytest.fixture(scope="module")
def g3_isa(tmp_path_factory):
    return _isa(tmp_path_factory, "conv3g.hip")


def test_lds_dma_convolution_kernels_keep_their_budget(g3_isa):
    """conv3h_fwd_k (8 accumulators per wave) and conv3s2d_k (8 parity accumulators) run two blocks per CU, i.e. within 256
    VGPRs; their epilogues hold prefetched addend rows while the accumulators die.  A spill would put scratch traffic into
    the K loop (the first form of the addend prefetch did: 40 spilled registers).  Also: both operands of the K loop arrive
    by LDS-DMA (`buffer_load_dwordx4 ... lds`), the loop has no `ds_write` of staged data, and the fragment reads are waited
    for with counted `lgkmcnt` (not 0) in front of MFMA groups."""
    meta = _kernel_meta(g3_isa)
    want = {k: v for k, v in meta.items() if "conv3h_fwd_k" in k or "conv3s2d_k" in k}
    assert len(want) == 3, sorted(meta)                  # conv3h with / without statistics, conv3s2d
    for name, (spill, scratch) in want.items():
        assert spill == 0 and scratch == 0, (name, spill, scratch)
    bodies = re.split(r"\n(_ZN3tsg\d+conv3(?:h_fwd|s2d)_k\w+):", g3_isa)
    checked = 0
    for name, body in zip(bodies[1::2], bodies[2::2]):
        body = body.split("s_endpgm")[0]
        lines = [l.strip() for l in body.split("\n")]
        mfma = [i for i, l in enumerate(lines) if l.startswith("v_mfma_f32_32x32x16_bf16")]
        assert len(mfma) >= 36, (name, len(mfma))          # 72 (+ 32 statistics MFMAs) in conv3h, 2 x 18 in conv3s2d
        loop = lines[mfma[0]:mfma[min(len(mfma), 72) - 1]]     # one chunk of the K loop (conv3h's statistics MFMAs come later)
        dma = [l for l in loop if l.startswith("buffer_load_dwordx4") and l.endswith("lds")]
        assert len(dma) >= 4, (name, len(dma))            # pieces of the next chunk are issued between the MFMA groups
        assert not any(l.startswith("ds_write") for l in loop), name
        counted = [l for l in loop if l.startswith("s_waitcnt lgkmcnt(") and l != "s_waitcnt lgkmcnt(0)"]
        assert len(counted) >= 4, (name, counted)
        checked += 1
    assert checked == 3
