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


# ---- round 5: the three ISA patterns of DESIGN.md 4.2 ("what the ISA said"), held by the build ------------------------

@pytest.fixture(scope="module")
def conv64_isa(tmp_path_factory):
    return _isa(tmp_path_factory, "conv64.hip")


@pytest.fixture(scope="module")
def ohem_isa(tmp_path_factory):
    return _isa(tmp_path_factory, "ohem.hip")


def _body(isa, needle):
    names = [n for n in re.findall(r"\n(_ZN3tsg\w+):", isa) if needle in n]
    assert names, needle
    i = isa.index("\n" + names[0] + ":")
    return isa[i:isa.index(".Lfunc_end", i)]


def _vmem_sequence(body):
    """loads ('L'), DMA loads to LDS ('D') and vmcnt waits ('W<n>') of a kernel, in program order"""
    seq = []
    for line in body.split("\n"):
        t = line.strip().split(";")[0].strip()
        if t.startswith(("global_load", "buffer_load", "scratch_load")):
            seq.append("D" if t.endswith(" lds") else ("S" if t.startswith("scratch") else "L"))
        m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", t)
        if m:
            seq.append("W" + m.group(1))
    return seq


def test_conv64_dma_kernel_does_not_spill_and_issues_its_pieces_back_to_back(conv64_isa):
    """A spilled register of the DMA geometry is reloaded in front of a `buffer_load ... lds`, and a scratch reload is an
    s_waitcnt vmcnt(0): a wait for the PREVIOUS piece's DMA (the first build of this kernel did exactly that)."""
    meta = _kernel_meta(conv64_isa)
    dma = {k: v for k, v in meta.items() if "conv64_dma_fwd_k" in k}
    assert len(dma) == 3, sorted(meta)            # <STATS, BSUM> = <0,0>, <1,0>, <0,1>
    for name, (spill, scratch) in dma.items():
        assert spill == 0 and scratch == 0, (name, spill, scratch)
    seq = _vmem_sequence(_body(conv64_isa, "conv64_dma_fwd_kILb0E"))
    assert "S" not in seq
    runs = re.findall(r"D+", "".join(s[0] if s[0] in "DS" else "x" for s in seq))
    assert runs and max(len(r) for r in runs) >= 7, seq[:60]      # the seven pieces of a wave, no wait between them


def test_fused_head_forward_keeps_its_staging_loads_in_flight(ohem_isa):
    """ohem_up_fwd2_k: the window and label loads are unconditional, so they issue in groups; the round-3 form (a load per
    divergent branch) shows `load, vmcnt(0)` pairs throughout."""
    seq = _vmem_sequence(_body(ohem_isa, "ohem_up_fwd2_kItLi1ELi20E"))
    s = "".join("L" if t == "L" else ("0" if t == "W0" else "w") for t in seq)
    assert len(re.findall(r"L{4,}", s)) >= 2, s                  # window loads and label loads issue in groups ...
    assert re.search(r"L{3,}w+L{2,}", s), s                      # ... that stream behind counted (non-zero) waits
    assert s.count("L0") <= 3, s                                 # no load-then-drain pairs in the staging loops
    old = "".join("L" if t == "L" else ("0" if t == "W0" else "w") for t in _vmem_sequence(_body(ohem_isa, "ohem_up_fwd_kItLi1ELi20E")))
    assert old.count("L0") >= 8                                  # ... which is what the round-3 form still looks like


def test_fused_head_backward_does_not_consume_its_prefetch_at_the_load(ohem_isa):
    """ohem_up_bwd_k: the next row's label / nll / lse loads are issued together with no wait between or right behind them
    (round 4: `s_waitcnt vmcnt(1)` two instructions after the label load: the widening sat at the load)."""
    for inst in ("ohem_up_bwd_kItLi1ELi20ELi1024E", "ohem_up_bwd_kItLi0ELi20ELi1024E"):
        seq = _vmem_sequence(_body(ohem_isa, inst))
        s = "".join("L" if t == "L" else "W" for t in seq)
        assert "LLL" in s, (inst, seq)
        k = [i for i in range(len(seq) - 2) if seq[i:i + 3] == ["L", "L", "L"]]
        # the prefetch triple inside the row loop is the last triple of the listing before the flush code; it must not be
        # followed at once by a counted wait on its first load
        last = k[-1]
        nxt = seq[last + 3] if last + 3 < len(seq) else ""
        assert nxt not in ("W2", "W1"), (inst, seq)
