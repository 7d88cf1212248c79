"""MFMA utilisation per kernel family from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES) over the
bench command: busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 (the
counter is summed over the 32 shader engines; MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per 32x32x16 bf16
MFMA).  usage: pmc_mfma_busy.py <dir with counter_collection.csv> <out.json>"""
import collections, csv, glob, json, os, re, sys
FAM = [("conv3x3_wrw", r"conv3_wrw_(gen_k<|tr_k|k\()"), ("conv3x3_c64_fwd", r"conv64_(dma_)?fwd(_s2)?_k<"), ("conv3x3_s2_dgrad", r"conv3s2d_k"), ("conv3x3_gen_fwd", r"conv3[gh]_fwd_k<"),
       ("conv3x3_c64_s2_dgrad", r"conv64_dgrad_s2_k<"), ("stem_conv_fwd_stats", r"stem_fwd_k<true>"), ("stem_conv_wrw", r"stem_wrw_k<false>"),
       ("stem_conv_wrw_bn", r"stem_wrw_k<true>"), ("psa_mm", r"psa_mm<")]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        for fam, pat in FAM:
            if re.search(pat, row["Kernel_Name"]):
                acc[fam][row["Counter_Name"]] += float(row["Counter_Value"])
out = {}
for fam, c in acc.items():
    if c.get("SQ_BUSY_CYCLES"):
        out[fam] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * c["SQ_BUSY_CYCLES"] / 32.0), 4)
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
