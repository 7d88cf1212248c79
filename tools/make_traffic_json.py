"""profiles/traffic.json from a `tools/pmc_traffic.sh` run: HBM bytes per launch (PMC) for each kernel family
bench.py reports a roofline for.  bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB for 16-B/lane streams (the gfx950
FETCH_SIZE correction of MI355X_MICROARCH.md); kernels whose loads are 4 B/lane or scalar use the uncorrected sum
(listed in UNCORRECTED).  Launch-weighted mean over the template instances of a family."""
import json, re, sys
src = json.load(open(sys.argv[1]))
FAMILIES = [("bn_stats", r"bn_reduce_(nhwc|nchw)<[^,]+, \d+, 0,"), ("bn_bwd_reduce", r"bn_reduce_(nhwc|nchw)<[^,]+, \d+, 1, [012]>"),
            ("bn_bwd_reduce_bits", r"bn_reduce_nhwc<[^,]+, \d+, 1, 3>"),
            ("bn_apply_fwd", r"bn_fwd_(nchw<|nhwc<[^,]+, \d+, (true|false), (true|false), false>)"),
            ("bn_apply_fwd_bits", r"bn_fwd_nhwc<[^,]+, \d+, (true|false), (true|false), true>"),
            ("bn_bwd_apply", r"bn_bwd_(nhwc|nchw)<[^,]+, \d+, [012],"), ("bn_bwd_apply_bits", r"bn_bwd_nhwc<[^,]+, \d+, 3,"),
            ("ohem_fwd", r"ohem_pass_a<"), ("ohem_bwd", r"ohem_bwd_k<"),
            ("upsample_fwd", r"tsg::up_fwd<"), ("upsample_bwd", r"tsg::up_bwd(_tiled)?<"),
            ("upsample_fwd_nhwc", r"up_fwd_nhwc<"), ("upsample_bwd_nhwc", r"up_bwd_nhwc<"),
            ("chanscale_fwd", r"cs_fwd_"), ("chanscale_bwd", r"cs_bwd_(nchw|nhwc<[^,]+, \d+, (true|false), true>)"),
            ("chanscale_bwd_ds", r"cs_bwd_nhwc<[^,]+, \d+, (true|false), false>"), ("chanscale_bwd_dx", r"cs_dx_nhwc<"), ("gap_fwd", r"gap_fwd"), ("gap_bwd", r"gap_bwd"),
            ("maxpool_fwd", r"maxpool_fwd_nhwc<"), ("maxpool_bwd", r"maxpool_bwd_nhwc<"),
            ("stem_conv_fwd", r"stem_fwd_k<false>"), ("stem_conv_fwd_stats", r"stem_fwd_k<true>"), ("stem_conv_wrw", r"stem_wrw_k<false>"), ("stem_conv_wrw_bn", r"stem_wrw_k<true>"),
            ("sgd_multi_step", r"sgd_multi_k"),
            ("conv3x3_wrw", r"conv3_wrw_(gen_k<|tr_k|k\()"), ("conv3x3_c64_fwd", r"conv64_(dma_)?fwd(_s2)?_k<"),
            ("conv3x3_gen_fwd", r"conv3[gh]_fwd_k<"),
            ("conv3x3_c64_s2_dgrad", r"conv64_dgrad_s2_k<"), ("conv3x3_s2_dgrad", r"conv3s2d_k"), ("ohem_up_fwd", r"ohem_up_fwd2?_k<"), ("ohem_up_bwd", r"ohem_up_bwd_k<"),
            ("bn_relu_pool_fwd", r"bn_relu_pool_fwd_k<"), ("bn_relu_pool_bwd_reduce", r"bn_relu_pool_bwd_reduce_k<"),
            ("bn_relu_pool_bwd_apply", r"bn_relu_pool_bwd_apply_k<")]
UNCORRECTED = {"ohem_bwd", "stem_conv_fwd", "stem_conv_fwd_stats", "ohem_up_fwd", "ohem_up_bwd"}   # scalar side-array loads / 4-B-per-lane loads
out = {"_source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 2`, "
                  "MI355X, tools/pmc_traffic.sh + tools/make_traffic_json.py; bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                  "(KiB units, FETCH_SIZE counts 1/2 of 16-B/lane streams on gfx950: MI355X_MICROARCH.md), uncorrected sum for "
                  + ", ".join(sorted(UNCORRECTED)) + " whose loads are not 16-B/lane streams; stem_conv_wrw mixes both (dy 16 B, x 4 B per lane) "
                  "and is reported corrected (upper bound)."}
for fam, pat in FAMILIES:
    tot = n = 0
    for k, v in src.items():
        if re.search(pat, k):
            f, w, l = v["FETCH_SIZE_KiB_per_launch"], v["WRITE_SIZE_KiB_per_launch"], v["launches"]
            b = ((1 if fam in UNCORRECTED else 2) * f + w) * 1024
            tot += b * l; n += l
    if n:
        out[fam] = int(tot / n)
out["_round"] = sys.argv[3] if len(sys.argv) > 3 else "r03"
if len(sys.argv) > 4:                       # per-family MFMA utilisation from an SQ counter pass (tools/pmc_mfma_busy.py)
    out["_mfma_busy_frac"] = json.load(open(sys.argv[4]))
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_source"}, indent=1))
