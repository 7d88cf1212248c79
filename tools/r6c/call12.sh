#!/bin/bash
# round 6c, call 11: the streaming passes' outputs as non-temporal stores (libtsg_hip_nt.so = the same sources with -DTSG_NT_STORE=1) against the default build, interleaved
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c_call12.txt
Q="--no-cpu-baseline --no-ohem-probe --no-psa-probe --i64-steps 0 --ref-steps 0 --fp32-steps 0 --forced-steps 0 --no-kernel-timing"
line() { grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; m=c.get('mode_probe') or {}
print('%-10s %.1f img/s %.3f ms  chosen %s whole %s segmented %s eager %s loss %s fallback %s' % ('$1', d['value'], d['ms_per_step'], m.get('chosen'), m.get('whole_graph_ms_per_step'), m.get('segmented_ms_per_step'), m.get('eager_ms_per_step'), c.get('final_loss'), str(c.get('hip_graph_fallback'))[:200]))
"; }
cp torchseg_amd/libtsg_hip.so /tmp/base.so
cp torchseg_amd/libtsg_hip_nt.so /tmp/nt.so
{
for i in 1 2 3 4; do
cp /tmp/base.so torchseg_amd/libtsg_hip.so; python bench.py $Q 2>/dev/null | line "default"
cp /tmp/nt.so torchseg_amd/libtsg_hip.so; python bench.py $Q 2>/dev/null | line "nt variant"
done
cp /tmp/nt.so torchseg_amd/libtsg_hip.so
true
cp /tmp/base.so torchseg_amd/libtsg_hip.so
true
true
} > $O 2>&1
cp /tmp/base.so torchseg_amd/libtsg_hip.so
cat $O | cut -c1-260
