#!/bin/bash
# eager step against hipGraph replay (--graph 2: optimizer inside the graph) on one box, current tree
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
for rep in 1 2; do for g in 0 2; do
  echo "graph=$g: $(timeout 300 python bench.py --graph $g --steps 30 --warmup 10 --no-cpu-baseline --no-ohem-probe 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
done; done
