#!/bin/bash
# Two trees on ONE device, alternating default bench runs (devices of the pool differ by +-4 %, so nothing else is comparable):
#   bash tools/ab_trees.sh _ab_r2 . 3      -> gpurun_out/ab_trees.txt
A=$1; B=$2; N=${3:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/ab_trees.txt
: > $out
for i in $(seq 1 $N); do
  for t in $A $B; do
    v=$(cd $R/$t && python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile $( [ "$t" = "." ] && echo --no-batch16 ) 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f images/s  %.1f ms/step  sampler %.1f ms' % (d['value'], d['ms_per_step'], d['stage_ms_per_batch']['sampler_ms']))")
    echo "round $i  tree $t : $v" | tee -a $out
  done
done
