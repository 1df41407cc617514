set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
A=gpurun_out/ab4.txt
: > $A
for v in base new base new; do
  if [ $v = new ]; then unset PBE_LIB_PATH; else export PBE_LIB_PATH=$GRAFT_REPO_ROOT/tools/_dbg/libpbe_hip_$v.so; fi
  echo "== $v" >> $A
  timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('img/s %.3f  sampler %.1f ms' % (d['value'], d['stage_ms_per_batch']['sampler_ms']))
for k,v in d['kernel_classes'].items(): print('   %-14s %8.2f ms  %7.1f %s' % (k, v['ms'], v['rate'], v['rate_unit']))
" >> $A
done
cat $A
