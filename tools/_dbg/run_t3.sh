set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py -m gpu -x -q 2>&1 | tail -5
O=gpurun_out/diag3_stamps.txt
: > $O
timeout -k 10 300 python tools/phase_stamps.py c:8:16:16:1280:0:1280 c:8:32:32:640:0:640 c:4:64:64:512:0:512 g:32768:320:1280:resid g:2048:10240:1280:geglu g:8192:640:2560:resid g:2048:1280:1280:resid >> $O 2>&1
PBE_STAMP_CFG=10 timeout -k 10 300 python tools/phase_stamps.py c:8:64:64:320:0:320 >> $O 2>&1
A=gpurun_out/ab3.txt
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
