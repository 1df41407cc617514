set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "cross_attention_general" 2>&1 | tail -3
O=gpurun_out/diag1_stamps.txt
: > $O
echo "== tuned table (isolated, back-to-back)" >> $O
timeout -k 10 300 python tools/phase_stamps.py c:8:64:64:320:0:320 c:8:16:16:1280:0:1280 c:8:32:32:640:0:640 c:4:64:64:512:0:512 g:2048:1280:1280:resid g:8192:640:640:resid g:2048:1280:5120:resid g:32768:320:1280:resid >> $O 2>&1
for c in 3 15 6 17 4 16 9 18; do
  echo "== forced cfg $c" >> $O
  PBE_STAMP_CFG=$c timeout -k 10 200 python tools/phase_stamps.py g:2048:1280:1280:resid g:8192:640:640:resid g:2048:1280:5120:resid g:512:1280:1280:resid >> $O 2>&1
done
echo "== in pipeline" >> $O
timeout -k 10 400 python tools/phase_stamps.py --pipeline "c:8:64:64:320:0:320:1:1:0" "c:8:16:16:1280:0:1280:1:1:0" "c:8:32:32:640:0:640:1:1:0" "g:2048:1280:1280:1|r" "g:8192:640:640:1|r" "g:32768:320:1280:1|r" "g:2048:10240:1280:1|geglu" >> $O 2>&1
tail -5 $O
