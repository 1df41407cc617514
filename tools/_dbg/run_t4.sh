set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/diag4_stamps.txt
: > $O
echo "== prio (shipped form)" >> $O
PBE_STAMP_CFG=10 timeout -k 10 300 python tools/phase_stamps.py c:8:64:64:320:0:320 c:8:32:32:640:0:640 >> $O 2>&1
echo "== no setprio" >> $O
PBE_STAMPS_LIB=$GRAFT_REPO_ROOT/tools/_dbg/libpbe_hip_noprio_stamps.so PBE_STAMP_CFG=10 timeout -k 10 300 python tools/phase_stamps.py c:8:64:64:320:0:320 c:8:32:32:640:0:640 g:32768:320:1280:resid >> $O 2>&1
tail -4 $O
