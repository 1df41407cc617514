set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
O=gpurun_out/stamps_after.txt
: > $O
timeout -k 10 300 python tools/phase_stamps.py c:8:16:16:1280:0:1280 c:8:32:32:640:0:640 c:4:64:64:512:0:512 g:32768:320:1280:resid g:2048:10240:1280:geglu g:8192:640:2560:resid g:2048:1280:1280:resid >> $O 2>&1
PBE_STAMP_CFG=10 timeout -k 10 300 python tools/phase_stamps.py c:8:64:64:320:0:320 >> $O 2>&1
echo "stamps done"
rm -rf gpurun_out/pmc_bench
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_bench/f -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> gpurun_out/pmc_f.err
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_bench/w -o p --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> gpurun_out/pmc_w.err
python tools/pmc_bench_traffic.py gpurun_out/pmc_bench/f gpurun_out/pmc_bench/w gpurun_out/igemm_traffic.json
rm -rf gpurun_out/pmc_bench
cp gpurun_out/igemm_traffic.json profiles/igemm_traffic.json
echo "traffic done"
timeout -k 10 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['traffic'], d['roofline']['frac'])"
