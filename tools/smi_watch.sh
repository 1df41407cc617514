#!/bin/bash
# sample GPU clocks / power while a command runs:  tools/smi_watch.sh out.txt -- cmd...
out=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ' ' ; echo; sleep 0.5; done ) > "$out" &
W=$!
"$@"
rc=$?
kill $W 2>/dev/null
exit $rc
