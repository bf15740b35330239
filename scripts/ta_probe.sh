#!/bin/bash
# ONE evidence pass for the TA_* counter question: what does rocprofv3 do with a TA_* group on this image?
# Everything it prints is kept (gpurun_out/ta_probe/): the counter list, the pass's stdout / stderr and exit code, dmesg, rocm-smi.
# Not to be looped; nothing else runs on the GPU in the same call.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ta_probe; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 5 60 rocprofv3 --list-avail > $O/list_avail.txt 2>&1; echo "list rc=$?" > $O/summary.txt
grep -o "TA_[A-Za-z0-9_]*" $O/list_avail.txt | sort -u | tr '\n' ' ' >> $O/summary.txt; echo >> $O/summary.txt
timeout -k 10 150 rocprofv3 --pmc TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr --output-format csv -d $O/run -- python3 $R/bench.py --steps 1 --warmup 1 --poses 256 --cpu-seconds 0 --no-extras --no-validate > $O/pass.log 2>&1
echo "pass rc=$?" >> $O/summary.txt
(dmesg 2>&1 | tail -30) > $O/dmesg.txt
(rocm-smi 2>&1 | head -30) > $O/smi_after.txt
ls -R $O/run 2>/dev/null | head -20 >> $O/summary.txt
cat $O/summary.txt; tail -15 $O/pass.log
