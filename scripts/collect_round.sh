set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/r02_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/r02_gputests.log
timeout -k 10 300 python bench.py > $O/r02_bench.json 2> $O/r02_bench.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --split-min 0 --cpu-seconds 0 --no-extras > $O/r02_bench_unsplit.json 2>/dev/null
timeout -k 10 200 python bench.py --config C4 --steps 50 --cpu-seconds 4 --no-extras > $O/r02_bench_C4.json 2>/dev/null
timeout -k 10 200 python bench.py --config C5 --steps 50 --cpu-seconds 4 --no-extras > $O/r02_bench_C5.json 2>/dev/null
timeout -k 10 200 python bench.py --poses 2048 --cpu-seconds 0 --no-extras > $O/r02_bench_2048.json 2>/dev/null
bash scripts/profile_round.sh r02 > /dev/null 2>&1; echo "profile rc=$?"
bash scripts/pmc_probe.sh r02v "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" > /dev/null 2>&1; echo "pmc rc=$?"
cd $R
timeout 300 python scripts/latency.py 2>&1 | grep -v amdgpu > $O/r02_latency.txt
timeout 300 python scripts/crossover.py 2>&1 | grep -v amdgpu > $O/r02_crossover.txt
gcc -O2 -I include examples/box_on_plane.c -L pressurefieldcontact.jl_amd/csrc -lpfc_hip -Wl,-rpath,$PWD/pressurefieldcontact.jl_amd/csrc -lm -o /tmp/box_on_plane && timeout 60 /tmp/box_on_plane 5000 > $O/r02_c_example.txt 2>&1
(export PFC_LIB=$PWD/pressurefieldcontact.jl_amd/csrc/exp/stamps.so PFC_ALLOW_DIAGNOSTIC=1; for c in c1 c2 c4 c3r pencil; do timeout 120 python scripts/small_scene.py $c 300 1; done) 2>&1 | grep -v amdgpu > $O/r02_fused_phases.txt
cd /tmp && export TMPDIR=/tmp
for c in c1 c2 c4; do rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_small_$c -- python3 $R/scripts/small_scene.py $c 300 1 > $O/r02_small_$c.log 2>&1; done
cd $R; timeout 200 python scripts/pencil_like.py 2>&1 | grep -v amdgpu > $O/r02_pencil_like.txt
echo collected
