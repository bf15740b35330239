# The evidence set of a round in one GPU call: bash scripts/collect_round.sh [tag]   (default r04) -> gpurun_out/<tag>_*
set -o pipefail
T=${1:-r04}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/${T}_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/${T}_gputests.log
timeout -k 10 300 python bench.py > $O/${T}_bench.json 2> $O/${T}_bench.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --split-min 0 --cpu-seconds 0 --no-extras > $O/${T}_bench_unsplit.json 2>/dev/null
timeout -k 10 200 python bench.py --config C4 --steps 50 --cpu-seconds 4 --no-extras > $O/${T}_bench_C4.json 2>/dev/null
timeout -k 10 200 python bench.py --config C5 --steps 50 --cpu-seconds 4 --no-extras > $O/${T}_bench_C5.json 2>/dev/null
timeout -k 10 200 python bench.py --poses 2048 --cpu-seconds 0 --no-extras > $O/${T}_bench_2048.json 2>/dev/null
bash scripts/profile_round.sh $T > /dev/null 2>&1; echo "profile rc=$?"
bash scripts/pmc_probe.sh ${T}v "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" > /dev/null 2>&1; echo "pmc rc=$?"
bash scripts/pmc_groups.sh ${T}x "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F64" "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU_IOPS SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_VMEM" > $O/${T}x.log 2>&1; echo "pmc mix rc=$?"
cd $R
bash scripts/timeline.sh $T > /dev/null 2>&1
timeout 300 python scripts/latency.py 2>&1 | grep -v amdgpu > $O/${T}_latency.txt
timeout 300 python scripts/crossover.py 2>&1 | grep -v amdgpu > $O/${T}_crossover.txt
gcc -O2 -I include examples/box_on_plane.c -L pressurefieldcontact.jl_amd/csrc -lpfc_hip -Wl,-rpath,$PWD/pressurefieldcontact.jl_amd/csrc -lm -o /tmp/box_on_plane && timeout 60 /tmp/box_on_plane 5000 > $O/${T}_c_example.txt 2>&1
(export PFC_LIB=$PWD/build/variants/stamps.so PFC_ALLOW_DIAGNOSTIC=1; for c in c1 c2 c4 c3r pencil c3; do timeout 120 python scripts/small_scene.py $c 300 1; done) 2>&1 | grep -v amdgpu > $O/${T}_fused_phases.txt
(timeout 200 python scripts/lat_c3.py; timeout 300 python scripts/lat_c3_poses.py 48,32,24,16,0) 2>&1 | grep -v amdgpu > $O/${T}_team_sizes.txt
timeout 200 python scripts/pencil_like.py 2>&1 | grep -v amdgpu > $O/${T}_pencil_like.txt
bash scripts/profile_dual.sh $T > /dev/null 2>&1; echo "dual profile rc=$?"
cd $R
timeout 200 python scripts/ab_fused_f32.py 2>&1 | grep -v amdgpu > $O/${T}_ab_fused_f32.txt
(for i in 1 2 3; do timeout 120 python scripts/dual_trace.py c5 40 6 4; PFC_NO_DUAL_FOLD=1 timeout 120 python scripts/dual_trace.py c5 40 6 4; done; timeout 120 python scripts/dual_trace.py c5 40 6 0; timeout 200 python scripts/dual_trace.py c3b 5 6 2; PFC_NO_DUAL_FOLD=1 timeout 200 python scripts/dual_trace.py c3b 5 6 2) 2>&1 | grep -v amdgpu > $O/${T}_dual_fold.txt
echo collected
