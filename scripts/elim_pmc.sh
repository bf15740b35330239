R=$GRAFT_REPO_ROOT; C=$R/pressurefieldcontact.jl_amd/csrc; cd /tmp; export TMPDIR=/tmp
for e in 0 3 4 9 7 5; do
  PFC_LIB=$R/build/variants/e$e.so PFC_ALLOW_DIAGNOSTIC=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/elim_pmc_$e -- python3 $R/bench.py --steps 2 --warmup 1 --reps 1 --cpu-seconds 0 --no-extras --no-validate --split-min 0 > $R/gpurun_out/elim_pmc_$e.log 2>&1
done
echo ok
