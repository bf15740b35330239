#!/bin/bash
# Elimination builds of k_narrow (see the PFC_EXP note in csrc/pfc_np.h): build the variants HERE (no GPU needed),
# then time them on the GPU box.  Variants live in build/variants/ and are selected with PFC_LIB (+ PFC_ALLOW_DIAGNOSTIC=1);
# the product library csrc/libpfc_hip.so is never touched.
#   bash scripts/elimination.sh build                      -> build/variants/e{0,3,4,9,7,5}.so
#   gpurun -- 'bash scripts/elimination.sh run'            -> gpurun_out/elim.txt (unsplit narrowphase ms per variant)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/pressurefieldcontact.jl_amd/csrc
V="0 3 4 9 7 5"
if [ "$1" = build ]; then
  mkdir -p $R/build/variants
  for e in $V; do
    (cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DPFC_EXP=$e -o $R/build/variants/e$e.so pfc_hip.hip pfc_tree.cpp pfc_sort.hip) &
  done
  wait; ls -la $R/build/variants
else
  mkdir -p $R/gpurun_out; : > $R/gpurun_out/elim.txt
  for e in $V; do
    (cd $R && PFC_LIB=$R/build/variants/e$e.so PFC_ALLOW_DIAGNOSTIC=1 timeout -k 10 120 python bench.py --cpu-seconds 0 --split-min 0 --steps 5 --no-validate > gpurun_out/elim_$e.json)
    python3 -c "import json; j=json.load(open('$R/gpurun_out/elim_$e.json')); print('PFC_EXP=$e  step %.3f ms  narrowphase %.3f ms' % (j['ms_per_step'], j['stage_ms_per_step']['narrowphase']))" >> $R/gpurun_out/elim.txt
  done
  cat $R/gpurun_out/elim.txt
fi
