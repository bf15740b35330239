cd $GRAFT_REPO_ROOT
for k in 1 2 3; do
  for v in product ${VARIANT:-eig2}; do
    if [ $v = product ]; then unset PFC_LIB; else export PFC_LIB=$PWD/build/variants/$v.so; fi
    a=$(timeout 100 python scripts/lat_c3.py 2>&1 | grep "team 48" | head -1)
    b=$(timeout 100 python bench.py --config C5 --steps 50 --cpu-seconds 0 --no-extras --no-validate 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 %.1f us' % (d['ms_per_step']*1e3))")
    c=$(timeout 200 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-extras --no-validate --reps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step %.3f ms' % d['ms_per_step'])")
    echo "$v | $a | $b | $c"
  done
done
