#!/bin/bash
# Variant library for a same-box A/B (scripts/ab_variants.sh): copies the sources, applies a Python patch script (run in
# the copy's csrc directory) and builds build/variants/<name>.so -- never the product library, never inside the package
# (csrc/ holds exactly one shared library: tests/test_host.py).  Every variant is compiled with -DPFC_VARIANT=1, so
# pfc_build_info() reports bit 16 and the Python binding refuses it without PFC_ALLOW_DIAGNOSTIC=1; bench.py prints the
# loaded library's path and build info in its JSON line.
# usage: [EXTRA_FLAGS="-mllvm ..."] bash scripts/mkvar.sh <patch.py> <name>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/pfc_var && mkdir -p /tmp/pfc_var/a/b /tmp/pfc_var/include && cp $R/include/pfc.h /tmp/pfc_var/include/
cp $R/pressurefieldcontact.jl_amd/csrc/*.h $R/pressurefieldcontact.jl_amd/csrc/*.hip $R/pressurefieldcontact.jl_amd/csrc/*.cpp /tmp/pfc_var/a/b/
cd /tmp/pfc_var/a/b && python3 "$1"
mkdir -p $R/build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -DPFC_VARIANT=1 $EXTRA_FLAGS -fPIC -shared -o $R/build/variants/$2.so pfc_hip.hip pfc_tree.cpp pfc_sort.hip 2>&1 | grep -i " error" || true
ls -la $R/build/variants/$2.so
