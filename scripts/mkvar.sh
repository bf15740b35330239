#!/bin/bash
# Variant library for a same-box A/B (scripts/ab_variants.sh): copies the sources, applies a Python patch script (run in
# the copy's csrc directory) and builds csrc/exp/<name>.so -- never the product library.
# usage: [EXTRA_FLAGS="-mllvm ..."] bash scripts/mkvar.sh <patch.py> <name>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/pfc_var && mkdir -p /tmp/pfc_var/a/b /tmp/pfc_var/include && cp $R/include/pfc.h /tmp/pfc_var/include/
cp $R/pressurefieldcontact.jl_amd/csrc/*.h $R/pressurefieldcontact.jl_amd/csrc/*.hip $R/pressurefieldcontact.jl_amd/csrc/*.cpp /tmp/pfc_var/a/b/
cd /tmp/pfc_var/a/b && python3 "$1"
mkdir -p $R/pressurefieldcontact.jl_amd/csrc/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize $EXTRA_FLAGS -fPIC -shared -o $R/pressurefieldcontact.jl_amd/csrc/exp/$2.so pfc_hip.hip pfc_tree.cpp 2>&1 | grep -i " error" || true
ls -la $R/pressurefieldcontact.jl_amd/csrc/exp/$2.so
