#!/bin/bash
# diagnostic build with in-kernel phase stamps -> build/variants/stamps.so (never the product library, never inside the package); use it with
#   PFC_LIB=build/variants/stamps.so PFC_ALLOW_DIAGNOSTIC=1 python scripts/stamps.py
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/variants && cd $R/pressurefieldcontact.jl_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DPFC_STAMPS -o $R/build/variants/stamps.so pfc_hip.hip pfc_tree.cpp pfc_sort.hip
