#!/bin/bash
# diagnostic build with in-kernel phase stamps -> csrc/exp/stamps.so (never the product library); use it with
#   PFC_LIB=pressurefieldcontact.jl_amd/csrc/exp/stamps.so PFC_ALLOW_DIAGNOSTIC=1 python scripts/stamps.py
cd "$(dirname "$0")/../pressurefieldcontact.jl_amd/csrc" && mkdir -p exp && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DPFC_STAMPS -o exp/stamps.so pfc_hip.hip pfc_tree.cpp
