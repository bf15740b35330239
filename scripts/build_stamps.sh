#!/bin/bash
# diagnostic build of libpfc_hip.so with in-kernel phase stamps; rebuild normally afterwards (touch the .hip)
cd "$(dirname "$0")/../pressurefieldcontact.jl_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DPFC_STAMPS -o libpfc_hip.so pfc_hip.hip pfc_tree.cpp
