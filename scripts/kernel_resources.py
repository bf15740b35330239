"""Per-kernel register / LDS / scratch figures of the product build (no GPU needed).

usage: python scripts/kernel_resources.py [extra hipcc flags] > out.txt
(hipcc -Rpass-analysis=kernel-resource-usage on the product sources and flags of pressurefieldcontact.jl_amd/_lib.py)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pressurefieldcontact.jl_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared"]


def main():
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + sys.argv[1:] + ["-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/pfc_res.so",
                                                            "pfc_hip.hip", "pfc_tree.cpp", "pfc_sort.hip"]
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if out.returncode != 0:
        sys.stderr.write(out.stderr[-4000:])
        sys.exit(1)
    cur, rows = None, {}
    for ln in out.stderr.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", ln)
        if m:
            cur = m.group(1); rows[cur] = {}
            continue
        m = re.search(r"remark: .*?(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs Spill|VGPRs Spill): (\d+)", ln)
        if m and cur:
            rows[cur][m.group(1)] = int(m.group(2))
    for k, v in rows.items():
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name))
        g = lambda key: v.get(key, 0)
        print(f"{name:48s} VGPR {g('VGPRs'):4d} AGPR {g('AGPRs'):4d} SGPR {g('SGPRs'):4d} sgpr-spill {g('SGPRs Spill'):4d} "
              f"vgpr-spill {g('VGPRs Spill'):4d} scratch {g('ScratchSize [bytes/lane]'):5d} LDS {g('LDS Size [bytes/block]'):6d} "
              f"occ {g('Occupancy [waves/SIMD]')}")


if __name__ == "__main__":
    main()
