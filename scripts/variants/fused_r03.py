"""Round 3's pfc_fused.h under the current host code (bisect of the team regression): the new FuArgs fields are added but unused."""
import subprocess
s = subprocess.run(["git", "-C", "/root/repo", "show", "96f8b7f:pressurefieldcontact.jl_amd/csrc/pfc_fused.h"], capture_output=True, text=True).stdout
a = "    unsigned long long *team;\n};"
assert s.count(a) == 1
s = s.replace(a, "    unsigned long long *team;\n    int team_fault;\n    const double *bp_pose;\n    int f32;\n};")
a = "    int model, nq, n_node1, n_node2, reserve, pad[11];"
assert s.count(a) == 1
s = s.replace(a, "    int model, nq, n_node1, n_node2, reserve, pad0;\n    double cmax12;\n    int pad[8];")
open("pfc_fused.h", "w").write(s)
