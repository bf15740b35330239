"""Bisect of the round-4 team regression (single C3 pose 83 -> 92 us at the commit that carried the status word through the friction
exchange): one of the changes reverted per variant.  usage (mkvar.sh): VAR=<nofault|noor|six|nobp> bash scripts/mkvar.sh scripts/variants/team_bisect.py <name>"""
import os
v = os.environ["VAR"]
p = "pfc_fused.h"; s = open(p).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, a[:60]
    s = s.replace(a, b)
if v == "nofault":
    rep("    const bool fault = phase == 0 && g.team_fault >= 0 && (int)blockIdx.x - item * nw == g.team_fault;      // uniform\n    if (fault && tid == 0) *s_flag = 1;\n", "")
    rep("(fault ? 0.5 : 1.0) * ", "")
elif v == "noor":
    rep("                if (__syncthreads_or((status & kStNonFinite) != 0)) status |= kStNonFinite;\n", "")
elif v == "six":
    rep("team_sum(g, item, nw, 2, 7, mine, 6, tid, s_team, &s_tflag, status);", "team_sum(g, item, nw, 2, 6, mine, -1, tid, s_team, &s_tflag, status);")
    rep("            if (MW) status |= (unsigned)s_tres[6];\n", "")
elif v == "nobp":
    rep("const double x = (g.bp_pose ? g.bp_pose : g.pose)[24 * (size_t)item + 12 + (tid - 160)];", "const double x = g.pose[24 * (size_t)item + 12 + (tid - 160)];")
open(p, "w").write(s)
