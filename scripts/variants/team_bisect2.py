"""Second bisect of the team regression on the current sources (build with EXTRA_FLAGS=-DPFC_FUSED_F32=0): VAR = a | b | d | e | all"""
import os
v = os.environ["VAR"]
p = "pfc_fused.h"; s = open(p).read()
def rep(a, b):
    global s
    assert s.count(a) == 1, a[:60]
    s = s.replace(a, b)
def cut(a, b, new=""):
    global s
    i = s.index(a); j = s.index(b, i) + len(b)
    s = s[:i] + new + s[j:]
if v in ("a", "all"):
    rep("    const double *s_pose = g.bp_pose ? s_bp : I.pose + 12;", "    const double *s_pose = I.pose + 12;")
    cut("        if (g.bp_pose && tid >= 160 && tid < 172) {", "        }\n")
    cut("    if (g.bp_pose) {\n#pragma unroll\n        for (int k = 0; k < 12; ++k) pose_ok", "    }\n")
if v in ("b", "all"):
    cut("    if (phase == 0 && g.team_fault >= 0", "        }\n    }\n")
if v in ("d", "all"):
    rep("            const bool nf_w = MW && __ballot((status & kStNonFinite) != 0) != 0ull;\n", "")
    rep(" s_cnt[wave][2] = nf_w ? 1 : 0; }", " }")
    rep("            if (MW && (((s_cnt[0][2] | s_cnt[1][2]) | s_cnt[2][2]) | s_cnt[3][2])) status |= kStNonFinite;\n", "")
if v in ("e", "all"):
    rep("                if (bad) mine = __builtin_nan(\"\");\n", "")
    rep("                if (tid == 0) s_tres[6] = (tot != tot) ? 1.0 : 0.0;\n", "")
    rep("            if (MW && s_tres[6] != 0.0) status |= kStFusedOvf;", "")
open(p, "w").write(s)
