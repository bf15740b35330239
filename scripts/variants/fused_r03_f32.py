"""Round 3's pfc_fused.h + ONLY the single-precision filter (no broadphase pose, no team status changes): bisect of what the
other round-4 changes of the one-launch kernel cost."""
import subprocess, re
s = subprocess.run(["git", "-C", "/root/repo", "show", "96f8b7f:pressurefieldcontact.jl_amd/csrc/pfc_fused.h"], capture_output=True, text=True).stdout
cur = open("/root/repo/pressurefieldcontact.jl_amd/csrc/pfc_fused.h").read()
def rep(a, b):
    global s
    assert s.count(a) == 1, a[:60]
    s = s.replace(a, b)
def between(text, a, b):
    i = text.index(a); j = text.index(b, i)
    return text[i:j]
rep("    unsigned long long *team;\n};", "    unsigned long long *team;\n    int team_fault;\n    const double *bp_pose;\n    int f32;\n};")
rep("    int model, nq, n_node1, n_node2, reserve, pad[11];", "    int model, nq, n_node1, n_node2, reserve, pad0;\n    double cmax12;\n    int pad[8];")
rep("    __shared__ double s_aR12[9];\n", "    __shared__ double s_aR12[9];\n    __shared__ float s_posef[13], s_q12[4];\n    __shared__ int s_pose_exact;\n")
# the tid == 192 block of the current file, on I.pose + 12
blk = between(cur, "        if (tid < 12) s_posef[tid] = (float)s_pose[tid];", "        __syncthreads();\n        FSTAMP(2);")
blk = blk.replace("s_pose[", "(I.pose + 12)[")
rep("        if (tid < 9) s_aR12[tid] = __builtin_fabs(I.pose[12 + tid]) + 1.0e-14;    // abs_R of an all-identity pair (:10)\n",
    "        if (tid < 9) s_aR12[tid] = __builtin_fabs(I.pose[12 + tid]) + 1.0e-14;    // abs_R of an all-identity pair (:10)\n" + blk)
rep("        bool bfs = MW;\n", "        bool bfs = MW;\n        bool f32_on = g.f32 != 0 && s_pose_exact == 0;\n")
# loop body
old_body = between(s, "            bool hit = false;\n            int a0 = 0, a1 = 0, b0 = 0, b1 = 0, la_id = 0, lb_id = 0;", "            STAMP(u2);")
new_body = between(cur, "            bool hit = false;\n            int a0 = 0, a1 = 0, b0 = 0, b1 = 0, la_id = 0, lb_id = 0;", "            STAMP(u2);")
rep(old_body, new_body)
old_cnt = between(s, "            const unsigned long long mc = __ballot(is_cand), m2 = __ballot(two), m4 = __ballot(four), ml = __ballot(live);", "            if (is_cand) {\n                const int pos = n_cand + c_off")
new_cnt = between(cur, "            const unsigned long long mc = __ballot(is_cand), m2 = __ballot(two), m4 = __ballot(four), ml = __ballot(live);", "            if (is_cand) {\n                const int pos = n_cand + c_off")
rep(old_cnt, new_cnt)
mac = between(cur, "        // the single-precision view of a node (NodeF, pfc_kernels.h)", "#ifdef PFC_STAMPS\n        unsigned long long cy[4]")
rep("        union NodeU { vec4i v[9]; NodeRec r; __device__ NodeU() {} };\n", "        union NodeU { vec4i v[9]; NodeRec r; __device__ NodeU() {} };\n" + mac)
open("pfc_fused.h", "w").write(s)
