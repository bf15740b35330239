"""Variant (with EXTRA_FLAGS=-DPFC_STAMPS): every rank of a team leaves the times at which its broadphase and its pass 0 ended;
pfc_debug_stamps returns, in slots 12..15, the earliest / latest broadphase end and the earliest / latest pass-0 end over the
workgroups of the launch, relative to block 0's start -- how unevenly the static share of the descent loads the ranks."""
import re
s = open("pfc_fused.h").read()
a = '''    FSTAMP(3);
#ifdef PFC_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.stamps) g.stamps[8] = (unsigned long long)n_test;'''
b = '''    FSTAMP(3);
#ifdef PFC_STAMPS
    if (threadIdx.x == 0 && g.stamps) { g.stamps[16 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); g.stamps[18 + 4 * blockIdx.x] = (unsigned long long)n_test; }
    if (blockIdx.x == 0 && threadIdx.x == 0 && g.stamps) g.stamps[8] = (unsigned long long)n_test;'''
assert s.count(a) == 1; s = s.replace(a, b)
a = '''        if (pass == 0) FSTAMP(5);'''
b = '''        if (pass == 0) FSTAMP(5);
#ifdef PFC_STAMPS
        if (pass == 0 && threadIdx.x == 0 && g.stamps) g.stamps[17 + 4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif'''
assert s.count(a) == 1; s = s.replace(a, b)
open("pfc_fused.h", "w").write(s)
s = open("pfc_hip.hip").read()
s = s.replace("h->stamps.ensure(16)", "h->stamps.ensure(16 + 4 * 256)")
a = '''    for (int k = 0; k < 16; ++k) out16[k] = (long long)v[k];
    out16[7] = h->last_undecided;'''
b = '''    for (int k = 0; k < 16; ++k) out16[k] = (long long)v[k];
    out16[7] = h->last_undecided;
    {
        unsigned long long r[4 * 256] = {0};
        if (h->stamps.p) HIP_TRY(h, copy_sync(h, r, h->stamps.p + 16, sizeof r, hipMemcpyDeviceToHost));
        const int nb = pfc_last_team(h) > 1 ? pfc_last_team(h) : 1;
        unsigned long long b0 = ~0ull, b1 = 0, p0 = ~0ull, p1 = 0, t0 = ~0ull, t1 = 0;
        for (int k = 0; k < nb; ++k) {
            if (r[4 * k] < b0) b0 = r[4 * k]; if (r[4 * k] > b1) b1 = r[4 * k];
            if (r[4 * k + 1] < p0) p0 = r[4 * k + 1]; if (r[4 * k + 1] > p1) p1 = r[4 * k + 1];
            if (r[4 * k + 2] < t0) t0 = r[4 * k + 2]; if (r[4 * k + 2] > t1) t1 = r[4 * k + 2];
        }
        out16[12] = (long long)(b0 - v[0]); out16[13] = (long long)(b1 - v[0]); out16[14] = (long long)(p0 - v[0]); out16[15] = (long long)(p1 - v[0]);
        out16[10] = (long long)t0; out16[11] = (long long)t1;
    }'''
assert s.count(a) == 1; s = s.replace(a, b)
open("pfc_hip.hip", "w").write(s)
