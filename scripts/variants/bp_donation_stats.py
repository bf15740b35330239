p='pfc_bp.h'; s=open(p).read()
def rep(a,b,c=1):
    global s
    assert s.count(a)==c,(a[:60],s.count(a)); s=s.replace(a,b)
rep('''struct Pool {
    int zero; ''','''struct Pool {
    unsigned long long *st;
    int zero; ''')
rep('''    P.ctl = nullptr; P.ring = nullptr; P.tag = g.pool_tag; P.pay = g.pool_pairs; P.ep = 0; P.zero = g.n_items >> 31;''','''    P.ctl = nullptr; P.ring = nullptr; P.tag = g.pool_tag; P.pay = g.pool_pairs; P.ep = 0; P.zero = g.n_items >> 31; P.st = g.stamps;
    const unsigned long long L_in = wall_clock64();
    unsigned long long L_iter = 0, L_chunk_iter = 0, L_units = 0, L_chunks = 0;''')
rep('''        int got = 0;
        unsigned long long *c01 = reinterpret_cast<unsigned long long *>(P.ctl);
        l2_write(my_tag, P.ep | kTagWaiting);''','''        int got = 0, polls = 0;
        const unsigned long long W0 = wall_clock64();
        unsigned long long *c01 = reinterpret_cast<unsigned long long *>(P.ctl);
        l2_write(my_tag, P.ep | kTagWaiting);''')
rep('''            const unsigned long long t = l2_read(my_tag);
            if (t == (P.ep | kTagFull)) { got = 1; break; }''','''            const unsigned long long t = l2_read(my_tag); ++polls;
            if (t == (P.ep | kTagFull)) { got = 1; break; }''')
rep('''        pool_count(P.ctl, got == 1 ? -kPoolIdle : -(kPoolIdle + kPoolRunning));''','''        if (P.st) { atomicAdd(&P.st[9], wall_clock64() - W0); atomicAdd(&P.st[4], (unsigned long long)polls); atomicAdd(&P.st[6], 1ull);
                    if (got == 1) atomicAdd(&P.st[5], wall_clock64() - l2_read(pay + kPoolChunk - 1)); }
        pool_count(P.ctl, got == 1 ? -kPoolIdle : -(kPoolIdle + kPoolRunning));''')
rep('''    if (tid == 0) l2_write(pay, ((unsigned long long)(unsigned)k << 32) | (unsigned)item);
    if (tid < k) {
        const int2 e = stk[2 * tid + 1];''','''    if (tid == 0) { l2_write(pay, ((unsigned long long)(unsigned)k << 32) | (unsigned)item); l2_write(pay + kPoolChunk - 1, wall_clock64()); if (P.st) atomicAdd(&P.st[14], 1ull); }
    if (tid < k) {
        const int2 e = stk[2 * tid + 1];''')
rep('const int k = sp / 2 < kPoolChunk - 1 ? sp / 2 : kPoolChunk - 1;','const int k = sp / 2 < kPoolChunk - 2 ? sp / 2 : kPoolChunk - 2;')
rep('''        for (int guard = 0; (sp > 0 || n_def > 0) && guard < (1 << 22); ++guard, par ^= 1) {''','''        L_units += 1; L_chunks += (src == 2);
        for (int guard = 0; (sp > 0 || n_def > 0) && guard < (1 << 22); ++guard, par ^= 1) {
            const unsigned long long L_i0 = wall_clock64();''')
rep('''            if (pool) {     // (uniform: s_want was written before the first barrier of this iteration)''','''            { const unsigned long long dt = wall_clock64() - L_i0; L_iter += dt; if (src == 2) L_chunk_iter += dt; }
            if (pool) {     // (uniform: s_want was written before the first barrier of this iteration)''')
rep('''        if (lane == 0 && g.stamps) { atomicAdd(&g.stamps[0], c_w1); atomicAdd(&g.stamps[1], c_w2); atomicAdd(&g.stamps[2], c_it); }
        if (tid == 0 && g.stamps) {
            atomicAdd(&g.stamps[8], c_a); atomicAdd(&g.stamps[9], c_b); atomicAdd(&g.stamps[10], c_c);
            atomicAdd(&g.stamps[13], c_d); atomicAdd(&g.stamps[11], c_it); atomicAdd(&g.stamps[12], c_p);
        }''','''        if (tid == 0 && g.stamps) { atomicAdd(&g.stamps[11], c_it); atomicAdd(&g.stamps[12], c_p); }''')
rep('''            if (n_und) atomicAdd(g.ucount, n_und);   // statistics
        }
    }
}

// first 64 bytes of a NodeRec''','''            if (n_und) atomicAdd(g.ucount, n_und);   // statistics
        }
    }
    if (tid == 0 && g.stamps) {
        const unsigned long long L_out = wall_clock64();
        atomicAdd(&g.stamps[0], L_out - L_in); atomicAdd(&g.stamps[1], 1ull); atomicAdd(&g.stamps[2], L_units);
        atomicAdd(&g.stamps[8], L_iter); atomicAdd(&g.stamps[10], L_chunk_iter); atomicAdd(&g.stamps[13], L_chunks);
        atomicMax(&g.stamps[3], L_out); atomicMax(&g.stamps[15], ~L_in);
    }
}

// first 64 bytes of a NodeRec''')
open(p,'w').write(s)
p='pfc_hip.hip'; s=open(p).read()
assert s.count('    np.stamps = h->stamps.p;')==1
s=s.replace('    np.stamps = h->stamps.p;','    np.stamps = nullptr;')
open(p,'w').write(s)
