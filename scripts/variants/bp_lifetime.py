p='pfc_bp.h'; s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)==cnt,(a,s.count(a)); s=s.replace(a,b)
# lifetime bookkeeping in k_bp_dfs32 only (first occurrence patterns are unique to it)
rep('''    const WorkRec s_first = g.seeds[blockIdx.x];
''','''    const WorkRec s_first = g.seeds[blockIdx.x];
    const unsigned long long L_in = wall_clock64();
    unsigned long long L_seed = 0, L_iter = 0, L_settle = 0, L_flush = 0, L_n = 0, L_tk = 0, L_t0 = 0;
''')
rep('''        const int item = __builtin_amdgcn_readfirstlane(s.item);   // uniform: scalar loads of the pose below
        const ItemRec *it = g.items + item;
        // the item's pose lives in LDS''','''        const int item = __builtin_amdgcn_readfirstlane(s.item);   // uniform: scalar loads of the pose below
        const ItemRec *it = g.items + item;
        L_t0 = __builtin_amdgcn_s_memtime(); L_n += 1;
        // the item's pose lives in LDS''')
rep('''        int par = 0;   // parity of the iteration: selects the parking counter
        for (int guard = 0; (sp > 0 || n_def > 0) && guard < (1 << 22); ++guard, par ^= 1) {''','''        int par = 0;   // parity of the iteration: selects the parking counter
        L_seed += __builtin_amdgcn_s_memtime() - L_t0;
        for (int guard = 0; (sp > 0 || n_def > 0) && guard < (1 << 22); ++guard, par ^= 1) {
            const unsigned long long L_i0 = __builtin_amdgcn_s_memtime();''')
rep('''            if (n_out > kOut - BLK || (sp == 0 && n_def == 0 && n_out > 0)) {
                flush_candidates<BLK>(g, ob, n_out, item, tid, &s_base);
                n_cand += n_out;
                n_out = 0;
            }
#ifdef PFC_STAMPS
            STAMP(u4);''','''            const unsigned long long L_f0 = __builtin_amdgcn_s_memtime();
            if (settle) L_settle += L_f0 - L_i0; else L_iter += L_f0 - L_i0;
            if (n_out > kOut - BLK || (sp == 0 && n_def == 0 && n_out > 0)) {
                flush_candidates<BLK>(g, ob, n_out, item, tid, &s_base);
                n_cand += n_out;
                n_out = 0;
                L_flush += __builtin_amdgcn_s_memtime() - L_f0;
            }
#ifdef PFC_STAMPS
            STAMP(u4);''')
rep('''        if (lane == 0 && g.stamps) { atomicAdd(&g.stamps[0], c_w1); atomicAdd(&g.stamps[1], c_w2); atomicAdd(&g.stamps[2], c_it); }
        if (tid == 0 && g.stamps) {
            atomicAdd(&g.stamps[8], c_a); atomicAdd(&g.stamps[9], c_b); atomicAdd(&g.stamps[10], c_c);
            atomicAdd(&g.stamps[13], c_d); atomicAdd(&g.stamps[11], c_it); atomicAdd(&g.stamps[12], c_p);
        }''','''        L_t0 = __builtin_amdgcn_s_memtime();''')
rep('''            if (n_und) atomicAdd(g.ucount, n_und);   // statistics
        }
    }
}''','''            if (n_und) atomicAdd(g.ucount, n_und);   // statistics
        }
        L_tk -= L_t0;   // closed below / at the next seed
        L_tk += __builtin_amdgcn_s_memtime();
    }
    if (tid == 0 && g.stamps) {
        const unsigned long long L_out = wall_clock64();
        atomicAdd(&g.stamps[0], L_out - L_in); atomicAdd(&g.stamps[1], 1ull); atomicAdd(&g.stamps[2], L_n);
        atomicAdd(&g.stamps[8], L_iter); atomicAdd(&g.stamps[9], L_settle); atomicAdd(&g.stamps[10], L_seed);
        atomicAdd(&g.stamps[13], L_flush); atomicAdd(&g.stamps[11], L_tk);
        atomicMax(&g.stamps[3], L_out); atomicMax(&g.stamps[4], ~L_in); atomicAdd(&g.stamps[5], L_out & 0xFFFFFFFFull); atomicMax(&g.stamps[6], ~L_out);
    }
}''')
open(p,'w').write(s)
