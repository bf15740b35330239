p='pfc_bp.h'; s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)==cnt,(a[:60],s.count(a)); s=s.replace(a,b)
# accumulators live across the seeds of a workgroup; one flush per workgroup at its exit
rep('''    const WorkRec s_first = g.seeds[blockIdx.x];
''','''    const WorkRec s_first = g.seeds[blockIdx.x];
    unsigned long long c_a = 0, c_b = 0, c_c = 0, c_d = 0, c_it = 0, c_p = 0, c_w1 = 0, c_w2 = 0;
    unsigned long long c_big = 0, c_bigsum = 0, c_max = 0, c_first = 0, c_seeds = 0, c_it0 = 0;
''')
rep('''#ifdef PFC_STAMPS
        unsigned long long c_a = 0, c_b = 0, c_c = 0, c_d = 0, c_it = 0, c_p = 0, c_w1 = 0, c_w2 = 0;   // c_w*: inside the two barriers, every wave
#endif''','''        c_seeds += 1; c_it0 = c_it;''')
rep('''                c_a += u1 - u0; c_b += u2 - u1; c_c += u3 - u2; c_d += u4 - u3; c_it += 1; c_p += (unsigned long long)p;''',
'''                c_a += u1 - u0; c_b += u2 - u1; c_c += u3 - u2; c_d += u4 - u3; c_it += 1; c_p += (unsigned long long)p;
                if (u1 - u0 > 4000) { c_big += 1; c_bigsum += u1 - u0; }
                if (u1 - u0 > c_max) c_max = u1 - u0;
                if (c_it == c_it0 + 1) c_first += u1 - u0;''')
rep('''#ifdef PFC_STAMPS
        if (lane == 0 && g.stamps) { atomicAdd(&g.stamps[0], c_w1); atomicAdd(&g.stamps[1], c_w2); atomicAdd(&g.stamps[2], c_it); }
        if (tid == 0 && g.stamps) {
            atomicAdd(&g.stamps[8], c_a); atomicAdd(&g.stamps[9], c_b); atomicAdd(&g.stamps[10], c_c);
            atomicAdd(&g.stamps[13], c_d); atomicAdd(&g.stamps[11], c_it); atomicAdd(&g.stamps[12], c_p);
        }
#endif''','')
rep('''            if (n_und) atomicAdd(g.ucount, n_und);   // statistics
        }
    }
}''','''            if (n_und) atomicAdd(g.ucount, n_und);   // statistics
        }
    }
    if (tid == 0 && g.stamps && c_it) {
        atomicAdd(&g.stamps[8], c_a); atomicAdd(&g.stamps[9], c_b); atomicAdd(&g.stamps[10], c_c);
        atomicAdd(&g.stamps[13], c_d); atomicAdd(&g.stamps[11], c_it); atomicAdd(&g.stamps[12], c_p);
        atomicAdd(&g.stamps[3], c_big); atomicAdd(&g.stamps[4], c_bigsum); atomicMax(&g.stamps[5], c_max); atomicAdd(&g.stamps[6], c_first); atomicAdd(&g.stamps[14], c_seeds);
    }
}''')
open(p,'w').write(s)
