# the Jacobi round with both rotation angles computed by every lane (the form before the lane reads)
p='pfc_br.h'; s=open(p).read()
a='''            aj = __shfl(ai, j, 64); bj = __shfl(bi, j, 64);
'''
assert s.count(a)==1
s=s.replace(a,'''            {
                const int p = j < pj ? j : pj, q = j < pj ? pj : j;
                double cs, sn;
                jacobi_angle(A[7 * p], A[7 * q], A[p + 6 * q], cs, sn);
                aj = cs; bj = (j == p) ? -sn : sn;
            }
''')
open(p,'w').write(s)
