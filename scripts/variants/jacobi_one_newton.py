# jacobi_angle with ONE Newton step on the two reciprocals and the first rsqrt (t only steers convergence; c keeps two steps: c^2 + s^2 = 1 to rounding)
p='pfc_br.h'; s=open(p).read()
for a in ('''    r = r * __builtin_fma(-(2.0 * apq), r, 2.0);
    r = r * __builtin_fma(-(2.0 * apq), r, 2.0);
''','''    y = y * __builtin_fma(-0.5 * h * y, y, 1.5);
    y = y * __builtin_fma(-0.5 * h * y, y, 1.5);
''','''    q = q * __builtin_fma(-den, q, 2.0);
    q = q * __builtin_fma(-den, q, 2.0);
'''):
    assert s.count(a)==1
    s=s.replace(a,a.splitlines(True)[0])
open(p,'w').write(s)
