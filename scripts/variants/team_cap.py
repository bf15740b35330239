p='pfc_hip.hip'; s=open(p).read()
a='''        int nw = (h->max_leaves + 128) / 256;
        if (nw > 8) nw = 8;'''
assert s.count(a)==1
s=s.replace(a,'''        static const int cap_env = std::getenv("PFC_TEAM_CAP") ? std::atoi(std::getenv("PFC_TEAM_CAP")) : 8;
        static const int per_env = std::getenv("PFC_TEAM_PER") ? std::atoi(std::getenv("PFC_TEAM_PER")) : 256;
        int nw = (h->max_leaves + per_env / 2) / per_env;
        if (nw > cap_env) nw = cap_env;''')
if '#include <cstdlib>' not in s: s=s.replace('#include <cmath>\n','#include <cmath>\n#include <cstdlib>\n',1)
open(p,'w').write(s)
