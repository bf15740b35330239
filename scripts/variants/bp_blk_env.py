# mkvar.sh patch: the workgroup size of the depth-first broadphase of small / mid launches from the environment
# (PFC_BP_BLK = 256 | 512) instead of bp_block_for -- the variant behind scripts/sweep_bp_blk_levels.py and sweep_bp_split.py.
p = 'pfc_hip.hip'; s = open(p).read()
a = '''            const int blk = bp_block_for(h, n_items);'''
assert s.count(a) == 1
s = s.replace(a, '''            static const int blk_env = std::getenv("PFC_BP_BLK") ? std::atoi(std::getenv("PFC_BP_BLK")) : 0;
            const int blk = (blk_env && n_items < kBpSmallBlockMin) ? blk_env : bp_block_for(h, n_items);''')
a = '''bool pile_mode(const pfc_context *h, int n_items) {
    return '''
assert s.count(a) == 1
s = s.replace(a, '''bool pile_mode(const pfc_context *h, int n_items) {
    if (std::getenv("PFC_BP_BLK")) return false;      // the sweeps choose split / whole themselves (option split_min)
    return ''')
s = s.replace('#include <cmath>\n', '#include <cmath>\n#include <cstdlib>\n', 1)
open(p, 'w').write(s)
