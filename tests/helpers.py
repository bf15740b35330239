"""Shared test helpers: run a configs.Workload through the CPU oracle."""
import numpy as np

from oracle import oracle as O


def oracle_meshes(w):
    return [O.OracleMesh(ms.mesh, ms.tree, ms.Ebar or 0.0) for ms in w.meshes]


def oracle_ins(pfc, c):
    mu_s, mu_d = pfc.scenario.determine_mu_s_mu_d(c.mu_s, c.mu_d)
    if c.model == "regularized":
        return O.make_ins(c.chi, c.n_quad_rule, O.REGULARIZED, mu_s, mu_d, v_c=c.v_tol)
    return O.make_ins(c.chi, c.n_quad_rule, O.BRISTLE, mu_s, mu_d, tau=c.tau, k_bar=c.k_bar, magic=c.magic)


def oracle_run(pfc, w, items=None, debug=True):
    """Per-item oracle evaluation (force_single_elastic_intersection!).  Returns a list of EvalResult."""
    om = oracle_meshes(w)
    oi = [oracle_ins(pfc, c) for c in w.instructions]
    out = []
    for k in (range(w.n_items) if items is None else items):
        c = w.instructions[int(w.ins_ids[k])]
        out.append(O.evaluate(om[c.id_1], om[c.id_2], oi[int(w.ins_ids[k])], w.pose[k], w.twist[k], w.s[k],
                              debug=debug))
    return out


def sorted_pairs(pairs, clip_n):
    """Canonical order for comparing candidate sets: sort by (i_1, i_2)."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    clip_n = np.asarray(clip_n).reshape(-1)
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    return pairs[order], clip_n[order]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    scale = max(np.linalg.norm(b), 1e-300)
    return float(np.linalg.norm(a - b) / scale)
