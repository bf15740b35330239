#!/usr/bin/env python3
"""bench.py — throughput of the contact hot path on MI355X.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the driver launches it under
``python -m torch.distributed.run --nproc-per-node N`` (one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs; ``--config``):
  C3 (default, the configuration the metric is quoted on): a 9 680-tet compliant blob against a 5 120-triangle rigid
     tool, bristle friction, a batch of independent Monte-Carlo poses per GPU.  Weak scaling: rank r evaluates its own
     ``--poses`` poses.
  C4: 256 independent box-on-plane scenes (regularized), sharded across the ranks in contiguous blocks
     (parallel.shard_block).  Strong scaling: the 256 scenes are the whole job.
  C5: pile of 64 compliant boxes, all 2 016 body pairs as bristle instructions, sharded by cost (leaf-count product,
     parallel.shard_by_cost).  Strong scaling.
One *step* = one pass of the whole hot path (OBB-tree broadphase -> tri/tet clip + quadrature -> friction reductions)
over the rank's items with meshes, trees, poses, twists and bristle states already resident in HBM, plus (N > 1) the
RCCL all-gather of the per-item result rows (parallel.all_gather_rows for C4/C5: [wrench | sdot | counts]).

metric  = tet-tri clip+integrate ops/s, one op = one broadphase candidate (triangle, tet) pair pushed through
          the narrowphase; contact-pairs/s (items/s) is reported beside it.
timing  = ``--reps`` repetitions of the K-step block, each bracketed by barrier + synchronize; ``value`` and
          ``ms_per_step`` are the MEDIAN repetition (max over ranks per repetition); the spread is reported.
roofline: the dominant kernel with the SURVEY §8(d) accounting (algorithmic bytes / kernel time vs the 8 TB/s HBM
          peak; kernel time from HIP events on the launch stream inside libpfc_hip) plus the measured PMC traffic and
          the vector-issue view (``valu``), which is what actually bounds these kernels.  The other hot kernel is
          reported under ``roofline_other`` against the bound that applies to it (vector issue), never against HBM with
          cache-served algorithmic bytes.
validation: after the timed region ``validated_items`` random items of the batch are re-evaluated by the CPU oracle
          and compared (counts bit-equal, wrench 1e-9, sdot 1e-6).
cpu_baseline: the plain-C oracle (a scalar port of the reference algorithm; the Julia reference cannot run here)
          timed single-threaded on a bounded sample of the same items, rank 0, N = 1 only.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_OP = 236          # pair 8 + tri idx 12 + 3 verts 72 + tet idx 16 + 4 verts 96 + 4 eps 32
BYTES_PER_NODE_TEST = 240   # 2 x (c 24 + e 24 + R 72)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec
N_SIMD, CLOCK_HZ = 1024, 2.4e9


def oracle_setup(pfc, w):
    from oracle import oracle as O
    O.build()
    om = [O.OracleMesh(ms.mesh, ms.tree, ms.Ebar or 0.0) for ms in w.meshes]
    oi, m1, m2 = [], [], []
    for c in w.instructions:
        mu_s, mu_d = pfc.scenario.determine_mu_s_mu_d(c.mu_s, c.mu_d)
        if c.model == "regularized":
            oi.append(O.make_ins(c.chi, c.n_quad_rule, O.REGULARIZED, mu_s, mu_d, v_c=c.v_tol))
        else:
            oi.append(O.make_ins(c.chi, c.n_quad_rule, O.BRISTLE, mu_s, mu_d, tau=c.tau, k_bar=c.k_bar, magic=c.magic))
        m1.append(c.id_1); m2.append(c.id_2)
    return O, om, oi, m1, m2


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pfc, w, budget_s: float, n_threads: int = 1, reps: int = 5):
    """Oracle ("port") timed on the host cores as BASELINE.md section 5 states it: the same items, a warm pass, then `reps` (>= 5)
    warm repetitions of ONE bounded sample of the batch, the MEDIAN repetition reported; items evaluated independently by
    pfo_eval_batch, serially (n_threads = 1, the reference's own execution model) or spread over OpenMP threads.  The sample
    (the first items of the batch) is sized from a calibration pass so that the repetitions together take about budget_s."""
    O, om, oi, m1, m2 = oracle_setup(pfc, w)
    chunk = min(max(8 * n_threads, 16), w.n_items)
    run = lambda k: O.evaluate_batch(om, oi, m1, m2, w.ins_ids[:k], w.pose[:k], w.twist[:k], w.s[:k], n_threads)
    run(chunk)                                                          # cold: page in meshes, start the thread team
    t0 = time.perf_counter(); run(chunk); t_cal = max(time.perf_counter() - t0, 1e-6)
    n_sample = int(min(w.n_items, max(chunk, chunk * (budget_s / (reps + 1)) / t_cal)))
    st, _, _, ct = run(n_sample)                                        # warm pass over the sample itself
    assert st == 0
    ops, items = int(ct[:, 1].sum()), int(ct.shape[0])
    dts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        st, _, _, ct = run(n_sample)
        dts.append(time.perf_counter() - t0)
        assert st == 0 and int(ct[:, 1].sum()) == ops
    dt = statistics.median(dts)
    return {"value": ops / dt, "unit": "ops/s", "cores": n_threads, "kind": "port",
            "sample": f"the first {items} items of the same batch ({ops} ops), median of {reps} warm repetitions "
                      f"({min(dts):.2f} .. {max(dts):.2f} s each), oracle/pfc_oracle.c "
                      f"({'single thread' if n_threads == 1 else str(n_threads) + ' OpenMP threads over items'})",
            "reps": reps, "seconds_per_rep": {"median": dt, "min": min(dts), "max": max(dts)},
            "contact_pairs_per_s": items / dt, "cpu_model": cpu_model()}


def validate(pfc, w, local_ids, wrench, sdot, counts, k: int, seed: int = 7):
    """Oracle check of k random items of what was just timed: counts bit-equal, wrench 1e-9, sdot 1e-6 (items whose K̄ has
    a near-null eigenvalue: wrench 1e-6; two or more: sdot 1e-3)."""
    import numpy as np
    O, om, oi, m1, m2 = oracle_setup(pfc, w)
    rng = np.random.default_rng(seed)
    pick = rng.choice(len(local_ids), size=min(k, len(local_ids)), replace=False)
    worst_w = worst_s = 0.0
    n_flat = 0
    for j in pick:
        g = int(local_ids[j])
        ins = int(w.ins_ids[g])
        r = O.evaluate(om[m1[ins]], om[m2[ins]], oi[ins], w.pose[g], w.twist[g], w.s[g], debug=True)
        if not np.array_equal(np.asarray(counts[j]), r.counts):
            raise AssertionError(f"validation: counts of item {g} differ: {counts[j]} vs oracle {r.counts}")
        for name, a, b, tol in (("wrench", wrench[j], r.wrench, 1e-9), ("sdot", sdot[j], r.sdot, 1e-6)):
            nb = float(np.linalg.norm(b))
            if nb == 0.0:
                if float(np.linalg.norm(a)) != 0.0:
                    raise AssertionError(f"validation: {name} of item {g} should be zero")
                continue
            err = float(np.linalg.norm(np.asarray(a) - b) / nb)
            if r.has_K:
                # decompose_K! clamps eigenvalues of K̄ at 1e-16 sigma_max (friction.jl:92): along a (near-)null direction
                # K̄^{-1/2} amplifies the last bits of K by up to 1e8, in the reference as much as here.  A flat patch has one such
                # direction: the friction part of its wrench and its ṡ are reproducible to ~1e-8 (north_star's 1e-6 is asserted);
                # two or more (sliver / edge contacts): ṡ to 1e-3 (tests/test_sdot_sensitivity.py).
                Kb = np.diag(r.Sinv) @ r.K @ np.diag(r.Sinv)
                ev = np.linalg.eigvalsh((Kb + Kb.T) / 2)
                n_null = int(np.sum(ev < 1e-12 * ev[-1]))
                if n_null >= 1:
                    tol = 1e-6
                if n_null >= 2 and name == "sdot":
                    tol = 1e-3
                if n_null >= 1 and name == "wrench":
                    n_flat += 1
            if err >= tol:
                raise AssertionError(f"validation: {name} of item {g} off by {err:.3e} (tolerance {tol})")
            if name == "wrench":
                worst_w = max(worst_w, err)
            else:
                worst_s = max(worst_s, err)
    return {"validated_items": int(len(pick)), "validation": "counts bit-equal; worst relative error wrench "
            f"{worst_w:.1e}, sdot {worst_s:.1e} (CPU oracle, same inputs"
            + (f"; {n_flat} items with a (near-)null direction of the patch stiffness held to 1e-6" if n_flat else "") + ")"}


def small_scene_latencies(pfc, reps: int = 200):
    """What one Radau stage evaluation costs (host buffers, Python caller: ~15 us of ctypes / numpy on top of the C ABI):
    C1 (test/boxes.jl) and C2, through pfc_eval; and a single C3 pose."""
    import numpy as np
    out = {}
    for name, w in (("C1", pfc.configs.c1_boxes()), ("C2", pfc.configs.c2_box_on_plane(1)), ("C3_single_pose", pfc.configs.c3_blob_tool(1))):
        m = pfc.configs.build_scenario(w)
        for _ in range(10):
            m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
        blocks = []
        for _ in range(max(reps // 25, 1)):          # median over blocks of 25 calls (one-off runtime stalls)
            t0 = time.perf_counter()
            for _ in range(25):
                wr, sd, ct = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
            blocks.append((time.perf_counter() - t0) / 25)
        dt = statistics.median(blocks)
        # the same scene on bound buffers (MechanismScenario.bind: one foreign call per evaluation, no array checks or
        # allocations) -- what a C or Julia caller of the ABI sees, within ~2 us
        bound = m.bind(w.pose, w.twist, w.s, w.ins_ids)
        for _ in range(10):
            bound()
        blocks_b = []
        for _ in range(max(reps // 25, 1)):
            t0 = time.perf_counter()
            for _ in range(25):
                bound()
            blocks_b.append((time.perf_counter() - t0) / 25)
        dt_b = statistics.median(blocks_b)
        assert np.array_equal(bound.counts, ct)
        out[name] = {"us_per_eval": dt * 1e6, "us_per_eval_bound_buffers": dt_b * 1e6, "ops": int(ct[:, 1].sum()), "ops_per_s": float(ct[:, 1].sum()) / dt,
                     "ops_per_s_bound_buffers": float(ct[:, 1].sum()) / dt_b,
                     "path": ({0: "fused", 1: "batched", 2: "batched, two halves"}[m.last_parts()] +
                              (f", team of {m.last_team()} workgroups" if m.last_team() > 1 else ""))}
        m.close()
    return out


def fixed_order_block(pfc):
    """Option fixed_order (bit-reproducible evaluations, DESIGN section 6 "Round 4" (3)) on BASELINE config 5 through host buffers: two
    fresh handles must return the same bits -- values and Dual(6) partials, the flat-patch pairs included -- and what the option costs."""
    import numpy as np
    w = pfc.configs.c5_pile()
    n = w.n_items
    rng = np.random.default_rng(5)
    seeds = (rng.standard_normal((n, 6, 24)) * 1e-2, rng.standard_normal((n, 6, 6)) * 0.1, rng.standard_normal((n, 6, 6)) * 1e-3)
    res, cost = {}, {}
    for fixed in (1, 0):
        outs = []
        for rep in range(2):
            m = pfc.configs.build_scenario(w)
            m.set_option("fixed_order", fixed)
            v = m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids)
            d = m.force_all_elastic_intersections_dual(w.pose, w.twist, w.s, *seeds, w.ins_ids)
            outs.append(tuple(v) + tuple(d))
            if rep == 1:
                ts = []
                for _ in range(12):
                    t0 = time.perf_counter(); m.force_all_elastic_intersections(w.pose, w.twist, w.s, w.ins_ids); ts.append(time.perf_counter() - t0)
                cost[fixed] = statistics.median(ts[2:]) * 1e6
            m.close()
        a, b = outs
        dw = np.abs(a[5] - b[5]).reshape(n, -1).max(1) / np.maximum(np.abs(b[5]).reshape(n, -1).max(1), 1e-300)
        res[fixed] = {"every_output_bit_equal": bool(all(np.array_equal(x, y) for x, y in zip(a, b))),
                      "items_whose_partials_differ_by_more_than_1e-12": int((dw > 1e-12).sum()), "worst_relative_difference_of_a_partial": float(dw.max())}
    return {"workload": "C5 pile, 2 016 instructions, value + Dual(6) on two fresh handles each", "fixed_order": res[1], "default": res[0],
            "value_evaluation_us": {"default": cost[0], "fixed_order": cost[1]}}


def dual_block(pfc, dev, n_dir: int = 6):
    """roofline_dual: what Radau's Jacobian evaluations cost on this path.  Device-resident Dual(n_dir) evaluations
    with dense seeds of C5 (2 016 pile instructions) and of a 2 048-pose C3 batch: a first chunk (value pass + Dual passes,
    pfc_eval_dual_device) and the further chunks of the same Jacobian (pfc_eval_dual_device_more: Dual passes on the kept value
    pass), medians; units = contributing (pair, direction) lanes per chunk; the wave-level instruction counts of the Dual kernels
    come from profiles/pmc_dual.json (rocprofv3 --pmc of the same calls, scripts/profile_dual.sh)."""
    import numpy as np
    import torch
    out = {}
    pj = os.path.join(ROOT, "profiles", "pmc_dual.json")
    pm = json.load(open(pj)) if os.path.exists(pj) else {}
    for name, key, w in (("C5", "c5", pfc.configs.c5_pile()), ("C3x2048", "c3b", pfc.configs.c3_blob_tool(2048))):
        m = pfc.configs.build_scenario(w)
        ni = w.n_items
        T = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
        rng = np.random.default_rng(7)
        t = [T(w.ins_ids, torch.int32), T(w.pose), T(w.twist), T(w.s), T(rng.standard_normal((ni, n_dir, 24)) * 1e-3),
             T(rng.standard_normal((ni, n_dir, 6)) * 1e-2), T(rng.standard_normal((ni, n_dir, 6)) * 1e-4)]
        o = [torch.zeros((ni, 6), dtype=torch.float64, device=dev), torch.zeros((ni, 6), dtype=torch.float64, device=dev),
             torch.zeros((ni, n_dir, 6), dtype=torch.float64, device=dev), torch.zeros((ni, n_dir, 6), dtype=torch.float64, device=dev),
             torch.zeros((ni, 4), dtype=torch.int32, device=dev)]
        st = torch.cuda.current_stream().cuda_stream

        def first():
            for _ in range(40):
                m.eval_dual_device(ni, n_dir, *[x.data_ptr() for x in t], *[x.data_ptr() for x in o], st)
                if m.check() == 0:
                    return
            raise RuntimeError("Dual work lists kept overflowing")

        def more():
            m.eval_dual_device_more(n_dir, t[4].data_ptr(), t[5].data_ptr(), t[6].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), st)
            assert m.check() == 0

        for _ in range(3):
            first(); more()
        tf, tm_ = [], []
        reps = 7 if key == "c5" else 3
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); first(); tf.append(time.perf_counter() - t0)
            for _ in range(3):
                t0 = time.perf_counter(); more(); tm_.append(time.perf_counter() - t0)
        cnt = o[4].cpu().numpy()
        pairs = int(cnt[:, 2].sum())               # pairs with a polygon: the Dual passes' work list (those without a traction point are skipped)
        units = pairs * n_dir
        e = {"items": ni, "n_dir": n_dir, "pairs_with_polygon": pairs, "pair_directions_per_chunk": units,
             "first_chunk_us": statistics.median(tf) * 1e6, "further_chunk_us": statistics.median(tm_) * 1e6,
             "pair_directions_per_s_further_chunks": units / statistics.median(tm_)}
        kern = pm.get(key, {})
        if kern:
            # the Dual kernels of ONE chunk: k_narrow_dual (pass A), k_dual_poly<1>, k_dual_eig, k_dual_poly<2>, k_dual_final
            tot_i = sum(v.get("valu_insts_per_launch", 0.0) for v in kern.values())
            tot_b = sum(v.get("issue_bound_us", 0.0) for v in kern.values())
            tot_t = sum(v.get("mean_us", 0.0) for k, v in kern.items() if "flags" not in k and "select" not in k)
            e["valu"] = {"wave_instructions_per_pair_direction": tot_i / max(units, 1) , "issue_bound_us_per_chunk": tot_b,
                         "kernel_us_per_chunk_profiled": tot_t, "frac_of_issue_bound": tot_b / tot_t if tot_t else None,
                         "per_kernel": {k.replace("pfc::", ""): {"us": round(v["mean_us"], 1), "wave_instr": int(v.get("valu_insts_per_launch", 0)),
                                                                 "frac_of_issue_bound": (round(v["frac_of_issue_bound"], 3) if v.get("frac_of_issue_bound") else None),
                                                                 "active_lanes": (round(v["active_lanes"], 1) if v.get("active_lanes") else None)}
                                        for k, v in kern.items()},
                         "note": "wave64 VALU instructions from profiles/pmc_dual.json (" + str(pm.get("measured")) + "); issue cost 4.2 cycles Float64 / 3.4 other, 1024 SIMDs, 2.4 GHz"}
        out[name] = e
        m.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["C3", "C4", "C5"], default="C3")
    ap.add_argument("--poses", type=int, default=8192, help="C3: Monte-Carlo poses (items) per GPU per step")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K-step block (median reported)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--validate", type=int, default=8, help="items re-evaluated by the CPU oracle after the timed region")
    ap.add_argument("--no-validate", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the small-scene latency block")
    ap.add_argument("--friction", choices=["bristle", "regularized"], default="bristle",
                    help="friction model of the C3 instruction (BASELINE: bristle; regularized is an experiment knob)")
    ap.add_argument("--split-min", type=int, default=-1, help="library option split_min (-1: library default 1025; 0: never split)")
    ap.add_argument("--fixed-order", action="store_true", help="library option fixed_order (bit-reproducible evaluations): what the option costs at bench scale")
    ap.add_argument("--clip-min", type=int, default=-1, help="library option clip_min (-1: library default 384; 0: one-kernel narrowphase)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="library option for A/B runs (pfc_set_option), e.g. --opt clip_dense=0")
    ap.add_argument("--bfs-levels", type=int, default=-1, help="broadphase BFS levels before the DFS kernel (-1 = auto)")
    ap.add_argument("--single-process", action="store_true",
                    help="ONE process drives the --gpus devices through a multi-device handle (pfc_create_multi: the form the "
                         "single-process Julia host uses) instead of one rank per GPU; PFC_BENCH_DEVICES=0,0 rehearses it on one GPU")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    multi_devs = None
    if args.single_process:
        if world != 1:
            print("bench.py: --single-process is ONE process (do not launch it under torch.distributed.run)", file=sys.stderr)
            sys.exit(2)
        multi_devs = ([int(x) for x in os.environ["PFC_BENCH_DEVICES"].split(",")] if os.environ.get("PFC_BENCH_DEVICES")
                      else list(range(args.gpus)))
    if world != args.gpus and not args.single_process:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE {world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    # PFC_BENCH_BACKEND=gloo is a rehearsal mode for a box with fewer GPUs than ranks (ranks share the visible GPUs and
    # the exchange goes through host memory); the measured configuration is nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("PFC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # PFC_BENCH_FORCE_EXCHANGE=1: a one-rank rehearsal of the RCCL exchange on a one-GPU box (process group of size 1, the
    # collective really issued): what a multi-GPU run does per step, minus the wire
    force_x = os.environ.get("PFC_BENCH_FORCE_EXCHANGE") == "1"
    if world > 1 or force_x:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:      # (a one-rank rehearsal: any free port, so that two of them can share a box)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if (world > 1 or force_x) and backend == "nccl":
        # One collective before the library creates its streams: the communicator's own streams exist from here on.  (With the
        # communicator initialised but idle the runtime put both streams of the handle on one hardware queue; the library
        # detects and repairs that -- pfc_hip.hip, make_twin -- at 1 % of the step, this order avoids it.)
        _t = torch.zeros(8, device=dev)
        dist.all_reduce(_t)
        torch.cuda.synchronize()
    import pfc_pkg
    pfc = pfc_pkg.load()
    P = pfc.parallel

    # ---- workload; meshes are identical on every rank ------------------------------------------------------------
    n_dev = len(multi_devs) if multi_devs else 1
    if args.config == "C3":
        # weak scaling: rank-specific poses (single process: n_dev x poses in the one batch)
        w = pfc.configs.c3_blob_tool(args.poses * n_dev, seed=20260103 + 7919 * rank)
        if args.friction == "regularized":
            w.instructions[0].model = "regularized"
        mine = np.arange(w.n_items)
        parts, n_global, scaling = None, w.n_items * world, "weak"
        desc = (f"C3: 9680-tet blob x 5120-tri tool, {args.friction} friction, quad rule 2, {args.poses} Monte-Carlo poses per GPU per step")
        if multi_devs:
            mine = np.arange(w.n_items)
    elif args.config == "C4":
        w = pfc.configs.c2_box_on_plane(256, montecarlo=True)
        parts = P.shard_block(w.n_items, world)
        mine, n_global, scaling = parts[rank], w.n_items, "strong"
        desc = "C4: 256 independent box-on-plane scenes (972-tet box, 2-triangle ground, regularized), contiguous blocks of scenes per GPU"
    else:
        w = pfc.configs.c5_pile()
        leaves = [ms.tree.n_leaf for ms in w.meshes]
        cost = [leaves[c.id_1] * leaves[c.id_2] for c in w.instructions]
        parts = P.shard_by_cost(cost, world)
        mine, n_global, scaling = parts[rank], w.n_items, "strong"
        desc = "C5: pile of 64 compliant boxes (108..2352 tets), all 2016 body pairs as bristle instructions, cost-weighted shards"
    if multi_devs:      # one process, every device behind one handle: the library cuts the items into ranges per device
        mine = np.arange(w.n_items)
        desc += f"; ONE process, multi-device handle over devices {multi_devs} (pfc_create_multi)"
    m = pfc.configs.build_scenario(w, device=local_rank, devices=multi_devs)
    if args.bfs_levels >= 0:
        m.set_option("bfs_levels", args.bfs_levels)
    if args.split_min >= 0:
        m.set_option("split_min", args.split_min)
    if args.fixed_order:
        m.set_option("fixed_order", 1)
    if args.clip_min >= 0:
        m.set_option("clip_min", args.clip_min)
    for kv in args.opt:
        name, _, val = kv.partition("=")
        m.set_option(name, int(val))
    if os.environ.get("PFC_NO_FILTER"):
        m.set_option("no_filter", int(os.environ["PFC_NO_FILTER"]))     # experiment knob (1: FP64 only)
    n = int(len(mine))
    d_ins = torch.from_numpy(np.ascontiguousarray(w.ins_ids[mine]).astype(np.int32)).to(dev)
    d_pose = torch.from_numpy(np.ascontiguousarray(w.pose[mine])).to(dev)
    d_twist = torch.from_numpy(np.ascontiguousarray(w.twist[mine])).to(dev)
    d_s = torch.from_numpy(np.ascontiguousarray(w.s[mine])).to(dev)
    d_wrench = torch.zeros((max(n, 1), 6), dtype=torch.float64, device=dev)
    d_sdot = torch.zeros((max(n, 1), 6), dtype=torch.float64, device=dev)
    d_counts = torch.zeros((max(n, 1), 4), dtype=torch.int32, device=dev)
    # C3 exchange: [wrench 6 | sdot 6] per item.  Two send / receive blocks: the all-gather of step k is issued asynchronously
    # (its own RCCL stream, behind the copies into the send block) and travels while step k + 1 is evaluated; a block is reused
    # two steps later, after its collective has been waited for, and every collective is waited for before the timed region ends.
    x_c3 = (world > 1 or force_x) and args.config == "C3"
    d_out = [torch.zeros((max(n, 1), 12), dtype=torch.float64, device=dev) for _ in range(2)]
    gathered = [torch.zeros((world * n, 12), dtype=torch.float64, device=dev) for _ in range(2)] if x_c3 else None
    pending = [None, None]
    x_k = [0]
    plan = (P.RowExchange(parts, n_global, dev, force_collective=force_x)
            if (args.config != "C3" and backend == "nccl" and (world > 1 or force_x)) else None)
    stream = torch.cuda.current_stream().cuda_stream
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    xchg_ms = [0.0]

    def exchange(timed: bool):
        if world == 1 and not force_x:
            return
        if timed:
            ev_a.record()
        if args.config == "C3":
            k = x_k[0] & 1
            x_k[0] += 1
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
            d_out[k][:, :6] = d_wrench
            d_out[k][:, 6:] = d_sdot
            if backend == "nccl":
                pending[k] = dist.all_gather_into_tensor(gathered[k], d_out[k], async_op=True)
                if timed:      # the exchange's own cost is measured with nothing beside it
                    pending[k].wait()
                    pending[k] = None
            else:
                parts_cpu = [torch.empty((n, 12), dtype=torch.float64) for _ in range(world)]
                dist.all_gather(parts_cpu, d_out[k].cpu())
                gathered[k].copy_(torch.cat(parts_cpu).to(dev))
        else:
            # the product exchange: per-item rows [wrench | sdot | counts] of the rank's own items -> item order, every rank
            if backend == "nccl":
                plan(d_wrench, d_sdot, d_counts)         # parallel.RowExchange: buffers and the item-order index built once
            else:
                rows = P.pack_rows(d_wrench[:n], d_sdot[:n], d_counts[:n]).cpu()
                P.all_gather_rows(rows, parts, n_global, force_collective=force_x)
        if timed:
            ev_b.record()
            ev_b.synchronize()
            xchg_ms[0] += ev_a.elapsed_time(ev_b)

    def step(timed: bool = False):
        # re-issue while a work list overflowed (only happens while buffers are still growing, i.e. in warmup)
        if n:
            for _ in range(40):
                m.eval_device(n, d_ins.data_ptr(), d_pose.data_ptr(), d_twist.data_ptr(), d_s.data_ptr(),
                              d_wrench.data_ptr(), d_sdot.data_ptr(), d_counts.data_ptr(), stream)
                if m.check() == 0:
                    break
            else:
                raise RuntimeError("work lists kept overflowing")
        exchange(timed)

    def fence():
        for k in range(2):      # collectives still travelling belong to the steps before the fence
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # ---- timed region: --reps repetitions of the K-step block ------------------------------------------------------
    rep_dt = []
    for _ in range(max(args.reps, 1)):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            if backend != "nccl":
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rep_dt.append(float(t.item()))
    dt = statistics.median(rep_dt)
    st = m.stats() if n else {"candidates": 0, "node_tests": 0, "tractions": 0}
    path = m.last_parts() if n else 1      # 0: fused small-scene kernel, 1: batched, 2: batched as two concurrent halves
    # ---- per-stage kernel times (HIP events inside the library, batched path) and the exchange's share: extra steps --
    stage = {}
    if n:
        m.set_option("profile", 1)
        for _ in range(args.steps):
            step()
            for k, v in m.stage_ms().items():
                stage[k] = stage.get(k, 0.0) + v
        st_prof = m.stats()
        parts_prof = max(m.last_parts(), 1)
        m.set_option("profile", 0)
    for _ in range(args.steps):
        step(timed=True)
    xchg_per_step = xchg_ms[0] / max(args.steps, 1)

    tot = torch.tensor([float(st["candidates"]), float(n), float(st["node_tests"]), float(st["tractions"])],
                       dtype=torch.float64, device=dev)
    if world > 1:
        if backend != "nccl":
            tot = tot.cpu()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    ops_step, items_step, nodes_step, trac_step = [float(v) for v in tot.tolist()]

    if rank == 0:
        K = args.steps
        np_ms = stage["narrowphase"] / K
        bp_ms = stage["broadphase"] / K
        parts_n = parts_prof * (max(m.last_shards(), 1) if multi_devs else 1)      # launches the step's units are spread over
        # SURVEY §8(d) accounting: algorithmic bytes / kernel time against the HBM peak (well below 1: these kernels re-read
        # their 64..256-byte records from L2 / Infinity Cache and are bound by vector issue, see "valu").
        roof_np = {"kernel": "k_clip_queue + k_integ (the narrowphase of a big batch: clip-only kernel -- k_narrow<.., 2 / 3> for scenarios with tet-tet instructions --, "
                             "then the integration over the kept polygons; ms_per_launch covers both, HIP events around the pair)", "bound": "hbm",
                   "limited_by": "vector-instruction issue and latency, not HBM: see 'valu' (the HBM fraction is the accounting SURVEY 8d prescribes)",
                   "achieved": BYTES_PER_OP * st_prof["candidates"] / parts_n / (np_ms * 1e-3) / 1e9,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "ms_per_launch": np_ms,
                   "units_per_launch": st_prof["candidates"] / parts_n, "bytes_per_unit": BYTES_PER_OP}
        roof_np["frac"] = roof_np["achieved"] / roof_np["peak"]
        # The broadphase reads two 64-byte NodeF lines per node test, all from L2 / Infinity Cache (PMC traffic ~0.1 % of the
        # 240 algorithmic bytes): HBM is not its roofline.  It is priced against vector-instruction issue instead.
        roof_bp = {"kernel": "k_bp_dfs32 (+ k_bp_expand seed levels for small batches)", "bound": "valu_fp32",
                   "unit": "wave-instructions/s", "traffic": None, "ms_per_launch": bp_ms,
                   "units_per_launch": st_prof["node_tests"] / parts_n,
                   "algorithmic_gbs_cache_served": BYTES_PER_NODE_TEST * st_prof["node_tests"] / parts_n / (bp_ms * 1e-3) / 1e9,
                   "peak": N_SIMD * CLOCK_HZ / 2.4}
        pv = os.path.join(ROOT, "profiles", "pmc_valu.json")      # counters were measured on the C3 default workload
        if os.path.exists(pv) and args.config == "C3":
            try:
                vj = json.load(open(pv))
                # Issue cost of a wave64 VALU instruction by type, measured on this chip at >= 2 waves per SIMD
                # (scripts/micro/valu_rate.hip, profiles/r02_valu_rate.txt): Float32 add/mul/fma 2.4 cycles, Float64 4.2,
                # v_pk_fma_f32 4.2, the integer add/logic/shift mix 3.4 (taken for every instruction the mix counters do not
                # classify: moves, selects, compares, conversions).  Mix from SQ_INSTS_VALU_{ADD,MUL,FMA}_F32/F64 (pmc_valu.json).
                CYC_F32, CYC_F64, CYC_OTHER = 2.4, 4.2, 3.4
                def vsum(kname, suffix):       # the narrowphase is two kernels: k_narrow (clip) + k_integ
                    ks = ("k_narrow", "k_integ") if kname == "k_narrow" else (kname,)
                    return sum(vj.get(f"{k}_{suffix}", 0) for k in ks)
                for r, kname, unit_key in ((roof_bp, "k_bp_dfs32", "node_tests"), (roof_np, "k_narrow", "candidates")):
                    n_valu = vsum(kname, "valu_insts")
                    f32 = sum(vsum(kname, f"valu_{t}_f32") for t in ("add", "mul", "fma"))
                    f64 = sum(vsum(kname, f"valu_{t}_f64") for t in ("add", "mul", "fma", "trans"))
                    have_mix = (kname + "_valu_fma_f64") in vj
                    if have_mix:
                        cyc = (f32 * CYC_F32 + f64 * CYC_F64 + (n_valu - f32 - f64) * CYC_OTHER) / n_valu
                    else:
                        cyc = CYC_F32 if kname == "k_bp_dfs32" else CYC_F64
                    per_unit = n_valu / vj[unit_key]
                    issue_ms = per_unit * r["units_per_launch"] * cyc / (N_SIMD * CLOCK_HZ) * 1e3
                    aia = kname + "_active_inst_any_cycles"
                    aia_v = vsum(kname, "active_inst_any_cycles")
                    r["valu"] = {"wave_instructions_per_unit": per_unit, "cycles_per_instruction": cyc,
                                 "mix": {"f32": f32 / n_valu, "f64": f64 / n_valu, "other": 1.0 - (f32 + f64) / n_valu} if have_mix else None,
                                 "active_lanes_per_instruction": (vsum(kname, "valu_thread_cycles") / n_valu
                                                                  if (kname + "_valu_thread_cycles") in vj else None),
                                 "issue_bound_ms": issue_ms, "frac_of_issue_peak": issue_ms / r["ms_per_launch"],
                                 # SQ_ACTIVE_INST_ANY (quad-cycles, summed over WAVES) over the SIMD-cycles of a launch: the mean
                                 # number of this kernel's waves per SIMD that have an instruction of any type (VALU, SALU, LDS,
                                 # VMEM) in flight; it is a sum over resident waves, so it is not capped at 1
                                 "waves_with_inst_in_flight_per_simd": (aia_v / parts_n * 4.0 /
                                                                        (N_SIMD * CLOCK_HZ * r["ms_per_launch"] * 1e-3)) if aia in vj else None,
                                 "note": "issue cost per wave64 VALU instruction weighted by the measured instruction mix (Float32 2.4, "
                                         "Float64 4.2, other 3.4 cycles: profiles/r02_valu_rate.txt); 1024 SIMDs, 2.4 GHz; counts from "
                                         f"profiles/pmc_valu.json ({vj.get('measured', 'see file')})"}
                roof_bp["peak"] = N_SIMD * CLOCK_HZ / roof_bp["valu"]["cycles_per_instruction"]
                roof_bp["achieved"] = roof_bp["valu"]["wave_instructions_per_unit"] * roof_bp["units_per_launch"] / (bp_ms * 1e-3)
                roof_bp["frac"] = roof_bp["achieved"] / roof_bp["peak"]
            except Exception:
                pass
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # measured offline with rocprofv3 --pmc
        if os.path.exists(pmc) and args.config == "C3" and args.poses == 8192:
            try:
                tj = json.load(open(pmc))
                roof_np["traffic"] = tj.get("k_narrow0_bytes_per_launch")
                roof_bp["traffic"] = tj.get("k_bp_dfs32_bytes_per_launch")
                roof_np["traffic_note"] = roof_bp["traffic_note"] = tj.get("note")
            except Exception:
                pass
        if parts_n > 1:
            for r in (roof_np, roof_bp):
                r["launch_note"] = (f"the step runs as {parts_n} concurrent half-batches on two streams: a launch processes "
                                    "1/2 of the step's units WHILE kernels of the other half share the CUs, so per-launch "
                                    "durations are longer than those of an exclusive launch (they overlap and do not add up to ms_per_step)")
        out = {
            "metric": "tet-tri clip+integrate ops/s",
            "value": ops_step * K / dt,
            "unit": "ops/s",
            "n_gpus": (len(set(multi_devs)) if multi_devs else world), "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": desc, "name": args.config, "items_per_step": items_step, "ops_per_step": ops_step,
                       "node_tests_per_step": nodes_step, "traction_points_per_step": trac_step,
                       "exchange": ("none" if (world == 1 and not force_x) else
                                    ("RCCL all-gather of [wrench, sdot] per item" if args.config == "C3" else
                                     "RCCL all-gather of [wrench | sdot | counts] rows (parallel.all_gather_rows)")
                                    if backend == "nccl" else f"REHEARSAL ({backend}, ranks share GPUs): all-gather through host memory")},
            "timing": {"reps": len(rep_dt), "median_ms_per_step": dt / K * 1e3,
                       "min_ms_per_step": min(rep_dt) / K * 1e3, "max_ms_per_step": max(rep_dt) / K * 1e3,
                       "seconds_timed": sum(rep_dt)},
            "contact_pairs_per_s": items_step * K / dt,
            "node_tests_per_s": nodes_step * K / dt,
            "stage_ms_per_step": {k: v / K for k, v in stage.items()},
            "exchange_ms_per_step": xchg_per_step,
            "exchange_share": (xchg_per_step / (dt / K * 1e3)) if (world > 1 or force_x) else 0.0,
            "path": {0: "fused small-scene kernel", 1: "batched", 2: "batched, two concurrent halves"}[path],
            "single_process_devices": multi_devs, "shards_used": (m.last_shards() if n else 0),
            # which library did the work: a variant build (scripts/mkvar.sh) reports bit 16, stamps bit 0, elimination bits 8..15
            "library": {"path": getattr(pfc._lib.lib(), "pfc_loaded_path", None), "build_info": int(pfc._lib.lib().pfc_build_info()),
                        "version": int(pfc._lib.lib().pfc_version())},
            "concurrent_parts": max(path, 1),
            "roofline": roof_np if (args.config != "C3" or np_ms >= bp_ms) else roof_bp,
            "roofline_other": roof_bp if (args.config != "C3" or np_ms >= bp_ms) else roof_np,
        }
        if not args.no_validate and n:
            out.update(validate(pfc, w, mine, d_wrench[:n].cpu().numpy(), d_sdot[:n].cpu().numpy(),
                                d_counts[:n].cpu().numpy(), args.validate))
        if world == 1 and not args.no_extras:
            # single-pose rate of the C3 scene and the small reference-sized scenes (latency, not throughput)
            out["small_scenes"] = small_scene_latencies(pfc)
            out["single_pose_ops_per_s"] = out["small_scenes"]["C3_single_pose"]["ops_per_s"]
            out["roofline_dual"] = dual_block(pfc, dev)
            out["reproducibility"] = fixed_order_block(pfc)
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(pfc, w, args.cpu_seconds, 1)
            usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
            out["cpu_baseline"]["host_cores_usable_by_this_process"] = usable
            import shutil
            # the Julia reference itself can only be timed where a Julia toolchain exists
            out["cpu_baseline"]["julia_on_box"] = shutil.which("julia")
            # second leg: the same port over ALL host cores the process may use, and beside it
            # over the box's CPU share for one GPU (16 cores)
            if usable > 1:
                out["cpu_baseline_multicore"] = cpu_baseline(pfc, w, args.cpu_seconds / 2, usable)
            if usable > 16:
                out["cpu_baseline_16_cores"] = cpu_baseline(pfc, w, args.cpu_seconds / 2, 16)
                out["cpu_baseline_16_cores"]["note"] = ("the CPU share of a one-GPU job on the bench box (gpurun: 16 cores); the scheduler may grant "
                                                        "the job less CPU time than the affinity mask shows, in which case the all-core leg is "
                                                        "time-sliced and comes out SLOWER than this one")
        print(json.dumps(out))
    m.close()
    if world > 1 or force_x:
        for k in range(2):
            if pending[k] is not None:
                pending[k].wait()
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
