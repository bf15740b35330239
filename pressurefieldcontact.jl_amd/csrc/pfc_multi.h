// pfc_multi.h -- multi-device handles (pfc_create_multi, SURVEY section 8(b) "pfc_create(device_mask)").  Host code only;
// included by pfc_hip.hip in front of the C ABI.
//
// The reference host is ONE Julia process whose calcXd! loops over the contact instructions
// (src/contact_algorithms_non_friction.jl:60-68); it cannot be launched once per GPU.  A multi-device handle lets that one
// process use every GPU of the node through the same entry points: the handle owns one ordinary handle ("shard") per entry
// of the device list -- meshes, trees and instructions replicated at pfc_add_mesh / pfc_add_instruction / pfc_finalize
// (a few MB) -- and every evaluation cuts its items into CONTIGUOUS ranges, one per shard, balanced by cost (the node tests and
// candidates each item had in the previous evaluation of the same item list; the leaf-count product of its two meshes the first
// time: SURVEY 8(e)).  Items are independent (each force_single_elastic_intersection!, :70-84, reads immutable meshes and its
// own pose / twist / state), so there is no collective in the data path:
//   * host-pointer entry points (pfc_eval, pfc_eval_dual[_bp]): one host thread per shard (persistent workers; the caller's
//     thread takes shard 0) calls the ordinary entry point on its range of the CALLER's arrays -- every device copies its rows
//     from / to the caller's memory itself, nothing is gathered;
//   * device-pointer entry points (pfc_eval_device, pfc_eval_dual_device[_more]): the buffers live on the FIRST device of the
//     list; ranges of shards 1.. are copied to staging buffers on their device (peer copies over xGMI), evaluated there on the
//     shard's own stream and copied back into the caller's buffers; events order everything against the caller's stream, one
//     host thread enqueues it all; pfc_check synchronises every shard.
// A contiguous range keeps the candidate lists grouped the way the single-device path has them, needs no index lists and no
// gather / scatter; what it gives up against a longest-first assignment is balance when single items dominate (they do not in
// the configurations of BASELINE.json: 256 equal scenes, 2 016 pile pairs, batches of poses).
#pragma once

struct pfc_multi {
    std::vector<pfc_context *> shard;
    std::vector<int> dev;
    int opt_min_items = 8;                 // option "multi_min": items per shard below which fewer shards are used
    // partition of the last evaluation
    std::vector<int> bound;                // items [bound[k], bound[k + 1]) on shard k
    int n_used = 1;
    int part_n = 0;
    bool part_ids = false, part_dev = false;   // ... made for an evaluation with ins_ids / through the device-pointer entry points
    std::vector<int> part_ins;             // the ins_ids the partition (and the costs) belong to (host-pointer entry points)
    std::vector<double> cost;
    bool counts_valid = false;
    int *h_counts = nullptr;               // pinned: n x 4 counters of the last evaluation
    size_t h_counts_cap = 0;
    std::vector<int> iota;
    long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // ---- worker threads (host-pointer entry points) ----
    struct Worker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        std::function<int()> job;
        std::atomic<int> state{0};         // 0 idle, 1 job posted, 2 done (result not collected yet), 3 quit
        int rc = 0;
    };
    std::vector<Worker *> workers;         // workers[k - 1] serves shard k
    // ---- device-pointer entry points ----
    struct Stage {
        DevBuf<double> in, out, din, dout;
        DevBuf<int> ids, cnt, iota;
        hipEvent_t done = nullptr;
    };
    std::vector<Stage> stage;
    hipEvent_t ev_fork = nullptr;
    bool dev_pending = false;              // a device-pointer evaluation is enqueued and not checked yet
    const int *dev_counts = nullptr;       // where shard k's counters lie on ITS device: stage[k].cnt (k = 0: the caller's array or stage[0].cnt)
    hipStream_t dev_stream = nullptr;
    bool dev_reuse_ok = false;             // pfc_eval_dual_device_more may follow (same partition)
    int dev_reuse_ndir = 0;
};

namespace {

void multi_worker_loop(pfc_multi::Worker *w, int device) {
    (void)hipSetDevice(device);
    for (;;) {
        int st = w->state.load(std::memory_order_acquire);
        // a simulation evaluates back to back: stay awake for a moment after a job before going to sleep on the condition variable
        for (int spin = 0; spin < 20000 && st != 1 && st != 3; ++spin) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            st = w->state.load(std::memory_order_acquire);
        }
        if (st != 1 && st != 3) {
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [&] { const int s2 = w->state.load(std::memory_order_acquire); return s2 == 1 || s2 == 3; });
            st = w->state.load(std::memory_order_acquire);
        }
        if (st == 3) return;
        w->rc = w->job();
        w->state.store(2, std::memory_order_release);
    }
}

void multi_post(pfc_multi::Worker *w, std::function<int()> f) {
    w->job = std::move(f);
    { std::lock_guard<std::mutex> lk(w->m); w->state.store(1, std::memory_order_release); }
    w->cv.notify_one();
}

int multi_collect(pfc_multi::Worker *w) {
    for (long spin = 0; w->state.load(std::memory_order_acquire) != 2; ++spin) {
        if (spin > 4000) std::this_thread::yield();
#if defined(__x86_64__)
        else __builtin_ia32_pause();
#endif
    }
    const int rc = w->rc;
    w->state.store(0, std::memory_order_release);
    return rc;
}

// jobs[k] for the shards k < n_used (empty function: nothing to do); the caller's thread runs jobs[0].  Returns the first
// non-zero status in shard order and copies that shard's message.
int multi_run(pfc_context *h, std::vector<std::function<int()>> &jobs) {
    pfc_multi *M = h->multi;
    for (size_t k = 1; k < jobs.size(); ++k)
        if (jobs[k]) multi_post(M->workers[k - 1], jobs[k]);
    int rc = jobs[0] ? jobs[0]() : PFC_OK;
    if (rc != PFC_OK) h->err = M->shard[0]->err;
    for (size_t k = 1; k < jobs.size(); ++k)
        if (jobs[k]) {
            const int rk = multi_collect(M->workers[k - 1]);
            if (rk != PFC_OK && rc == PFC_OK) { rc = rk; h->err = M->shard[k]->err; }
        }
    (void)hipSetDevice(M->dev[0]);
    return rc;
}

// Contiguous ranges balanced by cost.  The partition of the previous evaluation is kept while the item list is the same and
// the ranges stay within 15 % of balance under the new costs: the chunks of one Jacobian (same values, other partials,
// src/radau/radau_functions.jl:2-14) must see the same ranges for the shards to reuse their value pass.
// ins_ids: the item list if the host can read it (host-pointer entry points; null = item i is instruction i); on the
// device-pointer path the list is device data: has_ids tells whether there is one, is_dev that its content is not compared (the
// caller re-evaluating the same buffers is the case the cost feedback is for).
void multi_partition(pfc_context *h, int n, const int *ins_ids, bool has_ids, bool is_dev) {
    pfc_multi *M = h->multi;
    const int K = (int)M->shard.size();
    int k_use = n / (M->opt_min_items > 0 ? M->opt_min_items : 1);
    if (k_use > K) k_use = K;
    if (k_use < 1) k_use = 1;
    const bool same_list = M->part_n == n && M->part_ids == has_ids && M->part_dev == is_dev && (int)M->bound.size() == K + 1 &&
                           (is_dev || !ins_ids || std::memcmp(M->part_ins.data(), ins_ids, sizeof(int) * (size_t)n) == 0);
    M->cost.resize((size_t)n);
    if (same_list && M->counts_valid) {
        for (int i = 0; i < n; ++i) M->cost[i] = 64.0 + (double)M->h_counts[4 * (size_t)i] + 4.0 * (double)M->h_counts[4 * (size_t)i + 1];
    } else {
        const pfc_context *c0 = M->shard[0];
        for (int i = 0; i < n; ++i) {
            double c = 64.0;
            if (!(is_dev && has_ids)) {      // (a device-resident item list cannot be read here: equal costs until the counters are known)
                const int id = ins_ids ? ins_ids[i] : i;
                if (id >= 0 && id < (int)c0->ins.size())
                    c += 0.02 * (double)c0->meshes[c0->ins[id].m1].n_leaf * (double)c0->meshes[c0->ins[id].m2].n_leaf;
            }
            M->cost[i] = c;
        }
    }
    double total = 0.0;
    for (int i = 0; i < n; ++i) total += M->cost[i];
    if (same_list && M->n_used == k_use) {
        double worst = 0.0;
        for (int k = 0; k < k_use; ++k) {
            double c = 0.0;
            for (int i = M->bound[k]; i < M->bound[k + 1]; ++i) c += M->cost[i];
            if (c > worst) worst = c;
        }
        if (worst * k_use <= 1.15 * total) return;      // still balanced: keep the ranges
    }
    M->bound.assign((size_t)K + 1, n);
    M->bound[0] = 0;
    double acc = 0.0;
    int i = 0;
    for (int k = 1; k < k_use; ++k) {
        const double target = total * (double)k / (double)k_use;
        while (i < n && acc + 0.5 * M->cost[i] < target) acc += M->cost[i++];
        int b = i;
        if (b < M->bound[k - 1] + 1) b = M->bound[k - 1] + 1;      // every used shard gets an item
        if (b > n - (k_use - k)) b = n - (k_use - k);
        while (i < b) acc += M->cost[i++];
        M->bound[k] = b;
    }
    for (int k = k_use; k <= K; ++k) M->bound[k] = n;
    M->n_used = k_use;
    M->part_n = n; M->part_ids = has_ids; M->part_dev = is_dev;
    if (ins_ids && !is_dev) M->part_ins.assign(ins_ids, ins_ids + n); else M->part_ins.clear();
    M->dev_reuse_ok = false;
}

int multi_ensure_host(pfc_context *h, int n) {
    pfc_multi *M = h->multi;
    if (M->h_counts_cap < (size_t)n * 4) {
        if (M->h_counts) (void)hipHostFree(M->h_counts);
        M->h_counts = nullptr; M->h_counts_cap = 0; M->counts_valid = false;
        HIP_TRY(h, hipHostMalloc((void **)&M->h_counts, sizeof(int) * (size_t)n * 8));
        M->h_counts_cap = (size_t)n * 8;
    }
    if ((int)M->iota.size() < n) {
        const int n0 = (int)M->iota.size();
        M->iota.resize((size_t)n * 2);
        for (int i = n0; i < (int)M->iota.size(); ++i) M->iota[i] = i;
    }
    return PFC_OK;
}

void multi_merge_stats(pfc_context *h) {
    pfc_multi *M = h->multi;
    for (int k = 0; k < 8; ++k) M->stats[k] = 0;
    for (int k = 0; k < M->n_used; ++k) {
        const long long *s = M->shard[k]->stats;
        for (int j = 0; j < 4; ++j) M->stats[j] += s[j];
        if (s[4] > M->stats[4]) M->stats[4] = s[4];
        if (s[5] > M->stats[5]) M->stats[5] = s[5];
        M->stats[6] |= s[6];
        M->stats[7] += s[7];
    }
}

int multi_check_args(pfc_context *h, int n_items, const void *ins_ids, const void *pose, const void *twist, const void *wrench,
                     const void *sdot) {
    pfc_context *c0 = h->multi->shard[0];
    if (!c0->finalized) return fail(h, PFC_ERR_STATE, "pfc_eval before pfc_finalize");
    if (n_items < 0) return fail(h, PFC_ERR_BAD_ARG, "negative n_items");
    if (n_items == 0) return PFC_OK;
    if (c0->ins.empty()) return fail(h, PFC_ERR_STATE, "no contact instructions");
    if (!pose || !twist || !wrench || !sdot) return fail(h, PFC_ERR_BAD_ARG, "null buffer");
    if (!ins_ids && n_items > (int)c0->ins.size())
        return fail(h, PFC_ERR_BAD_ARG, "n_items exceeds the number of instructions and no ins_ids given");
    return PFC_OK;
}

// ---- host-pointer entry points -----------------------------------------------------------------------------------
int multi_eval(pfc_context *h, int n_items, const int *ins_ids, const double *pose, const double *twist, const double *s,
               double *wrench, double *sdot, int *counts) {
    pfc_multi *M = h->multi;
    { const int rc = multi_check_args(h, n_items, ins_ids, pose, twist, wrench, sdot); if (rc != PFC_OK || n_items == 0) return rc; }
    if (M->shard[0]->any_bristle && !s) return fail(h, PFC_ERR_BAD_ARG, "bristle instructions need the state buffer s");
    { const int rc = multi_ensure_host(h, n_items); if (rc != PFC_OK) return rc; }
    multi_partition(h, n_items, ins_ids, ins_ids != nullptr, false);
    M->dev_pending = false; M->dev_reuse_ok = false;
    const int *ids = ins_ids ? ins_ids : M->iota.data();
    int *cnt = M->h_counts;
    std::vector<std::function<int()>> jobs((size_t)M->n_used);
    for (int k = 0; k < M->n_used; ++k) {
        const int b0 = M->bound[k], nk = M->bound[k + 1] - b0;
        if (nk <= 0) continue;
        pfc_context *c = M->shard[k];
        const int *idk = (k == 0 && !ins_ids) ? nullptr : ids + b0;
        jobs[k] = [=]() {
            return pfc_eval(c, nk, idk, pose + 24 * (size_t)b0, twist + 6 * (size_t)b0, s ? s + 6 * (size_t)b0 : nullptr,
                            wrench + 6 * (size_t)b0, sdot + 6 * (size_t)b0, cnt + 4 * (size_t)b0);
        };
    }
    const int rc = multi_run(h, jobs);
    M->counts_valid = rc == PFC_OK;
    if (rc != PFC_OK) return rc;
    if (counts) std::memcpy(counts, cnt, sizeof(int) * (size_t)n_items * 4);
    multi_merge_stats(h);
    return PFC_OK;
}

int multi_eval_dual(pfc_context *h, int n_items, int n_dir, const int *ins_ids, const double *pose, const double *bp_pose,
                    const double *twist, const double *s, const double *d_pose, const double *d_twist, const double *d_s,
                    double *wrench, double *sdot, double *d_wrench, double *d_sdot, int *counts) {
    pfc_multi *M = h->multi;
    if (n_dir < 1 || n_dir > 16) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual: n_dir must be in 1..16");
    if (n_items > 0 && (!d_pose || !d_twist || !d_wrench || !d_sdot)) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual: null buffer");
    { const int rc = multi_check_args(h, n_items, ins_ids, pose, twist, wrench, sdot); if (rc != PFC_OK || n_items == 0) return rc; }
    if (M->shard[0]->any_bristle && !s) return fail(h, PFC_ERR_BAD_ARG, "bristle instructions need the state buffer s");
    { const int rc = multi_ensure_host(h, n_items); if (rc != PFC_OK) return rc; }
    multi_partition(h, n_items, ins_ids, ins_ids != nullptr, false);
    M->dev_pending = false; M->dev_reuse_ok = false;
    const int *ids = ins_ids ? ins_ids : M->iota.data();
    int *cnt = M->h_counts;
    const size_t nd = (size_t)n_dir;
    std::vector<std::function<int()>> jobs((size_t)M->n_used);
    for (int k = 0; k < M->n_used; ++k) {
        const size_t b0 = (size_t)M->bound[k];
        const int nk = M->bound[k + 1] - (int)b0;
        if (nk <= 0) continue;
        pfc_context *c = M->shard[k];
        const int *idk = (k == 0 && !ins_ids) ? nullptr : ids + b0;
        jobs[k] = [=]() {
            return pfc_eval_dual_bp(c, nk, n_dir, idk, pose + 24 * b0, bp_pose ? bp_pose + 24 * b0 : nullptr, twist + 6 * b0,
                                    s ? s + 6 * b0 : nullptr, d_pose + 24 * nd * b0, d_twist + 6 * nd * b0,
                                    d_s ? d_s + 6 * nd * b0 : nullptr, wrench + 6 * b0, sdot + 6 * b0, d_wrench + 6 * nd * b0,
                                    d_sdot + 6 * nd * b0, cnt + 4 * b0);
        };
    }
    const int rc = multi_run(h, jobs);
    M->counts_valid = rc == PFC_OK;
    if (rc != PFC_OK) return rc;
    if (counts) std::memcpy(counts, cnt, sizeof(int) * (size_t)n_items * 4);
    multi_merge_stats(h);
    return PFC_OK;
}

// ---- device-pointer entry points ---------------------------------------------------------------------------------
// The enqueueing thread walks over the devices of the list; whichever way an entry point is left (HIP_TRY returns from the middle
// of a shard's enqueue), the caller's thread is back on the first device -- the one its buffers and its stream live on.
struct MultiDeviceGuard {
    int dev;
    explicit MultiDeviceGuard(int d) : dev(d) {}
    ~MultiDeviceGuard() { (void)hipSetDevice(dev); }
};
hipError_t multi_copy(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    if (dst_dev == src_dev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    return hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st);
}

// device iota of shard k covering [0, n)
int multi_dev_iota(pfc_context *h, int k, int n) {
    pfc_multi *M = h->multi;
    pfc_multi::Stage &S = M->stage[k];
    if (S.iota.cap >= (size_t)n) return PFC_OK;
    HIP_TRY(h, S.iota.ensure((size_t)n));        // (multi_ensure_host has made the host iota at least n long)
    HIP_TRY(h, hipMemcpyAsync(S.iota.p, M->iota.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, M->shard[k]->stream));
    HIP_TRY(h, hipStreamSynchronize(M->shard[k]->stream));
    return PFC_OK;
}

// value evaluation (n_dir == 0) or Dual evaluation with every buffer on device dev[0]
int multi_eval_device(pfc_context *h, int n_items, int n_dir, const int *d_ins_ids, const double *d_pose, const double *d_bp,
                      const double *d_twist, const double *d_s, const double *d_dpose, const double *d_dtwist, const double *d_ds,
                      double *d_wrench, double *d_sdot, double *d_dwrench, double *d_dsdot, int *d_counts, void *stream) {
    pfc_multi *M = h->multi;
    { const int rc = multi_check_args(h, n_items, d_ins_ids, d_pose, d_twist, d_wrench, d_sdot); if (rc != PFC_OK) return rc; }
    M->dev_pending = false; M->dev_reuse_ok = false;
    if (n_items == 0) return PFC_OK;
    if (M->shard[0]->any_bristle && !d_s) return fail(h, PFC_ERR_BAD_ARG, "bristle instructions need the state buffer s");
    if (n_dir && (n_dir < 1 || n_dir > 16)) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual_device: n_dir must be in 1..16");
    if (n_dir && (!d_dpose || !d_dtwist || !d_dwrench || !d_dsdot)) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual_device: null buffer");
    { const int rc = multi_ensure_host(h, n_items); if (rc != PFC_OK) return rc; }
    // the item list is device data: the partition is made for (n_items, ids given or not) and balanced by the counters the
    // previous check brought back
    multi_partition(h, n_items, nullptr, d_ins_ids != nullptr, true);
    const int dev0 = M->dev[0];
    HIP_TRY(h, hipSetDevice(dev0));
    MultiDeviceGuard back_to_first(dev0);
    hipStream_t st0 = stream ? (hipStream_t)stream : M->shard[0]->stream;
    const size_t nd = (size_t)n_dir;
    if (M->n_used > 1) HIP_TRY(h, hipEventRecord(M->ev_fork, st0));
    int rc = PFC_OK;
    for (int k = 1; k < M->n_used && rc == PFC_OK; ++k) {
        const size_t b0 = (size_t)M->bound[k];
        const int nk = M->bound[k + 1] - (int)b0;
        if (nk <= 0) continue;
        pfc_context *c = M->shard[k];
        pfc_multi::Stage &S = M->stage[k];
        const int dk = M->dev[k];
        HIP_TRY(h, hipSetDevice(dk));
        hipStream_t sk = c->stream;
        const size_t n = (size_t)nk;
        HIP_TRY(h, S.in.ensure(n * 60)); HIP_TRY(h, S.out.ensure(n * 12)); HIP_TRY(h, S.cnt.ensure(n * 4));
        const int *idk = nullptr;
        if (d_ins_ids) { HIP_TRY(h, S.ids.ensure(n)); idk = S.ids.p; }
        else { const int r2 = multi_dev_iota(h, k, n_items); if (r2 != PFC_OK) return r2; idk = S.iota.p + b0; }
        HIP_TRY(h, hipStreamWaitEvent(sk, M->ev_fork, 0));
        double *ip = S.in.p, *it = ip + n * 24, *is = it + n * 6, *ib = is + n * 6;
        if (d_ins_ids) HIP_TRY(h, multi_copy(S.ids.p, dk, d_ins_ids + b0, dev0, sizeof(int) * n, sk));
        HIP_TRY(h, multi_copy(ip, dk, d_pose + 24 * b0, dev0, sizeof(double) * n * 24, sk));
        HIP_TRY(h, multi_copy(it, dk, d_twist + 6 * b0, dev0, sizeof(double) * n * 6, sk));
        if (d_s) HIP_TRY(h, multi_copy(is, dk, d_s + 6 * b0, dev0, sizeof(double) * n * 6, sk));
        if (d_bp) HIP_TRY(h, multi_copy(ib, dk, d_bp + 24 * b0, dev0, sizeof(double) * n * 24, sk));
        double *ow = S.out.p, *os = ow + n * 6;
        if (n_dir == 0) {
            rc = pfc_eval_device(c, nk, idk, ip, it, d_s ? is : nullptr, ow, os, S.cnt.p, sk);
        } else {
            HIP_TRY(h, S.din.ensure(n * nd * 36)); HIP_TRY(h, S.dout.ensure(n * nd * 12));
            double *dp = S.din.p, *dt = dp + n * nd * 24, *dsd = dt + n * nd * 6;
            HIP_TRY(h, multi_copy(dp, dk, d_dpose + 24 * nd * b0, dev0, sizeof(double) * n * nd * 24, sk));
            HIP_TRY(h, multi_copy(dt, dk, d_dtwist + 6 * nd * b0, dev0, sizeof(double) * n * nd * 6, sk));
            if (d_ds) HIP_TRY(h, multi_copy(dsd, dk, d_ds + 6 * nd * b0, dev0, sizeof(double) * n * nd * 6, sk));
            double *dw = S.dout.p, *dsdot = dw + n * nd * 6;
            rc = pfc_eval_dual_device_bp(c, nk, n_dir, idk, ip, d_bp ? ib : nullptr, it, d_s ? is : nullptr, dp, dt, d_ds ? dsd : nullptr,
                                         ow, os, dw, dsdot, S.cnt.p, sk);
            if (rc == PFC_OK) {
                HIP_TRY(h, multi_copy(d_dwrench + 6 * nd * b0, dev0, dw, dk, sizeof(double) * n * nd * 6, sk));
                HIP_TRY(h, multi_copy(d_dsdot + 6 * nd * b0, dev0, dsdot, dk, sizeof(double) * n * nd * 6, sk));
            }
        }
        if (rc != PFC_OK) { h->err = c->err; break; }
        HIP_TRY(h, multi_copy(d_wrench + 6 * b0, dev0, ow, dk, sizeof(double) * n * 6, sk));
        HIP_TRY(h, multi_copy(d_sdot + 6 * b0, dev0, os, dk, sizeof(double) * n * 6, sk));
        if (d_counts) HIP_TRY(h, multi_copy(d_counts + 4 * b0, dev0, S.cnt.p, dk, sizeof(int) * n * 4, sk));
        HIP_TRY(h, hipEventRecord(S.done, sk));
    }
    HIP_TRY(h, hipSetDevice(dev0));
    if (rc != PFC_OK) return rc;
    {   // shard 0: the caller's arrays in place, on the caller's stream
        const int n0 = M->bound[1];
        pfc_context *c = M->shard[0];
        int *cnt0 = d_counts;
        if (!cnt0) { HIP_TRY(h, M->stage[0].cnt.ensure((size_t)n0 * 4)); cnt0 = M->stage[0].cnt.p; }
        M->dev_counts = cnt0;
        if (n_dir == 0) rc = pfc_eval_device(c, n0, d_ins_ids, d_pose, d_twist, d_s, d_wrench, d_sdot, cnt0, st0);
        else rc = pfc_eval_dual_device_bp(c, n0, n_dir, d_ins_ids, d_pose, d_bp, d_twist, d_s, d_dpose, d_dtwist, d_ds, d_wrench, d_sdot,
                                          d_dwrench, d_dsdot, cnt0, st0);
        if (rc != PFC_OK) { h->err = c->err; return rc; }
    }
    for (int k = 1; k < M->n_used; ++k)
        if (M->bound[k + 1] > M->bound[k]) HIP_TRY(h, hipStreamWaitEvent(st0, M->stage[k].done, 0));
    M->dev_pending = true; M->dev_stream = st0; M->dev_reuse_ndir = n_dir;
    return PFC_OK;
}

int multi_eval_dual_device_more(pfc_context *h, int n_dir, const double *d_dpose, const double *d_dtwist, const double *d_ds,
                                double *d_dwrench, double *d_dsdot, void *stream) {
    pfc_multi *M = h->multi;
    if (n_dir < 1 || n_dir > 16) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual_device_more: n_dir must be in 1..16");
    if (!M->dev_reuse_ok)
        return fail(h, PFC_ERR_STATE, "pfc_eval_dual_device_more: no checked pfc_eval_dual_device evaluation on this handle to extend");
    if (!d_dpose || !d_dtwist || !d_dwrench || !d_dsdot) return fail(h, PFC_ERR_BAD_ARG, "pfc_eval_dual_device_more: null buffer");
    const int dev0 = M->dev[0];
    HIP_TRY(h, hipSetDevice(dev0));
    MultiDeviceGuard back_to_first(dev0);
    hipStream_t st0 = stream ? (hipStream_t)stream : M->shard[0]->stream;
    const size_t nd = (size_t)n_dir;
    if (M->n_used > 1) HIP_TRY(h, hipEventRecord(M->ev_fork, st0));
    int rc = PFC_OK;
    for (int k = 1; k < M->n_used; ++k) {
        const size_t b0 = (size_t)M->bound[k];
        const int nk = M->bound[k + 1] - (int)b0;
        if (nk <= 0) continue;
        pfc_context *c = M->shard[k];
        pfc_multi::Stage &S = M->stage[k];
        const int dk = M->dev[k];
        HIP_TRY(h, hipSetDevice(dk));
        hipStream_t sk = c->stream;
        const size_t n = (size_t)nk;
        HIP_TRY(h, S.din.ensure(n * nd * 36)); HIP_TRY(h, S.dout.ensure(n * nd * 12));
        double *dp = S.din.p, *dt = dp + n * nd * 24, *dsd = dt + n * nd * 6, *dw = S.dout.p, *dsdot = dw + n * nd * 6;
        HIP_TRY(h, hipStreamWaitEvent(sk, M->ev_fork, 0));
        HIP_TRY(h, multi_copy(dp, dk, d_dpose + 24 * nd * b0, dev0, sizeof(double) * n * nd * 24, sk));
        HIP_TRY(h, multi_copy(dt, dk, d_dtwist + 6 * nd * b0, dev0, sizeof(double) * n * nd * 6, sk));
        if (d_ds) HIP_TRY(h, multi_copy(dsd, dk, d_ds + 6 * nd * b0, dev0, sizeof(double) * n * nd * 6, sk));
        rc = pfc_eval_dual_device_more(c, n_dir, dp, dt, d_ds ? dsd : nullptr, dw, dsdot, sk);
        if (rc != PFC_OK) { h->err = c->err; (void)hipSetDevice(dev0); return rc; }
        HIP_TRY(h, multi_copy(d_dwrench + 6 * nd * b0, dev0, dw, dk, sizeof(double) * n * nd * 6, sk));
        HIP_TRY(h, multi_copy(d_dsdot + 6 * nd * b0, dev0, dsdot, dk, sizeof(double) * n * nd * 6, sk));
        HIP_TRY(h, hipEventRecord(S.done, sk));
    }
    HIP_TRY(h, hipSetDevice(dev0));
    rc = pfc_eval_dual_device_more(M->shard[0], n_dir, d_dpose, d_dtwist, d_ds, d_dwrench, d_dsdot, st0);
    if (rc != PFC_OK) { h->err = M->shard[0]->err; return rc; }
    for (int k = 1; k < M->n_used; ++k)
        if (M->bound[k + 1] > M->bound[k]) HIP_TRY(h, hipStreamWaitEvent(st0, M->stage[k].done, 0));
    M->dev_pending = true; M->dev_stream = st0; M->dev_reuse_ndir = -1;      // (-1: a further chunk -- the counters are not brought back again)
    return PFC_OK;
}

int multi_check(pfc_context *h) {
    pfc_multi *M = h->multi;
    if (!M->dev_pending) return PFC_OK;
    M->dev_pending = false;
    const bool more = M->dev_reuse_ndir < 0;
    int rc = PFC_OK;
    // every shard is checked (each grows its own lists on overflow); the first failure in shard order is reported
    for (int k = 0; k < M->n_used; ++k) {
        const int nk = M->bound[k + 1] - M->bound[k];
        if (nk <= 0) continue;
        pfc_context *c = M->shard[k];
        (void)hipSetDevice(M->dev[k]);
        if (!more) {
            const int *src = k == 0 ? M->dev_counts : M->stage[k].cnt.p;
            hipStream_t sk = k == 0 ? M->dev_stream : c->stream;
            if (hipMemcpyAsync(M->h_counts + 4 * (size_t)M->bound[k], src, sizeof(int) * 4 * (size_t)nk, hipMemcpyDeviceToHost, sk) != hipSuccess)
                (void)hipGetLastError();
        }
        const int rk = pfc_check(c);
        if (rk != PFC_OK && rc == PFC_OK) { rc = rk; h->err = c->err; }
    }
    (void)hipSetDevice(M->dev[0]);
    // the caller's stream has the shards' results behind their events: synchronise it too (pfc_check's contract)
    if (M->n_used > 1 && M->dev_stream) {
        const hipError_t e = hipStreamSynchronize(M->dev_stream);
        if (e != hipSuccess && rc == PFC_OK) rc = fail(h, PFC_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
    if (!more) M->counts_valid = rc == PFC_OK;
    if (rc == PFC_OK) {
        if (!more) multi_merge_stats(h);
        M->dev_reuse_ok = more ? true : M->dev_reuse_ndir > 0;
        if (M->dev_reuse_ok)
            for (int k = 0; k < M->n_used; ++k)
                if (M->bound[k + 1] > M->bound[k] && !M->shard[k]->dual_reuse_ok) M->dev_reuse_ok = false;
    }
    return rc;
}

// the shard that evaluated item `item` of the last evaluation and the item's index there
pfc_context *multi_locate(pfc_context *h, int item, int *local) {
    pfc_multi *M = h->multi;
    for (int k = 0; k < M->n_used && k + 1 < (int)M->bound.size(); ++k)
        if (item >= M->bound[k] && item < M->bound[k + 1]) { *local = item - M->bound[k]; return M->shard[k]; }
    return nullptr;
}

void multi_destroy(pfc_context *h) {
    pfc_multi *M = h->multi;
    for (pfc_multi::Worker *w : M->workers) {
        { std::lock_guard<std::mutex> lk(w->m); w->state.store(3, std::memory_order_release); }
        w->cv.notify_one();
        if (w->th.joinable()) w->th.join();
        delete w;
    }
    for (size_t k = 0; k < M->shard.size(); ++k) {
        (void)hipSetDevice(M->dev[k]);
        if (M->shard[k]->stream) (void)hipStreamSynchronize(M->shard[k]->stream);
        if (k < M->stage.size()) {
            pfc_multi::Stage &S = M->stage[k];
            S.in.release(); S.out.release(); S.din.release(); S.dout.release(); S.ids.release(); S.cnt.release(); S.iota.release();
            if (S.done) (void)hipEventDestroy(S.done);
        }
        pfc_destroy(M->shard[k]);
    }
    (void)hipSetDevice(M->dev.empty() ? 0 : M->dev[0]);
    if (M->ev_fork) (void)hipEventDestroy(M->ev_fork);
    if (M->h_counts) (void)hipHostFree(M->h_counts);
    delete M;
    h->multi = nullptr;
}

}  // namespace
