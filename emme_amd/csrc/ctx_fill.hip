// ctx_fill.hip -- the fill dispatcher: one batch of omegas -> which kernels, in which order, with which chunk tables
// (matrixAssembler, include/solver.h:417-515, batched over omega).
#include "ctx.hpp"

namespace emme {

namespace {

int items_per_group_for(const emme_ctx* c, long units) {
    // enough lane groups to give every SIMD several waves, but a few integrals per group
    // when the batch is large so the start-up cost (table staging) is amortised
    const int gw = c->p.integration_start_points == 15 ? 16 : 32;
    const long total = (long)c->npairs * c->nm * units;
    const long target_groups = 256L * 16 * (64 / gw) * 4;
    long ipg = total / target_groups;
    if (ipg < 1) ipg = 1;
    if (ipg > 8) ipg = 8;
    return (int)ipg;
}

}  // namespace

// host_active: which of the nbatch items to assemble (null = all).  Batches of wl_min or
// more items go through the omega-lane kernel, which shares the omega-independent node
// data between items; smaller ones through the lanes-are-nodes kernel.
int do_assemble(emme_ctx* c, int nbatch, const double* d_omega, const int* d_active,
                const int* host_active, double* d_M, const double* d_Mold, double* d_Mp,
                const double* d_domega, const unsigned long long* cost, const double* host_omega, bool newton_loop,
                bool force_uncached) {
    AssembleLaunch L;
    L.P = c->P;
    L.gk_points = c->p.integration_start_points;
    L.nbatch = nbatch;
    L.npairs = c->npairs;
    L.tab = c->d_tab;
    L.pairs = c->d_pairs;
    L.omega = d_omega;
    L.active = d_active;
    L.M = d_M;
    L.Mold = d_Mold;
    L.Mp = d_Mp;
    L.domega = d_domega;
    L.intervals = c->d_intervals;
    L.status = c->d_status;
    L.rounds = c->d_rounds;
    // inside a root search a matrix that already holds a non-finite integral is lost (k_newton_update retires its
    // chain): the fill kernels leave it alone.  Plain assembly calls always get the whole matrix.
    L.skip_lost = newton_loop && c->opt.skip_lost != 0;
    L.union_sel = c->opt.union_sel;
    L.union_walk = c->opt.fill != EMME_FILL_LANES;
    L.coop_wide_min = c->opt.coop_wide_min;
    L.defer_one_group = c->opt.defer_one_group;
    L.dense_min_cols = c->opt.dense_min_cols;
    std::vector<int>& idx = c->h_actidx;
    idx.clear();
    for (int b = 0; b < nbatch; ++b)
        if (!host_active || host_active[b] != 0) idx.push_back(b);
    const int n_act = (int)idx.size();
    if (n_act == 0) return EMME_OK;
    // Items that share a lane group walk the union of their quadrature trees, so a cheap
    // item next to an expensive one costs as much as the expensive one: group items of
    // similar cost (interval count of their previous assembly) together.
    if (cost)
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    // omegas whose level lists overflowed in their previous fill (root search only): first, a chunk each, through the
    // wide-list build of the dense fill
    int n_wide = 0;
    if (newton_loop && c->tiled && c->nm == 1 && !c->h_wide.empty()) {
        std::stable_partition(idx.begin(), idx.end(), [&](int b) { return c->h_wide[b] != 0; });
        for (int b : idx) n_wide += c->h_wide[b] != 0;
    }
    // contour classes present among the omegas (needs their host values)
    // The cache costs a few hundred ms of kernels plus the allocation of up to ~170 GB to build
    // and pays off after ~10 fills: a call with a handful of omegas (a single root of a
    // parameter scan) goes through the on-the-fly kernels unless the cache already exists.
    bool use_cache = !force_uncached && host_omega != nullptr && c->cache_depth != -2 &&
                     (nbatch >= c->opt.cache_min_batch || c->d_recs[0] != nullptr || c->d_recs[1] != nullptr);
    if (use_cache) {
        bool need[2] = {false, false};
        int count[2] = {0, 0};
        for (int b : idx) ++count[-std::copysign(1.0, host_omega[2 * b]) > 0.0 ? 0 : 1];
        need[0] = count[0] > 0, need[1] = count[1] > 0;
        // A contour class that holds only a few of the call's omegas and has no cache yet does not get one for their
        // sake: its main part costs as much to build as for a full batch (N = 512: 32 ms, and another 32 for the first
        // subtree), while those few omegas cost 0.7 ms each through the uncached kernel -- a context that lives for
        // one root search (BASELINE configs[4]: a fresh one per k_rho, one or two of 32 chains on the Re omega > 0
        // side) never earns it back.  The minority goes through the omega-lane kernel in a second pass of this call;
        // once it is more than a sixteenth of the batch its cache is built as before.
        for (int k = 0; k < 2; ++k) {
            if (need[0] && need[1] && !c->d_recs[k] && count[k] * 16 <= n_act && count[k] < count[1 - k]) {
                std::vector<int> major(nbatch, 0), minor(nbatch, 0);
                for (int b : idx) (((-std::copysign(1.0, host_omega[2 * b]) > 0.0 ? 0 : 1) == k) ? minor : major)[b] = 1;
                int rc = do_assemble(c, nbatch, d_omega, d_active, major.data(), d_M, d_Mold, d_Mp, d_domega, cost, host_omega,
                                     newton_loop, false);
                if (rc) return rc;
                const int mode = c->last_fill_mode;  // (the call's fill kernel, as reported, stays the majority's)
                rc = do_assemble(c, nbatch, d_omega, d_active, minor.data(), d_M, d_Mold, d_Mp, d_domega, cost, host_omega,
                                 newton_loop, true);
                c->last_fill_mode = mode;
                return rc;
            }
        }
        for (int k = 0; k < 2 && use_cache; ++k)
            if (need[k]) use_cache = ensure_node_cache(c, L, k);
        // the previous cached fill deferred a sizeable share of its integrals: look at which
        // intervals they were missing and cache a subtree around the most frequent one(s)
        if (use_cache && c->d_worklist_count && c->d_defer_info) {
            if (!c->pub_valid) {  // (the Newton loop gets the count from k_retire through pinned memory)
                HIP_TRY(hipMemcpyAsync(&c->last_deferred, c->d_worklist_count, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
            c->pub_valid = false;
            if (c->last_deferred >= 32) {
                const size_t cnt = std::min<size_t>(c->last_deferred, 1u << 16);
                std::vector<unsigned long long> info(cnt);
                HIP_TRY(hipMemcpy(info.data(), c->d_defer_info, cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                std::sort(info.begin(), info.end());
                // most frequent missing interval per contour class (bit 55 of an entry)
                unsigned long long best[2] = {0, 0};
                size_t best_n[2] = {0, 0}, total[2] = {0, 0};
                for (size_t q = 0; q < cnt;) {
                    size_t e = q;
                    while (e < cnt && info[e] == info[q]) ++e;
                    const int k = (int)((info[q] >> 55) & 1ull);
                    total[k] += e - q;
                    if (e - q > best_n[k]) best_n[k] = e - q, best[k] = info[q];
                    q = e;
                }
                if (std::getenv("EMME_DEBUG"))
                    for (int k = 0; k < 2; ++k)
                        if (total[k])
                            fprintf(stderr, "[emme] deferrals of class %d: %zu, most frequent missing interval depth %d path %llx (%zu)\n", k,
                                    total[k], (int)(best[k] >> 56), best[k] & 0x7fffffffffffffull, best_n[k]);
                for (int k = 0; k < 2; ++k)
                    if (total[k] >= 32 && best_n[k] * 4 >= total[k])
                        add_cache_subtree(c, L, (int)(best[k] >> 56), best[k] & 0x7fffffffffffffull, k);
            }
        }
    }
    if (use_cache) {
        // work list for integrals that outgrow the cache (worst case: every one of them)
        const size_t need = (size_t)c->npairs * c->nm * (size_t)nbatch;
        if (need > c->worklist_cap) {
            if (c->d_worklist) (void)hipFree(c->d_worklist);
            c->d_worklist = nullptr;
            HIP_TRY(malloc_retry((void**)&c->d_worklist, need * sizeof(unsigned long long)));
            if (c->d_defer_info) (void)hipFree(c->d_defer_info);
            c->d_defer_info = nullptr;
            HIP_TRY(malloc_retry((void**)&c->d_defer_info, need * sizeof(unsigned long long)));
            c->worklist_cap = need;
        }
        if (!c->d_worklist_count) HIP_TRY(malloc_retry((void**)&c->d_worklist_count, sizeof(unsigned int)));

    }
    if (use_cache) {
        const int gw = L.gk_points == 15 ? 16 : 32;
        // the union-walk kernel (electrostatic GK15 on folded records, assemble_cached.hip): lanes
        // that sit a round out cost little there, so its chunks are always full and each group
        // takes three items (measured optimum: 86.5 ms vs 104 with the policy below)
        const bool union_walk = (c->opt.fill != EMME_FILL_LANES || c->tiled) && c->folded && c->nm == 1 && L.gk_points == 15;
        // Omega chunks of unequal size.  Every lane walks ONE omega's trees, so an
        // omega whose integrals need 3x the intervals keeps its lane busy 3x longer than its
        // neighbours'.  A chunk of n omegas gives each of them gw/n lanes per group: expensive
        // omegas go into small chunks, cheap ones share a chunk 16 (32) at a time.  idx is
        // sorted by cost, most expensive first, so chunk capacities only grow along the list.
        std::vector<int>& ch = c->h_chunks;
        ch.clear();
        if (!idx.empty()) {
            std::vector<unsigned long long> cs;
            for (int b : idx) cs.push_back(cost ? cost[b] : 1ull);
            std::vector<unsigned long long> sorted = cs;
            std::sort(sorted.begin(), sorted.end());
            const double typical = (double)std::max<unsigned long long>(sorted[sorted.size() / 2], 1ull);
            // dense fill: one wave walks a (16-pair tile, chunk) serially, so (a) a chunk of omegas whose
            // trees do not overlap costs the SUM of their walks in one wave -- expensive omegas get narrow
            // chunks like in the independent-lane kernels -- and (b) a launch needs several times more
            // tile tasks than the chip holds waves: the widest chunk shrinks until there are at least
            // EMME_DENSE_MIN_TASKS (2000; 8000 while every lane ended with a global atomic -- with the counters
            // summed per workgroup 0 .. 3000 are equal, 44.7 ms of fill per bench search, and 8000 costs 45.8)
            // (dense fill: a chunk is 16 COLUMNS -- 16 omegas, or 5 omegas x 3 moments)
            const int tile_cap = 16 / c->nm;
            int dense_cap = c->tiled ? tile_cap : gw;
            if (c->tiled) {
                const long ntiles = (c->npairs + 15) / 16;
                const long min_tasks = c->opt.dense_min_tasks;
                while (dense_cap > 2 && ((long)idx.size() + dense_cap - 1) / dense_cap * ntiles < min_tasks) dense_cap >>= 1;
            }
            const double dense_ratio = c->opt.dense_cost_ratio;
            size_t q = 0;
            for (; q < (size_t)n_wide; ++q) ch.push_back((int)q), ch.push_back(1);
            while (q < idx.size()) {
                int cap = c->tiled ? dense_cap : gw;
                while ((!union_walk || c->tiled) && cap > (c->tiled ? 2 : 1) &&
                       (double)cs[q] * cap > typical * (c->tiled ? tile_cap * dense_ratio : gw * 1.5))
                    cap >>= 1;
                const int n = (int)std::min<size_t>((size_t)cap, idx.size() - q);
                ch.push_back((int)q);
                ch.push_back(n);
                q += (size_t)n;
            }
        }
        const int nchunks = (int)ch.size() / 2;
        const int n_lane = (int)idx.size();
        L.items_per_group = union_walk ? 3 : items_per_group_for(c, nchunks > 0 ? nchunks : 1);
        if (union_walk) {
            // up to three chunks (late Newton steps: <= 48 omegas) leave the SIMDs short of waves
            // with three items per group: two then (measured: one is worse again -- every
            // workgroup stages the grid tables; EMME_UNION_IPG_FEW / EMME_UNION_FEW_CHUNKS)
            const int ipg_few = c->opt.union_ipg_few, few = c->opt.union_few_chunks;
            if (nchunks <= few) L.items_per_group = std::max(1, ipg_few);
        }
        if (n_lane) {
            // omega order | chunk table: into a pinned slot, then ONE small kernel moves both to the device
            int* slot = c->p_lists + (size_t)(c->p_lists_turn++ & 1u) * c->p_lists_cap;
            std::copy(idx.begin(), idx.end(), slot);
            std::copy(ch.begin(), ch.end(), slot + n_lane);
            int n2 = (int)ch.size();
            if (c->tiled) {  // dense fill: position -> (chunk, column) map behind the chunk table
                for (int k = 0; k < nchunks; ++k)
                    for (int w = 0; w < ch[2 * k + 1]; ++w) slot[n_lane + n2 + ch[2 * k] + w] = (k << 8) | w;
                n2 += n_lane;
            }
            HIP_TRY(launch_stage_ints(slot, c->d_actidx, n_lane, c->d_chunks, n2, c->stream));
        }
        HIP_TRY(hipMemsetAsync(c->d_worklist_count, 0, sizeof(unsigned int), c->stream));
        c->last_fill_mode = c->tiled ? 4 : (union_walk ? 3 : 2);
        if (n_lane && c->tiled) {
            // dense fill: weighted phase tables for every cached interval and omega chunk, then one wave
            // per (16-pair tile, 16-omega chunk); chunk c = positions 16 c .. of the cost-sorted list
            const int n_int = node_cache_intervals(c->cache_geom);
            const int nch = nchunks;
            const size_t need = btab_bytes(n_int, nch, L.gk_points);
            if (need > c->btab_cap) {
                if (c->d_btab) (void)hipFree(c->d_btab);
                c->d_btab = nullptr, c->btab_cap = 0;
                HIP_TRY(malloc_retry(&c->d_btab, need + need / 4));
                c->btab_cap = need + need / 4;
            }
            {
                ScopedSpan s(c, K_OTHER);
                HIP_TRY(launch_btab(L.gk_points, c->nm, n_int, c->d_ttab, c->d_wtab, d_omega, c->d_actidx, n_lane,
                                    c->d_chunks + 2 * nchunks, nchunks, c->d_btab, c->stream));
            }
            {
                ScopedSpan s(c, K_ASM);
                static const bool stamps = std::getenv("EMME_DEBUG_STAMPS") != nullptr;
                hipEvent_t e0 = nullptr, e1 = nullptr;
                unsigned long long r0[16] = {};
                if (stamps) {  // diagnostic (EMME_DENSE_STAMPS build): this launch's tasks, their total and longest time
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    HIP_TRY(hipMemcpy(r0, c->d_rounds, sizeof r0, hipMemcpyDeviceToHost));
                    const unsigned long long zero = 0;
                    HIP_TRY(hipMemcpy(c->d_rounds + 9, &zero, sizeof zero, hipMemcpyHostToDevice));
                    HIP_TRY(hipEventCreate(&e0));
                    HIP_TRY(hipEventCreate(&e1));
                    HIP_TRY(hipEventRecord(e0, c->stream));
                }
                HIP_TRY(launch_assemble_dense(L, c->cache_geom, c->d_recs, c->d_recs_ext, c->d_scale, c->d_btab,
                                              c->d_worklist, c->d_worklist_count, c->d_defer_info, c->d_actidx, n_lane,
                                              c->d_chunks, nchunks, c->d_rounds, c->stream, c->d_tile_poison,
                                              (c->opt.dense_wide && c->nm == 1) ? nchunks : n_wide, newton_loop ? c->d_overflow : nullptr));
                if (stamps) {
                    HIP_TRY(hipEventRecord(e1, c->stream));
                    HIP_TRY(hipStreamSynchronize(c->stream));
                    float ms = 0.f;
                    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
                    unsigned long long r1[16] = {};
                    HIP_TRY(hipMemcpy(r1, c->d_rounds, sizeof r1, hipMemcpyDeviceToHost));
                    const double tasks = (double)(r1[3] - r0[3]), tot = (double)(r1[8] - r0[8]);
                    fprintf(stderr, "[emme] dense launch: %d omegas in %d chunks, %.0f tasks, %.3f ms; task ticks: mean %.0f, longest %.0f, "
                            "sum / 2048 wave slots %.0f; rounds dense %llu sparse %llu\n", n_lane, nchunks, tasks, ms,
                            tasks > 0 ? tot / tasks : 0.0, (double)r1[9], tot / 2048.0, r1[0] - r0[0], r1[1] - r0[1]);
                    {   // how the tiles' times are spread (all chunks of the launch added up per tile)
                        const size_t nt = std::min<size_t>(((size_t)c->npairs + 15) / 16, 8192);
                        std::vector<unsigned long long> tt(nt);
                        HIP_TRY(hipMemcpy(tt.data(), c->d_rounds + 16, nt * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                        HIP_TRY(hipMemset(c->d_rounds + 16, 0, nt * sizeof(unsigned long long)));
                        // first, middle and last tiles in index order, then percentiles
                        fprintf(stderr, "[emme]   tile ticks by index: %llu %llu %llu %llu %llu | ", tt[0], tt[nt / 4], tt[nt / 2], tt[3 * nt / 4], tt[nt - 1]);
                        std::sort(tt.begin(), tt.end());
                        fprintf(stderr, "sorted: min %llu p25 %llu p50 %llu p75 %llu p90 %llu p97 %llu max %llu\n", tt[0], tt[nt / 4], tt[nt / 2],
                                tt[3 * nt / 4], tt[nt * 9 / 10], tt[nt * 97 / 100], tt[nt - 1]);
                    }
                    (void)hipEventDestroy(e0);
                    (void)hipEventDestroy(e1);
                }
            }
        } else if (n_lane && c->folded) {
            // phase table of this launch: exp(T omega) for every cached interval, node and omega
            const int n_int = node_cache_intervals(c->cache_geom);
            const size_t need = (size_t)n_lane * n_int * gw * 2 * sizeof(double);
            if (need > c->etab_bytes) {
                if (c->d_etab) (void)hipFree(c->d_etab);
                c->d_etab = nullptr, c->etab_bytes = 0;
                HIP_TRY(malloc_retry(&c->d_etab, need + need / 4));
                c->etab_bytes = need + need / 4;
            }
            ScopedSpan s(c, K_OTHER);
            HIP_TRY(launch_phase_table(L.gk_points, n_int, c->d_ttab, d_omega, c->d_actidx, n_lane, c->d_etab,
                                       c->stream));
        }
        if (n_lane && !c->tiled) {
            ScopedSpan s(c, K_ASM);
            const void* etab = c->folded ? c->d_etab : nullptr;
            if (c->em_shared)
                HIP_TRY(launch_assemble_cached_em(L, c->cache_geom, c->d_recs, c->d_recs_ext, c->d_ttab, c->d_wtab,
                                                  c->d_scale, etab, c->d_worklist, c->d_worklist_count, c->d_defer_info,
                                                  c->d_actidx, n_lane, c->d_chunks, nchunks, c->stream));
            else
                HIP_TRY(launch_assemble_cached(L, c->cache_geom, c->d_recs, c->d_recs_ext, c->d_ttab, c->d_scale, etab,
                                               c->d_worklist, c->d_worklist_count, c->d_defer_info, c->d_actidx,
                                               n_lane, c->d_chunks, nchunks, c->stream));
        }
        if (n_lane) {
            ScopedSpan s(c, K_DEFER);
            // (tiled electromagnetic / GK31 contexts: the cooperative kernel reads the electrostatic GK15 tile blocks
            // only -- the few integrals that leave the cache are evaluated from scratch)
            const bool coop_cached = !(c->tiled && (c->nm > 1 || L.gk_points != 15));
            HIP_TRY(launch_assemble_list(L, c->d_worklist, c->d_worklist_count, coop_cached ? &c->cache_geom : nullptr, c->d_recs,
                                         c->d_recs_ext, c->d_ttab, c->em_shared ? c->d_wtab : nullptr, c->folded, c->stream,
                                         c->tiled && coop_cached, c->d_tile_poison));
        }
        if (std::getenv("EMME_DEBUG")) {
            unsigned int cnt = 0;
            (void)hipMemcpy(&cnt, c->d_worklist_count, sizeof cnt, hipMemcpyDeviceToHost);
            std::vector<unsigned long long> wl(cnt < 8 ? cnt : 8);
            if (!wl.empty()) (void)hipMemcpy(wl.data(), c->d_worklist, wl.size() * 8, hipMemcpyDeviceToHost);
            fprintf(stderr, "[emme] cached fill: %d items, %u integrals deferred (of %ld)", n_act, cnt,
                    (long)c->npairs * c->nm * n_act);
            std::vector<unsigned long long> dg(wl.size());
            if (!wl.empty() && c->d_defer_info) (void)hipMemcpy(dg.data(), c->d_defer_info, wl.size() * 8, hipMemcpyDeviceToHost);
            for (size_t q = 0; q < wl.size(); ++q)
                fprintf(stderr, " b%llu:i%llu@d%llu:p%llx", wl[q] >> 32, wl[q] & 0xffffffffull, dg[q] >> 56,
                        dg[q] & 0x7fffffffffffffull);
            fprintf(stderr, "\n");
        }
    } else if (n_act >= c->opt.wl_min || force_uncached) {
        const int gw = L.gk_points == 15 ? 16 : 32;
        L.items_per_group = items_per_group_for(c, (n_act + gw - 1) / gw);
        {
            int* slot = c->p_lists + (size_t)(c->p_lists_turn++ & 1u) * c->p_lists_cap;
            std::copy(idx.begin(), idx.end(), slot);
            HIP_TRY(launch_stage_ints(slot, c->d_actidx, n_act, nullptr, 0, c->stream));
        }
        c->last_fill_mode = 1;
        ScopedSpan s(c, K_ASM);
        HIP_TRY(launch_assemble_wl(L, c->d_actidx, n_act, c->stream));
    } else {
        L.items_per_group = items_per_group_for(c, nbatch);
        c->last_fill_mode = 0;
        ScopedSpan s(c, K_ASM);
        HIP_TRY(launch_assemble(L, c->stream));
    }
    c->acc.matrices += n_act;
    return EMME_OK;
}

}  // namespace emme
