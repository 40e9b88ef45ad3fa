// ctx_cache.hip -- host side of the HBM node cache: the process-wide pool of its buffers, the choice
// of the cached geometry, the builds of the main part and of run-time subtrees.
#include "ctx.hpp"

namespace emme {

// Process-wide pool of the big node-cache buffers.  Allocating ~150 GB takes seconds, far longer
// than filling it, and a parameter sweep creates one context per parameter set: buffers of a
// destroyed context are kept and handed to the next one (the records are recomputed anyway).
struct PoolEntry {
    void* ptr;
    size_t bytes;
    int device;
};
std::mutex g_pool_mu;
std::vector<PoolEntry> g_pool;

void pool_release_all() {
    std::lock_guard<std::mutex> g(g_pool_mu);
    for (auto& e : g_pool) {
        (void)hipSetDevice(e.device);
        (void)hipFree(e.ptr);
    }
    g_pool.clear();
}

hipError_t pool_alloc(void** out, size_t bytes, int device) {
    {
        std::lock_guard<std::mutex> g(g_pool_mu);
        int best = -1;
        for (int k = 0; k < (int)g_pool.size(); ++k)
            if (g_pool[k].device == device && g_pool[k].bytes >= bytes && g_pool[k].bytes <= bytes + bytes / 4 &&
                (best < 0 || g_pool[k].bytes < g_pool[best].bytes))
                best = k;
        if (best >= 0) {
            *out = g_pool[best].ptr;
            g_pool.erase(g_pool.begin() + best);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {  // give the pooled memory back to the driver and try once more
        (void)hipGetLastError();
        pool_release_all();
        e = hipMalloc(out, bytes);
    }
    return e;
}

// hipMalloc that gives pooled cache buffers back to the driver before reporting out-of-memory
hipError_t malloc_retry(void** out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        pool_release_all();
        e = hipMalloc(out, bytes);
    }
    return e;
}

void pool_free(void* p, size_t bytes, int device) {
    if (!p) return;
    std::lock_guard<std::mutex> g(g_pool_mu);
    g_pool.push_back({p, bytes, device});
    // keep at most ~one large context's worth; evict the oldest buffers beyond that
    size_t total = 0;
    for (const auto& e : g_pool) total += e.bytes;
    while (total > (size_t)200e9 && !g_pool.empty()) {
        (void)hipSetDevice(g_pool.front().device);
        (void)hipFree(g_pool.front().ptr);
        total -= g_pool.front().bytes;
        g_pool.erase(g_pool.begin());
    }
    (void)hipSetDevice(device);
}

// items the node cache is indexed by: (pair, moment), or pairs alone in the shared EM layout
long cache_items(const emme_ctx* c) { return (long)c->npairs * (c->em_shared ? 1 : c->nm); }
// bytes of one part of the node cache in this context's record layout
size_t cache_part_bytes(const emme_ctx* c, int gk_points, const NodeCacheGeom& g, int part) {
    return c->tiled ? node_cache_bytes_tiled(c->npairs, g, part, gk_points) : node_cache_bytes(gk_points, cache_items(c), g, part);
}
hipError_t build_cache_part(emme_ctx* c, const AssembleLaunch& L, const NodeCacheGeom& g, int part, int cls, void* recs) {
    const double omi = cls == 0 ? 1.0 : -1.0;
    if (c->tiled) {
        if (!c->d_tile_poison[cls]) {
            const size_t ntiles = ((size_t)c->npairs + 15) / 16;
            if (malloc_retry((void**)&c->d_tile_poison[cls], ntiles) != hipSuccess) return hipErrorOutOfMemory;
            const hipError_t e = hipMemsetAsync(c->d_tile_poison[cls], 0, ntiles, c->stream);
            if (e != hipSuccess) return e;
        }
        hipError_t e = launch_node_cache_tiled(L, g, part, omi, recs, c->d_ttab[cls], c->d_scale, c->stream, c->d_tile_poison[cls],
                                               c->nm > 1 ? c->d_wtab[cls] : nullptr);
        if (e == hipSuccess && std::getenv("EMME_DEBUG")) {
            const size_t ntiles = ((size_t)c->npairs + 15) / 16;
            std::vector<unsigned char> flags(ntiles);
            (void)hipMemcpyAsync(flags.data(), c->d_tile_poison[cls], ntiles, hipMemcpyDeviceToHost, c->stream);
            (void)hipStreamSynchronize(c->stream);
            int n_poison = 0;
            for (unsigned char f : flags) n_poison += f != 0;
            fprintf(stderr, "[emme] node cache: class %d part %d built; tiles with a poisoned block so far: %d\n", cls, part, n_poison);
        }
        return e;
    }
    return launch_node_cache(L, g, part, omi, recs, c->d_ttab[cls], c->d_wtab[cls], c->d_scale, c->folded, c->stream);
}

// Make sure the main part of the node cache of contour class `cls` (0: omi=+1, 1: omi=-1)
// exists.  Returns false (and disables the cache) if it does not fit the budget.
bool ensure_node_cache(emme_ctx* c, const AssembleLaunch& L, int cls) {
    if (c->cache_depth == -2) return false;
    const double budget = c->opt.node_cache_gb * (double)(1 << 30);
    if (c->cache_depth == -1) {
        // full tree to depth dfull + the fixed subtree under the rightmost depth-5 node (that is
        // where ordinary damped roots refine); the largest that leaves half the budget free
        static const int options[][3] = {{8, 5, 13}, {7, 5, 13}, {6, 5, 13}, {6, 5, 12}, {6, 5, 11},
                                         {5, 4, 10}, {5, 4, 9},  {4, 3, 7},  {3, 2, 5}};
        // A cache that reaches less deep than depth 5 sends most integrals of a damped omega to the
        // from-scratch kernel and is SLOWER than no cache at all (measured, N = 1024, where only depth 4
        // fits: 83 omega-points/s with that cache against 335 through the omega-lane kernel,
        // profiles/r02_size_sweep.jsonl): below that the context runs uncached (EMME_CACHE_MIN_DEPTH)
        // (tiled contexts: 6 -- their records are a third smaller, so depth 5 does fit at N = 1024, and is
        // as bad there: 97 omega-points/s)
        const int min_depth = c->opt.cache_min_depth > 0 ? c->opt.cache_min_depth : (c->tiled ? 6 : 5);
        bool found = false;
        for (const auto& o : options) {
            if (o[0] < min_depth) break;
            NodeCacheGeom g{};
            g.dfull = o[0], g.nsub = 1, g.rd[0] = o[1], g.dd[0] = o[2], g.rp[0] = (1ull << o[1]) - 1ull;
            if ((double)cache_part_bytes(c, L.gk_points, g, -1) <= 0.25 * budget) {
                c->cache_geom = g;
                found = true;
                break;
            }
        }
        if (!found) {
            c->cache_depth = -2;
            return false;
        }
        c->cache_depth = c->cache_geom.dfull;
        c->cache_max_intervals = node_cache_intervals(c->cache_geom) + (NODE_CACHE_MAX_SUB - 1) * 511;
    }
    if (c->d_recs[cls]) return true;
    const size_t bytes = cache_part_bytes(c, L.gk_points, c->cache_geom, -1);
    AllocTimer at(c);
    if (c->cache_bytes_used + (double)bytes > budget ||
        pool_alloc(&c->d_recs[cls], bytes, c->device) != hipSuccess ||
        malloc_retry(&c->d_ttab[cls], node_ttab_bytes(L.gk_points, c->cache_max_intervals)) != hipSuccess ||
        ((c->em_shared || (c->tiled && c->nm > 1)) &&
         malloc_retry(&c->d_wtab[cls], node_ttab_bytes(L.gk_points, c->cache_max_intervals)) != hipSuccess) ||
        (!c->d_scale &&
         malloc_retry((void**)&c->d_scale, sizeof(double) * c->cache_max_intervals) != hipSuccess)) {
        (void)hipGetLastError();
        pool_free(c->d_recs[cls], bytes, c->device);
        c->d_recs[cls] = nullptr;
        c->cache_depth = -2;  // fall back to the on-the-fly kernels for good
        return false;
    }
    at.stop();
    c->cache_bytes_used += (double)bytes;
    c->recs_bytes[cls] = bytes;
    ScopedSpan s(c, K_CACHE);
    if (build_cache_part(c, L, c->cache_geom, -1, cls, c->d_recs[cls]) != hipSuccess) {
        c->cache_depth = -2;
        return false;
    }
    return true;
}

// Which registered subtree covers interval (depth, path)?  -1 if none.
int find_subtree(const NodeCacheGeom& g, int depth, unsigned long long path) {
    for (int k = 0; k < g.nsub; ++k)
        if (depth >= g.rd[k] && depth <= g.dd[k] && (path >> (depth - g.rd[k])) == g.rp[k]) return k;
    return -1;
}

// The previous cached fill deferred integrals of contour class `cls` because interval
// (depth, path) was not cached for that class: build the subtree that covers it for this class,
// registering a new one around it (root 4 levels up, 8 levels deep = 511 intervals) if none does.
// Subtrees are built per class, on demand: the few omegas on the other side of the imaginary
// axis do not get 12 GB copies of regions they never visit.  Failure is harmless: those
// integrals keep going through the work list.
void add_cache_subtree(emme_ctx* c, const AssembleLaunch& L, int depth, unsigned long long path, int cls) {
    NodeCacheGeom& g = c->cache_geom;
    if (c->ext_failed || !c->d_recs[cls]) return;
    if (depth <= g.dfull) return;  // (inside the full tree: a poisoned tile's hand-over, not a missing interval)
    int k = find_subtree(g, depth, path);
    const bool fresh = k < 0;
    if (k == 0) return;  // the fixed subtree lives in the main buffer: nothing to add
    if (fresh) {
        if (g.nsub >= NODE_CACHE_MAX_SUB) return;
        const int rd = depth - 4 < 1 ? 1 : depth - 4;
        k = g.nsub;
        g.rd[k] = rd;
        g.rp[k] = path >> (depth - rd);
        g.dd[k] = rd + 8;
        g.nsub = k + 1;
    } else if (c->d_recs_ext[cls][k - 1]) {
        return;  // already there (the deferral was for an interval deeper than the subtree)
    }
    const double budget = c->opt.node_cache_gb * (double)(1 << 30);
    const size_t eb = cache_part_bytes(c, L.gk_points, g, k - 1);
    AllocTimer at(c);
    if (c->cache_bytes_used + (double)eb > budget ||
        pool_alloc(&c->d_recs_ext[cls][k - 1], eb, c->device) != hipSuccess) {
        (void)hipGetLastError();
        c->d_recs_ext[cls][k - 1] = nullptr;
        c->ext_failed = true;
        if (fresh) g.nsub = k;  // nothing built: forget the registration
        return;
    }
    at.stop();
    c->cache_bytes_used += (double)eb;
    c->recs_ext_bytes[cls][k - 1] = eb;
    {
        ScopedSpan s(c, K_CACHE);
        if (build_cache_part(c, L, g, k - 1, cls, c->d_recs_ext[cls][k - 1]) != hipSuccess) c->ext_failed = true;
    }
    if (std::getenv("EMME_DEBUG"))
        fprintf(stderr, "[emme] node cache: subtree %d (depth %d path %llx, to depth %d) built for class %d, %.1f GiB in use\n",
                k, g.rd[k], g.rp[k], g.dd[k], cls, c->cache_bytes_used / (double)(1 << 30));
}

}  // namespace emme
