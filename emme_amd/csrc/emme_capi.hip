// emme_capi.hip -- implementation of the C ABI declared in include/emme_hip.h:
// context (device tables, batch scratch, stream, profiling) and the batched drivers that
// stand where the reference has EigenSolver's constructor / matrixAssembler /
// newtonTraceSecantIteration (include/solver.h:396-415, 417-515, 113-160) and the
// solve_once_eigen loop (src/main.cpp:19-80).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/emme_hip.h"
#include "launch.hpp"

namespace emme {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

namespace {

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));             \
            return e_ == hipErrorOutOfMemory ? EMME_ENOMEM : EMME_EDEVICE;             \
        }                                                                              \
    } while (0)

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain host memory: clear the sticky error
        return false;
    }
    return attr.type == hipMemoryTypeDevice;
}

enum Kind { K_ASM = 0, K_LIN = 1, K_OTHER = 2, K_DEFER = 3, K_CACHE = 4, K_NULL = 5 };

}  // namespace
}  // namespace emme

using namespace emme;

struct emme_ctx {
    emme_params_t p;
    int device = 0;
    hipStream_t stream = nullptr;
    DevParams P;
    int N = 0, dim = 0, nm = 1, npairs = 0;
    double* d_tab = nullptr;
    ushort2* d_pairs = nullptr;
    // batch scratch
    int cap = 0;
    double *d_omega = nullptr, *d_domega = nullptr, *d_tr = nullptr;
    int *d_active = nullptr, *d_iters = nullptr, *d_info = nullptr, *d_status = nullptr;
    unsigned long long* d_intervals = nullptr;
    unsigned long long* d_rounds = nullptr;  // diagnostic counter of the omega-lane kernel
    int* d_actidx = nullptr;   // compacted list of batch items for the omega-lane kernel
    int* d_chunks = nullptr;   // (first, size) of every omega chunk of the cached kernel
    std::vector<int> h_chunks;
    std::vector<int> h_actidx; // its host image (kept alive across the async upload)
    int last_fill_mode = -1;   // kernel family of the last fill: 0 nodes, 1 omega-lane, 2 cached
    emme_options_t opt{};      // per-context options (emme_options_t; environment overrides applied at creation)
    // HBM cache of omega-independent node records, per contour class (omi = +1, -1)
    int cache_depth = -1;      // -1: not decided yet, -2: disabled / does not fit, else dfull
    NodeCacheGeom cache_geom{};
    int cache_max_intervals = 0;  // capacity of the T / scale tables
    void* d_recs[2] = {nullptr, nullptr};      // main part per contour class
    size_t recs_bytes[2] = {0, 0};
    size_t recs_ext_bytes[2][NODE_CACHE_MAX_SUB - 1] = {};
    void* d_recs_ext[2][NODE_CACHE_MAX_SUB - 1] = {};  // run-time subtrees per class
    void* d_ttab[2] = {nullptr, nullptr};      // T table per class
    void* d_wtab[2] = {nullptr, nullptr};      // moment-factor table per class (shared EM layout)
    unsigned char* d_tile_poison[2] = {nullptr, nullptr};  // tiled layout: tiles that hold a poisoned block, per class
    bool em_shared = false;    // nm == 3: one record per (pair, interval, node), three moments per lane
    bool folded = true;        // records carry exp(A0); exp(T omega) comes from a per-launch phase table
    bool tiled = false;        // electrostatic GK15: tiled record layout + dense (matrix-core) fill
    void* d_btab = nullptr;    // weighted phase tables of the current launch (dense fill)
    size_t btab_cap = 0;
    void* d_etab = nullptr;    // phase table of the current launch
    size_t etab_bytes = 0;
    int* h_lu_items = nullptr;     // blocked LU: the live matrices of the launch (pinned host / device)
    int* d_lu_items = nullptr;
    int lu_items_cap = 0;
    void* d_lu_scratch = nullptr;  // blocked LU: diagonal of X, hand-over flags, row-map snapshots
    size_t lu_scratch_bytes = 0;
    int n_cu = 256;                // compute units of the device
    int last_lu_nwg = 1;           // workgroups per matrix of the last LU launch
    bool lu_one_wg = false;        // a hand-over of the multi-workgroup LU timed out once: never again
    int* p_act = nullptr;          // pinned host copies of d_active / d_intervals / omega / the deferred
    unsigned long long* p_iv = nullptr;  // count, WRITTEN BY KERNELS (k_retire, k_newton_update): the
    double* p_w = nullptr;         // Newton loop reads them after its one synchronisation per step
    unsigned int* p_deferred = nullptr;
    unsigned int* d_overflow = nullptr;  // per item: integrals that left the dense fill because a level list was full
    unsigned int* p_overflow = nullptr;  // ... published by k_retire
    std::vector<unsigned char> h_wide;   // items whose chunks take the 128-entry build of the dense fill (root search)
    int p_cap = 0;
    bool pub_valid = false;        // last_deferred holds the previous fill's count (from p_deferred)
    int* p_lists = nullptr;        // pinned staging of the per-launch lists (omega order | chunks), two
    int p_lists_cap = 0;           // slots used in turn; k_stage_ints moves a slot to device memory
    unsigned int p_lists_turn = 0;
    unsigned int lu_items_turn = 0;
    bool ext_failed = false;
    unsigned long long* d_defer_info = nullptr;  // missing interval of every deferred integral
    double cache_bytes_used = 0.0;
    unsigned int last_deferred = 0;            // integrals the previous cached fill deferred
    double* d_scale = nullptr;  // half-widths of the cached intervals
    unsigned long long* d_worklist = nullptr;  // integrals deferred to the cooperative kernel
    unsigned int* d_worklist_count = nullptr;
    size_t worklist_cap = 0;
    int mat_cap = 0;  // matrices per set
    double *d_M = nullptr, *d_Mold = nullptr, *d_Mp = nullptr, *d_work = nullptr;
    double* d_iterates = nullptr;
    size_t iterates_cap = 0;
    int last_n = 0;
    // profiling
    bool prof = false;
    emme_profile_t acc{};
    struct Span {
        int kind;
        hipEvent_t a, b;
    };
    std::vector<Span> spans;
    std::vector<hipEvent_t> free_events;
};

namespace {

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

// host wall time of the cache allocations (hipMalloc of tens of GB: the cold cost of a context)
struct AllocTimer {
    emme_ctx* c;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool running = true;
    explicit AllocTimer(emme_ctx* ctx) : c(ctx) {}
    void stop() {
        if (running) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            c->acc.cache_alloc_ms += ms;
            if (std::getenv("EMME_DEBUG")) fprintf(stderr, "[emme] node cache: allocation took %.1f ms\n", ms);
        }
        running = false;
    }
    ~AllocTimer() { stop(); }
};

size_t mat_doubles(const emme_ctx* c) { return (size_t)c->dim * c->dim * 2; }

// Which contexts get the tiled record layout + dense (matrix-core) fill (assemble_dense.hip): electrostatic GK15 and
// electromagnetic GK31 (the two shapes BASELINE.json's configurations use), on folded records, with the default fill
// option.  The dense path carries the safe_exp-clamped tails (<= 4e-14 absolute), so inputs whose absolute quadrature
// goal (integration_accuracy) is tighter than 1e-9 keep the exact kernels.
bool wants_tiled(const emme_params_t& p, bool es, bool folded, int fill) {
    const bool shape = (es && p.integration_start_points == 15) || (!es && p.integration_start_points == 31);
    return shape && folded && p.integration_accuracy >= 1e-9 && fill == EMME_FILL_AUTO;
}

void options_default(emme_options_t& o) {
    o = emme_options_t{};
    o.size = (int)sizeof(emme_options_t);
    o.node_cache_gb = 176.0;  // both contour classes together (MI355X: 288 GB of HBM3E)
    o.cache_min_batch = 8;
    o.cache_min_depth = 0;
    o.fill = EMME_FILL_AUTO;
    o.phase_table = 1;
    o.em_shared = 1;
    o.wl_min = 4;
    o.union_sel = 2;
    o.union_ipg_few = 2, o.union_few_chunks = 3;
    o.coop_wide_min = 4096;
    o.defer_one_group = 0;
    o.dense_min_cols = 3;
    o.dense_min_tasks = 2000;
    o.dense_cost_ratio = 4.0;
    o.dense_wide = 0;
    o.skip_lost = 1;
    o.lu_split = 0;
    o.lu_group_min_n = 256;
    o.lu_spin_limit = 16000000;  // about 4 s
    o.lu_unblocked = 0;
}

// The EMME_* environment variables: developer overrides, read ONCE per context (at creation), winning over
// the caller's struct.  Library callers use emme_options_t.
void options_env_overrides(emme_options_t& o) {
    auto geti = [](const char* name, int& v) {
        if (const char* e = std::getenv(name)) v = std::atoi(e);
    };
    auto getd = [](const char* name, double& v) {
        if (const char* e = std::getenv(name)) v = std::atof(e);
    };
    getd("EMME_NODE_CACHE_GB", o.node_cache_gb);
    geti("EMME_CACHE_MIN_BATCH", o.cache_min_batch);
    geti("EMME_CACHE_MIN_DEPTH", o.cache_min_depth);
    if (const char* e = std::getenv("EMME_DENSE"))
        if (std::atoi(e) == 0 && o.fill == EMME_FILL_AUTO) o.fill = EMME_FILL_UNION;
    if (const char* e = std::getenv("EMME_UNION"))
        if (std::atoi(e) == 0) o.fill = EMME_FILL_LANES;
    geti("EMME_PHASE_TABLE", o.phase_table);
    geti("EMME_EM_SHARED", o.em_shared);
    geti("EMME_WL_MIN", o.wl_min);
    geti("EMME_UNION_SEL", o.union_sel);
    geti("EMME_UNION_IPG_FEW", o.union_ipg_few);
    geti("EMME_UNION_FEW_CHUNKS", o.union_few_chunks);
    geti("EMME_COOP_WIDE_MIN", o.coop_wide_min);
    if (std::getenv("EMME_DEFER_ONE_GROUP")) o.defer_one_group = 1;
    geti("EMME_DENSE_MIN_COLS", o.dense_min_cols);
    geti("EMME_DENSE_MIN_TASKS", o.dense_min_tasks);
    getd("EMME_DENSE_COST_RATIO", o.dense_cost_ratio);
    geti("EMME_DENSE_WIDE", o.dense_wide);
    geti("EMME_SKIP_LOST", o.skip_lost);
    geti("EMME_LU_SPLIT", o.lu_split);
    if (const char* e = std::getenv("EMME_LU_GROUP")) o.lu_group_min_n = std::atoi(e) <= 0 ? -1 : std::atoi(e);
    geti("EMME_LU_SPIN_LIMIT", o.lu_spin_limit);
    if (std::getenv("EMME_LU_UNBLOCKED")) o.lu_unblocked = 1;
}

int options_check(const emme_options_t* o) {
    if (o->size != (int)sizeof(emme_options_t)) {
        set_error("emme_options_t: size field does not match this library (use emme_options_default)");
        return EMME_EINVAL;
    }
    if (!(o->node_cache_gb >= 0.0) || o->cache_min_batch < 1 || o->cache_min_depth < 0 || o->fill < EMME_FILL_AUTO ||
        o->fill > EMME_FILL_LANES || o->wl_min < 1 || (o->union_sel != 1 && o->union_sel != 2 && o->union_sel != 4) ||
        o->union_ipg_few < 1 || o->union_few_chunks < 0 || o->coop_wide_min < -1 || o->dense_min_cols < 1 ||
        o->dense_min_cols > 17 || o->dense_min_tasks < 0 || !(o->dense_cost_ratio > 0.0) || o->lu_split < 0 ||
        o->lu_split > 16 || o->lu_spin_limit < 1) {
        set_error("emme_options_t: value out of range");
        return EMME_EINVAL;
    }
    return EMME_OK;
}

hipEvent_t get_event(emme_ctx* c) {
    if (!c->free_events.empty()) {
        hipEvent_t e = c->free_events.back();
        c->free_events.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct ScopedSpan {
    emme_ctx* c;
    int kind;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedSpan(emme_ctx* ctx, int k, hipStream_t on = nullptr, bool use_on = false)
        : c(ctx), kind(k), st(use_on ? on : ctx->stream) {
        if (c->prof) {
            a = get_event(c);
            b = get_event(c);
            if (a) (void)hipEventRecord(a, st);
        }
    }
    ~ScopedSpan() {
        if (c->prof && a && b) {
            (void)hipEventRecord(b, st);
            c->spans.push_back({kind, a, b});
        }
    }
};

int drain_spans(emme_ctx* c) {
    for (auto& s : c->spans) {
        HIP_TRY(hipEventSynchronize(s.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s.a, s.b));
        if (s.kind == K_ASM)
            c->acc.assemble_ms += ms, c->acc.assemble_launches++;
        else if (s.kind == K_LIN)
            c->acc.linstep_ms += ms, c->acc.linstep_launches++;
        else if (s.kind == K_DEFER)
            c->acc.deferred_ms += ms, c->acc.deferred_launches++;
        else if (s.kind == K_CACHE)
            c->acc.cache_build_ms += ms, c->acc.cache_build_launches++;
        else if (s.kind == K_NULL)
            c->acc.nullspace_ms += ms, c->acc.nullspace_launches++;
        else
            c->acc.other_ms += ms, c->acc.other_launches++;
        c->free_events.push_back(s.a);
        c->free_events.push_back(s.b);
    }
    c->spans.clear();
    return EMME_OK;
}

int ensure_batch(emme_ctx* c, int nb) {
    if (nb <= c->cap) return EMME_OK;
    auto F = [](auto*& p) {
        if (p) (void)hipFree(p);
        p = nullptr;
    };
    F(c->d_omega), F(c->d_domega), F(c->d_tr), F(c->d_active), F(c->d_iters), F(c->d_info),
        F(c->d_status), F(c->d_intervals), F(c->d_actidx), F(c->d_chunks), F(c->d_overflow);
    c->cap = 0;
    HIP_TRY(malloc_retry((void**)&c->d_overflow, sizeof(unsigned int) * nb));
    HIP_TRY(hipMemset(c->d_overflow, 0, sizeof(unsigned int) * nb));
    HIP_TRY(malloc_retry((void**)&c->d_omega, sizeof(double) * 2 * nb));
    HIP_TRY(malloc_retry((void**)&c->d_domega, sizeof(double) * 2 * nb));
    HIP_TRY(malloc_retry((void**)&c->d_tr, sizeof(double) * 2 * nb));
    HIP_TRY(malloc_retry((void**)&c->d_active, sizeof(int) * nb));
    HIP_TRY(malloc_retry((void**)&c->d_iters, sizeof(int) * nb));
    HIP_TRY(malloc_retry((void**)&c->d_info, sizeof(int) * nb));
    HIP_TRY(malloc_retry((void**)&c->d_status, sizeof(int) * nb));
    HIP_TRY(malloc_retry((void**)&c->d_intervals, sizeof(unsigned long long) * nb));
    HIP_TRY(malloc_retry((void**)&c->d_actidx, sizeof(int) * nb));
    HIP_TRY(malloc_retry((void**)&c->d_chunks, sizeof(int) * 3 * nb));  // (first, size) per chunk | position map
    {
        if (c->p_act) (void)hipHostFree(c->p_act);
        if (c->p_iv) (void)hipHostFree(c->p_iv);
        if (c->p_w) (void)hipHostFree(c->p_w);
        if (c->p_lists) (void)hipHostFree(c->p_lists);
        if (c->p_overflow) (void)hipHostFree(c->p_overflow);
        c->p_act = nullptr, c->p_iv = nullptr, c->p_w = nullptr, c->p_lists = nullptr, c->p_cap = 0, c->p_lists_cap = 0;
        c->p_overflow = nullptr;
        HIP_TRY(hipHostMalloc((void**)&c->p_overflow, sizeof(unsigned int) * nb));
        std::memset(c->p_overflow, 0, sizeof(unsigned int) * nb);
        HIP_TRY(hipHostMalloc((void**)&c->p_act, sizeof(int) * nb));
        HIP_TRY(hipHostMalloc((void**)&c->p_iv, sizeof(unsigned long long) * nb));
        HIP_TRY(hipHostMalloc((void**)&c->p_w, sizeof(double) * 2 * nb));
        HIP_TRY(hipHostMalloc((void**)&c->p_lists, sizeof(int) * 2 * 4 * nb));  // 2 slots x (order | chunks | map)
        if (!c->p_deferred) HIP_TRY(hipHostMalloc((void**)&c->p_deferred, sizeof(unsigned int)));
        c->p_cap = nb, c->p_lists_cap = 4 * nb;
    }
    if (!c->d_rounds) {
        HIP_TRY(malloc_retry((void**)&c->d_rounds, (16 + 8192) * sizeof(unsigned long long)));  // (+ per-tile ticks of the diagnostic build)
        HIP_TRY(hipMemset(c->d_rounds, 0, (16 + 8192) * sizeof(unsigned long long)));
    }
    c->cap = nb;
    return EMME_OK;
}

// which matrix sets a call needs: bit0 M, bit1 Mold, bit2 Mp, bit3 work
int ensure_mats(emme_ctx* c, int nb, int sets) {
    if (nb > c->mat_cap) {
        auto F = [](double*& p) {
            if (p) (void)hipFree(p);
            p = nullptr;
        };
        F(c->d_M), F(c->d_Mold), F(c->d_Mp), F(c->d_work);
        c->mat_cap = nb;
    }
    const size_t bytes = mat_doubles(c) * sizeof(double) * (size_t)c->mat_cap;
    if ((sets & 1) && !c->d_M) HIP_TRY(malloc_retry((void**)&c->d_M, bytes));
    if ((sets & 2) && !c->d_Mold) HIP_TRY(malloc_retry((void**)&c->d_Mold, bytes));
    if ((sets & 4) && !c->d_Mp) HIP_TRY(malloc_retry((void**)&c->d_Mp, bytes));
    if ((sets & 8) && !c->d_work) HIP_TRY(malloc_retry((void**)&c->d_work, bytes));
    return EMME_OK;
}

// the Newton linear step: blocked kernel while its panel fits in LDS, else the unblocked one
// `h_active`: host copy of `active` (null: all live).  With fewer live matrices than compute
// units each gets up to 8 workgroups (EMME_LU_SPLIT=k pins k; 1 = one workgroup per matrix).
hipError_t trace_solve(emme_ctx* c, int n, int nbatch, double* A, double* B, const int* active,
                       double* tr, int* info, const int* h_active) {
    const bool force_unblocked = c->opt.lu_unblocked != 0;
    const int split_env = c->opt.lu_split;
    // n <= ~560: the whole L21 panel fits in LDS; up to 1024 the chunked build takes over, which
    // needs helper workgroups (>= 2 per matrix, all resident); otherwise the unblocked kernel
    const bool fits = trace_solve_blocked_lds(n) <= 150 * 1024;
    if (!force_unblocked && (fits || n <= 1024)) {
        const size_t need = trace_solve_blocked_scratch(n, nbatch);
        if (need > c->lu_scratch_bytes) {
            if (c->d_lu_scratch) (void)hipFree(c->d_lu_scratch);
            c->d_lu_scratch = nullptr, c->lu_scratch_bytes = 0;
            hipError_t e = hipMalloc(&c->d_lu_scratch, need);
            if (e != hipSuccess) return e;
            c->lu_scratch_bytes = need;
        }
        // dense list of the live matrices (h_active: host copy of `active`, null = all live)
        int n_live = nbatch;
        int* lu_slot = nullptr;
        if (h_active) {
            if (nbatch > c->lu_items_cap) {
                if (c->d_lu_items) (void)hipFree(c->d_lu_items);
                if (c->h_lu_items) (void)hipHostFree(c->h_lu_items);
                c->d_lu_items = nullptr, c->h_lu_items = nullptr, c->lu_items_cap = 0;
                hipError_t e = hipMalloc((void**)&c->d_lu_items, sizeof(int) * nbatch);
                // pinned, two slots used in turn: the device reads a slot (k_stage_ints) while the host
                // may already be writing the next launch's list
                if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_lu_items, sizeof(int) * 2 * nbatch);
                if (e != hipSuccess) return e;
                c->lu_items_cap = nbatch;
            }
            lu_slot = c->h_lu_items + (size_t)(c->lu_items_turn++ & 1u) * c->lu_items_cap;
            n_live = 0;
            for (int b = 0; b < nbatch; ++b)
                if (h_active[b]) lu_slot[n_live++] = b;
            if (n_live == 0) return hipSuccess;
        }
        int nwg = 1;
        if (c->lu_one_wg) {
            nwg = 1;
        } else if (split_env > 0) {
            nwg = std::min(split_env, 16);
        } else if (n >= 128) {
            // every workgroup of a matrix must be resident at once (they wait for each other):
            // never more workgroups than compute units.  Below n = 128 the hand-over costs more
            // than the idle units are worth, and beyond 8 the factoring workgroup is the limit.
            // (n = 256: four are enough, role 0 is the limit then; n = 512: two A-helpers pay)
            nwg = std::max(1, std::min(n >= 768 ? 16 : (n >= 384 ? 8 : 4), c->n_cu / n_live));
        }
        if (!fits && nwg < 2 && !c->lu_one_wg && split_env != 1) nwg = 2;
        c->last_lu_nwg = nwg;
        const int* d_items = nullptr;
        if (nwg > 1 && h_active) {
            hipError_t e = launch_stage_ints(lu_slot, c->d_lu_items, n_live, nullptr, 0, c->stream);
            if (e != hipSuccess) return e;
            d_items = c->d_lu_items;
        }
        const hipError_t e = launch_trace_solve_blocked(n, nbatch, A, B, active, tr, info, nwg, d_items, n_live,
                                                        c->d_lu_scratch, c->stream, c->opt.lu_group_min_n, c->opt.lu_spin_limit);
        if (e != hipErrorNotSupported) return e;
        (void)hipGetLastError();  // chunked build not possible here (one workgroup per matrix, or no room)
        c->last_lu_nwg = 1;
    }
    return launch_trace_solve(n, nbatch, A, B, active, tr, info, c->stream);
}

// One Newton linear step on the batch: leaves tr[b] with domega = -1/tr[b].
//   trace-secant (include/solver.h:113-160): work <- M, LU of [work | Mp], tr(M^-1 M')
//   QR-secant    (include/solver.h:210-383): work <- M^T, pivoted QR of work, t_n / R_nn
hipError_t linear_step(emme_ctx* c, int method, int n, int nbatch, const double* M, double* work,
                       double* Mp, const int* active, double* tr, int* info,
                       const int* h_active = nullptr, bool work_ready = false) {
    const size_t mbytes = (size_t)n * n * 2 * sizeof(double) * nbatch;
    if (method == EMME_METHOD_QR_SECANT) {
        hipError_t e = launch_transpose(n, nbatch, M, work, active, c->stream);
        if (e != hipSuccess) return e;
        return launch_qr_secant(n, nbatch, work, Mp, active, tr, info, c->stream);
    }
    if (!work_ready) {  // (the root search copies M -> work together with M -> Mold)
        hipError_t e = hipMemcpyAsync(work, M, mbytes, hipMemcpyDeviceToDevice, c->stream);
        if (e != hipSuccess) return e;
    }
    return trace_solve(c, n, nbatch, work, Mp, active, tr, info, h_active);
}

int check_method(const emme_ctx* c, int method) {
    if (method != EMME_METHOD_TRACE_SECANT && method != EMME_METHOD_QR_SECANT) {
        set_error("unknown iteration method");
        return EMME_EINVAL;
    }
    if (method == EMME_METHOD_QR_SECANT && c->dim > 1024) {
        set_error("QR-secant step: matrix dimension above 1024 is not supported");
        return EMME_ECONFIG;
    }
    return EMME_OK;
}

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

// host_active: which of the nbatch items to assemble (null = all).  Batches of wl_min or
// more items go through the omega-lane kernel, which shares the omega-independent node
// data between items; smaller ones through the lanes-are-nodes kernel.
int do_assemble(emme_ctx* c, int nbatch, const double* d_omega, const int* d_active,
                const int* host_active, double* d_M, const double* d_Mold, double* d_Mp,
                const double* d_domega, const unsigned long long* cost = nullptr,
                const double* host_omega = nullptr, bool newton_loop = false) {
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
    bool use_cache = host_omega != nullptr && c->cache_depth != -2 &&
                     (nbatch >= c->opt.cache_min_batch || c->d_recs[0] != nullptr || c->d_recs[1] != nullptr);
    if (use_cache) {
        bool need[2] = {false, false};
        for (int b : idx) need[-std::copysign(1.0, host_omega[2 * b]) > 0.0 ? 0 : 1] = true;
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
            // (tiled electromagnetic contexts: the cooperative kernel does not read 32-slot tile blocks -- the few
            // integrals that leave the cache are evaluated from scratch)
            const bool coop_cached = !(c->tiled && c->nm > 1);
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
    } else if (n_act >= c->opt.wl_min) {
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

}  // namespace

extern "C" {

const char* emme_last_error(void) { return g_error.c_str(); }
int emme_version(void) { return 3; }

void emme_options_default(emme_options_t* opt) {
    if (opt) options_default(*opt);
}

int emme_ctx_create(const emme_params_t* p, int device, emme_ctx_t** out) {
    return emme_ctx_create_ex(p, device, nullptr, out);
}

int emme_ctx_create_ex(const emme_params_t* p, int device, const emme_options_t* opt, emme_ctx_t** out) {
    if (!p || !out) return EMME_EINVAL;
    *out = nullptr;
    emme_options_t o;
    options_default(o);
    if (opt) {
        const int rc = options_check(opt);
        if (rc) return rc;
        o = *opt;
    }
    options_env_overrides(o);
    {
        const int rc = options_check(&o);
        if (rc) return rc;
    }
    if (p->integration_start_points != 15 && p->integration_start_points != 31) {
        // include/functions.h:329
        set_error("integration_start_points should be 15 or 31");
        return EMME_ECONFIG;
    }
    if (p->npoints < 2 || p->npoints > 65535) {
        set_error("npoints must be in [2, 65535]");
        return EMME_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (the MI355X path has no CPU fallback)");
        return EMME_EDEVICE;
    }
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        set_error(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950");
        return EMME_EDEVICE;
    }

    emme_ctx* c = new emme_ctx;
    c->p = *p;
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->opt = o;
    const int N = p->npoints;
    c->N = N;
    const bool es = std::fpclassify(p->beta_e) == FP_ZERO;  // include/solver.h:406-407
    c->dim = es ? N : 2 * N;
    c->nm = es ? 1 : 3;
    // electromagnetic contexts share one node record per (pair, interval, node) between the three
    // moments (EMME_EM_SHARED=0: one record per moment, for A/B comparisons)
    c->em_shared = !es && o.em_shared != 0;
    // phase_table = 0: unfolded records and exp(A0 + T omega) per (pair, node, omega) in the fill
    c->folded = o.phase_table != 0;
    // electrostatic GK15 on folded records: tiled record layout + dense fill on the FP64 matrix cores
    // (assemble_dense.hip) instead of the union walk (EMME_DENSE=0 restores that).  It
    // carries the safe_exp-clamped tails (<= 4e-14 absolute), so inputs whose absolute quadrature goal
    // (integration_accuracy) is tighter than 1e-9 keep the exact union kernel.
    c->tiled = wants_tiled(*p, es, c->folded, o.fill);

    DevParams& P = c->P;
    std::vector<double> tab(3 * (size_t)N);
    double dx = 0;
    emme_tables(p, tab.data(), tab.data() + N, tab.data() + 2 * N, &dx);
    P.N = N, P.dim = c->dim, P.nm = c->nm, P.max_sub = p->integration_iteration_limit;
    P.dx = dx;
    P.inv_arc = 1.0 / p->arc_coeff;
    P.qR = p->q * p->R;
    P.vt = p->vt;
    P.cb = (p->q * p->R) / p->vt * (p->omega_d_bar);                                  // :88
    P.cbe = (p->q * p->R) / p->vt * (p->omega_d_bar * p->omega_s_e / p->omega_s_i);  // :93
    P.omega_s_i = p->omega_s_i, P.omega_s_e = p->omega_s_e;
    P.eta_i = p->eta_i, P.eta_e = p->eta_e, P.tau = p->tau;
    P.rel_tol = p->integration_precision;
    P.prec_goal = p->integration_accuracy;
    P.pref = (p->q * p->R) / (p->vt * std::sqrt(2.0 * M_PI));
    P.diag_a = 1.0 + 1.0 / p->tau;
    P.diag_d = es ? 0.0 : (2.0 * p->tau) / p->beta_e;

    // pair list ordered by diagonal offset (see assemble.hip header)
    std::vector<ushort2> pairs;
    pairs.reserve((size_t)N * (N - 1) / 2);
    for (int off = 1; off < N; ++off)
        for (int i = 0; i + off < N; ++i) pairs.push_back(make_ushort2((unsigned short)i, (unsigned short)(i + off)));
    c->npairs = (int)pairs.size();

    int rc = EMME_OK;
    auto fail = [&](int code) {
        emme_ctx_destroy(c);
        return code;
    };
    if (malloc_retry((void**)&c->d_tab, tab.size() * sizeof(double)) != hipSuccess ||
        malloc_retry((void**)&c->d_pairs, pairs.size() * sizeof(ushort2)) != hipSuccess) {
        set_error("hipMalloc failed for tables");
        return fail(EMME_ENOMEM);
    }
    if (hipMemcpy(c->d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(c->d_pairs, pairs.data(), pairs.size() * sizeof(ushort2), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("hipMemcpy failed for tables");
        return fail(EMME_EDEVICE);
    }
    (void)rc;
    *out = c;
    return EMME_OK;
}

void emme_ctx_destroy(emme_ctx_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    else (void)hipDeviceSynchronize();
    auto F = [](auto* p) {
        if (p) (void)hipFree((void*)p);
    };
    F(c->d_tab), F(c->d_pairs), F(c->d_omega), F(c->d_domega), F(c->d_tr), F(c->d_active),
        F(c->d_iters), F(c->d_info), F(c->d_status), F(c->d_intervals), F(c->d_actidx), F(c->d_chunks), F(c->d_M),
        F(c->d_Mold),
        F(c->d_Mp), F(c->d_work), F(c->d_iterates), F(c->d_rounds);
    for (int k = 0; k < 2; ++k) {
        // the big buffers go to the process-wide pool for the next context
        pool_free(c->d_recs[k], c->recs_bytes[k], c->device);
        F(c->d_ttab[k]), F(c->d_wtab[k]), F(c->d_tile_poison[k]);
        for (int e = 0; e < NODE_CACHE_MAX_SUB - 1; ++e) pool_free(c->d_recs_ext[k][e], c->recs_ext_bytes[k][e], c->device);
    }
    F(c->d_scale);
    F(c->d_etab);
    F(c->d_btab);
    F(c->d_lu_scratch);
    F(c->d_lu_items);
    if (c->h_lu_items) (void)hipHostFree(c->h_lu_items);
    if (c->p_act) (void)hipHostFree(c->p_act);
    if (c->p_iv) (void)hipHostFree(c->p_iv);
    if (c->p_w) (void)hipHostFree(c->p_w);
    if (c->p_lists) (void)hipHostFree(c->p_lists);
    if (c->p_deferred) (void)hipHostFree(c->p_deferred);
    if (c->p_overflow) (void)hipHostFree(c->p_overflow);
    F(c->d_overflow);
    F(c->d_worklist), F(c->d_worklist_count), F(c->d_defer_info);
    for (auto& s : c->spans) (void)hipEventDestroy(s.a), (void)hipEventDestroy(s.b);
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    delete c;
}

void emme_release_pooled_memory(void) { pool_release_all(); }

int emme_ctx_set_options(emme_ctx_t* c, const emme_options_t* opt) {
    if (!c || !opt) return EMME_EINVAL;
    const int rc = options_check(opt);
    if (rc) return rc;
    const bool layout_differs = opt->fill != c->opt.fill || (opt->phase_table != 0) != (c->opt.phase_table != 0) ||
                                (opt->em_shared != 0) != (c->opt.em_shared != 0);
    if (layout_differs) {
        if (c->d_recs[0] || c->d_recs[1]) {
            set_error("emme_ctx_set_options: fill / phase_table / em_shared fix the layout of the node cache, which exists already");
            return EMME_EINVAL;
        }
        const bool es = c->nm == 1;
        c->em_shared = !es && opt->em_shared != 0;
        c->folded = opt->phase_table != 0;
        c->tiled = wants_tiled(c->p, es, c->folded, opt->fill);
    }
    if (opt->node_cache_gb > 0.0 && c->cache_depth == -2 && !c->d_recs[0] && !c->d_recs[1])
        c->cache_depth = -1;  // a budget after "no cache": decide again
    if (opt->lu_split != c->opt.lu_split) c->lu_one_wg = false;
    c->opt = *opt;
    return EMME_OK;
}

int emme_ctx_get_options(const emme_ctx_t* c, emme_options_t* opt) {
    if (!c || !opt) return EMME_EINVAL;
    *opt = c->opt;
    return EMME_OK;
}

int emme_ctx_set_stream(emme_ctx_t* c, void* s) {
    if (!c) return EMME_EINVAL;
    c->stream = (hipStream_t)s;
    return EMME_OK;
}

int emme_ctx_dim(const emme_ctx_t* c) { return c ? c->dim : EMME_EINVAL; }

int emme_ctx_fill_mode(const emme_ctx_t* c) { return c ? c->last_fill_mode : EMME_EINVAL; }

double emme_ctx_node_cache_gib(const emme_ctx_t* c) {
    return c ? c->cache_bytes_used / (1024.0 * 1024.0 * 1024.0) : 0.0;
}

int emme_ctx_profile_enable(emme_ctx_t* c, int on) {
    if (!c) return EMME_EINVAL;
    c->prof = on != 0;
    return EMME_OK;
}

int emme_ctx_profile_read(emme_ctx_t* c, emme_profile_t* out, int reset) {
    if (!c || !out) return EMME_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    int rc = drain_spans(c);
    if (rc) return rc;
    c->acc.integrand_evals = c->acc.gk_intervals * c->p.integration_start_points;
    if (c->d_rounds) {
        unsigned long long r[16] = {};
        HIP_TRY(hipMemcpy(r, c->d_rounds, sizeof r, hipMemcpyDeviceToHost));
        if (std::getenv("EMME_DEBUG_STAMPS") && r[10])
            fprintf(stderr, "[emme] dense fill: %llu integrals handed over because a level list overflowed\n", r[10]);
        if (std::getenv("EMME_DEBUG_STAMPS") && r[8])  // diagnostic build (EMME_DENSE_STAMPS) only
            fprintf(stderr, "[emme] dense stamps: select %.3g  dense %.3g  sparse %.3g  decide %.3g  task total %.3g  "
                    "longest task %.3g cycles; per round: select %.0f dense %.0f sparse %.0f decide %.0f\n",
                    (double)r[4], (double)r[5], (double)r[6], (double)r[7], (double)r[8], (double)r[9],
                    (double)r[4] / (double)(r[0] + r[1] + 1), (double)r[5] / (double)(r[0] + 1),
                    (double)r[6] / (double)(r[1] + 1), (double)r[7] / (double)(r[0] + r[1] + 1));
        if (c->tiled) {
            c->acc.union_rounds = (long long)(r[0] + r[1]);
            c->acc.dense_rounds = (long long)r[0], c->acc.sparse_rounds = (long long)r[1];
            c->acc.sparse_columns = (long long)r[2], c->acc.tile_tasks = (long long)r[3];
        } else {
            c->acc.union_rounds = (long long)r[0];
        }
        if (reset) HIP_TRY(hipMemset(c->d_rounds, 0, sizeof r));
    }
    *out = c->acc;
    if (reset) c->acc = emme_profile_t{};
    return EMME_OK;
}

int emme_assemble_batch(emme_ctx_t* c, const double* omega, int nbatch, double* M,
                        long long* intervals) {
    if (!c || !omega || !M || nbatch < 1) return EMME_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_batch(c, nbatch);
    if (rc) return rc;
    const bool dev_out = is_device_ptr(M);
    double* dM = M;
    if (!dev_out) {
        rc = ensure_mats(c, nbatch, 1);
        if (rc) return rc;
        dM = c->d_M;
    }
    HIP_TRY(hipMemcpyAsync(c->d_omega, omega, sizeof(double) * 2 * nbatch, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_intervals, 0, sizeof(unsigned long long) * nbatch, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(int) * nbatch, c->stream));
    rc = do_assemble(c, nbatch, c->d_omega, nullptr, nullptr, dM, nullptr, nullptr, nullptr, nullptr, omega);
    if (rc) return rc;
    if (!dev_out)
        HIP_TRY(hipMemcpyAsync(M, dM, mat_doubles(c) * sizeof(double) * nbatch, hipMemcpyDeviceToHost, c->stream));
    std::vector<unsigned long long> iv(nbatch);
    std::vector<int> stv(nbatch);
    HIP_TRY(hipMemcpyAsync(iv.data(), c->d_intervals, sizeof(unsigned long long) * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(stv.data(), c->d_status, sizeof(int) * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    bool bad = false;
    for (int b = 0; b < nbatch; ++b) {
        c->acc.gk_intervals += (long long)iv[b];
        if (intervals) intervals[b] = (long long)iv[b];
        bad |= stv[b] != 0;
    }
    if (bad) {
        set_error("quadrature depth cap hit or non-finite integral in at least one item");
        return EMME_ENUMERIC;
    }
    return EMME_OK;
}

int emme_ctx_cache_settle(emme_ctx_t* c, const double* omega, int nbatch, int* fills_done) {
    if (!c || !omega || nbatch < 1) return EMME_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_batch(c, nbatch);
    if (rc) return rc;
    rc = ensure_mats(c, nbatch, 1);
    if (rc) return rc;
    int fills = 0;
    // a fill that deferred integrals makes the NEXT one cache a subtree around the interval most of
    // them were missing; at most NODE_CACHE_MAX_SUB - 1 run-time subtrees per contour class exist, so
    // the shape is final after at most that many growing fills plus one that finds nothing to add
    for (int round = 0; round < 2 * NODE_CACHE_MAX_SUB + 2; ++round) {
        const double before = c->cache_bytes_used;
        HIP_TRY(hipMemcpyAsync(c->d_omega, omega, sizeof(double) * 2 * nbatch, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_intervals, 0, sizeof(unsigned long long) * nbatch, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(int) * nbatch, c->stream));
        rc = do_assemble(c, nbatch, c->d_omega, nullptr, nullptr, c->d_M, nullptr, nullptr, nullptr, nullptr, omega);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        ++fills;
        if (round > 0 && c->cache_bytes_used == before) break;  // this fill found the shape it started with
    }
    if (fills_done) *fills_done = fills;
    return EMME_OK;
}

int emme_ctx_cache_state(const emme_ctx_t* c, int* full_depth, int* subtrees, double* gib) {
    if (!c) return EMME_EINVAL;
    if (full_depth) *full_depth = c->cache_depth;
    if (subtrees) *subtrees = c->cache_depth >= 0 ? c->cache_geom.nsub : 0;
    if (gib) *gib = c->cache_bytes_used / (1024.0 * 1024.0 * 1024.0);
    return EMME_OK;
}

int emme_trace_solve_batch(emme_ctx_t* c, int n, int nbatch, double* A, double* B, double* tr,
                           int* info) {
    if (!c || !A || !B || !tr || !info || n < 1 || nbatch < 1) return EMME_EINVAL;
    if ((size_t)2 * n * sizeof(double2) > 64 * 1024) {
        set_error("n too large for the LDS-staged pivot row");
        return EMME_EINVAL;
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_batch(c, nbatch);
    if (rc) return rc;
    const bool devA = is_device_ptr(A), devB = is_device_ptr(B);
    if (devA != devB) {
        set_error("A and B must both be host or both be device pointers");
        return EMME_EINVAL;
    }
    const size_t bytes = (size_t)n * n * 2 * sizeof(double) * nbatch;
    double *dA = A, *dB = B;
    struct Staging {  // device copies of host operands, released on every way out
        double *a = nullptr, *b = nullptr;
        ~Staging() {
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
        }
    } st;
    if (!devA) {
        HIP_TRY(malloc_retry((void**)&st.a, bytes));
        HIP_TRY(malloc_retry((void**)&st.b, bytes));
        dA = st.a, dB = st.b;
        HIP_TRY(hipMemcpyAsync(dA, A, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(dB, B, bytes, hipMemcpyHostToDevice, c->stream));
    }
    {
        ScopedSpan s(c, K_LIN);
        HIP_TRY(trace_solve(c, n, nbatch, dA, dB, nullptr, c->d_tr, c->d_info, nullptr));
    }
    HIP_TRY(hipMemcpyAsync(tr, c->d_tr, sizeof(double) * 2 * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(info, c->d_info, sizeof(int) * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return EMME_OK;
}

int emme_qr_secant_batch(emme_ctx_t* c, int n, int nbatch, const double* A, const double* B,
                         double* q, int* info) {
    if (!c || !A || !B || !q || !info || n < 1 || nbatch < 1) return EMME_EINVAL;
    if (n > 1024) {
        set_error("QR-secant step: matrix dimension above 1024 is not supported");
        return EMME_ECONFIG;
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_batch(c, nbatch);
    if (rc) return rc;
    const bool devA = is_device_ptr(A), devB = is_device_ptr(B);
    if (devA != devB) {
        set_error("A and B must both be host or both be device pointers");
        return EMME_EINVAL;
    }
    const size_t bytes = (size_t)n * n * 2 * sizeof(double) * nbatch;
    double *dA = nullptr, *dB = nullptr, *dW = nullptr;
    auto release = [&]() {
        if (dW) (void)hipFree(dW);
        if (!devA && dA) (void)hipFree(dA);
        if (!devA && dB) (void)hipFree(dB);
    };
    if (malloc_retry((void**)&dW, bytes) != hipSuccess) {
        set_error("hipMalloc failed");
        return EMME_ENOMEM;
    }
    if (devA) {
        dA = const_cast<double*>(A), dB = const_cast<double*>(B);
    } else {
        if (malloc_retry((void**)&dA, bytes) != hipSuccess || malloc_retry((void**)&dB, bytes) != hipSuccess) {
            release();
            set_error("hipMalloc failed");
            return EMME_ENOMEM;
        }
        (void)hipMemcpyAsync(dA, A, bytes, hipMemcpyHostToDevice, c->stream);
        (void)hipMemcpyAsync(dB, B, bytes, hipMemcpyHostToDevice, c->stream);
    }
    hipError_t e;
    {
        ScopedSpan s(c, K_LIN);
        e = launch_transpose(n, nbatch, dA, dW, nullptr, c->stream);
        if (e == hipSuccess) e = launch_qr_secant(n, nbatch, dW, dB, nullptr, c->d_tr, c->d_info, c->stream);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(q, c->d_tr, sizeof(double) * 2 * nbatch, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(info, c->d_info, sizeof(int) * nbatch, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    release();
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        return EMME_EDEVICE;
    }
    return EMME_OK;
}

int emme_newton_step_batch(emme_ctx_t* c, double* omega, double* domega, int nbatch, double* M,
                           double* Mp, int method, int* info) {
    if (!c || !omega || !domega || !M || !Mp || !info || nbatch < 1) return EMME_EINVAL;
    int rc = check_method(c, method);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    rc = ensure_batch(c, nbatch);
    if (rc) return rc;
    const bool dev = is_device_ptr(M);
    if (dev != is_device_ptr(Mp)) {
        set_error("M and Mp must both be host or both be device pointers");
        return EMME_EINVAL;
    }
    const size_t mbytes = mat_doubles(c) * sizeof(double) * nbatch;
    rc = ensure_mats(c, nbatch, dev ? (2 | 8) : (1 | 2 | 4 | 8));
    if (rc) return rc;
    double *dM = M, *dMp = Mp;
    if (!dev) {
        dM = c->d_M, dMp = c->d_Mp;
        HIP_TRY(hipMemcpyAsync(dM, M, mbytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(dMp, Mp, mbytes, hipMemcpyHostToDevice, c->stream));
    }
    const hipMemcpyKind in_kind = is_device_ptr(omega) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const hipMemcpyKind out_kind = is_device_ptr(omega) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    HIP_TRY(hipMemcpyAsync(c->d_omega, omega, sizeof(double) * 2 * nbatch, in_kind, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_intervals, 0, sizeof(unsigned long long) * nbatch, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(int) * nbatch, c->stream));
    {
        // eigen_matrix_old = eigen_matrix (include/solver.h:114); the factorisation then
        // consumes a scratch copy so M_old survives for the secant update
        ScopedSpan s(c, K_OTHER);
        HIP_TRY(hipMemcpyAsync(c->d_Mold, dM, mbytes, hipMemcpyDeviceToDevice, c->stream));
    }
    {
        ScopedSpan s(c, K_LIN);
        HIP_TRY(linear_step(c, method, c->dim, nbatch, dM, c->d_work, dMp, nullptr, c->d_tr, c->d_info));
    }
    {
        ScopedSpan s(c, K_OTHER);
        HIP_TRY(launch_newton_update(nbatch, c->d_tr, c->d_omega, c->d_domega, nullptr, nullptr,
                                     c->d_info, 0.0, nullptr, 0, 0, c->stream));
    }
    std::vector<double> h_w(2 * (size_t)nbatch);
    HIP_TRY(hipMemcpyAsync(h_w.data(), c->d_omega, sizeof(double) * 2 * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    rc = do_assemble(c, nbatch, c->d_omega, nullptr, nullptr, dM, c->d_Mold, dMp, c->d_domega, nullptr, h_w.data());
    if (rc) return rc;
    if (!dev) {
        HIP_TRY(hipMemcpyAsync(M, dM, mbytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(Mp, dMp, mbytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(omega, c->d_omega, sizeof(double) * 2 * nbatch, out_kind, c->stream));
    HIP_TRY(hipMemcpyAsync(domega, c->d_domega, sizeof(double) * 2 * nbatch, out_kind, c->stream));
    std::vector<unsigned long long> iv(nbatch);
    HIP_TRY(hipMemcpyAsync(iv.data(), c->d_intervals, sizeof(unsigned long long) * nbatch, hipMemcpyDeviceToHost, c->stream));
    if (is_device_ptr(info))
        HIP_TRY(hipMemcpyAsync(info, c->d_info, sizeof(int) * nbatch, hipMemcpyDeviceToDevice, c->stream));
    else
        HIP_TRY(hipMemcpyAsync(info, c->d_info, sizeof(int) * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int b = 0; b < nbatch; ++b) c->acc.gk_intervals += (long long)iv[b];
    return EMME_OK;
}

int emme_solve_roots(emme_ctx_t* c, const double* guesses, int n, double tol, int step_limit,
                     double* roots, int* iters, int* info, double* iterates) {
    if (!c || !guesses || !roots || !iters || !info || n < 1 || step_limit < 0) return EMME_EINVAL;
    const int method = c->p.iteration_method;  // src/main.cpp:45-49
    int rc = check_method(c, method);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    rc = ensure_batch(c, n);
    if (rc) return rc;
    rc = ensure_mats(c, n, 1 | 2 | 4 | 8);
    if (rc) return rc;
    const int stride = step_limit + 1;
    if (iterates) {
        const size_t need = (size_t)n * stride * 2;
        if (need > c->iterates_cap) {
            if (c->d_iterates) (void)hipFree(c->d_iterates);
            c->d_iterates = nullptr;
            HIP_TRY(malloc_retry((void**)&c->d_iterates, need * sizeof(double)));
            c->iterates_cap = need;
        }
        std::vector<double> nanv(need, std::numeric_limits<double>::quiet_NaN());
        HIP_TRY(hipMemcpyAsync(c->d_iterates, nanv.data(), need * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }

    // EigenSolver ctor (include/solver.h:396-415): eigen_value = 0.99 g, d = 0.01 g;
    // M_old = M(eigen_value); eigen_value += d; M = M(eigen_value); M' = (M - M_old)/d
    std::vector<double> w0(2 * (size_t)n), dw(2 * (size_t)n), w1(2 * (size_t)n);
    for (int b = 0; b < 2 * n; ++b) {
        w0[b] = 0.99 * guesses[b];
        dw[b] = 0.01 * guesses[b];
        w1[b] = w0[b] + dw[b];
    }
    std::vector<int> ones(n, 1), zeros(n, 0);
    HIP_TRY(hipMemcpyAsync(c->d_active, ones.data(), sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_iters, zeros.data(), sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_info, zeros.data(), sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_intervals, 0, sizeof(unsigned long long) * n, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(int) * n, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_omega, w0.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_domega, dw.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, c->stream));
    c->h_wide.assign(n, 0);
    HIP_TRY(hipMemsetAsync(c->d_overflow, 0, sizeof(unsigned int) * n, c->stream));
    std::vector<int> act(n, 1);
    std::vector<double> h_w(2 * (size_t)n);
    std::vector<unsigned long long> iv_prev(n, 0), iv_now(n, 0), cost(n, 0), iv_prev_dbg(n, 0);
    auto refresh_cost = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(iv_now.data(), c->d_intervals, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int b = 0; b < n; ++b) {
            if (iv_now[b] != iv_prev[b]) cost[b] = iv_now[b] - iv_prev[b];
            iv_prev[b] = iv_now[b];
        }
        return EMME_OK;
    };
    rc = do_assemble(c, n, c->d_omega, nullptr, nullptr, c->d_Mold, nullptr, nullptr, nullptr, nullptr, w0.data(), true);
    if (rc) return rc;
    rc = refresh_cost();  // synchronises; the first fill's interval counts order the second
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_omega, w1.data(), sizeof(double) * 2 * n, hipMemcpyHostToDevice, c->stream));
    // (the fills of a root search write M only: the secant M' = (M - M_old) / d, include/solver.h:157 and :412, is
    // taken by k_secant_copy at the top of the step that uses it, in one coalesced pass)
    rc = do_assemble(c, n, c->d_omega, nullptr, nullptr, c->d_M, nullptr, nullptr, nullptr,
                     cost.data(), w1.data(), true);
    if (rc) return rc;

    rc = refresh_cost();
    if (rc) return rc;
    // One stream synchronisation per Newton step: the host needs the new omegas (contour
    // classes, cache growth) before it can launch the fill.  The active flags and interval
    // counts a fill leaves behind travel to pinned memory asynchronously and are read after the
    // NEXT step's synchronisation, so the LU and the update of that step are queued behind the
    // fill without a bubble (their list of live matrices is one step old: a superset).
    bool pending = false;
    int j_pending = 0;
    auto take_pending = [&]() {  // results of the previous step's fill + retire
        for (int b = 0; b < n; ++b) {
            act[b] = c->p_act[b];
            iv_now[b] = c->p_iv[b];
            // (an eighth of its integrals did not fit the 64-entry level lists: 128 entries from now on)
            if (c->p_overflow[b] * 8u >= (unsigned)c->npairs) c->h_wide[b] = 1;
            c->last_deferred = *c->p_deferred, c->pub_valid = true;
            if (iv_now[b] != iv_prev[b]) cost[b] = iv_now[b] - iv_prev[b];
            iv_prev[b] = iv_now[b];
        }
        pending = false;
        if (std::getenv("EMME_DEBUG")) {
            unsigned long long tot = 0, mx = 0;
            int na = 0, nprev = 0;
            for (int b = 0; b < n; ++b) {
                if (iv_now[b] != iv_prev_dbg[b]) {
                    const unsigned long long d = iv_now[b] - iv_prev_dbg[b];
                    tot += d, mx = d > mx ? d : mx, ++nprev;
                }
                iv_prev_dbg[b] = iv_now[b];
                na += act[b] != 0;
            }
            fprintf(stderr, "[emme] LU workgroups per matrix %d\n", c->last_lu_nwg);
            fprintf(stderr, "[emme] iter %2d: assembled %3d, lane-intervals %10llu (max/item %9llu), still active %d\n",
                    j_pending, nprev, tot, mx, na);
        }
    };
    for (int j = 0; j <= step_limit; ++j) {  // src/main.cpp:43
        const bool fused_copy = method == EMME_METHOD_TRACE_SECANT;
        {
            // the secant M' of the step just taken, then this step's matrix becomes the "previous" one (and the
            // LU's work copy): one pass, for the chains still iterating only
            ScopedSpan s(c, K_OTHER);
            HIP_TRY(launch_secant_copy_sym(c->dim, n, c->d_M, c->d_Mold, fused_copy ? c->d_work : nullptr, c->d_Mp,
                                           c->d_domega, c->d_active, c->stream));
        }
        {
            ScopedSpan s(c, K_LIN);
            HIP_TRY(linear_step(c, method, c->dim, n, c->d_M, c->d_work, c->d_Mp, c->d_active, c->d_tr, c->d_info,
                                act.data(), fused_copy));
        }
        {
            ScopedSpan s(c, K_OTHER);
            HIP_TRY(launch_newton_update(n, c->d_tr, c->d_omega, c->d_domega, c->d_active, c->d_iters,
                                         c->d_info, tol, c->d_iterates, j, stride, c->stream, c->p_w,
                                         c->opt.skip_lost ? c->d_status : nullptr));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        std::copy(c->p_w, c->p_w + 2 * (size_t)n, h_w.begin());
        if (pending) {
            take_pending();
            bool any = false;
            for (int b = 0; b < n; ++b) any |= act[b] != 0;
            if (!any) break;  // (this step's LU and update found nothing active: no-ops)
        }
        rc = do_assemble(c, n, c->d_omega, c->d_active, act.data(), c->d_M, nullptr, nullptr, nullptr,
                         cost.data(), h_w.data(), true);
        if (rc) return rc;
        {
            ScopedSpan s(c, K_OTHER);
            HIP_TRY(launch_retire(n, c->d_active, c->stream, c->p_act, c->d_intervals, c->p_iv,
                                  c->d_worklist_count, c->p_deferred, c->d_overflow, c->p_overflow));
        }
        pending = true, j_pending = j;
    }
    std::vector<unsigned long long> iv(n);
    std::vector<int> stv(n);
    HIP_TRY(hipMemcpyAsync(roots, c->d_omega, sizeof(double) * 2 * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(iters, c->d_iters, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(info, c->d_info, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(iv.data(), c->d_intervals, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(stv.data(), c->d_status, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
    if (iterates)
        HIP_TRY(hipMemcpyAsync(iterates, c->d_iterates, sizeof(double) * 2 * (size_t)n * stride, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->last_n = n;
    for (int b = 0; b < n; ++b) {
        c->acc.gk_intervals += (long long)iv[b];
        // a chain that met a non-finite integral or the quadrature depth cap is reported
        // per item (the reference would carry the NaN to its "eigenvalue": "NaN" record,
        // src/main.cpp:311-316); the other chains of the batch are unaffected
        if (stv[b] != 0 && info[b] == 0) info[b] = EMME_ENUMERIC;
        // whatever the cause, a non-finite omega is never handed back as a root
        if (info[b] == 0 && !(std::isfinite(roots[2 * b]) && std::isfinite(roots[2 * b + 1]))) info[b] = EMME_ENUMERIC;
    }
    // The multi-workgroup LU needs its workgroups resident together; if something else held
    // compute units for seconds (a foreign kernel on a shared device) a hand-over wait timed out
    // and retired those chains with EMME_EDEVICE.  Do the search again with one workgroup per
    // matrix, and keep it that way for this context.
    if (!c->lu_one_wg) {
        bool timed_out = false;
        for (int b = 0; b < n; ++b) timed_out |= info[b] == EMME_EDEVICE;
        if (timed_out) {
            c->lu_one_wg = true;
            if (std::getenv("EMME_DEBUG")) fprintf(stderr, "[emme] LU hand-over timed out: repeating the search with one workgroup per matrix\n");
            return emme_solve_roots(c, guesses, n, tol, step_limit, roots, iters, info, iterates);
        }
    }
    return EMME_OK;
}

int emme_bessel_batch(const double* z, int n, double* out) {
    if (!z || !out || n < 1) return EMME_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        set_error("no HIP device available (the MI355X path has no CPU fallback)");
        return EMME_EDEVICE;
    }
    double *dz = nullptr, *dout = nullptr;
    struct Free {
        double*& a;
        double*& b;
        ~Free() {
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
        }
    } guard{dz, dout};
    HIP_TRY(hipMalloc((void**)&dz, sizeof(double) * 2 * n));
    HIP_TRY(hipMalloc((void**)&dout, sizeof(double) * 8 * n));
    HIP_TRY(hipMemcpy(dz, z, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
    HIP_TRY(launch_bessel_probe(dz, n, dout, nullptr));
    HIP_TRY(hipMemcpy(out, dout, sizeof(double) * 8 * n, hipMemcpyDeviceToHost));
    return EMME_OK;
}

// nullSpace (reference include/solver.h:58-112), batched on the device: see nullspace.hip
int emme_null_vectors_batch(emme_ctx_t* c, int n, int nbatch, const double* M, double* vecs, int* info) {
    if (!c || !vecs || !info || n < 1 || nbatch < 1) return EMME_EINVAL;
    if (!M && (n != c->dim || nbatch > c->last_n || !c->d_M)) {
        set_error("emme_null_vectors_batch: M = NULL needs a preceding emme_solve_roots call (n = emme_ctx_dim, nbatch <= its n)");
        return EMME_EINVAL;
    }
    if (n > 2048) {
        set_error("emme_null_vectors_batch: order above 2048 is not supported");
        return EMME_ECONFIG;
    }
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_batch(c, nbatch);
    if (rc) return rc;
    const size_t mbytes = (size_t)n * n * 2 * sizeof(double);
    struct Tmp {  // device scratch of this call, released on every way out
        double *a = nullptr, *b = nullptr, *v = nullptr;
        int *info = nullptr, *maps = nullptr;
        ~Tmp() {
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
            if (v) (void)hipFree(v);
            if (info) (void)hipFree(info);
            if (maps) (void)hipFree(maps);
        }
    } t;
    // work copy the factorisation overwrites: the context's LU work set after a root search, else a buffer of its own
    double* work = nullptr;
    if (!M && c->d_work && c->mat_cap >= nbatch) {
        work = c->d_work;
        HIP_TRY(hipMemcpyAsync(work, c->d_M, mbytes * nbatch, hipMemcpyDeviceToDevice, c->stream));
    } else {
        HIP_TRY(malloc_retry((void**)&t.a, mbytes * nbatch));
        work = t.a;
        const double* src = M ? M : c->d_M;
        HIP_TRY(hipMemcpyAsync(work, src, mbytes * nbatch, is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(malloc_retry((void**)&t.v, sizeof(double) * 2 * (size_t)n * nbatch));
    HIP_TRY(malloc_retry((void**)&t.info, sizeof(int) * nbatch));
    const bool one_wg = trace_solve_blocked_lds(n) <= 150 * 1024;  // the whole L21 panel in one workgroup's LDS
    const int max_sweeps = 60;  // (two at a converged root; the rest is for matrices that are not singular)
    if (one_wg || n <= 1024) {
        // blocked factorisations: row orders in the LU scratch.  Above the one-workgroup panel the chunked
        // multi-workgroup kernel of the Newton step factors (two workgroups per matrix, which must be resident
        // together: slices of at most half the compute units; its right-hand side is a dummy).
        const int slice_max = one_wg ? nbatch : std::max(1, c->n_cu / 2);
        const size_t need = trace_solve_blocked_scratch(n, std::min(nbatch, slice_max));
        if (need > c->lu_scratch_bytes) {
            if (c->d_lu_scratch) (void)hipFree(c->d_lu_scratch);
            c->d_lu_scratch = nullptr, c->lu_scratch_bytes = 0;
            HIP_TRY(malloc_retry(&c->d_lu_scratch, need));
            c->lu_scratch_bytes = need;
        }
        if (!one_wg) HIP_TRY(malloc_retry((void**)&t.b, mbytes * std::min(nbatch, slice_max)));
        for (int b0 = 0; b0 < nbatch; b0 += slice_max) {
            const int nb = std::min(slice_max, nbatch - b0);
            double* a0 = work + (size_t)b0 * n * n * 2;
            ScopedSpan sp(c, K_NULL);
            if (one_wg) {
                HIP_TRY(launch_lu_inplace(n, nb, a0, nullptr, nb, c->d_info, c->d_lu_scratch, c->stream));
            } else {
                HIP_TRY(hipMemsetAsync(t.b, 0, mbytes * nb, c->stream));
                const hipError_t e = launch_trace_solve_blocked(n, nb, a0, t.b, nullptr, c->d_tr, c->d_info, 2, nullptr, nb,
                                                                c->d_lu_scratch, c->stream, -1, c->opt.lu_spin_limit);
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    set_error("emme_null_vectors_batch: the chunked factorisation could not be launched (its two workgroups per matrix must be resident together)");
                    return EMME_EDEVICE;
                }
            }
            HIP_TRY(launch_null_iterate(n, a0, trace_solve_rowmaps(c->d_lu_scratch, n, nb), trace_solve_nb(), nullptr, nb,
                                        c->d_info, t.v + (size_t)b0 * n * 2, t.info + b0, max_sweeps, c->stream));
        }
    } else {
        HIP_TRY(malloc_retry((void**)&t.maps, sizeof(int) * (size_t)n * nbatch));
        ScopedSpan sp(c, K_NULL);
        HIP_TRY(launch_lu_unblocked_inplace(n, nbatch, work, t.maps, c->d_info, c->stream));
        HIP_TRY(launch_null_iterate(n, work, t.maps, n, nullptr, nbatch, c->d_info, t.v, t.info, max_sweeps, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(vecs, t.v, sizeof(double) * 2 * (size_t)n * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(info, t.info, sizeof(int) * nbatch, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return EMME_OK;
}

int emme_ctx_get_matrix(emme_ctx_t* c, int b, double* M_host) {
    if (!c || !M_host || b < 0 || b >= c->last_n || !c->d_M) return EMME_EINVAL;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpy(M_host, c->d_M + mat_doubles(c) * (size_t)b, mat_doubles(c) * sizeof(double), hipMemcpyDeviceToHost));
    return EMME_OK;
}

}  // extern "C"
