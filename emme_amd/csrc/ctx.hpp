// ctx.hpp -- the context behind the C ABI (include/emme_hip.h) and the host-side helpers its translation
// units share: emme_capi.hip (the ABI entry points and the Newton loop), ctx_cache.hip (buffer pool and the
// HBM node cache's host side), ctx_fill.hip (the fill dispatcher).
#pragma once
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

void set_error(const std::string& msg);

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            emme::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));       \
            return e_ == hipErrorOutOfMemory ? EMME_ENOMEM : EMME_EDEVICE;             \
        }                                                                              \
    } while (0)

enum Kind { K_ASM = 0, K_LIN = 1, K_OTHER = 2, K_DEFER = 3, K_CACHE = 4, K_NULL = 5 };

}  // namespace emme

// (the C ABI's opaque context type lives at global scope; its members are types of namespace emme)
using emme::DevParams;
using emme::NodeCacheGeom;
using emme::NODE_CACHE_MAX_SUB;

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

namespace emme {

// ---- ctx_cache.hip: buffer pool and node cache --------------------------------------------------------
void pool_release_all();
hipError_t pool_alloc(void** out, size_t bytes, int device);
hipError_t malloc_retry(void** out, size_t bytes);
void pool_free(void* p, size_t bytes, int device);
long cache_items(const emme_ctx* c);
size_t cache_part_bytes(const emme_ctx* c, int gk_points, const NodeCacheGeom& g, int part);
bool ensure_node_cache(emme_ctx* c, const AssembleLaunch& L, int cls);
void add_cache_subtree(emme_ctx* c, const AssembleLaunch& L, int depth, unsigned long long path, int cls);

// ---- ctx_fill.hip: the fill dispatcher ------------------------------------------------------------------
int do_assemble(emme_ctx* c, int nbatch, const double* d_omega, const int* d_active,
                const int* host_active, double* d_M, const double* d_Mold, double* d_Mp,
                const double* d_domega, const unsigned long long* cost = nullptr,
                const double* host_omega = nullptr, bool newton_loop = false, bool force_uncached = false);

// ---- emme_capi.hip ----------------------------------------------------------------------------------------
hipEvent_t get_event(emme_ctx* c);

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

}  // namespace emme
