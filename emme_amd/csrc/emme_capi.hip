// emme_capi.hip -- implementation of the C ABI declared in include/emme_hip.h:
// context (device tables, batch scratch, stream, profiling) and the batched drivers that
// stand where the reference has EigenSolver's constructor / matrixAssembler /
// newtonTraceSecantIteration (include/solver.h:396-415, 417-515, 113-160) and the
// solve_once_eigen loop (src/main.cpp:19-80).
#include "ctx.hpp"

namespace emme {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }

namespace {

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

}  // namespace
}  // namespace emme

using namespace emme;

namespace emme {

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
}  // namespace emme

namespace {

size_t mat_doubles(const emme_ctx* c) { return (size_t)c->dim * c->dim * 2; }

// Which contexts get the tiled record layout + dense (matrix-core) fill (assemble_dense.hip): both quadrature orders,
// electrostatic and electromagnetic (BASELINE.json's configurations use electrostatic GK15 and electromagnetic GK31),
// on folded records, with the default fill option.  The dense path carries the safe_exp-clamped tails (<= 4e-14 absolute), so inputs whose absolute quadrature
// goal (integration_accuracy) is tighter than 1e-9 keep the exact kernels.
bool wants_tiled(const emme_params_t& p, bool es, bool folded, int fill) {
    const bool shape = p.integration_start_points == 15 || p.integration_start_points == 31;
    (void)es;
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
    // two sweeps at a converged root; the rest is for matrices that are not singular (chains that never converged):
    // a launch lasts as long as its slowest matrix, 0.18 ms per sweep at n = 256.  Measured on the 128 matrices of the
    // headline search (worst 1 - overlap against the SVD where the SVD itself determines the vector): 60 sweeps
    // 11.6 ms / 2.7e-14, 30 sweeps 6.7 ms / 8.9e-14, 20 sweeps 4.9 ms / 2.2e-9
    const int max_sweeps = 30;
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
