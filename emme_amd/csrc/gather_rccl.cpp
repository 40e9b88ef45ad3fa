// gather_rccl.cpp -- the ONE collective of the multi-GPU scan: an RCCL all-gather of the found roots.
//
// The reference runs its parameter scan sequentially in one process (src/main.cpp:264-324).  Here
// the (parameter set, omega guess) items are dealt round-robin to one process per GPU (item k ->
// rank k mod world), every rank runs its Newton chains with no exchange at all, and at the end
// each rank contributes {omega_re, omega_im, iters, info} = 32 B per item to a single
// ncclAllGather over xGMI.  4 KiB per GPU at BASELINE configs[3]: latency-bound,
// one call, nothing to bucket.
//
// RCCL is bound at run time (dlopen): a single-GPU user of libemme_hip.so does not load the
// ~500 MB library, and a process that already carries a copy (PyTorch ships one as "librccl.so")
// shares it instead of loading a second one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/emme_hip.h"

namespace emme {
void set_error(const std::string& msg);
}

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy that is already in the process first (torch's has no soname and is known by its
        // file name), then the ROCm one
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* p : paths)
            if (!r.handle) r.handle = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) {
            r.why = std::string("cannot load RCCL: ") + dlerror();
            return;
        }
        auto sym = [&](const char* s) {
            void* p = dlsym(r.handle, s);
            if (!p && r.why.empty()) r.why = std::string("RCCL lacks symbol ") + s;
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return &r;
}

// null (and the reason in emme_last_error) when RCCL could not be bound
Rccl* rccl_or_error() {
    Rccl* r = rccl();
    if (r->why.empty()) return r;
    emme::set_error(r->why);
    return nullptr;
}

int rccl_fail(const char* what, ncclResult_t rc) {
    Rccl* r = rccl();
    emme::set_error(std::string(what) + ": " + (r->GetErrorString ? r->GetErrorString(rc) : "RCCL error"));
    return EMME_EDEVICE;
}

}  // namespace

struct emme_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    double *d_send = nullptr, *d_recv = nullptr;
    size_t cap = 0;  // items per rank the device buffers hold
};

static_assert(EMME_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ncclUniqueId is passed through as bytes");

extern "C" {

// ---- the slot mapping of the gather, host only (no RCCL, no device: unit-tested on the CPU build) -----------------
// Every rank contributes m = ceil(n_total / world) slots of 4 doubles {w_re, w_im, iters, info}; its n_local results
// fill the first slots, the rest is NaN padding.  Item k of the scan was solved by rank k mod world and sits in that
// rank's slot k / world.
int emme_gather_slots(int n_total, int world) {
    if (n_total < 1 || world < 1) return EMME_EINVAL;
    return (n_total + world - 1) / world;
}

int emme_gather_share(int n_total, int world, int rank) {
    if (n_total < 1 || world < 1 || rank < 0 || rank >= world) return EMME_EINVAL;
    return rank < n_total ? (n_total - rank + world - 1) / world : 0;
}

int emme_gather_pack(int rank, int world, const double* roots, const int* iters, const int* info, int n_local,
                     int n_total, double* send) {
    if (!send || n_local < 0 || (n_local > 0 && (!roots || !iters || !info))) return EMME_EINVAL;
    const int share = emme_gather_share(n_total, world, rank);
    if (share < 0) return EMME_EINVAL;
    // the deal is fixed: checked BEFORE any collective -- a rank with a different idea of its share would enter the
    // all-gather with the wrong contents
    if (n_local != share) {
        emme::set_error("emme_gather_roots: n_local does not match the round-robin share of this rank");
        return EMME_EINVAL;
    }
    const size_t m = (size_t)emme_gather_slots(n_total, world);
    for (size_t q = 0; q < 4 * m; ++q) send[q] = std::numeric_limits<double>::quiet_NaN();
    for (int k = 0; k < n_local; ++k) {
        send[4 * (size_t)k + 0] = roots[2 * k];
        send[4 * (size_t)k + 1] = roots[2 * k + 1];
        send[4 * (size_t)k + 2] = (double)iters[k];
        send[4 * (size_t)k + 3] = (double)info[k];
    }
    return EMME_OK;
}

int emme_gather_unpack(int world, int n_total, const double* all, double* roots_all, int* iters_all, int* info_all) {
    if (!all || !roots_all || !iters_all || !info_all || n_total < 1 || world < 1) return EMME_EINVAL;
    const size_t m = (size_t)emme_gather_slots(n_total, world);
    for (int k = 0; k < n_total; ++k) {
        const int rk = k % world;
        const size_t slot = (size_t)rk * m + (size_t)(k / world);
        roots_all[2 * k] = all[4 * slot + 0];
        roots_all[2 * k + 1] = all[4 * slot + 1];
        iters_all[k] = (int)all[4 * slot + 2];
        info_all[k] = (int)all[4 * slot + 3];
    }
    return EMME_OK;
}

// 0 when RCCL can be bound in this process (not a collective: every rank asks for itself BEFORE the collective
// emme_comm_create, so that all ranks can agree on a fall-back instead of some of them waiting in ncclCommInitRank)
int emme_comm_available(void) { return rccl_or_error() ? EMME_OK : EMME_EDEVICE; }

int emme_comm_unique_id(unsigned char* id) {
    if (!id) return EMME_EINVAL;
    Rccl* r = rccl_or_error();
    if (!r) return EMME_EDEVICE;
    ncclUniqueId u;
    const ncclResult_t rc = r->GetUniqueId(&u);
    if (rc != ncclSuccess) return rccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return EMME_OK;
}

int emme_comm_create(const unsigned char* id, int rank, int world, int device, emme_comm_t** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return EMME_EINVAL;
    *out = nullptr;
    Rccl* r = rccl_or_error();
    if (!r) return EMME_EDEVICE;
    if (device < 0 && hipGetDevice(&device) != hipSuccess) {
        emme::set_error("no HIP device");
        return EMME_EDEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) {
        emme::set_error("hipSetDevice failed");
        return EMME_EDEVICE;
    }
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    emme_comm* c = new emme_comm;
    c->rank = rank, c->world = world, c->device = device;
    const ncclResult_t rc = r->CommInitRank(&c->comm, world, u, rank);
    if (rc != ncclSuccess) {
        delete c;
        return rccl_fail("ncclCommInitRank", rc);
    }
    *out = c;
    return EMME_OK;
}

void emme_comm_destroy(emme_comm_t* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    Rccl* r = rccl();
    if (r->why.empty() && c->comm) (void)r->CommDestroy(c->comm);
    delete c;
}

int emme_gather_roots(emme_comm_t* c, void* hip_stream, const double* roots, const int* iters,
                      const int* info, int n_local, int n_total, double* roots_all, int* iters_all,
                      int* info_all) {
    if (!c || !roots || !iters || !info || !roots_all || !iters_all || !info_all || n_total < 1 || n_local < 0)
        return EMME_EINVAL;
    const size_t m = (size_t)emme_gather_slots(n_total, c->world);  // slots per rank (largest share)
    std::vector<double> pack(m * 4);
    {   // the share is checked here, BEFORE the collective
        const int rc = emme_gather_pack(c->rank, c->world, roots, iters, info, n_local, n_total, pack.data());
        if (rc != EMME_OK) return rc;
    }
    Rccl* r = rccl_or_error();
    if (!r) return EMME_EDEVICE;
    if (hipSetDevice(c->device) != hipSuccess) return EMME_EDEVICE;
    hipStream_t st = (hipStream_t)hip_stream;
    if (m > c->cap) {
        if (c->d_send) (void)hipFree(c->d_send);
        if (c->d_recv) (void)hipFree(c->d_recv);
        c->d_send = c->d_recv = nullptr, c->cap = 0;
        if (hipMalloc((void**)&c->d_send, m * 4 * sizeof(double)) != hipSuccess ||
            hipMalloc((void**)&c->d_recv, m * 4 * sizeof(double) * c->world) != hipSuccess) {
            emme::set_error("hipMalloc failed for the gather buffers");
            return EMME_ENOMEM;
        }
        c->cap = m;
    }
    std::vector<double> all(m * 4 * c->world);
    if (hipMemcpyAsync(c->d_send, pack.data(), pack.size() * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess)
        return EMME_EDEVICE;
    const ncclResult_t rc = r->AllGather(c->d_send, c->d_recv, m * 4, ncclDouble, c->comm, st);
    if (rc != ncclSuccess) return rccl_fail("ncclAllGather", rc);
    if (hipMemcpyAsync(all.data(), c->d_recv, all.size() * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        emme::set_error("copy of the gathered roots failed");
        return EMME_EDEVICE;
    }
    return emme_gather_unpack(c->world, n_total, all.data(), roots_all, iters_all, info_all);
}

}  // extern "C"
