// cg_group: a column grid cut into slabs that the LIBRARY drives -- one process, one handle per slab, every slab on a
// device of the caller's choice (several slabs may share one).  What fem_amd/slabs.py does with one process per GPU over
// torch.distributed, below the C ABI, so that any host that can load the library (the Node addon, a C program) gets all
// GPUs of the node behind one object with the reference's step names.  The reference is single-threaded
// (template/script_columngrid.js); this file has no counterpart there.
//
// Written against the handle API of include/colgrid.h alone.  Rows that cross a slab edge travel device to device:
//   - RCCL (ncclSend / ncclRecv between the communicators of one ncclCommInitAll) when every slab has a device of its own
//     and librccl can be loaded;
//   - hipMemcpyPeerAsync otherwise (slabs sharing a device, CG_GROUP_TRANSPORT=copy, no librccl).
// Either way the transfer runs on a stream of its own per slab, ordered by events against the slab's kernels, so the rows
// of a lateral step that need no neighbour are updated while the halo rows travel.  Whole-domain scalars (heights, flags,
// CMP's target) are combined on the host: the handles hand them back anyway.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/colgrid.h"

namespace {

// ---- the few RCCL entry points used, resolved at run time (the library must load on hosts without librccl) ----------
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;                       // 0 = ncclSuccess
constexpr int NCCL_UINT8 = 1;                   // ncclUint8 (ncclInt8 = 0, ncclUint8 = 1)
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
#define SYM(field, name) field = (decltype(field))dlsym(lib, name)
        SYM(CommInitAll, "ncclCommInitAll"); SYM(CommDestroy, "ncclCommDestroy"); SYM(GroupStart, "ncclGroupStart");
        SYM(GroupEnd, "ncclGroupEnd"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Send || !Recv) { dlclose(lib); lib = nullptr; return false; }
        return true;
    }
};

thread_local std::string t_group_create_error;

struct GMask { std::vector<int> ids; int any = -1; bool live = false; };      // any: does it open a cell anywhere (-1: not asked yet)
struct Bufs { void* p[4] = {nullptr, nullptr, nullptr, nullptr}; size_t cap = 0; };      // send lower / upper, received lower / upper

}  // namespace

struct cg_group {
    int n = 0;
    int nx_global = 0, ny = 0;
    double dx = 0, dy = 0, zs = 0.1;
    std::vector<cg_grid*> s;
    std::vector<int> dev, i0, nxl;
    std::vector<hipStream_t> st;                   // every slab's own stream
    std::vector<hipStream_t> xfer;                 // one transfer stream per slab
    std::vector<hipEvent_t> packed, arrived, taken;   // per slab: its send rows are written / its received rows are in / its send rows have been read
    std::vector<GMask> masks;
    std::vector<Bufs> ghost, seedbuf;              // rows of whole columns (strip, wet etch, ALD) and ALD seed rows
    int gaps_anywhere = -1;                        // "some slab may hold a gap or a cavity": -1 = ask the slabs again
    bool use_rccl = false;
    Rccl rccl;
    std::vector<ncclComm_t> comm;
    uint64_t halo_bytes = 0, exchanges = 0;
    std::string err;
};

namespace {

int gfail(cg_group* G, int code, const char* fmt, ...) {
    char buf[600];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (G) G->err = buf; else t_group_create_error = buf;
    return code;
}
// a handle call failed: its message becomes the group's
int from_slab(cg_group* G, int k, int rc) {
    if (rc) { char buf[600]; snprintf(buf, sizeof buf, "slab %d: %s", k, cg_last_error(G->s[k])); G->err = buf; }
    return rc;
}
#define SLAB(G, k, call) do { int rc__ = from_slab((G), (k), (call)); if (rc__) return rc__; } while (0)
#define GHIP(G, call)                                                                                          \
    do { hipError_t e__ = (call);                                                                              \
         if (e__ != hipSuccess) return gfail((G), e__ == hipErrorOutOfMemory ? CG_ERR_NOMEM : CG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); } while (0)
#define GNEED(G) do { if (!(G)) return gfail(nullptr, CG_ERR_ARG, "null group"); } while (0)

double js_round(double x) { double r = std::floor(x); if (x - r >= 0.5) r += 1.0; return r; }

// (the stream a slab was made with: the group owns its slabs and never re-binds them, and asking the slab itself would
// launch the vertical steps it holds back for one fused launch)
hipStream_t slab_stream(cg_group* G, int k) { return G->st[k]; }

int ensure_bufs(cg_group* G, std::vector<Bufs>& set, size_t bytes) {
    for (int k = 0; k < G->n; k++) {
        Bufs& b = set[k];
        if (b.cap >= bytes) continue;
        GHIP(G, hipSetDevice(G->dev[k]));
        GHIP(G, hipDeviceSynchronize());
        for (auto& p : b.p) { if (p) (void)hipFree(p); p = nullptr; }
        b.cap = 0;
        const size_t want = std::max<size_t>(bytes, 16);
        for (auto& p : b.p) { GHIP(G, hipMalloc(&p, want)); GHIP(G, hipMemset(p, 0, want)); }
        b.cap = want;
    }
    return CG_OK;
}

// Before slab k's send rows are written again, its neighbours must have read the last ones.
int wait_taken(cg_group* G, int k) {
    GHIP(G, hipSetDevice(G->dev[k]));
    const hipStream_t st = slab_stream(G, k);
    GHIP(G, hipStreamWaitEvent(st, G->taken[k], 0));                      // (RCCL: the sends of these rows run on slab k's own transfer stream)
    if (k > 0) GHIP(G, hipStreamWaitEvent(st, G->taken[k - 1], 0));
    if (k < G->n - 1) GHIP(G, hipStreamWaitEvent(st, G->taken[k + 1], 0));
    return CG_OK;
}

// send_lower of slab k arrives as received-upper of slab k-1, send_upper as received-lower of slab k+1.  `bufs[k]` = the
// four device buffers of slab k; their contents have been written on slab k's own stream.  When the call returns every
// slab's stream has been told to wait for its rows; nothing is waited for on the host.  `between` runs after the
// transfers have been started (work that needs no neighbour rows).
template <class Between>
int exchange(cg_group* G, const std::vector<std::array<void*, 4>>& bufs, size_t bytes, Between between) {
    const int n = G->n;
    for (int k = 0; k < n; k++) {
        GHIP(G, hipSetDevice(G->dev[k]));
        GHIP(G, hipEventRecord(G->packed[k], slab_stream(G, k)));
    }
    // every transfer stream first waits for its own slab's stream: the rows are packed, and whatever read the receive
    // buffers last (the kernels of the step before) is through
    for (int k = 0; k < n; k++) { GHIP(G, hipSetDevice(G->dev[k])); GHIP(G, hipStreamWaitEvent(G->xfer[k], G->packed[k], 0)); }
    if (G->use_rccl) {
        ncclResult_t r = G->rccl.GroupStart();
        for (int k = 0; k < n && !r; k++) {
            if (k > 0) { r = G->rccl.Send(bufs[k][0], bytes, NCCL_UINT8, k - 1, G->comm[k], G->xfer[k]); if (!r) r = G->rccl.Recv(bufs[k][2], bytes, NCCL_UINT8, k - 1, G->comm[k], G->xfer[k]); }
            if (k < n - 1 && !r) { r = G->rccl.Send(bufs[k][1], bytes, NCCL_UINT8, k + 1, G->comm[k], G->xfer[k]); if (!r) r = G->rccl.Recv(bufs[k][3], bytes, NCCL_UINT8, k + 1, G->comm[k], G->xfer[k]); }
        }
        const ncclResult_t r2 = G->rccl.GroupEnd();
        if (r || r2) return gfail(G, CG_ERR_HIP, "RCCL halo exchange failed: %s", G->rccl.GetErrorString ? G->rccl.GetErrorString(r ? r : r2) : "?");
    } else {
        // every slab pulls its neighbours' rows on its transfer stream
        for (int k = 0; k < n; k++) {
            GHIP(G, hipSetDevice(G->dev[k]));
            if (k > 0) {
                GHIP(G, hipStreamWaitEvent(G->xfer[k], G->packed[k - 1], 0));
                GHIP(G, hipMemcpyPeerAsync(bufs[k][2], G->dev[k], bufs[k - 1][1], G->dev[k - 1], bytes, G->xfer[k]));
            }
            if (k < n - 1) {
                GHIP(G, hipStreamWaitEvent(G->xfer[k], G->packed[k + 1], 0));
                GHIP(G, hipMemcpyPeerAsync(bufs[k][3], G->dev[k], bufs[k + 1][0], G->dev[k + 1], bytes, G->xfer[k]));
            }
        }
    }
    // arrived[k]: slab k's received rows are in.  taken[k]: slab k's transfer stream is through with this exchange -- with
    // copies it has read its neighbours' send rows, with RCCL its neighbours have received what it sent (a receive completes
    // with its send) -- so a slab whose neighbours' `taken` have fired may write its send rows again (wait_taken).
    for (int k = 0; k < n; k++) {
        GHIP(G, hipSetDevice(G->dev[k]));
        GHIP(G, hipEventRecord(G->arrived[k], G->xfer[k]));
        GHIP(G, hipEventRecord(G->taken[k], G->xfer[k]));
    }
    { int rc = between(); if (rc) return rc; }
    for (int k = 0; k < n; k++) {
        GHIP(G, hipSetDevice(G->dev[k]));
        GHIP(G, hipStreamWaitEvent(slab_stream(G, k), G->arrived[k], 0));
    }
    G->halo_bytes += (uint64_t)bytes * 2 * (uint64_t)(n - 1);
    G->exchanges += 1;
    return CG_OK;
}
int no_work() { return CG_OK; }

std::vector<std::array<void*, 4>> as_arrays(const std::vector<Bufs>& set) {
    std::vector<std::array<void*, 4>> v(set.size());
    for (size_t k = 0; k < set.size(); k++) v[k] = {set[k].p[0], set[k].p[1], set[k].p[2], set[k].p[3]};
    return v;
}

// ---- whole-domain scalars ----------------------------------------------------------------------------------------------
int domain_state(cg_group* G, int* max_len, uint32_t* top_q) {
    int ml = 0; uint32_t tq = 0;
    for (int k = 0; k < G->n; k++) {
        int a = 0, b = 0, m = 0; uint32_t q = 0;
        SLAB(G, k, cg_cavity_state(G->s[k], &a, &b, &m, &q));
        ml = std::max(ml, m); tq = std::max(tq, q);
    }
    if (max_len) *max_len = ml;
    if (top_q) *top_q = tq;
    return CG_OK;
}

int sync_lmax(cg_group* G) {
    int m = 0;
    for (int k = 0; k < G->n; k++) { cg_info inf; SLAB(G, k, cg_get_info(G->s[k], &inf)); m = std::max(m, inf.lmax); }
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_raise_lmax(G->s[k], m));
    return CG_OK;
}

bool lateral_step(const cg_group* G, double thickness, double conformality) {
    const double t = std::isnan(thickness) ? 0.0 : std::max(0.0, thickness);
    const double S = std::max(0.0, std::min(1.0, conformality));
    return t > 0 && !(t * S < std::min(G->dx, G->dy));                    // cg1:445, 609
}

int mask_ids(cg_group* G, int mask_id, const GMask** out) {
    *out = nullptr;
    if (mask_id == CG_MASK_NONE) return CG_OK;
    if (mask_id < 0 || (size_t)mask_id >= G->masks.size() || !G->masks[mask_id].live) return gfail(G, CG_ERR_ARG, "unknown mask id %d", mask_id);
    *out = &G->masks[mask_id];
    return CG_OK;
}

int new_mask(cg_group* G, int* id) {
    for (size_t k = 0; k < G->masks.size(); k++)
        if (!G->masks[k].live) { G->masks[k] = GMask(); G->masks[k].live = true; *id = (int)k; return CG_OK; }      // (a fresh entry: nothing of the old mask survives)
    G->masks.emplace_back(); G->masks.back().live = true; *id = (int)G->masks.size() - 1;
    return CG_OK;
}

// ghost rows: one boundary row of whole columns per side, flood marks included (cg_ghost_pack's wire format)
int swap_ghosts(cg_group* G, int planes) {
    size_t bytes = 0;
    SLAB(G, 0, cg_ghost_bytes(G->s[0], planes, &bytes));
    { int rc = ensure_bufs(G, G->ghost, bytes); if (rc) return rc; }
    for (int k = 0; k < G->n; k++) {
        { int rc = wait_taken(G, k); if (rc) return rc; }
        SLAB(G, k, cg_ghost_pack(G->s[k], planes, G->ghost[k].p[0], G->ghost[k].p[1]));
    }
    return exchange(G, as_arrays(G->ghost), bytes, no_work);
}
const void* ghost_lo(const cg_group* G, int k) { return k > 0 ? G->ghost[k].p[2] : nullptr; }
const void* ghost_up(const cg_group* G, int k) { return k < G->n - 1 ? G->ghost[k].p[3] : nullptr; }

int group_strip(cg_group* G, int mat);

int group_identify_cavity(cg_group* G) {
    // the flags only change in strip / CMP / cavity passes and the steps that lay down material 2, all of them calls of
    // this file: the answer of the last look stands until one of those has run
    if (G->gaps_anywhere < 0) {
        int any = 0;
        for (int k = 0; k < G->n; k++) { int a = 0, b = 0; SLAB(G, k, cg_cavity_state(G->s[k], &a, &b, nullptr, nullptr)); any |= a | b; }
        G->gaps_anywhere = any ? 1 : 0;
    }
    if (!G->gaps_anywhere) return CG_OK;
    uint32_t top_q = 0;
    { int rc = domain_state(G, nullptr, &top_q); if (rc) return rc; }
    const uint32_t hmax_q = top_q > 0 ? top_q : ((uint32_t)js_round(1.0 / G->zs) & 0xffffu);      // maxHeight() is 1 on an empty grid (cg2:103)
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_cavity_fill(G->s[k], hmax_q));
    { int rc = group_strip(G, CG_MAT_CAVITY); if (rc) return rc; }                               // cg2:79
    G->gaps_anywhere = -1;
    return sync_lmax(G);
}

// strip_connected (cg1:867 - cg2:1) across slab edges: local fixpoints, one boundary row swapped per round
int group_strip(cg_group* G, int mat) {
    G->gaps_anywhere = -1;
    int planes = 0;
    { int rc = domain_state(G, &planes, nullptr); if (rc) return rc; }
    { int rc = swap_ghosts(G, planes); if (rc) return rc; }
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_strip_seed(G->s[k], mat, planes, ghost_lo(G, k), ghost_up(G, k)));
    { int rc = swap_ghosts(G, planes); if (rc) return rc; }
    for (int round = 0;; round++) {
        if (round > 1000000) return gfail(G, CG_ERR_STATE, "strip_connected did not converge across the slabs");
        int changed = 0;
        for (int k = 0; k < G->n; k++) { int ch = 0; SLAB(G, k, cg_strip_sweep(G->s[k], mat, planes, ghost_lo(G, k), ghost_up(G, k), &ch)); changed |= ch; }
        if (!changed) break;
        { int rc = swap_ghosts(G, planes); if (rc) return rc; }
    }
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_strip_finish(G->s[k]));
    return CG_OK;
}

int group_lateral(cg_group* G, int kind, int mask_id, int mat, double thickness, double conformality) {
    const GMask* gm = nullptr;
    { int rc = mask_ids(G, mask_id, &gm); if (rc) return rc; }
    const int n = G->n;
    int R = 0;
    for (int k = 0; k < n; k++) {
        { int rc = wait_taken(G, k); if (rc) return rc; }
        int Rk = 0;
        SLAB(G, k, cg_stencil_begin(G->s[k], kind, gm ? gm->ids[k] : CG_MASK_NONE, mat, thickness, conformality, k > 0, k < n - 1, &Rk));
        if (k == 0) R = Rk;
        else if (Rk != R) return gfail(G, CG_ERR_STATE, "slab %d reports a stencil radius of %d rows, slab 0 one of %d", k, Rk, R);
    }
    if (R == 0) return CG_OK;                              // a vertical step or a no-op: nothing crosses a slab edge
    std::vector<std::array<void*, 4>> bufs(n);
    size_t bytes = 0;
    for (int k = 0; k < n; k++) {
        size_t b = 0;
        SLAB(G, k, cg_halo_buffers(G->s[k], &bufs[k][0], &bufs[k][1], &bufs[k][2], &bufs[k][3], &b));
        if (k == 0) bytes = b;
    }
    // the tile rows whose stencil stays inside a slab are updated while the halo rows travel
    { int rc = exchange(G, bufs, bytes, [&]() -> int { for (int k = 0; k < n; k++) SLAB(G, k, cg_stencil_interior(G->s[k])); return CG_OK; }); if (rc) return rc; }
    for (int k = 0; k < n; k++) SLAB(G, k, cg_stencil_finish(G->s[k]));
    return CG_OK;
}

}  // namespace

// ===================================================================================================================
extern "C" {

const char* cg_group_last_error(const cg_group* G) { return G ? G->err.c_str() : t_group_create_error.c_str(); }

int cg_group_create(int nx_global, int ny, double dx, double dy, int lmax_init, double zscale, const int* devices, int nslabs, cg_group** out) {
    if (!out) return gfail(nullptr, CG_ERR_ARG, "out is null");
    *out = nullptr;
    if (nslabs < 1 || nslabs > 64 || !devices) return gfail(nullptr, CG_ERR_ARG, "need 1..64 slabs and a device for each");
    if (nx_global < nslabs || ny < 1) return gfail(nullptr, CG_ERR_ARG, "a %d x %d grid cannot be cut into %d slabs", nx_global, ny, nslabs);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return gfail(nullptr, CG_ERR_HIP, "no HIP device");
    for (int k = 0; k < nslabs; k++)
        if (devices[k] < 0 || devices[k] >= ndev) return gfail(nullptr, CG_ERR_ARG, "slab %d: device %d, the process sees %d", k, devices[k], ndev);
    int before = 0; (void)hipGetDevice(&before);
    cg_group* G = new cg_group();
    G->n = nslabs; G->nx_global = nx_global; G->ny = ny; G->dx = dx; G->dy = dy; G->zs = zscale;
    G->s.assign(nslabs, nullptr); G->dev.assign(devices, devices + nslabs); G->i0.resize(nslabs); G->nxl.resize(nslabs);
    G->st.assign(nslabs, nullptr); G->xfer.assign(nslabs, nullptr); G->packed.assign(nslabs, nullptr); G->arrived.assign(nslabs, nullptr); G->taken.assign(nslabs, nullptr);
    G->ghost.resize(nslabs); G->seedbuf.resize(nslabs);
    auto body = [&]() -> int {
        // rows are dealt out as evenly as whole rows allow, the first slabs taking the extra ones
        int at = 0;
        for (int k = 0; k < nslabs; k++) {
            G->i0[k] = at; G->nxl[k] = nx_global / nslabs + (k < nx_global % nslabs ? 1 : 0); at += G->nxl[k];
            GHIP(G, hipSetDevice(G->dev[k]));
            int rc = cg_create_slab(nx_global, ny, G->i0[k], G->nxl[k], dx, dy, lmax_init, zscale, &G->s[k]);
            if (rc) return gfail(G, rc, "slab %d on device %d: %s", k, G->dev[k], cg_last_error(nullptr));
            { void* p = nullptr; SLAB(G, k, cg_get_stream(G->s[k], &p)); G->st[k] = (hipStream_t)p; }
            GHIP(G, hipStreamCreateWithFlags(&G->xfer[k], hipStreamNonBlocking));
            GHIP(G, hipEventCreateWithFlags(&G->packed[k], hipEventDisableTiming));
            GHIP(G, hipEventCreateWithFlags(&G->arrived[k], hipEventDisableTiming));
            GHIP(G, hipEventCreateWithFlags(&G->taken[k], hipEventDisableTiming));
            GHIP(G, hipEventRecord(G->taken[k], G->xfer[k]));          // (so that the first wait_taken finds a recorded event)
        }
        // RCCL wants one device per rank: only when no two slabs share one
        bool distinct = true;
        for (int a = 0; a < nslabs; a++) for (int b = a + 1; b < nslabs; b++) if (G->dev[a] == G->dev[b]) distinct = false;
        const char* tr = getenv("CG_GROUP_TRANSPORT");
        if (nslabs > 1 && distinct && !(tr && !strcmp(tr, "copy")) && G->rccl.load()) {
            G->comm.assign(nslabs, nullptr);
            const ncclResult_t r = G->rccl.CommInitAll(G->comm.data(), nslabs, G->dev.data());
            if (r == 0) G->use_rccl = true;
            else { G->comm.clear(); if (tr && !strcmp(tr, "rccl")) return gfail(G, CG_ERR_HIP, "ncclCommInitAll failed: %s", G->rccl.GetErrorString ? G->rccl.GetErrorString(r) : "?"); }
        } else if (tr && !strcmp(tr, "rccl") && nslabs > 1) {
            return gfail(G, CG_ERR_UNSUPPORTED, "CG_GROUP_TRANSPORT=rccl needs librccl and a device of its own for every slab");
        }
        if (!G->use_rccl) {
            for (int a = 0; a + 1 < nslabs; a++) {                     // neighbours copy from each other's memory
                if (G->dev[a] == G->dev[a + 1]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, G->dev[a], G->dev[a + 1]) == hipSuccess && can) {
                    (void)hipSetDevice(G->dev[a]); (void)hipDeviceEnablePeerAccess(G->dev[a + 1], 0);
                    (void)hipSetDevice(G->dev[a + 1]); (void)hipDeviceEnablePeerAccess(G->dev[a], 0);
                    (void)hipGetLastError();                           // (already enabled is fine)
                }
            }
        }
        G->gaps_anywhere = 0;                                          // an empty grid has neither gaps nor cavities
        return CG_OK;
    };
    const int rc = body();
    (void)hipSetDevice(before);
    if (rc) { t_group_create_error = G->err; cg_group_destroy(G); return rc; }
    *out = G;
    return CG_OK;
}

int cg_group_destroy(cg_group* G) {
    if (!G) return CG_OK;
    for (int k = 0; k < G->n; k++) {
        (void)hipSetDevice(G->dev[k]);
        if (G->s[k]) (void)cg_sync(G->s[k]);
        if (G->xfer[k]) (void)hipStreamSynchronize(G->xfer[k]);
    }
    if (G->use_rccl) for (auto c : G->comm) if (c) (void)G->rccl.CommDestroy(c);
    for (int k = 0; k < G->n; k++) {
        (void)hipSetDevice(G->dev[k]);
        for (auto set : {&G->ghost, &G->seedbuf}) for (auto& p : (*set)[k].p) if (p) (void)hipFree(p);
        if (G->packed[k]) (void)hipEventDestroy(G->packed[k]);
        if (G->arrived[k]) (void)hipEventDestroy(G->arrived[k]);
        if (G->taken[k]) (void)hipEventDestroy(G->taken[k]);
        if (G->xfer[k]) (void)hipStreamDestroy(G->xfer[k]);
        if (G->s[k]) (void)cg_destroy(G->s[k]);
    }
    delete G;
    return CG_OK;
}

int cg_group_slab_count(const cg_group* G, int* nslabs) { if (!G || !nslabs) return CG_ERR_ARG; *nslabs = G->n; return CG_OK; }

int cg_group_slab(cg_group* G, int k, cg_grid** slab) {
    GNEED(G);
    if (k < 0 || k >= G->n || !slab) return gfail(G, CG_ERR_ARG, "slab %d of %d", k, G->n);
    *slab = G->s[k];
    return CG_OK;
}

int cg_group_transport(const cg_group* G, int* rccl, uint64_t* exchanges, uint64_t* bytes_moved) {
    if (!G) return CG_ERR_ARG;
    if (rccl) *rccl = G->use_rccl ? 1 : 0;
    if (exchanges) *exchanges = G->exchanges;
    if (bytes_moved) *bytes_moved = G->halo_bytes;
    return CG_OK;
}

int cg_group_get_info(cg_group* G, cg_info* o) {
    GNEED(G);
    if (!o) return gfail(G, CG_ERR_ARG, "null output");
    { int rc = sync_lmax(G); if (rc) return rc; }
    SLAB(G, 0, cg_get_info(G->s[0], o));
    o->nx = G->nx_global; o->i0 = 0;                      // the whole domain; `device` is the first slab's
    return CG_OK;
}

int cg_group_sync(cg_group* G) {
    GNEED(G);
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_sync(G->s[k]));
    for (int k = 0; k < G->n; k++) { GHIP(G, hipSetDevice(G->dev[k])); GHIP(G, hipStreamSynchronize(G->xfer[k])); }
    return CG_OK;
}

int cg_group_reset(cg_group* G, int lmax_init) {
    GNEED(G);
    int lm = lmax_init;
    if (lm <= 0) { cg_info inf; int rc = cg_group_get_info(G, &inf); if (rc) return rc; lm = inf.lmax; }
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_reset(G->s[k], lm));
    G->gaps_anywhere = 0;
    return CG_OK;
}

int cg_group_clear(cg_group* G) {
    GNEED(G);
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_clear(G->s[k]));
    G->gaps_anywhere = 0;
    return CG_OK;
}

int cg_group_set_plane_budget(cg_group* G, int max_planes) {
    GNEED(G);
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_set_plane_budget(G->s[k], max_planes));
    return CG_OK;
}

// ---- masks -----------------------------------------------------------------------------------------------------------
int cg_group_mask_from_shapes(cg_group* G, const cg_shape* shapes, int nshapes, const double* pts, int npts, int* mask_id) {
    GNEED(G);
    if (!mask_id) return gfail(G, CG_ERR_ARG, "null output");
    std::vector<int> ids(G->n, -1);
    for (int k = 0; k < G->n; k++) {
        const int rc = cg_mask_from_shapes(G->s[k], shapes, nshapes, pts, npts, &ids[k]);      // every slab rasterises its own rows (global cell centres)
        if (rc) { from_slab(G, k, rc); for (int q = 0; q < k; q++) (void)cg_mask_destroy(G->s[q], ids[q]); return rc; }
    }
    int id = -1; new_mask(G, &id);
    G->masks[id].ids = ids;
    *mask_id = id;
    return CG_OK;
}

int cg_group_mask_from_raster(cg_group* G, const uint8_t* host_raster, size_t n, int* mask_id) {
    GNEED(G);
    if (!mask_id || !host_raster || n != (size_t)G->nx_global * G->ny) return gfail(G, CG_ERR_ARG, "mask raster: need %zu bytes, got %zu", (size_t)G->nx_global * G->ny, n);
    std::vector<int> ids(G->n, -1);
    for (int k = 0; k < G->n; k++) {
        const int rc = cg_mask_from_raster(G->s[k], host_raster + (size_t)G->i0[k] * G->ny, (size_t)G->nxl[k] * G->ny, &ids[k]);
        if (rc) { from_slab(G, k, rc); for (int q = 0; q < k; q++) (void)cg_mask_destroy(G->s[q], ids[q]); return rc; }
    }
    int id = -1; new_mask(G, &id);
    G->masks[id].ids = ids;
    *mask_id = id;
    return CG_OK;
}

int cg_group_mask_download(cg_group* G, int mask_id, uint8_t* host_out, size_t n) {
    GNEED(G);
    const GMask* gm = nullptr;
    { int rc = mask_ids(G, mask_id, &gm); if (rc) return rc; }
    if (!gm || !host_out || n != (size_t)G->nx_global * G->ny) return gfail(G, CG_ERR_ARG, "bad mask id %d or size", mask_id);
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_mask_download(G->s[k], gm->ids[k], host_out + (size_t)G->i0[k] * G->ny, (size_t)G->nxl[k] * G->ny));
    return CG_OK;
}

int cg_group_mask_destroy(cg_group* G, int mask_id) {
    GNEED(G);
    const GMask* gm = nullptr;
    { int rc = mask_ids(G, mask_id, &gm); if (rc) return rc; }
    if (!gm) return gfail(G, CG_ERR_ARG, "bad mask id %d", mask_id);
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_mask_destroy(G->s[k], gm->ids[k]));
    G->masks[mask_id] = GMask();                           // the id may be handed out again; nothing of this mask is kept
    return CG_OK;
}

// ---- process steps ---------------------------------------------------------------------------------------------------
int cg_group_deposit_general(cg_group* G, int mask_id, int mat, double thickness, double conformality) {
    GNEED(G);
    if (G->n == 1) {
        const GMask* gm = nullptr;
        { int rc = mask_ids(G, mask_id, &gm); if (rc) return rc; }
        SLAB(G, 0, cg_deposit_general(G->s[0], gm ? gm->ids[0] : CG_MASK_NONE, mat, thickness, conformality));
        return CG_OK;
    }
    if (std::isnan(conformality)) return gfail(G, CG_ERR_ARG, "conformality is not a number");
    const bool lateral = lateral_step(G, thickness, conformality);
    if (lateral && mask_id != CG_MASK_NONE) {
        // A lateral deposit whose mask opens no cell of the DOMAIN returns before anything changes, identify_cavity
        // included -- after its kernel has entered the kernel cache (cg1:470 vs 474-485).
        const GMask* gm = nullptr;
        { int rc = mask_ids(G, mask_id, &gm); if (rc) return rc; }
        GMask& m = G->masks[mask_id];
        if (m.any < 0) {
            int any = 0;
            for (int k = 0; k < G->n; k++) { int a = 1; SLAB(G, k, cg_mask_any_open(G->s[k], m.ids[k], &a)); any |= a; }
            m.any = any ? 1 : 0;
        }
        if (!m.any) {
            for (int k = 0; k < G->n; k++) SLAB(G, k, cg_kernel_cache_touch(G->s[k], thickness, conformality));
            return CG_OK;
        }
    }
    if (mat == CG_MAT_CAVITY) G->gaps_anywhere = -1;       // material 2 is what identify_cavity strips
    { int rc = group_lateral(G, CG_STENCIL_DEPOSIT, mask_id, mat, thickness, conformality); if (rc) return rc; }
    { int rc = sync_lmax(G); if (rc) return rc; }
    if (lateral) return group_identify_cavity(G);          // cg1:543; a slab handle leaves it to its driver
    return CG_OK;
}

int cg_group_etch_general(cg_group* G, int mask_id, int mat, double thickness, double conformality) {
    GNEED(G);
    if (G->n == 1) {
        const GMask* gm = nullptr;
        { int rc = mask_ids(G, mask_id, &gm); if (rc) return rc; }
        SLAB(G, 0, cg_etch_general(G->s[0], gm ? gm->ids[0] : CG_MASK_NONE, mat, thickness, conformality));
        return CG_OK;
    }
    if (std::isnan(conformality)) return gfail(G, CG_ERR_ARG, "conformality is not a number");
    return group_lateral(G, CG_STENCIL_ETCH, mask_id, mat, thickness, conformality);
}

int cg_group_strip_connected(cg_group* G, int mat) {
    GNEED(G);
    if (G->n == 1) { SLAB(G, 0, cg_strip_connected(G->s[0], mat)); return CG_OK; }
    if (mat < 0 || mat > 255) return gfail(G, CG_ERR_ARG, "material id %d outside 0..255", mat);
    return group_strip(G, mat);
}

int cg_group_identify_cavity(cg_group* G) {
    GNEED(G);
    if (G->n == 1) { SLAB(G, 0, cg_identify_cavity(G->s[0])); return CG_OK; }
    return group_identify_cavity(G);
}

int cg_group_cmp(cg_group* G, double depth, int stop_mat) {                // cg1:747-859
    GNEED(G);
    if (G->n == 1) { SLAB(G, 0, cg_cmp(G->s[0], depth, stop_mat)); return CG_OK; }
    if (!(depth > 0)) return CG_OK;                                        // cg1:748
    uint32_t top_q = 0, stop_q = 0; int found = 0;
    for (int k = 0; k < G->n; k++) {
        uint32_t a = 0, b = 0; int f = 0;
        SLAB(G, k, cg_cmp_scan(G->s[k], stop_mat, &a, &b, &f));
        top_q = std::max(top_q, a); stop_q = std::max(stop_q, b); found |= f;
    }
    const double zTop = (double)top_q * G->zs;                             // cg1:755-781
    const double target = (stop_mat == 0 || !found) ? zTop - depth : std::max(zTop - depth, (double)stop_q * G->zs);
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_cmp_apply(G->s[k], target, stop_mat));
    G->gaps_anywhere = -1;
    return group_identify_cavity(G);                                       // cg1:858
}

int cg_group_etch_wet_step(cg_group* G, int mat, int dt_q, int use_diagonal) {      // stepOnce (cg3:250-268), identify_cavity included
    GNEED(G);
    if (G->n == 1) { SLAB(G, 0, cg_etch_wet_step(G->s[0], mat, dt_q, use_diagonal)); return CG_OK; }
    int planes = 0; uint32_t hmax_q = 0;
    { int rc = domain_state(G, &planes, &hmax_q); if (rc) return rc; }
    { int rc = swap_ghosts(G, planes); if (rc) return rc; }                 // a cut reaches one column sideways
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_etch_wet_step_slab(G->s[k], mat, dt_q, use_diagonal, hmax_q, planes, ghost_lo(G, k), ghost_up(G, k)));
    G->gaps_anywhere = -1;
    { int rc = sync_lmax(G); if (rc) return rc; }
    return group_identify_cavity(G);                                       // cg3:267
}

int cg_group_release_scratch(cg_group* G) {
    GNEED(G);
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_release_scratch(G->s[k]));
    return CG_OK;
}

int cg_group_etch_wet(cg_group* G, int mat, double total_thickness, double dt) {    // cg2:893 - cg3:308, without the slider cache
    GNEED(G);
    if (G->n == 1) { SLAB(G, 0, cg_etch_wet(G->s[0], mat, total_thickness, dt)); return CG_OK; }
    const double T = std::isnan(total_thickness) ? 0 : std::max(0.0, total_thickness);           // cg2:894-895
    if (T <= 0) return CG_OK;
    double base_dt = std::isnan(dt) ? std::min(G->dx, G->dy) : dt;                               // cg2:905
    if (!(base_dt > 1e-6)) base_dt = 1e-6;
    double R = 0, t = 0;
    int rc = CG_OK;
    while (t < T - 1e-9) {                                                                       // cg3:286-306
        R += 0.7071067811865476;
        int diag = 0;
        if (R > 1) { R -= 1; diag = 1; }
        int dt_q = (int)js_round(std::min(base_dt, T - t) / G->zs);
        if (dt_q < 1) dt_q = 1;
        rc = cg_group_etch_wet_step(G, mat, dt_q, diag); if (rc) break;
        t += (double)dt_q * G->zs;
    }
    if (!rc) rc = group_identify_cavity(G);                                                      // cg3:307
    (void)cg_group_release_scratch(G);
    return rc;
}

// ---- deposit_ALD (cg2:111-868): the seeds are collected once (cg2:826), so the slabs talk once ---------------------------
int cg_group_ald_collect_seeds(cg_group* G, double total_thickness, uint64_t* count) {
    GNEED(G);
    if (G->n == 1) { SLAB(G, 0, cg_ald_collect_seeds(G->s[0], total_thickness, count)); return CG_OK; }
    if (!(total_thickness > 0)) return gfail(G, CG_ERR_ARG, "deposit_ALD thickness %g", total_thickness);
    int planes = 0; uint32_t hmax_q = 0;
    { int rc = domain_state(G, &planes, &hmax_q); if (rc) return rc; }
    { int rc = swap_ghosts(G, planes); if (rc) return rc; }                 // the neighbours' boundary rows decide the side faces of ours
    int seed_planes = 1; uint64_t total = 0;
    for (int k = 0; k < G->n; k++) {
        int sp = 1; uint64_t c = 0;
        SLAB(G, k, cg_ald_collect_seeds_slab(G->s[k], total_thickness, hmax_q, planes, ghost_lo(G, k), ghost_up(G, k), &sp, &c));
        seed_planes = std::max(seed_planes, sp); total += c;
    }
    // the seeds of the ceil(T/dx) rows next to each edge go to the neighbour: every later coat reads them in place
    size_t bytes = 0;
    SLAB(G, 0, cg_ald_seed_halo_bytes(G->s[0], seed_planes, &bytes));
    { int rc = ensure_bufs(G, G->seedbuf, bytes); if (rc) return rc; }
    for (int k = 0; k < G->n; k++) {
        { int rc = wait_taken(G, k); if (rc) return rc; }
        SLAB(G, k, cg_ald_seed_halo_pack(G->s[k], seed_planes, G->seedbuf[k].p[0], G->seedbuf[k].p[1]));
    }
    { int rc = exchange(G, as_arrays(G->seedbuf), bytes, no_work); if (rc) return rc; }
    for (int k = 0; k < G->n; k++)
        SLAB(G, k, cg_ald_seed_halo_unpack(G->s[k], seed_planes, k > 0 ? G->seedbuf[k].p[2] : nullptr, k < G->n - 1 ? G->seedbuf[k].p[3] : nullptr));
    if (count) *count = total;
    return CG_OK;
}

int cg_group_ald_drop_seeds(cg_group* G) {
    GNEED(G);
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_ald_drop_seeds(G->s[k]));
    return CG_OK;
}

int cg_group_ald_coat(cg_group* G, int mat, double t) {                    // cg2:861-862; the caller then calls identify_cavity
    GNEED(G);
    if (G->n == 1) { SLAB(G, 0, cg_ald_coat(G->s[0], mat, t)); return CG_OK; }
    uint32_t hbase_q = 0;
    { int rc = domain_state(G, nullptr, &hbase_q); if (rc) return rc; }
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_ald_coat_slab(G->s[k], mat, t, hbase_q));
    G->gaps_anywhere = -1;
    return sync_lmax(G);
}

int cg_group_ald_increment(cg_group* G, int mat, double t) {               // one pass of the loop body cg2:844-854
    GNEED(G);
    if (G->n == 1) { SLAB(G, 0, cg_ald_increment(G->s[0], mat, t)); return CG_OK; }
    uint32_t hbase_q = 0;
    { int rc = domain_state(G, nullptr, &hbase_q); if (rc) return rc; }
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_ald_coat_slab(G->s[k], mat, t, hbase_q));
    G->gaps_anywhere = -1;
    { int rc = group_identify_cavity(G); if (rc) return rc; }               // cg2:851
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_ald_merge_slab(G->s[k], hbase_q));
    G->gaps_anywhere = -1;
    return sync_lmax(G);
}

int cg_group_deposit_ald(cg_group* G, int mat, double total_thickness, double dt) {
    GNEED(G);
    if (G->n == 1) { SLAB(G, 0, cg_deposit_ald(G->s[0], mat, total_thickness, dt)); return CG_OK; }
    const double T = std::isnan(total_thickness) ? 0 : std::max(0.0, total_thickness);           // cg2:112-113
    if (T <= 0) return CG_OK;
    double step = std::isnan(dt) ? std::max(std::max(G->dx, G->dy), std::min(G->dx, G->dy)) : dt;   // cg2:124-127
    if (!(step > 1e-3)) step = 1e-3;
    int rc = cg_group_ald_collect_seeds(G, T, nullptr);
    for (double t = step; !rc && t < T - 1e-9; t += step) rc = cg_group_ald_increment(G, mat, t);   // cg2:843-858
    if (!rc) rc = cg_group_ald_coat(G, mat, T);                                                  // cg2:861-862
    if (!rc) rc = group_identify_cavity(G);                                                      // cg2:867
    (void)cg_group_release_scratch(G);
    (void)cg_group_ald_drop_seeds(G);
    return rc;
}

// ---- single columns, scalars, host exchange ------------------------------------------------------------------------------
static int owner_of(const cg_group* G, int i) {
    for (int k = 0; k < G->n; k++) if (i >= G->i0[k] && i < G->i0[k] + G->nxl[k]) return k;
    return -1;
}

int cg_group_add_layer(cg_group* G, int i, int j, int mat, double z0_nm, double z1_nm) {
    GNEED(G);
    const int k = owner_of(G, i);
    if (k < 0) return gfail(G, CG_ERR_ARG, "column (%d, %d) is outside the %d x %d grid", i, j, G->nx_global, G->ny);
    SLAB(G, k, cg_add_layer(G->s[k], i - G->i0[k], j, mat, z0_nm, z1_nm));
    G->gaps_anywhere = -1;
    return sync_lmax(G);
}

int cg_group_remove_top_layer(cg_group* G, int i, int j) {
    GNEED(G);
    const int k = owner_of(G, i);
    if (k < 0) return gfail(G, CG_ERR_ARG, "column (%d, %d) is outside the %d x %d grid", i, j, G->nx_global, G->ny);
    SLAB(G, k, cg_remove_top_layer(G->s[k], i - G->i0[k], j));
    return CG_OK;
}

int cg_group_max_height(cg_group* G, double* z_nm, uint32_t* top_q) {
    GNEED(G);
    uint32_t q = 0;
    for (int k = 0; k < G->n; k++) { uint32_t a = 0; SLAB(G, k, cg_max_height(G->s[k], nullptr, &a)); q = std::max(q, a); }
    if (top_q) *top_q = q;
    if (z_nm) { const double z = (double)q * G->zs; *z_nm = z > 0 ? z : 1; }      // cg2:103
    return CG_OK;
}

int cg_group_checksum(cg_group* G, uint64_t* sum) {
    GNEED(G);
    if (!sum) return gfail(G, CG_ERR_ARG, "null output");
    uint64_t t = 0;
    for (int k = 0; k < G->n; k++) { uint64_t a = 0; SLAB(G, k, cg_checksum(G->s[k], &a)); t += a; }      // mod 2^64: the checksum is a sum over columns keyed by global index
    *sum = t;
    return CG_OK;
}

int cg_group_download(cg_group* G, uint8_t* mat, uint16_t* zpair, uint8_t* len) {
    GNEED(G);
    if (!mat || !zpair || !len) return gfail(G, CG_ERR_ARG, "null output array");
    cg_info inf;
    { int rc = cg_group_get_info(G, &inf); if (rc) return rc; }            // (brings every slab to the domain's Lmax: one stride)
    const size_t L = (size_t)inf.lmax;
    for (int k = 0; k < G->n; k++) {
        const size_t c0 = (size_t)G->i0[k] * G->ny;
        SLAB(G, k, cg_download(G->s[k], mat + c0 * L, zpair + 2 * c0 * L, len + c0));
    }
    return CG_OK;
}

int cg_group_upload(cg_group* G, const uint8_t* mat, const uint16_t* zpair, const uint8_t* len, int lmax) {
    GNEED(G);
    if (!mat || !zpair || !len || lmax < 1 || lmax > 256) return gfail(G, CG_ERR_ARG, "bad upload arguments (lmax=%d)", lmax);
    const size_t L = (size_t)lmax;
    for (int k = 0; k < G->n; k++) {
        const size_t c0 = (size_t)G->i0[k] * G->ny;
        SLAB(G, k, cg_upload(G->s[k], mat + c0 * L, zpair + 2 * c0 * L, len + c0, lmax));
    }
    G->gaps_anywhere = -1;
    return CG_OK;
}

int cg_group_snapshot_save(cg_group* G, int slot) { GNEED(G); for (int k = 0; k < G->n; k++) SLAB(G, k, cg_snapshot_save(G->s[k], slot)); return CG_OK; }
int cg_group_snapshot_drop(cg_group* G, int slot) { GNEED(G); for (int k = 0; k < G->n; k++) SLAB(G, k, cg_snapshot_drop(G->s[k], slot)); return CG_OK; }
int cg_group_snapshot_load(cg_group* G, int slot) {
    GNEED(G);
    for (int k = 0; k < G->n; k++) SLAB(G, k, cg_snapshot_load(G->s[k], slot));
    G->gaps_anywhere = -1;
    return CG_OK;
}

int cg_group_index_from_pos(const cg_group* G, int axis, double pos_nm, int32_t* index) {
    if (!G) return CG_ERR_ARG;
    return cg_index_from_pos(G->s[0], axis, pos_nm, index);               // (a slab answers for the whole domain)
}

int cg_group_slices(cg_group* G, int axis, const int32_t* index, int count, uint8_t* mat, uint16_t* zpair, uint8_t* len) {
    GNEED(G);
    if ((axis != 0 && axis != 1) || !index || count < 1 || !mat || !zpair || !len) return gfail(G, CG_ERR_ARG, "bad slice request");
    cg_info inf;
    { int rc = cg_group_get_info(G, &inf); if (rc) return rc; }
    const size_t L = (size_t)inf.lmax;
    if (axis == 0) {                                       // 'YZ': a whole line belongs to one slab
        const size_t line = (size_t)G->ny;
        for (int q = 0; q < count; q++) {
            const int k = owner_of(G, index[q]);
            if (k < 0) return gfail(G, CG_ERR_ARG, "slice index %d outside [0,%d)", index[q], G->nx_global);
            const int32_t local = index[q] - G->i0[k];
            SLAB(G, k, cg_slices(G->s[k], 0, &local, 1, mat + (size_t)q * line * L, zpair + 2 * (size_t)q * line * L, len + (size_t)q * line));
        }
        return CG_OK;
    }
    // 'XZ': every slab contributes its rows of every line
    const size_t line = (size_t)G->nx_global;
    for (int k = 0; k < G->n; k++) {
        const size_t part = (size_t)G->nxl[k];
        std::vector<uint8_t> m(part * count * L), l(part * count);
        std::vector<uint16_t> z(2 * part * count * L);
        SLAB(G, k, cg_slices(G->s[k], 1, index, count, m.data(), z.data(), l.data()));
        for (int q = 0; q < count; q++) {
            const size_t dst = (size_t)q * line + (size_t)G->i0[k];
            memcpy(mat + dst * L, m.data() + (size_t)q * part * L, part * L);
            memcpy(zpair + 2 * dst * L, z.data() + 2 * (size_t)q * part * L, 4 * part * L);
            memcpy(len + dst, l.data() + (size_t)q * part, part);
        }
    }
    return CG_OK;
}

}  // extern "C"
