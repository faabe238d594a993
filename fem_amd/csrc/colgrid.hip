// libcolgrid: host side of the C ABI declared in include/colgrid.h.
// Owns the device arrays of a grid (both forms of the columns, see cg_kernels.hip.h), the mask
// rasters, the conformal-kernel cache and the scratch of the incremental models; enqueues the
// kernels of cg_kernels.hip.h / cg_interval_kernels.hip.h on the handle's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/colgrid.h"
#include "cg_kernels.hip.h"
#include "cg_interval_kernels.hip.h"
#include "cg_mask_bin.hip.h"
#include "cg_launch.h"

using namespace cgk;

// groups of 4 columns in flight per thread in the vertical kernels (tuning knob: env CG_VU = 1, 2, 4)
static int vu_setting() {
    static int v = [] { const char* e = getenv("CG_VU"); int x = e ? atoi(e) : 4; return (x == 1 || x == 2 || x == 4) ? x : 4; }();
    return v;
}
template <bool ETCH>
static void launch_vertical(cg_grid* g, const uint8_t* mask, int mat, double t, int P, bool fresh_grid);
// columns per tile row of the symmetric stencil kernel: 128 for the memory-bound radii, 64 above
// (measured: -5 % at R <= 2, +6 % at R = 7 for 128); env CG_TJS = 64, 128, 256 overrides
static int sym_tile_cols(int Re) {
    static int v = [] { const char* e = getenv("CG_TJS"); int x = e ? atoi(e) : 0; return (x == 64 || x == 128 || x == 256) ? x : 0; }();
    return v ? v : (Re <= 2 ? 128 : 64);
}

// the row-streaming lateral kernel: smallest radius it takes (env CG_STREAM_MINR; 9 = never) and rows per tile (env CG_STREAM_TI)
static int stream_min_radius() {
    static int v = [] { const char* e = getenv("CG_STREAM_MINR"); int x = e ? atoi(e) : 3; return (x >= 1 && x <= 9) ? x : 3; }();
    return v;
}
// Rows per tile of the row-streaming lateral kernels.  A tile stages 2R rows before its first output row, so wide kernels
// gain from taller tiles (16384^2: R = 7 etch 1.75 -> 1.69 ms, R = 6 deposit 0.82 -> 0.78 with 128 rows) while the narrow
// ones prefer more, smaller tiles; CG_STREAM_TI sets one height for all.
static int stream_tile_rows(int R) {
    static int v = [] { const char* e = getenv("CG_STREAM_TI"); int x = e ? atoi(e) : 0; return (x >= 8 && x <= 4096) ? x : 0; }();
    return v ? v : (R >= 6 ? 128 : 64);
}

namespace {

thread_local std::string t_create_error;

// device address of the index array of cg_slices_noalloc on every GPU a handle has been created on (see there)
constexpr int MAX_DEVICES = 64;
int32_t* g_scratch_ptr[MAX_DEVICES] = {nullptr};
std::mutex g_scratch_mutex;

// One cached conformal kernel (script_columngrid.js:324-357): radius, the thickness it was
// built for (the reference reuses it as `baseAdd` on a cache hit) and its taps on the device.
struct ConfKernel {
    int key[4];
    int Rx, Ry = 0;
    double baseAdd;
    std::vector<int> du, dv;
    std::vector<float> add;
    Tap* d_taps = nullptr;  int ntaps = 0;       // float32-exact taps
    Tap* d_taps64 = nullptr; int ntaps64 = 0;    // taps that must be evaluated in double
    int dev_R = -1;                              // pitch the device offsets were built for
    // symmetric form for k_stencil_sym: effective radius, quadrant table, fine taps at its pitch
    int Re = 0; bool sym_ok = false;
    std::vector<float> quad;                     // (Re+1)^2 increments, -inf = not a float32 tap
    Tap* d_taps64_sym = nullptr; int ntaps64_sym = 0;
    float* d_quad = nullptr;                     // the quadrant table on the device (k_stencil_stream)
};

// A device buffer that grows and shrinks in place: a reserved address range of which a prefix is
// backed by physical chunks (HIP virtual memory management).  Growing maps one more chunk behind
// the data -- nothing is copied, pointers stay valid -- and shrinking unmaps whole chunks from the end.
struct Elastic {
    char* base = nullptr; size_t reserved = 0, mapped = 0, unit = 0;      // unit: bytes of every chunk
    std::vector<hipMemGenericAllocationHandle_t> chunks;
    bool plain = false;     // device without virtual memory management: one hipMalloc block, growth copies (base changes)
};

struct Snapshot {
    uint8_t* len = nullptr; uint32_t* rec = nullptr; uint32_t* top = nullptr;
    int nextp = 0;
    int planes = 0;                                       // planes copied (the ones in use when it was taken)
    bool split = true, len_valid = true;
    int lmax = 0; int lmax_vis = 0; int len_bound = 0; bool has_gaps = false, has_cavity = false;
};

struct Pending {
    bool prefold = false;       // the queue of vertical steps on the reset grid was left waiting for this deposit (see cg_stencil_begin)
    bool active = false;
    int kind = 0, mask_id = CG_MASK_NONE, mat = 0, R = 0;
    double t_eff = 0;
    ConfKernel* K = nullptr;
    bool has_lower = false, has_upper = false;
    bool interior_done = false, may_overflow = false; int P = -1;     // set by cg_stencil_interior
    bool held = false;          // cg_stencil_interior launched nothing: the deposit will wait for the steps behind it (cg_stencil_finish)
};

}  // namespace

struct cg_grid {
    int device = 0;
    int nx = 0, ny = 0, nx_global = 0, i0 = 0;
    size_t N = 0;
    double dx = 0, dy = 0, zscale = 0.1;
    int lmax = 32;                                        // planes allocated
    int lmax_vis = 32;                                    // the Lmax the reference would have now (stride of host-side arrays)
    uint8_t* len = nullptr; uint32_t* rec = nullptr;      // see cg_kernels.hip.h for the layout
    uint32_t* top = nullptr;                              // split form: top interval of every column
    uint32_t* top_alt = nullptr;                          // the buffer a lateral step writes its new tops to (then swapped)
    int nextp = 0;                                        // split form: first plane no interval lives in yet
    bool split = true;                                    // which form the columns are in right now
    bool len_valid = true;                                // split form: len[] agrees with the chains (see ensure_len)
    bool pristine = false;                                // every column is empty, but len / top still hold old bytes (see zero_planes)
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<uint8_t*> masks;
    std::vector<int8_t> mask_any;                         // per mask: does it open any column of this handle (-1: not looked at yet)
    std::deque<ConfKernel*> kcache;
    // halo messages of a lateral step on a slab
    uint8_t* halo[4] = {nullptr, nullptr, nullptr, nullptr}; size_t halo_cap = 0;  // send lo/up, recv lo/up
    Pending pend;
    std::map<int, Snapshot> snaps;
    // invariants tracked on the host
    StreamOpts sopt = {};       // float-arithmetic licences of the streaming lateral kernel that depend on zscale alone (see StreamOpts)
    int len_bound = 0;          // upper bound of max(len); INT32_MAX = unknown
    bool has_gaps = false;      // some column may have a vertical gap between layers
    bool has_cavity = false;    // some column may hold a cavity layer (material 2)
    // small device words: flags[0] overflow, [1] any etchable, [2] gap seen; red[0..1] reductions
    int* d_flags = nullptr; unsigned* d_red = nullptr; unsigned long long* d_sum = nullptr;
    unsigned long long* d_stats = nullptr; bool stats_on = false; uint64_t stats_host[CG_NSTAT] = {0};
    // slice staging
    uint8_t* sl_mat = nullptr; uint16_t* sl_zp = nullptr; uint8_t* sl_len = nullptr; size_t sl_cap = 0, sl_cols_cap = 0;   // slots / columns allocated
    int32_t* sl_idx = nullptr; int sl_idx_cap = 0; std::vector<int32_t> sl_idx_host; int sl_idx_axis = -1;   // the request last uploaded
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // per-kernel timing (cg_profile): event pairs recorded around launches, resolved on read
    bool prof_on = false;
    struct ProfRec { int kind; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[CG_NKIND] = {0}; uint64_t prof_cnt[CG_NKIND] = {0};
    // caller-owned halo buffers (cg_set_halo_buffers)
    bool halo_external = false;
    // etch_wet / deposit_ALD write a second set of planes (ALD's cavity merge a third) and swap
    uint8_t* alt_len[2] = {nullptr, nullptr}; uint32_t* alt_rec[2] = {nullptr, nullptr}; int alt_lmax[2] = {0, 0};
    // rec (the grid's and the two spare sets') live in elastic buffers; rec / alt_rec are their bases
    Elastic rec_el, alt_el[2];
    int plane_budget = 256;                                // most planes the handle may map before it compacts instead (cg_set_plane_budget)
    // deposit_ALD: seeds of the exposed faces (plane-major like the grid), their bounding box, the tap table
    uint16_t* seed_n = nullptr; uint2* seed = nullptr; int seed_planes = 0; bool has_seeds = false;
    uint32_t* ald_runs = nullptr; uint8_t* ald_runs_n = nullptr; size_t ald_runs_cap = 0;      // k_ald_grow's output for the coat in hand (ald_runs_cap = classes x seed_stride entries)
    int seed_bbox[4] = {0, -1, 0, -1}; uint64_t seed_total = 0;
    // a slab also holds the seeds of seed_rows rows of each neighbour: column (i, j) is entry seed_off + i*ny + j,
    // planes are seed_stride entries apart (a whole grid: seed_off = 0, seed_stride = N, no neighbour rows)
    size_t seed_stride = 0, seed_off = 0; int seed_rows = 0; bool seed_lo = false, seed_hi = false;
    int* d_bbox = nullptr;
    uint32_t* d_sink = nullptr;                            // 4 MiB that nobody reads (StreamOpts::sink)
    char* mask_tab = nullptr; size_t mask_tab_cap = 0;     // cg_mask_from_shapes: shape table, points, bins (grow-only, see cg_release_scratch)
    int* mask_list = nullptr; size_t mask_list_cap = 0;    // ... and the per-tile shape lists
    uint8_t* cand = nullptr;                               // strip_connected: columns that still hold an unmarked node
    uint32_t* uf = nullptr; size_t uf_cap = 0;             // strip_connected: the union-find forest over the nodes (uf_cap words), kept until cg_release_scratch
    bool uf_on = false, uf_joined = false;                 // this flood runs on the forest / the forest has been joined
    unsigned uf_nblk = 0;                                  // blocks of the launch that fills the lists (the launches that read them take the same grid)
    uint32_t* uf_list = nullptr; uint32_t* uf_counts = nullptr; size_t uf_segcap = 0;    // (inside uf) columns with an unmarked node, listed per seed block
    // vertical steps queued for one fused launch (see k_vertical_fused): nothing has looked at the grid since the first
    struct VQOp { const uint8_t* mask; bool etch; int mat; double t; int P; bool fresh; };
    // A lateral deposit whose launch waits for the vertical steps that follow it (they go out inside its kernel, see
    // k_stencil_lean's FOLD form).  Everything the host tracks is settled as if it had run, except the swap of the two
    // top buffers, which follows the launch.
    // pre: the run of vertical steps that was waiting on the reset grid when the deposit came (applied in the same launch,
    // k_stencil_lean's FRESH form).
    struct LatPending { bool active = false; const uint8_t* mask = nullptr; int mat = 0; double t_eff = 0; int R = 0; ConfKernel* K = nullptr; int P = -1;
                        const uint8_t* halo_lo = nullptr; const uint8_t* halo_up = nullptr;      // a slab's deposit: the neighbours' rows it received
                        std::vector<VQOp> pre; } lat;
    std::vector<VQOp> vq;
    bool fuse = true;                                      // CG_NO_FUSE=1: every vertical step is launched on its own
    struct VOptEntry { double t; bool etch; VertOpts vo; };
    std::vector<VOptEntry> vopt_cache;                     // integer licences of the last vertical steps (vert_opts)
    AldTap* d_ald_taps = nullptr; int ald_taps_cap = 0;
    std::string err;
};

namespace {

int fail(cg_grid* g, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (g) g->err = buf; else t_create_error = buf;
    return code;
}

hipEvent_t prof_event(cg_grid* g) {
    if (!g->prof_pool.empty()) { hipEvent_t e = g->prof_pool.back(); g->prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; (void)hipEventCreate(&e); return e;
}
// RAII bracket: records an event pair around the launches made while it is alive
struct ProfScope {
    cg_grid* g; int kind; hipEvent_t a = nullptr;
    ProfScope(cg_grid* g_, int kind_) : g(g_), kind(kind_) {
        if (g->prof_on) { a = prof_event(g); if (a) (void)hipEventRecord(a, g->stream); }
    }
    ~ProfScope() {
        if (a) { hipEvent_t b = prof_event(g); if (b) { (void)hipEventRecord(b, g->stream); g->prof_recs.push_back({kind, a, b}); } }
    }
};

#define HIPCHK(g, call)                                                                       \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            return fail(g, e__ == hipErrorOutOfMemory ? CG_ERR_NOMEM : CG_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                        hipGetErrorString(e__), __FILE__, __LINE__);                          \
    } while (0)

int materialize(cg_grid* g);
int flush_vertical(cg_grid* g);
}
static int scratch_resolve(cg_grid* g);
namespace {
#define NEED_H0(g)                                                                            \
    do { if (!(g)) return fail(nullptr, CG_ERR_ARG, "null handle");                           \
         hipError_t e__ = hipSetDevice((g)->device);                                          \
         if (e__ != hipSuccess) return fail(g, CG_ERR_HIP, "hipSetDevice(%d): %s", (g)->device, hipGetErrorString(e__)); } while (0)
// every entry point but the one that queues vertical steps first launches what is queued
#define NEED_H(g)                                                                             \
    do { NEED_H0(g); if (!(g)->vq.empty() || (g)->lat.active) { int rcf__ = flush_vertical(g); if (rcf__) return rcf__; } } while (0)
// entry points that read or write the columns: a grid that was reset but whose streams have not
// been cleared yet (see zero_planes) is cleared first
#define NEED(g)                                                                               \
    do { NEED_H(g); { int rcm__ = materialize(g); if (rcm__) return rcm__; } } while (0)

double js_round(double x) { double r = std::floor(x); if (x - r >= 0.5) r += 1.0; return r; }

int next_pow2(int n) {                       // script_columngrid.js:302-304
    int v = std::max(1, n - 1);
    return 1 << (32 - __builtin_clz((unsigned)v));
}

int js_to_int32(double v) {                  // (v | 0)
    if (!std::isfinite(v)) return 0;
    double m = std::fmod(std::trunc(v), 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int)(uint32_t)m;
}

GridView view(const cg_grid* g) {
    GridView v;
    v.len = g->len; v.rec = g->rec; v.top = g->top; v.top_w = g->top; v.split = g->split ? 1 : 0;
    v.P = -1;
    v.N = g->N; v.nx = g->nx; v.ny = g->ny; v.lmax = g->lmax; v.i0 = g->i0; v.tile_i0 = 0; v.tile_in = 0; v.sup_i = 1; v.sup_j = 1; v.ti = 0; v.zs = g->zscale; v.inv_zs = 1.0 / g->zscale;
    return v;
}

unsigned blocks_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (unsigned)std::min<size_t>(std::max<size_t>(b, 1), 256 * 16);
}

// ---- elastic buffers ---------------------------------------------------------------------------
int el_reserve(cg_grid* g, Elastic& e, size_t bytes, size_t unit_bytes) {
    if (e.base || e.plain) return CG_OK;
    int vmm = 0;
    if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, g->device) != hipSuccess) { (void)hipGetLastError(); vmm = 0; }
    if (!vmm || getenv("CG_NO_VMM")) { e.plain = true; e.reserved = bytes; e.unit = std::max<size_t>(unit_bytes, 1); e.mapped = 0; return CG_OK; }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = g->device;
    size_t gran = 0;
    HIPCHK(g, hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    gran = std::max<size_t>(gran, (size_t)2 << 20);
    // All chunks of one range have the same size: with chunks of different sizes in one range
    // hipMemSetAccess refused some of them ("invalid argument", ROCm 7.2), with equal ones never.
    e.unit = (std::max<size_t>(unit_bytes, 1) + gran - 1) / gran * gran;
    e.reserved = (std::max<size_t>(bytes, 1) + e.unit - 1) / e.unit * e.unit;
    void* p = nullptr;
    HIPCHK(g, hipMemAddressReserve(&p, e.reserved, 0, nullptr, 0));
    e.base = (char*)p; e.mapped = 0;
    return CG_OK;
}
// Make at least `bytes` of the buffer usable (growth), or give back the chunks beyond them (shrink).
int el_resize(cg_grid* g, Elastic& e, size_t bytes) {
    if (e.plain) {
        if (bytes <= e.mapped && e.base) return CG_OK;                 // never shrinks
        const size_t want = std::max<size_t>(bytes, 16);
        char* nb = nullptr;
        if (hipMalloc(&nb, want) != hipSuccess) { (void)hipGetLastError(); return fail(g, CG_ERR_NOMEM, "no device memory for %zu bytes of planes", want); }
        if (e.base) {
            HIPCHK(g, hipMemcpyAsync(nb, e.base, e.mapped, hipMemcpyDeviceToDevice, g->stream));
            HIPCHK(g, hipStreamSynchronize(g->stream));
            (void)hipFree(e.base);
        }
        e.base = nb; e.mapped = want;
        return CG_OK;
    }
    const size_t want = (bytes + e.unit - 1) / e.unit * e.unit;
    if (want > e.reserved) return fail(g, CG_ERR_NOMEM, "elastic buffer: %zu bytes asked, %zu reserved", want, e.reserved);
    while (e.mapped < want) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = g->device;
        hipMemGenericAllocationHandle_t h;
        hipError_t er = hipMemCreate(&h, e.unit, &prop, 0);
        if (er != hipSuccess) { (void)hipGetLastError(); return fail(g, CG_ERR_NOMEM, "no device memory for %zu more bytes of planes (%s)", e.unit, hipGetErrorString(er)); }
        er = hipMemMap(e.base + e.mapped, e.unit, 0, h, 0);
        if (er != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); return fail(g, CG_ERR_HIP, "hipMemMap: %s", hipGetErrorString(er)); }
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice; acc.location.id = g->device; acc.flags = hipMemAccessFlagsProtReadWrite;
        er = hipMemSetAccess(e.base + e.mapped, e.unit, &acc, 1);
        if (er != hipSuccess) {
            (void)hipGetLastError(); (void)hipMemUnmap(e.base + e.mapped, e.unit); (void)hipMemRelease(h);
            return fail(g, CG_ERR_HIP, "hipMemSetAccess(base %p + %zu, %zu bytes): %s", (void*)e.base, e.mapped, e.unit, hipGetErrorString(er));
        }
        e.chunks.push_back(h); e.mapped += e.unit;
    }
    bool synced = false;
    while (e.mapped >= want + e.unit) {
        if (!synced) { HIPCHK(g, hipStreamSynchronize(g->stream)); synced = true; }
        const hipMemGenericAllocationHandle_t h = e.chunks.back(); e.chunks.pop_back();
        e.mapped -= e.unit;
        HIPCHK(g, hipMemUnmap(e.base + e.mapped, e.unit));
        HIPCHK(g, hipMemRelease(h));
    }
    return CG_OK;
}
void el_free(Elastic& e) {
    if (e.plain) { if (e.base) (void)hipFree(e.base); e.base = nullptr; e.mapped = 0; e.plain = false; return; }
    while (!e.chunks.empty()) {
        const hipMemGenericAllocationHandle_t h = e.chunks.back(); e.chunks.pop_back();
        e.mapped -= e.unit;
        (void)hipMemUnmap(e.base + e.mapped, e.unit); (void)hipMemRelease(h);
    }
    if (e.base) (void)hipMemAddressFree(e.base, e.reserved);
    e.base = nullptr; e.reserved = 0; e.mapped = 0;
}

constexpr int MAX_PLANES = 256;
constexpr int SPLIT_PLANES = LNK_NONE;                  // planes a record's link can name: the split form holds columns of up to SPLIT_PLANES + 1 intervals

void free_planes(cg_grid* g) {
    if (g->len) (void)hipFree(g->len);
    if (g->top) (void)hipFree(g->top);
    if (g->top_alt) (void)hipFree(g->top_alt);
    el_free(g->rec_el);
    g->len = nullptr; g->rec = nullptr; g->top = nullptr; g->top_alt = nullptr;
    g->lmax = 0;
}

// Planes for `lmax` intervals per column.  The column streams are allocated once; the planes are
// address ranges for MAX_PLANES planes of which the first lmax are backed (see Elastic).
int alloc_planes(cg_grid* g, int lmax) {
    if (!g->len) {
        HIPCHK(g, hipMalloc(&g->len, std::max<size_t>(g->N, 4)));
        HIPCHK(g, hipMalloc(&g->top, std::max<size_t>(4 * g->N, 16)));
        HIPCHK(g, hipMalloc(&g->top_alt, std::max<size_t>(4 * g->N, 16)));
    }
    int rc = el_reserve(g, g->rec_el, 4 * g->N * (size_t)MAX_PLANES, 4 * g->N * 16); if (rc) return rc;        // chunks of 16 planes
    rc = el_resize(g, g->rec_el, 4 * g->N * (size_t)lmax); if (rc) return rc;
    g->rec = (uint32_t*)g->rec_el.base;
    g->lmax = lmax;
    return CG_OK;
}

// Empty every column.  Nothing is cleared here: the grid is marked pristine, and either the next
// step is a deposit without a mask -- then the kernel treats every column as empty and writes the top
// of all of them itself -- or whatever touches the columns first clears len / top (materialize).
// Plane slots hold unspecified data anyway and nothing reads them.
int zero_planes(cg_grid* g, bool all = true) {
    (void)all;
    g->pristine = true;
    g->split = true; g->nextp = 0; g->len_valid = false;
    g->len_bound = 0; g->has_gaps = false; g->has_cavity = false;
    return CG_OK;
}
int materialize(cg_grid* g) {
    if (!g->pristine) return CG_OK;
    ProfScope ps(g, CG_K_CLEAR);
    HIPCHK(g, hipMemsetAsync(g->len, 0, g->N, g->stream));
    HIPCHK(g, hipMemsetAsync(g->top, 0, 4 * g->N, g->stream));
    g->pristine = false; g->len_valid = true;
    return CG_OK;
}

// The kernels of the vertical and lateral steps work on the split form, the ones that
// restructure whole columns (strip, cavity, CMP, wet etch, ALD, upload) on whole stacks.
int refresh_len_bound(cg_grid* g, unsigned* topq_out);
int trim_planes(cg_grid* g);
// The vertical and lateral kernels that own 4 columns per thread work from the chains alone and
// leave len[] behind; whatever reads len[] of a grid in split form calls this first.
int ensure_len(cg_grid* g) {
    if (g->len_valid || !g->split) { g->len_valid = true; return CG_OK; }
    ProfScope ps(g, CG_K_OTHER);
    hipLaunchKernelGGL(k_relen, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g));
    HIPCHK(g, hipGetLastError());
    g->len_valid = true;
    return CG_OK;
}
int ensure_split(cg_grid* g) {
    if (g->split) return CG_OK;
    if (g->len_bound > std::min(g->lmax, SPLIT_PLANES + 1)) { int rc = refresh_len_bound(g, nullptr); if (rc) return rc; }
    if (g->len_bound > SPLIT_PLANES + 1) return CG_OK;        // too tall for the links: the steps run on the whole form (see run_deep_*)
    ProfScope ps(g, CG_K_OTHER);
    hipLaunchKernelGGL(k_promote, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g));
    HIPCHK(g, hipGetLastError());
    g->split = true;
    g->nextp = std::max(0, g->len_bound - 1);              // interval k lives in plane k: planes 0 .. max(len)-2 are taken
    return CG_OK;
}
int ensure_whole(cg_grid* g) {
    if (!g->split) return CG_OK;
    { int rc = ensure_len(g); if (rc) return rc; }
    ProfScope ps(g, CG_K_OTHER);
    hipLaunchKernelGGL(k_demote, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g));
    HIPCHK(g, hipGetLastError());
    g->split = false;
    return CG_OK;
}

// More planes (the device side of _expandBuffers, script_columngrid.js:129-168; the Lmax a caller
// sees is tracked apart, see sync_vis).  Plane-major storage makes the old contents a prefix of
// the new extent, and the buffers grow in place: nothing is copied.
int grow_planes(cg_grid* g, int add = 16) {
    const int newL = std::min(MAX_PLANES, g->lmax + add);
    if (newL == g->lmax) return CG_OK;
    int rc = el_resize(g, g->rec_el, 4 * g->N * (size_t)newL); if (rc) return rc;
    g->rec = (uint32_t*)g->rec_el.base;                     // (unchanged unless the device lacks virtual memory management)
    g->lmax = newL;
    return CG_OK;
}

// Give back the planes no column reaches (whole form only): scratch sets and snapshots are sized
// by the planes in use, and a long run of deposits may have mapped many more than that.
int trim_planes(cg_grid* g) {
    if (g->split) return CG_OK;
    int rc = refresh_len_bound(g, nullptr); if (rc) return rc;
    const int want = std::max(std::min(g->lmax_vis, MAX_PLANES), (g->len_bound + 15) / 16 * 16);
    if (want + 16 > g->lmax) return CG_OK;
    rc = el_resize(g, g->rec_el, 4 * g->N * (size_t)want); if (rc) return rc;
    g->lmax = want;
    return CG_OK;
}

// Is there device memory for `add` more planes, with room left for a copy of the planes in use
// (a snapshot, or the scratch set of etch_wet / deposit_ALD) and some slack?
bool memory_allows_planes(cg_grid* g, int add) {
    size_t fr = 0, to = 0;
    if (hipMemGetInfo(&fr, &to) != hipSuccess) { (void)hipGetLastError(); return false; }
    const size_t plane = 4 * g->N;
    return fr > plane * (size_t)add + plane * (size_t)std::max(g->len_bound, 16) + to / 16;
}

int read_flags(cg_grid* g, int out[4]) {
    HIPCHK(g, hipMemcpyAsync(out, g->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    return CG_OK;
}

int refresh_len_bound(cg_grid* g, unsigned* topq_out) {
    { int rc = ensure_len(g); if (rc) return rc; }
    HIPCHK(g, hipMemsetAsync(g->d_red, 0, 2 * sizeof(unsigned), g->stream));
    { ProfScope ps(g, CG_K_REDUCE);
      hipLaunchKernelGGL(k_max_top, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g), g->d_red); }
    unsigned r[2];
    HIPCHK(g, hipMemcpyAsync(r, g->d_red, sizeof r, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    g->len_bound = (int)r[1];
    if (topq_out) *topq_out = r[0];
    return CG_OK;
}

// The reference's Lmax: += 16 whenever an append finds its column full (script_columngrid.js:129-168,
// 418; script_columngrid2.js:658).  Gaps take intervals of their own on the device but no slot in
// the reference, so the planes (lmax) may outgrow it; only layers count here.
int sync_vis(cg_grid* g) {
    if (g->len_bound <= g->lmax_vis) return CG_OK;
    int m = 0;
    if (!g->has_gaps) { int rc = refresh_len_bound(g, nullptr); if (rc) return rc; m = g->len_bound; }
    else {
        int rc = ensure_whole(g); if (rc) return rc;
        HIPCHK(g, hipMemsetAsync(g->d_red, 0, 2 * sizeof(unsigned), g->stream));
        { ProfScope ps(g, CG_K_REDUCE);
          hipLaunchKernelGGL(k_max_layers, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g), g->d_red); }
        unsigned r[2];
        HIPCHK(g, hipMemcpyAsync(r, g->d_red, sizeof r, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
        m = (int)r[0]; g->len_bound = (int)r[1];
    }
    while (m > g->lmax_vis) g->lmax_vis += 16;
    return CG_OK;
}

const uint8_t* mask_ptr(cg_grid* g, int mask_id, bool* ok) {
    *ok = true;
    if (mask_id == CG_MASK_NONE) return nullptr;
    if (mask_id < 0 || (size_t)mask_id >= g->masks.size() || !g->masks[mask_id]) { *ok = false; return nullptr; }
    return g->masks[mask_id];
}

void stats_begin(cg_grid* g) {
    if (g->stats_on) (void)hipMemsetAsync(g->d_stats, 0, CG_NSTAT * sizeof(unsigned long long), g->stream);
}

void free_kernel(ConfKernel* k) {
    if (k->d_taps) (void)hipFree(k->d_taps);
    if (k->d_taps64) (void)hipFree(k->d_taps64);
    if (k->d_taps64_sym) (void)hipFree(k->d_taps64_sym);
    if (k->d_quad) (void)hipFree(k->d_quad);
    delete k;
}

// _getConformalKernelArray (script_columngrid.js:324-357) with its 32-entry FIFO cache whose key
// truncates t, S, dx, dy to 1e-3: two thicknesses closer than that share one kernel.
ConfKernel* get_kernel(cg_grid* g, double t, double S) {
    int key[4] = {js_to_int32(t * 1000), js_to_int32(S * 1000), js_to_int32(g->dx * 1000), js_to_int32(g->dy * 1000)};
    for (ConfKernel* k : g->kcache)
        if (!memcmp(k->key, key, sizeof key)) return k;
    ConfKernel* k = new ConfKernel();
    memcpy(k->key, key, sizeof key);
    const double dx = g->dx, dy = g->dy;
    const int Rx = (int)std::ceil(t / dx), Ry = (int)std::ceil(t / dy);
    const double t2 = t * t, Sinv = 1 / std::max(S, 1e-6), dx2 = dx * dx, dy2 = dy * dy;
    k->Rx = Rx; k->Ry = Ry; k->baseAdd = t;
    for (int du = -Rx; du <= Rx; du++) {
        const double du2dx2 = (double)du * du * dx2;
        for (int dv = -Ry; dv <= Ry; dv++) {
            const double r2 = du2dx2 + (double)dv * dv * dy2;
            const double r_lat = std::sqrt(r2) * Sinv;
            if (r_lat <= t + 1e-9) {
                const double add = std::sqrt(std::max(0.0, t2 - r_lat * r_lat));
                k->du.push_back(du); k->dv.push_back(dv); k->add.push_back((float)add);
            }
        }
    }
    g->kcache.push_back(k);
    if (g->kcache.size() > 32) { free_kernel(g->kcache.front()); g->kcache.pop_front(); }
    return k;
}

// device copies of the taps for the LDS pitch of radius R
int upload_taps(cg_grid* g, ConfKernel* k, int R) {
    if (k->dev_R == R) return CG_OK;
    const int pitch = TJ + 2 * R;
    std::vector<Tap> safe, fine;
    for (size_t q = 0; q < k->add.size(); q++) {
        Tap tp; tp.off = k->du[q] * pitch + k->dv[q]; tp.add = k->add[q];
        const double scaled = std::ldexp((double)k->add[q], 40);
        if (scaled == std::floor(scaled)) safe.push_back(tp); else fine.push_back(tp);
    }
    if (k->d_taps) { (void)hipFree(k->d_taps); k->d_taps = nullptr; }
    if (k->d_taps64) { (void)hipFree(k->d_taps64); k->d_taps64 = nullptr; }
    HIPCHK(g, hipMalloc(&k->d_taps, std::max<size_t>(1, safe.size()) * sizeof(Tap)));
    HIPCHK(g, hipMalloc(&k->d_taps64, std::max<size_t>(1, fine.size()) * sizeof(Tap)));
    if (!safe.empty()) HIPCHK(g, hipMemcpy(k->d_taps, safe.data(), safe.size() * sizeof(Tap), hipMemcpyHostToDevice));
    if (!fine.empty()) HIPCHK(g, hipMemcpy(k->d_taps64, fine.data(), fine.size() * sizeof(Tap), hipMemcpyHostToDevice));
    k->ntaps = (int)safe.size(); k->ntaps64 = (int)fine.size(); k->dev_R = R;

    // symmetric quadrant form (k_stencil_sym): usable when the effective radius is 1..8 and the
    // four mirror images of every tap exist with the very same increment
    int Re = 0;
    for (size_t q = 0; q < k->add.size(); q++) Re = std::max(Re, std::max(std::abs(k->du[q]), std::abs(k->dv[q])));
    k->Re = Re; k->sym_ok = false;
    if (Re >= 1 && Re <= 8 && g->dx == g->dy) {          // (the quadrant table is read transposed as well: square cells only)
        const int n = Re + 1;
        std::vector<float> quad(n * n, -INFINITY);
        std::vector<int> seen(n * n, 0);
        bool ok = true;
        std::vector<Tap> fine_sym;
        const int pitch_sym = sym_tile_cols(Re) + 2 * ((Re + 3) & ~3);
        for (size_t q = 0; q < k->add.size() && ok; q++) {
            const int a = std::abs(k->du[q]), b = std::abs(k->dv[q]);
            const double scaled = std::ldexp((double)k->add[q], 40);
            const bool is_safe = scaled == std::floor(scaled);
            if (!seen[a * n + b]) {
                seen[a * n + b] = 1;
                if (is_safe) quad[a * n + b] = k->add[q];
            }
            uint32_t b0, b1; float ref = k->add[q];
            // all mirror images must carry the same bits
            for (size_t r = 0; r < k->add.size(); r++)
                if (std::abs(k->du[r]) == a && std::abs(k->dv[r]) == b) { memcpy(&b0, &k->add[r], 4); memcpy(&b1, &ref, 4); if (b0 != b1) ok = false; }
            seen[a * n + b] += 1;
            if (!is_safe) { Tap tp; tp.off = k->du[q] * pitch_sym + k->dv[q]; tp.add = k->add[q]; fine_sym.push_back(tp); }
        }
        // every quadrant entry must have been seen 1 (centre), 2 (axes) or 4 times
        for (int a = 0; a <= Re && ok; a++) for (int b = 0; b <= Re; b++) {
            const int cnt = seen[a * n + b] ? seen[a * n + b] - 1 : 0;
            const int want = (a == 0 && b == 0) ? 1 : ((a == 0 || b == 0) ? 2 : 4);
            if (cnt != 0 && cnt != want) ok = false;
        }
        if (ok) {
            k->quad = quad; k->sym_ok = true;
            if (k->d_taps64_sym) { (void)hipFree(k->d_taps64_sym); k->d_taps64_sym = nullptr; }
            HIPCHK(g, hipMalloc(&k->d_taps64_sym, std::max<size_t>(1, fine_sym.size()) * sizeof(Tap)));
            if (!fine_sym.empty()) HIPCHK(g, hipMemcpy(k->d_taps64_sym, fine_sym.data(), fine_sym.size() * sizeof(Tap), hipMemcpyHostToDevice));
            k->ntaps64_sym = (int)fine_sym.size();
            if (!k->d_quad) HIPCHK(g, hipMalloc(&k->d_quad, quad.size() * sizeof(float)));
            HIPCHK(g, hipMemcpy(k->d_quad, quad.data(), quad.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    return CG_OK;
}

int ensure_stencil_scratch(cg_grid* g, int R, bool need_halo) {
    if (!need_halo) return CG_OK;
    size_t hneed = (size_t)R * g->ny * 3;
    if (hneed > g->halo_cap && g->halo_external)
        return fail(g, CG_ERR_ARG, "caller's halo buffers hold %zu bytes each, this step needs %zu", g->halo_cap, hneed);
    if (hneed > g->halo_cap) {
        for (auto& h : g->halo) { if (h) (void)hipFree(h); h = nullptr; }
        g->halo_cap = 0;
        for (auto& h : g->halo) HIPCHK(g, hipMalloc(&h, hneed));
        g->halo_cap = hneed;
    }
    return CG_OK;
}

// The plane a depositing launch pushes old tops into: every such launch gets a fresh one, so
// its writes are dense whatever the column lengths.  When the planes run out, more are mapped
// behind them as long as memory and the handle's budget allow (nothing moves); otherwise the
// columns are compacted (whole form and back: interval k in plane k), which frees every plane
// above max(len)-2 at the price of moving every interval.
int take_append_plane(cg_grid* g, int* P) {
    // planes come in chunks of 16, a link names planes 0 .. SPLIT_PLANES-1: the cap is 64 planes of which 63 are usable
    const int cap = std::min(std::min(g->plane_budget, MAX_PLANES), (SPLIT_PLANES + 15) / 16 * 16);
    const auto usable = [&]() { return std::min(g->lmax, SPLIT_PLANES); };
    if (g->nextp >= usable()) {
        if (g->lmax + 16 <= cap && memory_allows_planes(g, 16)) {
            int rc = grow_planes(g); if (rc) return rc;                          // in place, nothing moves
        } else {
            int rc = ensure_whole(g); if (rc) return rc;
            rc = refresh_len_bound(g, nullptr); if (rc) return rc;
            rc = ensure_split(g); if (rc) return rc;
            if (!g->split) { *P = -1; return CG_OK; }                             // a column is too tall for the links: the caller runs on the whole form
            if (g->nextp >= usable() && g->lmax < SPLIT_PLANES) { rc = grow_planes(g); if (rc) return rc; }      // the columns need them, budget or not
            if (g->nextp >= usable()) {                                          // max(len) - 1 == SPLIT_PLANES: as tall as the links reach, nothing to append into
                rc = ensure_whole(g); if (rc) return rc;
                *P = -1; return CG_OK;
            }
        }
    }
    *P = g->nextp++;
    return CG_OK;
}

// vertical deposit: the 4-column kernel while every column has a plane to spare, otherwise the
// one-column kernel with the growth protocol of the reference (an append that finds no room)
int run_depo_vertical(cg_grid* g, const uint8_t* mask, int mat, double t) {
    int P = -1;
    if (g->split) { int rcp = take_append_plane(g, &P); if (rcp) return rcp; }      // (may find a column too tall and leave the grid whole)
    bool may_overflow = g->len_bound >= g->lmax;
    if (may_overflow) { int rc = refresh_len_bound(g, nullptr); if (rc) return rc; may_overflow = g->len_bound >= g->lmax; }
    const dim3 grid(blocks_for(g->N)), blk(256);
    GridView gv = view(g); gv.P = P;
    if (g->split && g->N % 4 == 0 && !may_overflow) {        // 4 columns per thread, no look at the lengths
        ProfScope ps(g, CG_K_DEPO_VERTICAL);
        launch_vertical<false>(g, mask, mat, t, P, g->pristine);
        HIPCHK(g, hipGetLastError());
        g->len_valid = false;
    } else {
        // a column may be full: the one-column kernel reads the lengths, reports an append that
        // found no room (script_columngrid.js:418) and is re-run for those columns after the growth
        int rc = ensure_len(g); if (rc) return rc;
        HIPCHK(g, hipMemsetAsync(g->d_flags, 0, sizeof(int), g->stream));
        ProfScope ps(g, CG_K_DEPO_VERTICAL);
        if (g->stats_on) hipLaunchKernelGGL((k_vertical_1<false, true>), grid, blk, 0, g->stream, gv, mask, mat, t, -1, g->d_flags, g->d_stats);
        else hipLaunchKernelGGL((k_vertical_1<false, false>), grid, blk, 0, g->stream, gv, mask, mat, t, -1, g->d_flags, g->d_stats);
        HIPCHK(g, hipGetLastError());
        if (may_overflow) {
            int fl[4]; rc = read_flags(g, fl); if (rc) return rc;
            if (fl[0]) {
                const int oldL = g->lmax;
                rc = grow_planes(g); if (rc) return rc;
                gv = view(g); gv.P = P;
                if (g->stats_on) hipLaunchKernelGGL((k_vertical_1<false, true>), grid, blk, 0, g->stream, gv, mask, mat, t, oldL, g->d_flags, g->d_stats);
                else hipLaunchKernelGGL((k_vertical_1<false, false>), grid, blk, 0, g->stream, gv, mask, mat, t, oldL, g->d_flags, g->d_stats);
                HIPCHK(g, hipGetLastError());
            }
        }
    }
    if (g->len_bound < 255) g->len_bound += 1;
    return sync_vis(g);
}

// New tops go to top_alt (the caller swaps the buffers when the step is complete).
// part: PART_ALL every tile; PART_INTERIOR the tile rows whose stencil stays inside the slab
// (they need no halo message); PART_BOUNDARY the others.
enum { PART_ALL = 0, PART_INTERIOR = 1, PART_BOUNDARY = 2 };
struct Band { int t0, n; };
static int stencil_bands(int nx, int th, int R, bool has_lower, bool has_upper, int part, Band out[2]) {
    const int ntr = (nx + th - 1) / th;
    int lo = has_lower ? std::min(ntr, (R + th - 1) / th) : 0;              // tile rows reaching below row 0
    int up = has_upper ? std::max(lo, std::min(ntr, std::max(0, (nx - R) / th))) : ntr;   // first tile row reaching past row nx-1
    if (part == PART_ALL) { out[0] = {0, ntr}; return 1; }
    if (part == PART_INTERIOR) { out[0] = {lo, up - lo}; return up > lo ? 1 : 0; }
    int n = 0;
    if (lo > 0) out[n++] = {0, lo};
    if (ntr > up) out[n++] = {up, ntr - up};
    return n;
}

// tiles per super-tile of the lateral kernels (tuning knobs: env CG_SUPI, CG_SUPJ)
static int sup_tiles(bool along_j) {
    static int vi = [] { const char* e = getenv("CG_SUPI"); int x = e ? atoi(e) : 2; return (x >= 1 && x <= 16) ? x : 2; }();
    static int vj = [] { const char* e = getenv("CG_SUPJ"); int x = e ? atoi(e) : 4; return (x >= 1 && x <= 16) ? x : 4; }();
    return along_j ? vj : vi;
}
// blocks of a launch over tile_in x tiles_j tiles: whole super-tiles, a multiple of 8 of them
static unsigned sup_grid(GridView& gv, int t0, int tile_in, int tiles_j) {
    gv.tile_i0 = t0; gv.tile_in = tile_in; gv.sup_i = sup_tiles(false); gv.sup_j = sup_tiles(true);
    const long n_s = (long)((tile_in + gv.sup_i - 1) / gv.sup_i) * ((tiles_j + gv.sup_j - 1) / gv.sup_j);
    return (unsigned)(((n_s + 7) / 8) * 8 * gv.sup_i * gv.sup_j);
}

// The kernel families a lateral step can take: sym = the forms that leave the lengths behind, lean = the lean
// row-streaming form (cg_stencil_lean.hip.h) for the tiles that read no message, streams = a row-streaming form at all.
struct StencilForm { bool sym, lean, streams; };
static StencilForm stencil_form(const cg_grid* g, const ConfKernel* K, bool etch, int mat, int only_len, bool keep_len) {
    StencilForm f;
    // keep_len: a column may be full, so the lengths matter -> the general kernel, which reads and
    // maintains them; otherwise the symmetric kernel where it applies, which leaves them behind
    f.sym = g->split && only_len < 0 && !keep_len && K->sym_ok && g->ny % 4 == 0 && !getenv("CG_NO_SYM");
    // The lean form when everything it takes for granted holds: the float licences, a centre increment well above the
    // spacing of float32 heights (an open column then always grows), a material that an empty column's record cannot be
    // mistaken for, 32-bit column indices.
    static const bool no_lean = getenv("CG_NO_LEAN") != nullptr;
    const float a00 = K->quad.empty() ? 0.f : K->quad[0];
    f.lean = f.sym && !no_lean && g->sopt.deq_f32 && g->sopt.grow_f32 && !getenv("CG_STREAM_NOF32") && mat != 0 && a00 >= 0.01f &&
             std::isfinite(a00) && g->N < ((size_t)1 << 31);
    // Radii the streaming forms take: the lean deposit beats the gathering kernel at every radius (16384^2: R = 1 0.64 vs
    // 0.67 ms, R = 2 0.66 vs 0.70); etches and the general streaming form from CG_STREAM_MINR (3) on.
    const int min_r = (f.lean && !etch) ? std::min(1, stream_min_radius()) : stream_min_radius();
    f.streams = f.sym && K->ntaps64_sym == 0 && K->Re >= min_r && !g->stats_on;
    return f;
}

int launch_stencil(cg_grid* g, bool etch, const uint8_t* mask, int mat, double t_eff, int R, ConfKernel* K, int only_len,
                   const uint8_t* halo_lo, const uint8_t* halo_up, int P, int part = PART_ALL, bool keep_len = false, const FusedOps* fold = nullptr,
                   const FreshTab* fresh = nullptr) {
    const StencilForm form = stencil_form(g, K, etch, mat, only_len, keep_len);
    const bool sym = form.sym, lean = form.lean;
    if (!sym) { int rcl = ensure_len(g); if (rcl) return rcl; }
    GridView gv = view(g);
    gv.top_w = g->split ? g->top_alt : g->top; gv.P = P;      // (whole form: the planes are rewritten in place, top is only the copy of the old tops)
    const size_t lds = stencil_lds_bytes(R);
    const dim3 blk(256);
    Band bands[2];
    ProfScope ps(g, etch ? CG_K_STENCIL_ETCH : ((fold || fresh) ? CG_K_STENCIL_DEPOSIT_STEPS : CG_K_STENCIL_DEPOSIT));
    if (form.streams) {
        // row-streaming form: one wave per tile of ti rows x STRIP columns
        g->len_valid = false;
        const int ti = stream_tile_rows(K->Re);
        gv.ti = ti;
        // tile rows that reach a neighbour's rows read its message (MSG form), the others never look at one
        Band bi[2], bb[2];
        const int n_int = stencil_bands(g->nx, ti, R, halo_lo != nullptr, halo_up != nullptr, PART_INTERIOR, bi);
        const int n_bnd = stencil_bands(g->nx, ti, R, halo_lo != nullptr, halo_up != nullptr, PART_BOUNDARY, bb);
        const int tiles_j = (g->ny + STRIP - 1) / STRIP;
        StreamOpts sopt = g->sopt;
        sopt.skip_init = ((double)(float)t_eff == t_eff && !K->quad.empty() && K->quad[0] == (float)t_eff) ? 1 : 0;
        if (getenv("CG_STREAM_NOF32")) { sopt.deq_f32 = 0; sopt.skip_init = 0; sopt.grow_f32 = 0; }
        sopt.a00 = K->quad.empty() ? 0.f : K->quad[0];
        if (!g->d_sink) HIPCHK(g, hipMalloc(&g->d_sink, (size_t)4 << 20));
        sopt.sink = g->d_sink;
        sopt.fast_rows = (etch && !K->quad.empty() && K->quad[0] != -INFINITY && !getenv("CG_STREAM_NOFAST")) ? 1 : 0;
        StreamArgs sa{gv, mask, mat, t_eff, R, halo_lo, halo_up, K->d_quad, sopt, g->d_flags, g->d_stats, lean, fold, fresh};
        // The rows that read a message are few (a tile row or two per slab edge) and a wave walks its tile row by row:
        // in tiles as tall as the others' that launch would last as long as one whole tile takes, on a nearly empty
        // device.  It is cut into tiles of edge_ti rows instead (env CG_EDGE_TI; more waves, each with fewer rows).
        static const int edge_ti_env = [] { const char* e = getenv("CG_EDGE_TI"); const int v = e ? atoi(e) : 8; return (v >= 4 && v <= 256) ? v : 8; }();
        const int edge_ti = (ti % edge_ti_env == 0) ? edge_ti_env : ti;
        const auto go = [&](const Band& band, bool msg) {
            int t0 = band.t0, n = band.n;
            sa.gv.ti = ti;
            if (msg && edge_ti != ti) {
                const int r0 = band.t0 * ti, r1 = std::min(g->nx, (band.t0 + band.n) * ti);
                t0 = r0 / edge_ti; n = (r1 - r0 + edge_ti - 1) / edge_ti; sa.gv.ti = edge_ti;
            }
            sa.gv.tile_i0 = t0; sa.gv.tile_in = n; sa.gv.sup_i = 1; sa.gv.sup_j = 1;
            launch_stream(K->Re, etch, msg, mask != nullptr, (unsigned)(((long)n * tiles_j + 7) / 8 * 8), g->stream, sa);
        };
        if (part != PART_BOUNDARY) for (int b_ = 0; b_ < n_int; b_++) go(bi[b_], false);
        if (part != PART_INTERIOR) for (int b_ = 0; b_ < n_bnd; b_++) go(bb[b_], true);
        HIPCHK(g, hipGetLastError());
        return CG_OK;
    }
    if (sym) {
        g->len_valid = false;
        const int tjs = sym_tile_cols(K->Re);
        const int tis = 1024 / tjs * sym_passes(K->Re);
        const int nb = stencil_bands(g->nx, tis, R, halo_lo != nullptr, halo_up != nullptr, part, bands);
        SymArgs ya{gv, mask, mat, t_eff, R, halo_lo, halo_up, K->quad.data(), K->d_taps64_sym, K->ntaps64_sym, g->d_flags, g->d_stats};
        for (int b_ = 0; b_ < nb; b_++) {
            const unsigned blocks = sup_grid(ya.gv, bands[b_].t0, bands[b_].n, (g->ny + tjs - 1) / tjs);
            launch_sym(K->Re, tjs, etch, g->stats_on, blocks, g->stream, ya);
        }
        HIPCHK(g, hipGetLastError());
        return CG_OK;
    }
#define LAUNCH(E, S)                                                                                           \
    do {                                                                                                       \
        if (lds > 48 * 1024)                                                                                   \
            HIPCHK(g, hipFuncSetAttribute((const void*)k_stencil_apply<E, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        const int nb = stencil_bands(g->nx, TI, R, halo_lo != nullptr, halo_up != nullptr, part, bands);       \
        for (int b_ = 0; b_ < nb; b_++) {                                                                      \
            const dim3 grid(sup_grid(gv, bands[b_].t0, bands[b_].n, (g->ny + TJ - 1) / TJ));                   \
            hipLaunchKernelGGL((k_stencil_apply<E, S>), grid, blk, lds, g->stream, gv, mask, mat, t_eff, R, halo_lo, halo_up, \
                               K->d_taps, K->ntaps, K->d_taps64, K->ntaps64, only_len, g->d_flags, g->d_stats); \
        }                                                                                                      \
    } while (0)
    if (etch) { if (g->stats_on) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (g->stats_on) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    HIPCHK(g, hipGetLastError());
    return CG_OK;
}


// ---- second / third set of planes for the out-of-place models ---------------------------------
void free_alt(cg_grid* g, int which) {
    if (g->alt_len[which]) (void)hipFree(g->alt_len[which]);
    el_free(g->alt_el[which]);
    g->alt_len[which] = nullptr; g->alt_rec[which] = nullptr; g->alt_lmax[which] = 0;
}

// a spare set of planes as tall as the grid's (its contents are whatever the last user left)
int ensure_alt(cg_grid* g, int which) {
    if (g->alt_rec[which] && g->alt_lmax[which] == g->lmax) return CG_OK;
    if (!g->alt_len[which]) {
        HIPCHK(g, hipMalloc(&g->alt_len[which], std::max<size_t>(g->N, 4)));
        HIPCHK(g, hipMemsetAsync(g->alt_len[which], 0, g->N, g->stream));
    }
    int rc = el_reserve(g, g->alt_el[which], 4 * g->N * (size_t)MAX_PLANES, 4 * g->N * 16); if (rc) return rc;
    rc = el_resize(g, g->alt_el[which], 4 * g->N * (size_t)g->lmax); if (rc) return rc;
    g->alt_rec[which] = (uint32_t*)g->alt_el[which].base;
    g->alt_lmax[which] = g->lmax;
    return CG_OK;
}

GridView alt_view(const cg_grid* g, int which) {
    GridView v = view(g);
    v.len = g->alt_len[which]; v.rec = g->alt_rec[which]; v.lmax = g->alt_lmax[which];
    return v;
}

void swap_alt(cg_grid* g, int which) {
    std::swap(g->len, g->alt_len[which]); std::swap(g->rec, g->alt_rec[which]); std::swap(g->lmax, g->alt_lmax[which]);
    std::swap(g->rec_el, g->alt_el[which]);
    g->split = false;                            // the out-of-place kernels write whole stacks
}

// Run `launch(dst)` with dst = spare planes `which`, grow and repeat while a column does not
// fit, then make dst the grid (the old planes become the spare set).
template <class Launch>
int run_out_of_place(cg_grid* g, int which, const char* what, Launch launch) {
    { int rc = trim_planes(g); if (rc) return rc; }        // the spare set is as tall as the grid: no taller than needed
    for (int attempt = 0; attempt < 17; attempt++) {
        int rc = ensure_alt(g, which); if (rc) return rc;
        HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
        launch(alt_view(g, which));
        HIPCHK(g, hipGetLastError());
        int fl[4]; rc = read_flags(g, fl); if (rc) return rc;
        if (fl[3] > 255 || fl[0])
            return fail(g, CG_ERR_UNSUPPORTED, "%s: a column would need more than 255 intervals or layers that overlap", what);
        if (fl[3] <= g->lmax) {
            swap_alt(g, which);
            g->len_bound = fl[3]; g->has_gaps = fl[2] != 0; g->has_cavity = fl[1] != 0;
            return sync_vis(g);
        }
        rc = grow_planes(g, ((fl[3] - g->lmax + 15) / 16) * 16); if (rc) return rc;      // the spare set follows in ensure_alt
    }
    return fail(g, CG_ERR_STATE, "%s: plane growth did not converge", what);
}

// ---- deposit_ALD helpers -----------------------------------------------------------------------
AldLayout make_bit_layout(const cg_grid* g, double HmaxPlus) {                  // script_columngrid2.js:161-167
    const double hc = std::ceil(HmaxPlus / g->zscale);
    AldLayout L;
    L.Zmax = hc > 1 ? (hc < 2.0e9 ? (int)hc : 2000000000) : 1;
    L.wpc = (int)(((unsigned)L.Zmax + 31u) >> 5);
    const int lastBits = L.Zmax & 31;
    L.lastMask = lastBits == 0 ? 0xFFFFFFFFu : ((1u << lastBits) - 1u);
    return L;
}

// threads per block such that `nsets` bitsets of L.wpc words per thread fit 64 KiB of LDS
int ald_block(const AldLayout& L, int nsets, size_t* lds_bytes) {
    const size_t per_thread = (size_t)nsets * L.wpc * 4;
    int bt = 256;
    while (bt > 1 && bt * per_thread > 65536) bt >>= 1;
    if (bt * per_thread > 65536) return 0;
    *lds_bytes = bt * per_thread;
    return bt;
}

unsigned ald_blocks(size_t n, int bt) { return (unsigned)std::min<size_t>(std::max<size_t>((n + bt - 1) / bt, 1), (size_t)1 << 20); }

void drop_seeds(cg_grid* g) {
    if (g->seed_n) (void)hipFree(g->seed_n);
    if (g->seed) (void)hipFree(g->seed);
    if (g->ald_runs) (void)hipFree(g->ald_runs);
    if (g->ald_runs_n) (void)hipFree(g->ald_runs_n);
    g->ald_runs = nullptr; g->ald_runs_n = nullptr; g->ald_runs_cap = 0;
    g->seed_n = nullptr; g->seed = nullptr; g->seed_planes = 0; g->has_seeds = false; g->seed_total = 0;
    g->seed_bbox[0] = 0; g->seed_bbox[1] = -1; g->seed_bbox[2] = 0; g->seed_bbox[3] = -1;
    g->seed_stride = g->seed_off = 0; g->seed_rows = 0; g->seed_lo = g->seed_hi = false;
}

ArrLayout seed_layout(const cg_grid* g) {
    return ArrLayout{g->seed_stride, g->seed_off, g->seed_lo ? g->seed_rows : 0, g->seed_hi ? g->seed_rows : 0};
}

// buildKernelRows (script_columngrid2.js:466-496) flattened to (du, dv, add)
int build_ald_taps(const cg_grid* g, double t, std::vector<AldTap>* taps) {
    const int Rx = js_to_int32(std::ceil(t / g->dx));
    const double t2 = t * t;
    taps->clear();
    for (int du = -Rx; du <= Rx; du++) {
        const double x = du * g->dx, x2 = x * x, rem = t2 - x2;
        if (rem < 0) continue;
        const int maxDv = js_to_int32(std::floor(std::sqrt(rem) / g->dy));
        for (int dv = -maxDv; dv <= maxDv; dv++) {
            const double y = dv * g->dy;
            double add = rem - (y * y);
            if (add < 0) add = 0;
            taps->push_back(AldTap{du, dv, std::sqrt(add), 0, 0, 0, 0});
        }
    }
    return Rx;
}

// coat(t) from the held seeds, inserted as `mat`: the grid becomes grid + coat, spare set 0 keeps the grid as it was
// hmax_q: the height of the whole domain when the handle is a slab of it, null for the handle's own
int ald_coat(cg_grid* g, int mat, double t, const uint32_t* hmax_q = nullptr) {
    if (!g->has_seeds) return fail(g, CG_ERR_STATE, "deposit_ALD: no seeds held (cg_ald_collect_seeds first)");
    unsigned qtop = 0;
    int rc = ensure_whole(g); if (rc) return rc;
    rc = refresh_len_bound(g, &qtop); if (rc) return rc;
    if (hmax_q) qtop = *hmax_q;
    const double Hnow = qtop > 0 ? (double)qtop * g->zscale : 1.0;                               // maxHeight()
    const double Hcap = Hnow + t + g->zscale;                                                    // cg2:501, 702
    const AldLayout L = make_bit_layout(g, Hcap);
    size_t lds = 0;
    const int bt = ald_block(L, 1, &lds);
    if (!bt) return fail(g, CG_ERR_UNSUPPORTED, "deposit_ALD: %d height bins per column do not fit the bitsets", L.Zmax);
    std::vector<AldTap> taps;
    const int Rx = build_ald_taps(g, t, &taps);
    // the taps whose quantised range ends are the seed's ends moved by whole steps, whatever the seed (see AldTap)
    const double inv_zs = 1.0 / g->zscale;
    const int HcapQ = (int)std::min(2.0e9, js_round(Hcap * inv_zs));
    static const bool no_int = getenv("CG_ALD_NOINT") != nullptr;
    for (AldTap& tp : taps) {
        const double a = tp.add * inv_zs, f = a - std::floor(a);
        tp.exact = (!no_int && tp.add > 1e-6 && a < 1.0e6 && std::fabs(f - 0.5) > 1e-6 && g->zscale > 1e-6 && Hcap > 1e-6) ? 1 : 0;
        tp.lo_q = (int)std::ceil(std::min(a, 1.0e6) - 0.5); tp.hi_q = (int)std::floor(std::min(a, 1.0e6) + 0.5);
    }
    // taps at the same distance share their `add`: one class each (k_ald_grow); the classes' taps go behind the kernel's own
    const int ntaps = (int)taps.size();
    std::vector<AldTap> classes;
    for (AldTap& tp : taps) {
        int c = -1;
        for (size_t k = 0; k < classes.size(); k++) if (classes[k].add == tp.add) c = (int)k;
        if (c < 0) { c = (int)classes.size(); classes.push_back(tp); }
        tp.cls = c;
    }
    const int ncls = (int)classes.size();
    taps.insert(taps.end(), classes.begin(), classes.end());
    if ((int)taps.size() > g->ald_taps_cap) {
        HIPCHK(g, hipStreamSynchronize(g->stream));
        if (g->d_ald_taps) (void)hipFree(g->d_ald_taps);
        g->d_ald_taps = nullptr; g->ald_taps_cap = 0;
        HIPCHK(g, hipMalloc(&g->d_ald_taps, taps.size() * sizeof(AldTap)));
        g->ald_taps_cap = (int)taps.size();
    }
    if (!taps.empty()) {
        HIPCHK(g, hipMemcpyAsync(g->d_ald_taps, taps.data(), taps.size() * sizeof(AldTap), hipMemcpyHostToDevice, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));            // taps is a local
    }
    int imin = 1, imax = 0, jmin = 1, jmax = 0;                // no seed: nothing is filled (cg2:517)
    if (g->seed_bbox[1] >= g->seed_bbox[0]) {                  // cg2:521-524 (both directions grow by Rx)
        imin = std::max(0, g->seed_bbox[0] - Rx); imax = std::min(g->nx - 1, g->seed_bbox[1] + Rx);
        jmin = std::max(0, g->seed_bbox[2] - Rx); jmax = std::min(g->ny - 1, g->seed_bbox[3] + Rx);
    }
    ProfScope ps(g, CG_K_OTHER);
    // The seeds grown once per class and united (k_ald_grow), then the coat from those runs: unless a height does not fit the
    // 16 bits a run's end is kept in, the runs find no memory, or CG_ALD_SEEDS=1 asks for the kernel that grows every seed
    // under every tap.
    static const bool from_seeds = getenv("CG_ALD_SEEDS") != nullptr;
    bool use_runs = !from_seeds && HcapQ <= 65535 && ncls >= 1 && (size_t)ncls * ALD_RUNS * g->seed_stride < ((size_t)1 << 32);      // (32-bit entry numbers in the kernel)
    if (use_runs) {
        const size_t need = (size_t)ncls * g->seed_stride;
        if (g->ald_runs_cap < need) {
            HIPCHK(g, hipStreamSynchronize(g->stream));
            if (g->ald_runs) (void)hipFree(g->ald_runs);
            if (g->ald_runs_n) (void)hipFree(g->ald_runs_n);
            g->ald_runs = nullptr; g->ald_runs_n = nullptr; g->ald_runs_cap = 0;
            if (hipMalloc(&g->ald_runs, need * ALD_RUNS * sizeof(uint32_t)) != hipSuccess || hipMalloc(&g->ald_runs_n, need) != hipSuccess) {
                (void)hipGetLastError();
                if (g->ald_runs) (void)hipFree(g->ald_runs);
                g->ald_runs = nullptr; g->ald_runs_n = nullptr; use_runs = false;
            } else g->ald_runs_cap = need;
        }
    }
    if (use_runs) {
        hipLaunchKernelGGL(k_ald_grow, dim3(ald_blocks(g->seed_stride, 256)), dim3(256), 0, g->stream, g->seed_n, g->seed, g->seed_planes, g->seed_stride,
                           g->d_ald_taps + ntaps, ncls, g->zscale, 1.0 / g->zscale, Hcap, HcapQ, g->ald_runs_n, g->ald_runs);
        HIPCHK(g, hipGetLastError());
    }
    return run_out_of_place(g, 0, "deposit_ALD", [&](GridView dst) {
        if (use_runs)
            hipLaunchKernelGGL(k_ald_coat_runs, dim3(ald_blocks(g->N, bt)), dim3(bt), lds, g->stream, view(g), dst, L, Hcap, g->d_ald_taps, ntaps,
                               g->seed_n, g->seed, g->seed_planes, seed_layout(g), g->ald_runs_n, g->ald_runs, imin, imax, jmin, jmax, mat, HcapQ, g->d_flags);
        else
            hipLaunchKernelGGL(k_ald_coat, dim3(ald_blocks(g->N, bt)), dim3(bt), lds, g->stream, view(g), dst, L, Hcap, g->d_ald_taps, ntaps,
                               g->seed_n, g->seed, g->seed_planes, seed_layout(g), imin, imax, jmin, jmax, mat, HcapQ, g->d_flags);
    });
}

}  // namespace

// The integer licences of a vertical step of thickness t (VertOpts), found by trying: every height step for the
// quantiser identities, the neighbouring layer thicknesses for the two thresholds.  Kept per handle for the last few
// thicknesses (a flow repeats them).
static uint16_t quantz_host(double z, double zs) {
    const double v = js_round(z / zs);
    if (!(std::fabs(v) < 9.0e18)) return 0;
    return (uint16_t)(long long)v;
}

static VertOpts vert_opts(cg_grid* g, double t, bool etch) {
    static const bool off = getenv("CG_VERT_NOINT") != nullptr;
    VertOpts none{-1, -1, -1, -1};
    if (off || !(t > 0) || !(g->zscale > 1e-6) || !(t / g->zscale < 60000.0)) return none;
    for (const auto& e : g->vopt_cache) if (e.t == t && e.etch == etch) return e.vo;
    // thick = q1*zs - q0*zs carries the rounding of two products of up to 65535*zs each: the thresholds must stay
    // clear of the reference's 1e-9 by more than that, whatever the step size
    const double zs = g->zscale, margin = std::max(1e-10, 4.0 * 65535.0 * zs * 2.220446049250313e-16);
    if (margin > 2.5e-10) return none;                         // (steps above ~4 nm: the double form decides)
    VertOpts vo = none;
    if (!etch) {
        const int K = (int)quantz_host(0.0 * zs + t, zs);
        bool ok = true;
        for (int q = 0; ok && q + K < 65536; q++) ok = (int)quantz_host((double)q * zs + t, zs) == q + K;
        if (ok) vo.depo_k = K;
    } else {
        // smallest d with t < d*zs - 1e-9, and largest d with t - d*zs > 1e-9, each clear of its boundary by the margin
        const int d0 = (int)std::floor(t / zs);
        int kc = -1, kr = -1;
        bool ok = true;
        for (int d = std::max(0, d0 - 2); d <= d0 + 3 && ok; d++) {
            const double x = (double)d * zs;
            const double a = x - CG_EPS - t;                   // > 0: the cut stays inside a layer d steps thick
            const double c = t - x - CG_EPS;                   // > 0: something is left to take after such a layer
            if (std::fabs(a) < margin || std::fabs(c) < margin) ok = false;
            if (a > 0 && kc < 0) kc = d;
            if (c > 0) kr = d;
        }
        if (d0 - 2 > 0 && kr < 0) ok = false;                  // (the scan must have seen the lower threshold)
        if (ok && kc >= 1 && kr < kc) {
            const int ke = kc - (int)quantz_host((double)kc * zs - t, zs);
            for (int q = kc; ok && q < 65536; q++) ok = (int)quantz_host((double)q * zs - t, zs) == q - ke;
            if (ok) { vo.etch_kc = kc; vo.etch_kr = kr; vo.etch_ke = ke; }
        }
    }
    if (g->vopt_cache.size() >= 16) g->vopt_cache.erase(g->vopt_cache.begin());
    g->vopt_cache.push_back({t, etch, vo});
    return vo;
}

template <bool ETCH>
static void launch_vertical(cg_grid* g, const uint8_t* mask, int mat, double t, int P, bool fresh_grid) {
    GridView gv = view(g); gv.P = P;
    const VertOpts vo = g->stats_on ? VertOpts{-1, -1, -1, -1} : vert_opts(g, t, ETCH);
    const int fresh = (!ETCH && fresh_grid) ? 1 : 0;
    // groups per thread: 1 for the read-modify-write mixes (measured: appends 0.55 ms, 0.61 with 2 or 4 at 16384^2), 4 for the
    // launch that only writes (the first deposit on a reset grid: 0.36 -> 0.21 ms)
    const int U = getenv("CG_VU") ? vu_setting() : (fresh ? 4 : 1);
    // one tile per workgroup: measured 25 % faster than a capped, sweeping grid for this access mix
    static const long cap = [] { const char* e = getenv("CG_VGRID"); return e ? atol(e) : 0L; }();
    size_t nb = (g->N / 4 + (size_t)256 * U - 1) / ((size_t)256 * U);
    if (cap > 0) nb = std::min<size_t>(nb, (size_t)cap);
    const dim3 grid((unsigned)std::max<size_t>(1, std::min<size_t>(nb, 0x7fffffff))), blk(256);
#define LV(S, UU, I) hipLaunchKernelGGL((k_vertical_g4<ETCH, S, UU, I>), grid, blk, 0, g->stream, gv, mask, mat, t, fresh, vo, g->d_flags, g->d_stats)
    const bool lic = ETCH ? vo.etch_kc >= 0 : vo.depo_k >= 0;
    if (g->stats_on) { if (U == 1) LV(true, 1, false); else if (U == 2) LV(true, 2, false); else LV(true, 4, false); }
    else if (lic) { if (U == 1) LV(false, 1, true); else if (U == 2) LV(false, 2, true); else LV(false, 4, true); }
    else { if (U == 1) LV(false, 1, false); else if (U == 2) LV(false, 2, false); else LV(false, 4, false); }
#undef LV
    if (fresh) g->pristine = false;          // the tops are defined now; len is behind as after any such launch
}

// ---- runs of vertical steps ---------------------------------------------------------------------
// A vertical step changes a column from that column alone, so consecutive vertical steps are queued and go out as ONE
// pass over the columns (k_vertical_fused) when something else wants the grid: any other entry point, a lateral
// step, or a full queue.  What the host tracks per step (the plane a deposit pushes into, the bound on the column
// lengths, the pristine state) is settled when the step is queued; a deposit that could meet a full column, outgrow
// the Lmax the caller sees or find no plane to spare is not queued: it flushes and takes the one-step protocol.
namespace {
// One column of the reference on the host, in the device's record format: _depoAtTop_ (cg1:397-426) and _applyEtchAt
// (cg1:553-589) with the literal quantiser.  `planes` is what the deposits of the run have pushed at this column so far.
struct HostColumn {
    uint32_t top = 0;
    std::map<int, uint32_t> planes;
    static uint32_t z1(uint32_t r) { return r & 0xffffu; }
    static int mat(uint32_t r) { return (int)((r >> 16) & 255u); }
    static int lnk(uint32_t r) { return (int)(r >> LNK_SHIFT); }
    static uint32_t make(uint32_t q, int m, int l) { return (q & 0xffffu) | ((uint32_t)(m & 255) << 16) | REC_LAYER | ((uint32_t)l << LNK_SHIFT); }
    uint32_t below(uint32_t r) const { const int l = lnk(r); if (l == LNK_NONE) return 0; auto it = planes.find(l); return it == planes.end() ? 0 : it->second; }
    // returns what was pushed into plane P (0: nothing)
    uint32_t deposit(int m, double t, int P, double zs) {
        if (!(t > 0)) return 0;
        const uint32_t q = quantz_host((double)z1(top) * zs + t, zs);
        if (top != 0 && mat(top) == m) { top = (top & 0xffff0000u) | q; return 0; }
        const uint32_t old = top;
        top = make(q, m, old != 0 ? P : LNK_NONE);
        if (old != 0) planes[P] = old;
        return old;
    }
    void etch(int m, double t, double zs) {
        const bool isAll = (m == 255);
        double remaining = t;
        while (remaining > CG_EPS && top != 0) {
            if (!isAll && mat(top) != m) break;
            const uint32_t b = below(top);
            const double zt = (double)z1(top) * zs, zb = (double)z1(b) * zs, thick = zt - zb;
            if (remaining < thick - CG_EPS) { top = (top & 0xffff0000u) | quantz_host(zt - remaining, zs); break; }
            remaining -= thick;
            top = b;                                           // (a reset grid holds no gaps: nothing else to drop)
        }
    }
};

// The run `ops` starts on a reset grid.  When it has at most FRESH_MAX_MASKS distinct masks its outcome per column is one
// of 2^masks columns worked out here; k_vertical_fresh stores them.  Returns false when the run does not qualify.
bool fresh_table(cg_grid* g, const std::vector<cg_grid::VQOp>& ops, FreshTab& T) {
    static const bool off = getenv("CG_NO_FRESH_RUN") != nullptr;
    if (off || ops.empty() || !ops[0].fresh || (int)ops.size() > FRESH_MAX_STEPS || g->N % 4 != 0) return false;
    T = FreshTab{};
    std::vector<int> mask_of(ops.size(), -1);
    for (size_t k = 0; k < ops.size(); k++) {
        if (!ops[k].mask) continue;
        int m = -1;
        for (int q = 0; q < T.nmask; q++) if (T.mask[q] == ops[k].mask) m = q;
        if (m < 0) { if (T.nmask == FRESH_MAX_MASKS) return false; m = T.nmask; T.mask[T.nmask++] = ops[k].mask; }
        mask_of[k] = m;
    }
    for (size_t k = 0; k < ops.size(); k++)
        if (!ops[k].etch) { if (T.ndep == FRESH_MAX_DEP) return false; T.plane[T.ndep++] = ops[k].P; }
    const int ncls = 1 << T.nmask;
    for (int c = 0; c < ncls; c++) {
        HostColumn col;
        int d = 0;
        for (size_t k = 0; k < ops.size(); k++) {
            const bool open = mask_of[k] < 0 || ((c >> mask_of[k]) & 1);
            if (ops[k].etch) { if (open) col.etch(ops[k].mat, ops[k].t, g->zscale); continue; }
            T.pushed[d][c] = open ? col.deposit(ops[k].mat, ops[k].t, ops[k].P, g->zscale) : 0u;
            d++;
        }
        T.top[c] = col.top;
        // a record the final chain does not reach need not be stored (plane slots outside a column's chain are unspecified)
        for (int q = 0; q < T.ndep; q++) {
            bool reached = false;
            for (uint32_t r = col.top; r != 0 && HostColumn::lnk(r) != LNK_NONE; r = col.below(r)) if (HostColumn::lnk(r) == T.plane[q]) { reached = true; break; }
            if (!reached) T.pushed[q][c] = 0;
        }
    }
    T.rt_shift = -1;
    return true;
}

// off / count: the columns [off, off + count) only (count a multiple of the 4096 columns a workgroup takes; the kernel then
// sees a grid that begins at `off`, its plane stride stays N).  What the kernel stores depends on the masks alone, so a
// column may be written by several such launches, or by one of them and by a lateral launch of the FRESH form.
bool fresh_run(cg_grid* g, const std::vector<cg_grid::VQOp>& ops, int* rc, size_t off = 0, size_t count = 0) {
    *rc = CG_OK;
    FreshTab T;
    if (!fresh_table(g, ops, T)) return false;
    GridView gv = view(g);
    if (count) {
        if (count % (1024 * FRESH_U) != 0 || off % 4 != 0 || off + count > g->N) { *rc = fail(g, CG_ERR_STATE, "a run on the reset grid over columns %zu + %zu", off, count); return true; }
        gv.top += off; gv.top_w += off; gv.rec += off;
        for (int q = 0; q < T.nmask; q++) T.mask[q] += off;
    }
    const size_t ncol = count ? count : g->N;
    const size_t nb = (ncol / 4 + (size_t)256 * FRESH_U - 1) / ((size_t)256 * FRESH_U);
    ProfScope ps(g, CG_K_VERTICAL_FUSED);
    const dim3 grid((unsigned)std::max<size_t>(1, nb));
    switch (T.nmask) {
    case 0: hipLaunchKernelGGL(k_vertical_fresh<0>, grid, dim3(256), 0, g->stream, gv, T); break;
    case 1: hipLaunchKernelGGL(k_vertical_fresh<1>, grid, dim3(256), 0, g->stream, gv, T); break;
    case 2: hipLaunchKernelGGL(k_vertical_fresh<2>, grid, dim3(256), 0, g->stream, gv, T); break;
    default: hipLaunchKernelGGL(k_vertical_fresh<3>, grid, dim3(256), 0, g->stream, gv, T); break;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) *rc = fail(g, CG_ERR_HIP, "k_vertical_fresh: %s", hipGetErrorString(e));
    return true;
}

// One k_vertical_fused launch for `ops` (2..FUSE_MAX of them, or 1 on a range) over the columns [off, off + count): the
// whole grid, or whole rows of it whose count is a multiple of the 1024 columns a workgroup takes (the kernel then sees
// a grid that begins at `off`; its plane stride stays N).
int launch_fused(cg_grid* g, const std::vector<cg_grid::VQOp>& ops, size_t off, size_t count) {
    const bool whole = off == 0 && count == g->N;
    FusedOps F = {};
    F.n = (int)ops.size(); F.fresh = (whole && ops[0].fresh) ? 1 : 0;
    for (int k = 0; k < F.n; k++) {
        const cg_grid::VQOp& o = ops[k];
        F.op[k].t = o.t; F.op[k].vo = vert_opts(g, o.t, o.etch);
        F.op[k].etch = o.etch ? 1 : 0; F.op[k].mat = o.mat; F.op[k].P = o.P;
        F.masks[k] = o.mask ? o.mask + off : nullptr;
    }
    F.rt_shift = -1;
    if ((g->ny >> 2) % 256 == 0) {
        const unsigned rt = (unsigned)((g->ny >> 2) / 256);
        if (rt > 1 && (rt & (rt - 1)) == 0) F.rt_shift = __builtin_ctz(rt);
    }
    if (F.fresh) {
        // The leading deposits without a mask leave every column of the reset grid the same: _depoAtTop_ (cg1:397-426)
        // on that one column here, with the quantiser in its literal form (which the device's forms equal by construction).
        uint32_t top = 0;
        const auto z1 = [](uint32_t r) { return r & 0xffffu; };
        const auto make = [](uint32_t q, int mat, int lnk) { return (q & 0xffffu) | ((uint32_t)(mat & 255) << 16) | REC_LAYER | ((uint32_t)lnk << LNK_SHIFT); };
        int k = 0;
        for (; k < F.n && k < FUSE_UNI_MAX && !ops[k].etch && !ops[k].mask; k++) {
            const uint32_t q = quantz_host((double)z1(top) * g->zscale + ops[k].t, g->zscale);
            F.uni_pushed[k] = 0;
            if (top != 0 && (int)((top >> 16) & 255u) == ops[k].mat) top = (top & 0xffff0000u) | q;
            else { F.uni_pushed[k] = top; top = make(q, ops[k].mat, top != 0 ? ops[k].P : LNK_NONE); }
        }
        F.nuni = k; F.uni_top = top;
    }
    GridView gv = view(g);
    if (!whole) {
        if (count % 1024 != 0 || off % 4 != 0 || off + count > g->N) return fail(g, CG_ERR_STATE, "a run of vertical steps over columns %zu + %zu", off, count);
        gv.top += off; gv.top_w += off; gv.rec += off;
    }
    const size_t nb = whole ? (g->N / 4 + 255) / 256 : count / 1024;
    ProfScope ps(g, CG_K_VERTICAL_FUSED);
    const dim3 gridF((unsigned)std::max<size_t>(1, nb));
    switch (F.nuni) {
    case 0: hipLaunchKernelGGL(k_vertical_fused<0>, gridF, dim3(256), 0, g->stream, gv, F); break;
    case 1: hipLaunchKernelGGL(k_vertical_fused<1>, gridF, dim3(256), 0, g->stream, gv, F); break;
    case 2: hipLaunchKernelGGL(k_vertical_fused<2>, gridF, dim3(256), 0, g->stream, gv, F); break;
    default: hipLaunchKernelGGL(k_vertical_fused<3>, gridF, dim3(256), 0, g->stream, gv, F); break;
    }
    HIPCHK(g, hipGetLastError());
    return CG_OK;
}

// The rows of a slab whose lateral step reads a neighbour's message, as ranges of columns: [off, off + count) each.
int boundary_column_ranges(const cg_grid* g, const ConfKernel* K, int R, bool lower, bool upper, size_t off[2], size_t count[2]) {
    Band bb[2];
    const int ti = stream_tile_rows(K->Re);
    const int n = stencil_bands(g->nx, ti, R, lower, upper, 2 /* PART_BOUNDARY */, bb);
    for (int b = 0; b < n; b++) {
        const int r0 = bb[b].t0 * ti, r1 = std::min(g->nx, (bb[b].t0 + bb[b].n) * ti);
        off[b] = (size_t)r0 * g->ny; count[b] = (size_t)(r1 - r0) * g->ny;
    }
    return n;
}

// The columns a slab's edge launches READ: the rows of the edge bands and R rows beyond them, as ranges of whole
// workgroups of k_vertical_fresh (4096 columns), the upper one ending with the slab.  False: the slab is too thin for that
// to leave anything out.
bool edge_read_ranges(const cg_grid* g, const ConfKernel* K, int R, bool lower, bool upper, size_t off[2], size_t count[2], int* n) {
    Band bb[2];
    const int ti = stream_tile_rows(K->Re);
    const int nb = stencil_bands(g->nx, ti, R, lower, upper, 2 /* PART_BOUNDARY */, bb);
    const size_t unit = (size_t)1024 * FRESH_U;
    size_t covered = 0;
    *n = 0;
    for (int b = 0; b < nb; b++) {
        const int r0 = std::max(0, bb[b].t0 * ti - R), r1 = std::min(g->nx, (bb[b].t0 + bb[b].n) * ti + R);
        const size_t a = (size_t)r0 * g->ny, e = (size_t)r1 * g->ny;
        size_t o, c;
        if (bb[b].t0 == 0) { o = 0; c = (e + unit - 1) / unit * unit; }
        else { c = (g->N - a + unit - 1) / unit * unit; if (c > g->N) return false; o = g->N - c; }
        if (o + c > g->N || o % 4 != 0) return false;
        off[*n] = o; count[*n] = c; (*n)++;
        covered += c;
    }
    return covered < g->N;
}

int flush_vertical(cg_grid* g) {
    if (g->lat.active) {
        // the waiting lateral deposit, with the vertical steps queued behind it inside its launch when there are any
        const cg_grid::LatPending L = g->lat;
        g->lat.active = false; g->lat.pre.clear();
        std::vector<cg_grid::VQOp> behind; behind.swap(g->vq);
        FusedOps F = {};
        const bool fold = !behind.empty() && (int)behind.size() < FUSE_MAX && L.K->Re <= LEAN_FOLD_MAX_R;
        if (fold) {
            F.n = 1 + (int)behind.size();
            F.op[0].etch = 0; F.op[0].mat = L.mat; F.op[0].P = L.P; F.op[0].t = L.t_eff; F.op[0].vo = VertOpts{-1, -1, -1, -1};
            for (int k = 1; k < F.n; k++) {
                const cg_grid::VQOp& o = behind[k - 1];
                F.op[k].t = o.t; F.op[k].vo = vert_opts(g, o.t, o.etch);
                F.op[k].etch = o.etch ? 1 : 0; F.op[k].mat = o.mat; F.op[k].P = o.P;
                F.masks[k] = o.mask;
            }
            F.rt_shift = -1;
        }
        FreshTab T;
        const bool pre = !L.pre.empty();
        if (pre && !fresh_table(g, L.pre, T)) return fail(g, CG_ERR_STATE, "the waiting run of vertical steps no longer fits the table form");
        if (L.halo_lo || L.halo_up) {
            // a slab: the steps behind go out inside the launch over the rows that read no message; the other rows get
            // the deposit from the kernel that reads the messages and then the steps in a pass of their own.  A run on the
            // reset grid in front of the deposit is put together from its masks by the first launch; the rows the second
            // one reads got it when the step was opened (cg_stencil_begin).
            int rc = launch_stencil(g, false, L.mask, L.mat, L.t_eff, L.R, L.K, -1, L.halo_lo, L.halo_up, L.P, 1 /* PART_INTERIOR */, false, fold ? &F : nullptr, pre ? &T : nullptr);
            if (rc) return rc;
            rc = launch_stencil(g, false, L.mask, L.mat, L.t_eff, L.R, L.K, -1, L.halo_lo, L.halo_up, L.P, 2 /* PART_BOUNDARY */, false, nullptr, nullptr);
            if (rc) return rc;
            std::swap(g->top, g->top_alt);
            if (fold) {
                size_t off[2], count[2];
                const int nr = boundary_column_ranges(g, L.K, L.R, L.halo_lo != nullptr, L.halo_up != nullptr, off, count);
                for (int b = 0; b < nr; b++) { rc = launch_fused(g, behind, off[b], count[b]); if (rc) return rc; }
                return CG_OK;
            }
            g->vq.swap(behind);
        } else {
            int rc = launch_stencil(g, false, L.mask, L.mat, L.t_eff, L.R, L.K, -1, nullptr, nullptr, L.P, 0 /* PART_ALL */, false, fold ? &F : nullptr, pre ? &T : nullptr);
            if (rc) return rc;
            std::swap(g->top, g->top_alt);
            if (fold) return CG_OK;
            g->vq.swap(behind);
        }
    }
    if (g->vq.empty()) return CG_OK;
    std::vector<cg_grid::VQOp> ops; ops.swap(g->vq);
    if (ops.size() == 1) {
        const cg_grid::VQOp& o = ops[0];
        ProfScope ps(g, o.etch ? CG_K_ETCH_VERTICAL : CG_K_DEPO_VERTICAL);
        if (o.etch) launch_vertical<true>(g, o.mask, o.mat, o.t, -1, false);
        else launch_vertical<false>(g, o.mask, o.mat, o.t, o.P, o.fresh);
        HIPCHK(g, hipGetLastError());
        return CG_OK;
    }
    if (ops[0].fresh) {
        int rcf = 0;
        if (fresh_run(g, ops, &rcf)) return rcf;            // a run on a reset grid: one column per mask class, looked up
    }
    if ((int)ops.size() > FUSE_MAX) {
        // (a long run that the class form could not take after all: in pieces)
        for (size_t b = 0; b < ops.size(); b += FUSE_MAX) {
            g->vq.assign(ops.begin() + b, ops.begin() + std::min(ops.size(), b + FUSE_MAX));
            if (b > 0) g->vq[0].fresh = false;
            int rcp = flush_vertical(g); if (rcp) return rcp;
        }
        return CG_OK;
    }
    return launch_fused(g, ops, 0, g->N);
}

int enqueue_vertical(cg_grid* g, bool etch, const uint8_t* mask, int mat, double t) {
    cg_grid::VQOp op{mask, etch, mat, t, -1, false};
    if (!etch) {
        if (g->len_bound >= g->lmax || g->len_bound + 1 > g->lmax_vis || g->nextp >= std::min(g->lmax, SPLIT_PLANES)) {
            int rc = flush_vertical(g); if (rc) return rc;
            return run_depo_vertical(g, mask, mat, t);
        }
        op.P = g->nextp++;
        op.fresh = g->pristine;
        g->pristine = false;
        if (g->len_bound < 255) g->len_bound += 1;
    }
    g->len_valid = false;
    g->vq.push_back(op);
    // the run of a reset grid goes out as a table look-up whatever its length (fresh_run); any other run holds FUSE_MAX steps
    const int cap = g->vq[0].fresh ? FRESH_MAX_STEPS : (g->lat.active ? FUSE_MAX - 1 : FUSE_MAX);      // (a waiting lateral deposit takes the first place of its launch)
    if ((int)g->vq.size() >= cap) return flush_vertical(g);
    return CG_OK;
}
}  // namespace

// =============================================================================================
extern "C" {

const char* cg_version(void) { return "colgrid 0.1 gfx950"; }

const char* cg_last_error(const cg_grid* g) { return g ? g->err.c_str() : t_create_error.c_str(); }

int cg_create_slab(int nx_global, int ny, int i0, int nx_local, double dx, double dy, int lmax_init, double zscale,
                   cg_grid** out) {
    if (!out) return fail(nullptr, CG_ERR_ARG, "out is null");
    *out = nullptr;
    if (nx_global < 1 || ny < 1 || nx_local < 1 || i0 < 0 || i0 + nx_local > nx_global || !(dx > 0) || !(dy > 0) || !(zscale > 0))
        return fail(nullptr, CG_ERR_ARG, "bad domain: nx_global=%d ny=%d i0=%d nx_local=%d dx=%g dy=%g zscale=%g", nx_global, ny, i0,
                    nx_local, dx, dy, zscale);
    cg_grid* g = new cg_grid();
    g->fuse = getenv("CG_NO_FUSE") == nullptr;
    hipError_t e = hipGetDevice(&g->device);
    if (e != hipSuccess) { int rc = fail(nullptr, CG_ERR_HIP, "hipGetDevice: %s", hipGetErrorString(e)); delete g; return rc; }
    g->nx = nx_local; g->ny = ny; g->nx_global = nx_global; g->i0 = i0;
    g->N = (size_t)nx_local * ny; g->dx = dx; g->dy = dy; g->zscale = zscale;
    int lmax = next_pow2(lmax_init);
    if (lmax > 256) lmax = 256;
    {   // what float arithmetic may stand in for double arithmetic at this zscale (StreamOpts)
        StreamOpts& o = g->sopt;
        o.zs_hi = (float)zscale; o.zs_lo = (float)(zscale - (double)o.zs_hi); o.izf = (float)(1.0 / zscale);
        o.deq_f32 = 1;
        for (unsigned q = 0; q < 65536u && o.deq_f32; q++) {
            const float qf = (float)q;
            if (std::fmaf(qf, o.zs_hi, qf * o.zs_lo) != (float)((double)q * zscale)) o.deq_f32 = 0;
        }
        o.grow_f32 = zscale >= 0.02 ? 1 : 0;               // 2^-24 * zscale > 1e-9 with room to spare
        o.skip_init = 0;                                   // per step
    }
    int rc = CG_OK;
    auto body = [&]() -> int {
        HIPCHK(g, hipStreamCreateWithFlags(&g->own_stream, hipStreamNonBlocking));
        g->stream = g->own_stream;
        HIPCHK(g, hipMalloc(&g->d_flags, 4 * sizeof(int)));
        HIPCHK(g, hipMalloc(&g->d_red, 8 * sizeof(unsigned)));
        HIPCHK(g, hipMalloc(&g->d_sum, sizeof(unsigned long long)));
        HIPCHK(g, hipMalloc(&g->d_bbox, 4 * sizeof(int)));
        HIPCHK(g, hipMalloc(&g->d_stats, CG_NSTAT * sizeof(unsigned long long)));
        HIPCHK(g, hipMemset(g->d_flags, 0, 4 * sizeof(int)));
        HIPCHK(g, hipMemset(g->d_stats, 0, CG_NSTAT * sizeof(unsigned long long)));
        HIPCHK(g, hipEventCreate(&g->ev0)); HIPCHK(g, hipEventCreate(&g->ev1));
        int r = scratch_resolve(g); if (r) return r;
        r = alloc_planes(g, lmax); if (r) return r;
        g->lmax_vis = lmax;
        r = zero_planes(g); if (r) return r;
        HIPCHK(g, hipStreamSynchronize(g->stream));
        return CG_OK;
    };
    rc = body();
    if (rc) { t_create_error = g->err; cg_destroy(g); return rc; }
    *out = g;
    return CG_OK;
}

int cg_create(double lx, double ly, double dx, double dy, int lmax_init, double zscale, cg_grid** out) {
    if (!(dx > 0) || !(dy > 0)) return fail(nullptr, CG_ERR_ARG, "bad pitch dx=%g dy=%g", dx, dy);
    // setDomain (script_columngrid.js:49-60)
    double nx = js_round(lx / dx), ny = js_round(ly / dy);
    int NX = nx < 1 ? 1 : (int)nx, NY = ny < 1 ? 1 : (int)ny;
    return cg_create_slab(NX, NY, 0, NX, dx, dy, lmax_init, zscale, out);
}

int cg_destroy(cg_grid* g) {
    if (!g) return CG_OK;
    (void)hipSetDevice(g->device);
    g->vq.clear(); g->lat.active = false; g->lat.pre.clear();
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    free_planes(g);
    for (auto m : g->masks) if (m) (void)hipFree(m);
    for (auto k : g->kcache) free_kernel(k);
    for (auto& s : g->snaps) { (void)hipFree(s.second.len); (void)hipFree(s.second.rec); (void)hipFree(s.second.top); }
    if (!g->halo_external) for (auto h : g->halo) if (h) (void)hipFree(h);
    for (auto& r : g->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : g->prof_pool) (void)hipEventDestroy(e);
    if (g->d_flags) (void)hipFree(g->d_flags);
    if (g->d_red) (void)hipFree(g->d_red);
    if (g->d_sum) (void)hipFree(g->d_sum);
    if (g->d_bbox) (void)hipFree(g->d_bbox);
    if (g->cand) (void)hipFree(g->cand);
    if (g->d_sink) (void)hipFree(g->d_sink);
    if (g->mask_tab) (void)hipFree(g->mask_tab);
    if (g->mask_list) (void)hipFree(g->mask_list);
    if (g->uf) (void)hipFree(g->uf);
    if (g->d_ald_taps) (void)hipFree(g->d_ald_taps);
    free_alt(g, 0); free_alt(g, 1); drop_seeds(g);
    if (g->d_stats) (void)hipFree(g->d_stats);
    if (g->sl_mat) (void)hipFree(g->sl_mat);
    if (g->sl_zp) (void)hipFree(g->sl_zp);
    if (g->sl_len) (void)hipFree(g->sl_len);
    if (g->sl_idx) (void)hipFree(g->sl_idx);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->own_stream) (void)hipStreamDestroy(g->own_stream);
    delete g;
    return CG_OK;
}

int cg_reset(cg_grid* g, int lmax_init) {
    // (steps still waiting in the queue are launched first although the reset wipes out what they do: a step that was
    // asked for is a step that runs, whoever times the calls)
    NEED_H(g);
    int lmax = next_pow2(lmax_init > 0 ? lmax_init : g->lmax_vis);
    if (lmax > 256) lmax = 256;
    g->pend.active = false;
    g->lmax_vis = lmax;
    // the planes mapped so far stay (capacity, not state); more are mapped if the new Lmax asks for them
    if (lmax > g->lmax) { int rc = grow_planes(g, lmax - g->lmax); if (rc) return rc; }
    return zero_planes(g, true);
}

int cg_set_plane_budget(cg_grid* g, int max_planes) {
    NEED_H(g);
    if (max_planes < 1) return fail(g, CG_ERR_ARG, "plane budget %d", max_planes);
    g->plane_budget = std::min(max_planes, MAX_PLANES);
    return CG_OK;
}

int cg_raise_lmax(cg_grid* g, int lmax) {
    NEED_H0(g);
    if (lmax < 1 || lmax > 256) return fail(g, CG_ERR_ARG, "Lmax %d outside 1..256", lmax);
    if (lmax > g->lmax_vis) g->lmax_vis = lmax;
    return CG_OK;
}

int cg_clear(cg_grid* g) { NEED_H(g); g->pend.active = false; return zero_planes(g, g->len_bound > g->lmax); }

int cg_get_info(const cg_grid* g, cg_info* o) {
    if (!g || !o) return CG_ERR_ARG;
    o->nx = g->nx; o->ny = g->ny; o->nx_global = g->nx_global; o->i0 = g->i0; o->lmax = g->lmax_vis; o->device = g->device;
    o->dx = g->dx; o->dy = g->dy; o->zscale = g->zscale;
    o->offset_x = -(g->nx_global - 1) * g->dx / 2; o->offset_y = -(g->ny - 1) * g->dy / 2;
    o->lx_eff = g->nx_global * g->dx; o->ly_eff = g->ny * g->dy;
    return CG_OK;
}

int cg_set_stream(cg_grid* g, void* s) {
    NEED_H(g);
    HIPCHK(g, hipStreamSynchronize(g->stream));
    g->stream = s ? (hipStream_t)s : g->own_stream;
    return CG_OK;
}

int cg_get_stream(cg_grid* g, void** s) {
    NEED_H(g);
    if (!s) return fail(g, CG_ERR_ARG, "null output");
    *s = (void*)g->stream;
    return CG_OK;
}

int cg_sync(cg_grid* g) { NEED_H(g); HIPCHK(g, hipStreamSynchronize(g->stream)); return CG_OK; }

// ---- masks ----------------------------------------------------------------------------------
static int new_mask_slot(cg_grid* g, uint8_t** buf, int* id) {
    HIPCHK(g, hipMalloc(buf, std::max<size_t>(g->N, 1)));
    for (size_t k = 0; k < g->masks.size(); k++)
        if (!g->masks[k]) { g->masks[k] = *buf; g->mask_any[k] = -1; *id = (int)k; return CG_OK; }
    g->masks.push_back(*buf); g->mask_any.push_back(-1); *id = (int)g->masks.size() - 1;
    return CG_OK;
}

// Does the mask open any column of this handle?  Looked up once per mask (rasters never change).
static int mask_opens_any(cg_grid* g, int mask_id, bool* any) {
    *any = true;
    if (mask_id == CG_MASK_NONE) return CG_OK;
    if (mask_id < 0 || (size_t)mask_id >= g->masks.size() || !g->masks[mask_id]) return fail(g, CG_ERR_ARG, "unknown mask id %d", mask_id);
    if (g->mask_any[mask_id] < 0) {
        unsigned r = 0;
        HIPCHK(g, hipMemsetAsync(g->d_red + 4, 0, sizeof(unsigned), g->stream));
        hipLaunchKernelGGL(k_mask_any, dim3(blocks_for(g->N / 4 + 1)), dim3(256), 0, g->stream, g->masks[mask_id], g->N, g->d_red + 4);
        HIPCHK(g, hipGetLastError());
        HIPCHK(g, hipMemcpyAsync(&r, g->d_red + 4, sizeof r, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
        g->mask_any[mask_id] = r ? 1 : 0;
    }
    *any = g->mask_any[mask_id] != 0;
    return CG_OK;
}

int cg_mask_any_open(cg_grid* g, int mask_id, int* any) {
    NEED_H(g);
    if (!any) return fail(g, CG_ERR_ARG, "null output");
    bool a = true;
    int rc = mask_opens_any(g, mask_id, &a); if (rc) return rc;
    *any = a ? 1 : 0;
    return CG_OK;
}

int cg_mask_from_raster(cg_grid* g, const uint8_t* host, size_t n, int* mask_id) {
    NEED_H(g);
    if (!host || !mask_id || n != g->N) return fail(g, CG_ERR_ARG, "mask raster: need %zu bytes, got %zu", g->N, n);
    uint8_t* buf; int rc = new_mask_slot(g, &buf, mask_id); if (rc) return rc;
    HIPCHK(g, hipMemcpyAsync(buf, host, n, hipMemcpyHostToDevice, g->stream));
    hipLaunchKernelGGL(k_mask_normalise, dim3(blocks_for(n)), dim3(256), 0, g->stream, buf, buf, n);
    HIPCHK(g, hipStreamSynchronize(g->stream));      // the host buffer is the caller's again
    return CG_OK;
}

int cg_mask_from_device_raster(cg_grid* g, const void* dev, size_t n, int* mask_id) {
    NEED_H(g);
    if (!dev || !mask_id || n != g->N) return fail(g, CG_ERR_ARG, "mask raster: need %zu bytes, got %zu", g->N, n);
    uint8_t* buf; int rc = new_mask_slot(g, &buf, mask_id); if (rc) return rc;
    hipLaunchKernelGGL(k_mask_normalise, dim3(blocks_for(n)), dim3(256), 0, g->stream, (const uint8_t*)dev, buf, n);
    HIPCHK(g, hipGetLastError());
    return CG_OK;
}

int cg_mask_from_shapes(cg_grid* g, const cg_shape* shapes, int nshapes, const double* pts, int npts, int* mask_id) {
    NEED_H(g);
    if (!mask_id || nshapes < 0 || npts < 0 || (nshapes > 0 && !shapes) || (npts > 0 && !pts))
        return fail(g, CG_ERR_ARG, "bad shape list");
    // (a polygon must name points inside the table: checked here for the lists the one-pass rasteriser walks, by
    // k_shape_ranges for the long ones -- a pass of the host over half a million records costs a millisecond)
    const auto check_polygons = [&]() -> int {
        for (int s = 0; s < nshapes; s++)
            if (shapes[s].type == CG_SHAPE_POLYGON && (shapes[s].pt_off < 0 || shapes[s].pt_cnt < 0 || (long long)shapes[s].pt_off + shapes[s].pt_cnt > npts))
                return fail(g, CG_ERR_ARG, "shape %d: polygon points [%d,+%d) outside the %d given", s, shapes[s].pt_off, shapes[s].pt_cnt, npts);
        return CG_OK;
    };
    static_assert(sizeof(Shape) == sizeof(cg_shape), "Shape mirrors cg_shape");
    static const bool tm_on = getenv("CG_MASK_TIMING") != nullptr;
    auto tm_now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tm0 = tm_now();
    auto tm = [&](const char* what) { if (!tm_on) return; (void)hipStreamSynchronize(g->stream); const double t1 = tm_now(); fprintf(stderr, "  [mask] %-18s %.3f ms\n", what, t1 - tm0); tm0 = t1; };
    tm("start");
    // From a handful of shapes on they are binned by tile, on the device (cg_mask_bin.hip.h): count, scan, fill.  The one
    // number the host needs in between is the length of the lists.
    static const int min_shapes = [] { const char* e = getenv("CG_MASK_BIN_MIN"); return e ? atoi(e) : 8; }();
    const int tn_i = (g->nx + MASK_TILE - 1) / MASK_TILE, tn_j = (g->ny + MASK_TILE - 1) / MASK_TILE;
    const size_t ntiles = (size_t)tn_i * tn_j;
    constexpr int WIDE_CAP = 64;                                // more wide shapes than this: every cell would walk them all anyway
    const bool bin = nshapes >= min_shapes && ntiles <= (size_t)INT32_MAX / 2;
    if (!bin) { int rcc = check_polygons(); if (rcc) return rcc; }
    // one piece of device memory that the handle keeps between calls: shapes | points | ranges | counts | offsets | cursors | wide | counters
    const auto up16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t b_sh = up16(std::max(1, nshapes) * sizeof(Shape)), b_pts = up16((size_t)std::max(1, npts) * 2 * sizeof(double));
    const size_t b_rg = bin ? (size_t)nshapes * sizeof(TileRange) : 0, b_cnt = bin ? ntiles * sizeof(int) : 0, b_off = bin ? (ntiles + 1) * sizeof(int) : 0;
    const size_t o_pts = b_sh, o_rg = o_pts + b_pts, o_cnt = o_rg + up16(b_rg), o_off = o_cnt + up16(b_cnt), o_cur = o_off + up16(b_off);
    const size_t o_wide = o_cur + up16(b_cnt), o_ctr = o_wide + WIDE_CAP * sizeof(int), total_b = o_ctr + 4 * sizeof(unsigned);
    if (g->mask_tab_cap < total_b) {
        if (g->mask_tab) { HIPCHK(g, hipStreamSynchronize(g->stream)); (void)hipFree(g->mask_tab); g->mask_tab = nullptr; g->mask_tab_cap = 0; }
        HIPCHK(g, hipMalloc(&g->mask_tab, total_b + total_b / 4));
        g->mask_tab_cap = total_b + total_b / 4;
    }
    char* const tab = g->mask_tab;
    Shape* d_sh = (Shape*)tab; double* d_pts = (double*)(tab + o_pts);
    if (nshapes) HIPCHK(g, hipMemcpyAsync(d_sh, shapes, nshapes * sizeof(Shape), hipMemcpyHostToDevice, g->stream));
    if (npts) HIPCHK(g, hipMemcpyAsync(d_pts, pts, (size_t)npts * 2 * sizeof(double), hipMemcpyHostToDevice, g->stream));
    tm("tables + h2d");
    uint8_t* buf = nullptr;
    int* d_list = nullptr;
    ProfScope ps(g, CG_K_MASK);
    bool binned = false;
    if (bin) {
        TileRange* d_rg = (TileRange*)(tab + o_rg); int* d_cnt = (int*)(tab + o_cnt); int* d_off = (int*)(tab + o_off); int* d_cur = (int*)(tab + o_cur);
        int* d_wide = (int*)(tab + o_wide); unsigned* d_ctr = (unsigned*)(tab + o_ctr);
        HIPCHK(g, hipMemsetAsync(d_cnt, 0, b_cnt, g->stream));
        HIPCHK(g, hipMemsetAsync(d_ctr, 0, 4 * sizeof(unsigned), g->stream));
        const unsigned wide_area = (unsigned)std::max<size_t>(64, ntiles / 64);
        hipLaunchKernelGGL(k_shape_ranges, dim3((unsigned)((nshapes + 255) / 256)), dim3(256), 0, g->stream, g->nx, g->ny, g->i0, g->dx, g->dy, d_sh, nshapes,
                           d_pts, tn_i, tn_j, wide_area, d_rg, d_cnt, d_wide, WIDE_CAP, d_ctr, npts);
        hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, g->stream, d_cnt, (int)ntiles, d_off);
        HIPCHK(g, hipGetLastError());
        unsigned ctr[4] = {0, 0, 0, 0};
        HIPCHK(g, hipMemcpyAsync(ctr, d_ctr, sizeof ctr, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
        tm("ranges + scan");
        if (ctr[3]) {
            const cg_shape& b = shapes[ctr[3] - 1];
            return fail(g, CG_ERR_ARG, "shape %u: polygon points [%d,+%d) outside the %d given", ctr[3] - 1, b.pt_off, b.pt_cnt, npts);
        }
        { int rc = new_mask_slot(g, &buf, mask_id); if (rc) return rc; }
        tm("raster alloc");
        if (!ctr[2] && ctr[0] <= (unsigned)WIDE_CAP && ctr[1] <= (1u << 27)) {
            if (g->mask_list_cap < ctr[1]) {
                if (g->mask_list) { (void)hipFree(g->mask_list); g->mask_list = nullptr; g->mask_list_cap = 0; }      // (the stream is idle: synchronised above)
                const size_t want = (size_t)ctr[1] + ctr[1] / 4 + 1;
                HIPCHK(g, hipMalloc(&g->mask_list, want * sizeof(int)));
                g->mask_list_cap = want;
            }
            d_list = g->mask_list;
            HIPCHK(g, hipMemcpyAsync(d_cur, d_off, b_cnt, hipMemcpyDeviceToDevice, g->stream));
            hipLaunchKernelGGL(k_shape_fill, dim3((unsigned)((nshapes + 255) / 256)), dim3(256), 0, g->stream, d_rg, nshapes, tn_j, d_cur, d_list);
            hipLaunchKernelGGL(k_mask_shapes_tiles, dim3((unsigned)ntiles), dim3(256), 0, g->stream, g->nx, g->ny, g->i0, g->dx, g->dy, d_sh, d_pts,
                               tn_j, d_off, d_list, d_wide, (int)ctr[0], buf);
            binned = true;
        }
    }
    if (!binned) {
        if (!buf) { int rc = new_mask_slot(g, &buf, mask_id); if (rc) return rc; }
        hipLaunchKernelGGL(k_mask_shapes, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, g->nx, g->ny, g->i0, g->dx, g->dy, d_sh, nshapes,
                           d_pts, buf);
    }
    HIPCHK(g, hipGetLastError());
    HIPCHK(g, hipStreamSynchronize(g->stream));
    tm("fill + raster");
    // (the host's shape table has been read and the raster is complete; the device tables stay with the handle)
    return CG_OK;
}

int cg_mask_download(cg_grid* g, int mask_id, uint8_t* out, size_t n) {
    NEED_H(g);
    bool ok; const uint8_t* m = mask_ptr(g, mask_id, &ok);
    if (!ok || !m || !out || n != g->N) return fail(g, CG_ERR_ARG, "bad mask id %d or size", mask_id);
    HIPCHK(g, hipMemcpyAsync(out, m, n, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    return CG_OK;
}

int cg_mask_destroy(cg_grid* g, int mask_id) {
    NEED_H(g);
    bool ok; const uint8_t* m = mask_ptr(g, mask_id, &ok);
    if (!ok || !m) return fail(g, CG_ERR_ARG, "bad mask id %d", mask_id);
    HIPCHK(g, hipStreamSynchronize(g->stream));
    (void)hipFree(g->masks[mask_id]); g->masks[mask_id] = nullptr;
    return CG_OK;
}

// ---- steps ----------------------------------------------------------------------------------
int cg_stencil_begin(cg_grid* g, int kind, int mask_id, int mat, double thickness, double conformality, int has_lower, int has_upper,
                     int* radius_rows) {
    NEED_H0(g);
    if (radius_rows) *radius_rows = 0;
    if (kind != CG_STENCIL_DEPOSIT && kind != CG_STENCIL_ETCH) return fail(g, CG_ERR_ARG, "bad stencil kind %d", kind);
    if (g->pend.active) return fail(g, CG_ERR_STATE, "a stencil step is already open");
    bool ok; const uint8_t* mask = mask_ptr(g, mask_id, &ok);
    if (!ok) return fail(g, CG_ERR_ARG, "unknown mask id %d", mask_id);
    if (mat < 0 || mat > 255) return fail(g, CG_ERR_ARG, "material id %d outside 0..255", mat);
    // a thickness that is not a number never passes "thickness <= 0" in the runtime's guard nor "t <= 0" in the grid
    // (rt:159, cg1:437, 600: both comparisons are false for NaN), and every height it then produces is NaN, which a
    // Uint16Array stores as 0: columns collapse.  Nothing a caller can want; taken as "no thickness", a no-op.
    const double t = std::isnan(thickness) ? 0.0 : std::max(0.0, thickness);
    if (!(t > 0)) return CG_OK;                                           // cg1:437, 600
    // A conformality that is not a number passes the runtime's typeof test (rt:156) and the clamp (cg1:440, 603)
    // unchanged, fails "latRange < min(dx, dy)" and takes the lateral path with a kernel that has no taps:
    // not a step of the reference's model, refused.
    if (std::isnan(conformality)) return fail(g, CG_ERR_ARG, "conformality is not a number");
    const double S = std::max(0.0, std::min(1.0, conformality));          // cg1:440, 603
    const double latRange = t * S;
    // a vertical step joins the queue of vertical steps when the fused kernel can take it; anything else first
    // launches what is queued
    const bool can_fuse = latRange < std::min(g->dx, g->dy) && g->fuse && !g->stats_on && g->N % 4 == 0;
    // A lateral deposit without a mask that finds a run of vertical steps waiting on the reset grid leaves it waiting:
    // when the deposit takes the lean kernel (known further down) the run goes out inside that launch.
    static const bool no_fold = getenv("CG_NO_FOLD") != nullptr;
    const bool slab = has_lower || has_upper;
    bool prefold = !no_fold && !can_fuse && kind == CG_STENCIL_DEPOSIT && !mask && (slab || g->nx == g->nx_global) && g->fuse &&
                   !g->stats_on && !g->lat.active && !g->vq.empty() && g->vq[0].fresh && g->split && !g->has_gaps && !g->has_cavity &&
                   g->len_bound + 1 <= g->lmax_vis && g->len_bound < g->lmax && g->nextp < std::min(g->lmax, SPLIT_PLANES);
    if (prefold) { FreshTab probe; prefold = fresh_table(g, g->vq, probe); }
    if (prefold && slab && g->N % ((size_t)1024 * FRESH_U) != 0) prefold = false;
    if (!can_fuse && !prefold) { int rcf = flush_vertical(g); if (rcf) return rcf; }
    // a reset grid whose first step is a deposit everywhere needs no clearing pass: the kernel writes
    // top of every column itself
    const bool fresh = g->pristine && kind == CG_STENCIL_DEPOSIT && latRange < std::min(g->dx, g->dy) && !mask && g->N % 4 == 0 && g->lmax >= 1;
    if (!fresh) { int rcm = materialize(g); if (rcm) return rcm; }
    { int rcs = ensure_split(g); if (rcs) return rcs; }
    stats_begin(g);
    if (kind == CG_STENCIL_DEPOSIT && mat == CG_MAT_CAVITY) g->has_cavity = true;      // a "cavity" laid down as a film
    if (latRange < std::min(g->dx, g->dy)) {                              // cg1:445, 609
        if (can_fuse && g->split) return enqueue_vertical(g, kind == CG_STENCIL_ETCH, mask, mat, t);
        { int rcf = flush_vertical(g); if (rcf) return rcf; }
        if (kind == CG_STENCIL_DEPOSIT) return run_depo_vertical(g, mask, mat, t);
        const dim3 grid(blocks_for(g->N)), blk(256);
        if (g->split && g->N % 4 == 0) {    // 4 columns per thread, no look at the lengths
            ProfScope ps(g, CG_K_ETCH_VERTICAL);
            launch_vertical<true>(g, mask, mat, t, -1, false);
            g->len_valid = false;
        } else {
            int rcl = ensure_len(g); if (rcl) return rcl;
            ProfScope ps(g, CG_K_ETCH_VERTICAL);
            if (g->stats_on) hipLaunchKernelGGL((k_vertical_1<true, true>), grid, blk, 0, g->stream, view(g), mask, mat, t, -1, g->d_flags, g->d_stats);
            else hipLaunchKernelGGL((k_vertical_1<true, false>), grid, blk, 0, g->stream, view(g), mask, mat, t, -1, g->d_flags, g->d_stats);
        }
        HIPCHK(g, hipGetLastError());
        return CG_OK;
    }
    ConfKernel* K = get_kernel(g, t, S);
    const int R = K->Rx;
    size_t pre_off[2] = {0, 0}, pre_count[2] = {0, 0}; int pre_n = 0;
    if (prefold) {
        // (the form is known once the kernel's device tables exist: a kernel seen for the first time gets them here)
        if (K->Ry <= K->Rx) { int rcu = upload_taps(g, K, R); if (rcu) return rcu; }
        const StencilForm form = stencil_form(g, K, false, mat, -1, false);
        bool ok_pre = form.streams && form.lean;
        // on a slab the rows that the edge launches read get the run right away (the rows to send are packed from them)
        if (ok_pre && slab) ok_pre = edge_read_ranges(g, K, R, has_lower != 0, has_upper != 0, pre_off, pre_count, &pre_n);
        if (!ok_pre) { prefold = false; int rcf = flush_vertical(g); if (rcf) return rcf; }
    }
    // The reference pads its height map by Rx cells in BOTH directions (script_columngrid.js:471, 666) while the taps reach
    // Ry = ceil(t/dy) cells along j.  With Ry <= Rx (any dy >= dx) every tap stays inside the pad and the step is well
    // defined; with Ry > Rx its reads run into the neighbouring pad row or past the end of a scratch buffer that keeps
    // what earlier, larger steps left there: nothing a caller could rely on, refused.
    if (K->Ry > K->Rx)
        return fail(g, CG_ERR_UNSUPPORTED, "this lateral step reaches %d cells along y but the reference pads its height map by Rx = %d "
                                            "(script_columngrid.js:471, 666): undefined there when dy < dx", K->Ry, K->Rx);
    // A lateral deposit whose mask opens no column returns before anything changes, identify_cavity
    // included (cg1:474-485; the kernel cache above has its entry by then, cg1:470).  On a slab the
    // question is about the whole domain: the driver asks cg_mask_any_open on every slab.
    if (kind == CG_STENCIL_DEPOSIT && mask && g->nx == g->nx_global) {
        bool any = true;
        int rca = mask_opens_any(g, mask_id, &any); if (rca) return rca;
        if (!any) return CG_OK;
    }
    if (stencil_lds_bytes(R) > 160 * 1024) return fail(g, CG_ERR_UNSUPPORTED, "stencil radius %d cells does not fit the LDS tile", R);
    if ((has_lower || has_upper) && g->nx < R)
        return fail(g, CG_ERR_UNSUPPORTED, "slab of %d rows is thinner than the stencil radius %d", g->nx, R);
    int rc = ensure_stencil_scratch(g, R, has_lower || has_upper); if (rc) return rc;
    rc = upload_taps(g, K, R); if (rc) return rc;
    if (!g->split) {
        // columns too tall for the split form: the step rewrites the planes in place and reads the OLD heights
        // from a copy of the tops made now
        ProfScope ps(g, CG_K_OTHER);
        hipLaunchKernelGGL(k_snapshot_tops, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g));
        HIPCHK(g, hipGetLastError());
    }
    if (prefold && slab) {
        for (int b = 0; b < pre_n; b++) {
            int rcr = 0;
            if (!fresh_run(g, g->vq, &rcr, pre_off[b], pre_count[b])) return fail(g, CG_ERR_STATE, "the waiting run of vertical steps no longer fits the table form");
            if (rcr) return rcr;
        }
    }
    if (has_lower || has_upper) {
        ProfScope ps(g, CG_K_HALO);
        hipLaunchKernelGGL(k_halo_pack, dim3(blocks_for((size_t)R * g->ny)), dim3(256), 0, g->stream, view(g), mask, mat,
                           kind == CG_STENCIL_ETCH ? 1 : 0, R, g->halo[0], g->halo[1]);
        HIPCHK(g, hipGetLastError());
    }
    g->pend.active = true; g->pend.kind = kind; g->pend.mask_id = mask_id; g->pend.mat = mat; g->pend.R = R; g->pend.K = K;
    g->pend.t_eff = (kind == CG_STENCIL_DEPOSIT) ? K->baseAdd : t;        // cg1:470/499 vs cg1:702
    g->pend.has_lower = has_lower != 0; g->pend.has_upper = has_upper != 0;
    g->pend.interior_done = false; g->pend.may_overflow = false; g->pend.P = -1; g->pend.prefold = prefold; g->pend.held = false;
    if (radius_rows) *radius_rows = R;
    return CG_OK;
}

int cg_kernel_cache_touch(cg_grid* g, double thickness, double conformality) {
    NEED_H0(g);
    if (std::isnan(conformality)) return fail(g, CG_ERR_ARG, "conformality is not a number");
    const double t = std::isnan(thickness) ? 0.0 : std::max(0.0, thickness);
    const double S = std::max(0.0, std::min(1.0, conformality));
    if (!(t > 0) || t * S < std::min(g->dx, g->dy)) return CG_OK;         // cg1:437, 445: such a step never asks for a kernel
    (void)get_kernel(g, t, S);
    return CG_OK;
}

int cg_halo_buffers(cg_grid* g, void** send_lower, void** send_upper, void** recv_lower, void** recv_upper, size_t* bytes_each) {
    NEED_H0(g);
    if (!g->pend.active) { NEED_H(g); return fail(g, CG_ERR_STATE, "no stencil step is open"); }      // (an open step may have left a run of vertical steps waiting for it)
    if (send_lower) *send_lower = g->halo[0];
    if (send_upper) *send_upper = g->halo[1];
    if (recv_lower) *recv_lower = g->halo[2];
    if (recv_upper) *recv_upper = g->halo[3];
    if (bytes_each) *bytes_each = (size_t)g->pend.R * g->ny * 3;
    return CG_OK;
}

int cg_set_halo_buffers(cg_grid* g, void* send_lower, void* send_upper, void* recv_lower, void* recv_upper, size_t bytes_each) {
    NEED_H(g);
    if (g->pend.active) return fail(g, CG_ERR_STATE, "a stencil step is open");
    HIPCHK(g, hipStreamSynchronize(g->stream));
    if (!g->halo_external) for (auto& h : g->halo) { if (h) (void)hipFree(h); h = nullptr; }
    g->halo_cap = 0; g->halo_external = false;
    for (auto& h : g->halo) h = nullptr;
    if (!send_lower && !send_upper && !recv_lower && !recv_upper) return CG_OK;      // back to library-owned
    if (!send_lower || !send_upper || !recv_lower || !recv_upper || bytes_each == 0)
        return fail(g, CG_ERR_ARG, "need four buffers and a size");
    g->halo[0] = (uint8_t*)send_lower; g->halo[1] = (uint8_t*)send_upper;
    g->halo[2] = (uint8_t*)recv_lower; g->halo[3] = (uint8_t*)recv_upper;
    g->halo_cap = bytes_each; g->halo_external = true;
    return CG_OK;
}

// What a deposit has to settle before its first launch: the overflow watch and the plane it
// pushes old tops into (one for the whole step, however many launches it takes).
static int stencil_prepare_deposit(cg_grid* g) {
    Pending& p = g->pend;
    p.may_overflow = g->len_bound >= g->lmax;
    if (p.may_overflow) { int rc = refresh_len_bound(g, nullptr); if (rc) return rc; p.may_overflow = g->len_bound >= g->lmax; }
    if (p.may_overflow) HIPCHK(g, hipMemsetAsync(g->d_flags, 0, sizeof(int), g->stream));
    p.P = -1;
    if (!g->split) return CG_OK;                           // whole form: appends go to plane len
    const bool was_split = g->split;
    int rc = take_append_plane(g, &p.P); if (rc) return rc;
    if (was_split && !g->split) {
        // the compaction found a column as tall as the links reach: the step runs on the whole form, whose old
        // heights come from a copy of the tops (the halo rows packed in cg_stencil_begin hold the same heights)
        ProfScope ps(g, CG_K_OTHER);
        hipLaunchKernelGGL(k_snapshot_tops, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g));
        HIPCHK(g, hipGetLastError());
        p.may_overflow = g->len_bound >= g->lmax;
        if (p.may_overflow) HIPCHK(g, hipMemsetAsync(g->d_flags, 0, sizeof(int), g->stream));
    }
    return CG_OK;
}

// Can the open deposit wait for the vertical steps that may follow it (they then go out inside its launch, see
// flush_vertical)?  slab: the step reads rows of a neighbour.
static bool deposit_can_wait(cg_grid* g, const Pending& p, bool slab) {
    static const bool no_fold = getenv("CG_NO_FOLD") != nullptr;
    if (no_fold || !g->fuse || g->stats_on || !g->split || p.may_overflow || g->N % 4 != 0 || g->has_gaps || g->has_cavity || g->len_bound + 1 > g->lmax_vis)
        return false;
    if (!slab) {
        if (g->nx != g->nx_global || !(p.K->Re <= LEAN_FOLD_MAX_R || (p.prefold && !g->vq.empty()))) return false;
    } else {
        if (!(p.K->Re <= LEAN_FOLD_MAX_R || (p.prefold && !g->vq.empty()))) return false;
        // the rows that read a message take the steps in launches of whole workgroups; the others must exist
        size_t off[2], count[2];
        const int nr = boundary_column_ranges(g, p.K, p.R, p.has_lower, p.has_upper, off, count);
        size_t edge = 0;
        for (int b = 0; b < nr; b++) { if (count[b] % 1024 != 0 || off[b] % 4 != 0) return false; edge += count[b]; }
        if (edge >= g->N) return false;
    }
    const StencilForm form = stencil_form(g, p.K, false, p.mat, -1, false);
    return form.streams && form.lean;
}

int cg_stencil_interior(cg_grid* g) {
    NEED_H0(g);
    if (!g->pend.active) { NEED(g); return fail(g, CG_ERR_STATE, "no stencil step is open"); }
    // (a run of vertical steps left waiting for this deposit stays in its queue for now)
    if (!(g->pend.prefold && !g->lat.active && !g->vq.empty() && g->vq[0].fresh)) { g->pend.prefold = false; NEED(g); }
    if (g->pend.interior_done) return fail(g, CG_ERR_STATE, "the interior of this step has been launched already");
    if (!g->pend.has_lower && !g->pend.has_upper) return CG_OK;            // nothing is in flight: finish does it all
    bool ok; const uint8_t* mask = mask_ptr(g, g->pend.mask_id, &ok);
    if (!ok) return fail(g, CG_ERR_ARG, "mask %d vanished during the step", g->pend.mask_id);
    const bool etch = g->pend.kind == CG_STENCIL_ETCH;
    if (!etch) { int rc = stencil_prepare_deposit(g); if (rc) return rc; }
    const Pending p = g->pend;
    if (!etch && deposit_can_wait(g, p, true)) {
        // the deposit will wait for the vertical steps behind it (cg_stencil_finish): nothing goes out yet
        g->pend.interior_done = true; g->pend.held = true;
        return CG_OK;
    }
    if (g->pend.prefold) {                                 // not this time: the run that was left waiting goes out first
        g->pend.prefold = false;
        int rcf = flush_vertical(g); if (rcf) return rcf;
        rcf = materialize(g); if (rcf) return rcf;
    }
    const uint8_t* halo_lo = p.has_lower ? g->halo[2] : nullptr;           // not read by interior tiles; they tell the bands
    const uint8_t* halo_up = p.has_upper ? g->halo[3] : nullptr;
    int rc = launch_stencil(g, etch, mask, p.mat, p.t_eff, p.R, p.K, -1, halo_lo, halo_up, etch ? -1 : p.P, PART_INTERIOR, !etch && p.may_overflow);
    if (rc) return rc;
    g->pend.interior_done = true;
    return CG_OK;
}

int cg_stencil_finish(cg_grid* g) {
    NEED_H0(g);
    if (!g->pend.active) { NEED(g); return fail(g, CG_ERR_STATE, "no stencil step is open"); }
    // (a run of vertical steps left waiting for this deposit stays in its queue for now)
    const bool waiting = g->pend.prefold && !g->lat.active && !g->vq.empty() && g->vq[0].fresh;
    if (!waiting) { g->pend.prefold = false; NEED(g); }
    const bool etch = g->pend.kind == CG_STENCIL_ETCH;
    if (!etch && !g->pend.interior_done) { int rc = stencil_prepare_deposit(g); if (rc) { g->pend.active = false; return rc; } }
    Pending p = g->pend;
    g->pend.active = false;
    bool ok; const uint8_t* mask = mask_ptr(g, p.mask_id, &ok);
    if (!ok) return fail(g, CG_ERR_ARG, "mask %d vanished during the step", p.mask_id);
    const uint8_t* halo_lo = p.has_lower ? g->halo[2] : nullptr;
    const uint8_t* halo_up = p.has_upper ? g->halo[3] : nullptr;
    const int part = p.interior_done ? PART_BOUNDARY : PART_ALL;
    if (etch) {
        int rc = launch_stencil(g, true, mask, p.mat, p.t_eff, p.R, p.K, -1, halo_lo, halo_up, -1, part); if (rc) return rc;
        if (g->split) std::swap(g->top, g->top_alt);
        return CG_OK;
    }

    // A deposit that the lean kernel takes waits for the vertical steps behind it: they go out inside its launch
    // (flush_vertical).  Only when nothing of what the host tracks needs the device before that.  On a slab the rows
    // that read a neighbour's message get the steps in a pass of their own; the messages stay where they arrived.
    const bool slab = halo_lo || halo_up;
    if (p.held || (part == PART_ALL && deposit_can_wait(g, p, slab))) {
        g->lat.pre.clear();
        if (p.prefold) g->lat.pre.swap(g->vq);             // the waiting run goes with it
        g->lat.active = true; g->lat.mask = mask; g->lat.mat = p.mat; g->lat.t_eff = p.t_eff; g->lat.R = p.R; g->lat.K = p.K; g->lat.P = p.P;
        g->lat.halo_lo = halo_lo; g->lat.halo_up = halo_up;
        g->len_valid = false;
        if (g->len_bound < 255) g->len_bound += 1;
        return CG_OK;
    }
    if (p.prefold) { int rcf = flush_vertical(g); if (rcf) return rcf; rcf = materialize(g); if (rcf) return rcf; }      // not this time: the run first
    int rc = launch_stencil(g, false, mask, p.mat, p.t_eff, p.R, p.K, -1, halo_lo, halo_up, p.P, part, p.may_overflow); if (rc) return rc;
    if (p.may_overflow) {
        int fl[4]; rc = read_flags(g, fl); if (rc) return rc;
        if (fl[0]) {
            const int oldL = g->lmax;
            rc = grow_planes(g); if (rc) return rc;
            rc = launch_stencil(g, false, mask, p.mat, p.t_eff, p.R, p.K, oldL, halo_lo, halo_up, p.P); if (rc) return rc;
        }
    }
    if (g->split) std::swap(g->top, g->top_alt);
    if (g->len_bound < 255) g->len_bound += 1;
    rc = sync_vis(g); if (rc) return rc;
    // identify_cavity (cg1:543) -- a no-op while no column has a vertical gap.  On a slab the
    // driver runs it over all slabs (the decision and the flood are whole-domain matters).
    if ((g->has_gaps || g->has_cavity) && g->nx == g->nx_global) return cg_identify_cavity(g);
    return CG_OK;
}

int cg_deposit_general(cg_grid* g, int mask_id, int mat, double thickness, double conformality) {
    int R = 0;
    int rc = cg_stencil_begin(g, CG_STENCIL_DEPOSIT, mask_id, mat, thickness, conformality, 0, 0, &R);
    if (rc || !g->pend.active) return rc;
    return cg_stencil_finish(g);
}

int cg_etch_general(cg_grid* g, int mask_id, int mat, double thickness, double conformality) {
    int R = 0;
    int rc = cg_stencil_begin(g, CG_STENCIL_ETCH, mask_id, mat, thickness, conformality, 0, 0, &R);
    if (rc || !g->pend.active) return rc;
    return cg_stencil_finish(g);
}

// ---- whole-column steps, in the pieces a decomposed domain needs between its exchanges ---------
int cg_cmp_scan(cg_grid* g, int stop_mat, uint32_t* top_q, uint32_t* stopper_q, int* stopper_found) {
    NEED(g);
    if (stop_mat < 0 || stop_mat > 255) return fail(g, CG_ERR_ARG, "material id %d outside 0..255", stop_mat);
    g->pend.active = false;
    unsigned qtop = 0;
    int rc = ensure_whole(g); if (rc) return rc;
    rc = refresh_len_bound(g, &qtop); if (rc) return rc;
    unsigned st[2] = {0, 0};
    if (stop_mat != 0) {
        ProfScope ps(g, CG_K_OTHER);
        HIPCHK(g, hipMemsetAsync(g->d_red + 2, 0, 2 * sizeof(unsigned), g->stream));
        hipLaunchKernelGGL(k_cmp_stopper_top, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g), stop_mat, g->d_red + 2);
        HIPCHK(g, hipGetLastError());
        HIPCHK(g, hipMemcpyAsync(st, g->d_red + 2, sizeof st, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
    }
    if (top_q) *top_q = qtop;
    if (stopper_q) *stopper_q = st[0];
    if (stopper_found) *stopper_found = st[1] ? 1 : 0;
    return CG_OK;
}

int cg_cmp_apply(cg_grid* g, double target_z, int stop_mat) {
    NEED(g);
    if (stop_mat < 0 || stop_mat > 255) return fail(g, CG_ERR_ARG, "material id %d outside 0..255", stop_mat);
    g->pend.active = false;
    int rc = ensure_whole(g); if (rc) return rc;
    ProfScope ps(g, CG_K_OTHER);
    hipLaunchKernelGGL(k_cmp_apply, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g), target_z, stop_mat);
    HIPCHK(g, hipGetLastError());
    return CG_OK;
}

int cg_cmp(cg_grid* g, double depth, int stop_mat) {
    NEED(g);
    if (!(depth > 0)) return CG_OK;                                        // cg1:748
    if (g->nx != g->nx_global)
        return fail(g, CG_ERR_UNSUPPORTED, "cmp on a slab needs the whole domain's heights: use cg_cmp_scan / cg_cmp_apply and reduce in between");
    uint32_t qtop = 0, qstop = 0; int found = 0;
    int rc = cg_cmp_scan(g, stop_mat, &qtop, &qstop, &found); if (rc) return rc;
    const double zTop = (double)qtop * g->zscale;                          // cg1:755-781
    const double targetZ = (stop_mat == 0 || !found) ? zTop - depth : std::max(zTop - depth, (double)qstop * g->zscale);
    rc = cg_cmp_apply(g, targetZ, stop_mat); if (rc) return rc;
    return cg_identify_cavity(g);                                          // cg1:858
}

namespace {
size_t ghost_len_bytes(const cg_grid* g) { return ((size_t)g->ny + 3) & ~(size_t)3; }
Ghost make_ghost(const cg_grid* g, int planes, const void* lower, const void* upper) {
    Ghost gh;
    gh.planes = planes;
    gh.lo_len = (const uint8_t*)lower; gh.lo_rec = lower ? (const uint32_t*)((const uint8_t*)lower + ghost_len_bytes(g)) : nullptr;
    gh.up_len = (const uint8_t*)upper; gh.up_rec = upper ? (const uint32_t*)((const uint8_t*)upper + ghost_len_bytes(g)) : nullptr;
    return gh;
}
}  // namespace

int cg_ghost_bytes(const cg_grid* g, int planes, size_t* bytes) {
    if (!g || !bytes || planes < 0) return CG_ERR_ARG;
    *bytes = ghost_len_bytes(g) + (size_t)planes * g->ny * 4;
    return CG_OK;
}

int cg_ghost_pack(cg_grid* g, int planes, void* send_lower, void* send_upper) {
    NEED(g);
    if (planes < 0 || planes > 256 || !send_lower || !send_upper) return fail(g, CG_ERR_ARG, "bad ghost arguments");
    int rc = ensure_whole(g); if (rc) return rc;
    ProfScope ps(g, CG_K_HALO);
    hipLaunchKernelGGL(k_ghost_pack, dim3(blocks_for((size_t)g->ny * (planes + 1))), dim3(256), 0, g->stream, view(g), planes,
                       (uint8_t*)send_lower, (uint8_t*)send_upper);
    HIPCHK(g, hipGetLastError());
    return CG_OK;
}

// What a flood needs: cand[], and -- when node numbers fit 32 bits and the memory is there -- the union-find forest over
// the nodes (interval k of column c = number k*N + c) with, behind it, the seed blocks' lists of columns that hold an
// unmarked node and the lengths of those lists.  Without the forest the flood sweeps to its fixpoint (CG_STRIP_SWEEPS=1
// asks for that).
static int strip_prepare(cg_grid* g, int cols_per_thread = 1) {
    if (!g->cand) HIPCHK(g, hipMalloc(&g->cand, std::max<size_t>(g->N, 16)));
    g->uf_on = false; g->uf_joined = false;
    // (the kernel that fills the lists visits runs of 256 x cols_per_thread columns per block)
    const size_t per = (size_t)256 * cols_per_thread;
    const unsigned nblk = blocks_for((g->N + cols_per_thread - 1) / cols_per_thread);
    g->uf_nblk = nblk;
    const size_t nodes = (size_t)std::max(1, std::min(g->len_bound, g->lmax)) * g->N;
    const size_t segcap = ((g->N + (size_t)nblk * per - 1) / ((size_t)nblk * per)) * per;
    const size_t words = nodes + (size_t)nblk * segcap + nblk;
    const char* sw = getenv("CG_STRIP_SWEEPS");
    if (!(sw && sw[0] == '1') && nodes < 0xffffffffull) {          // (the highest 32-bit number stands for "marked")
        if (g->uf_cap < words) {
            if (g->uf) { (void)hipFree(g->uf); g->uf = nullptr; g->uf_cap = 0; }
            if (hipMalloc(&g->uf, words * sizeof(uint32_t)) == hipSuccess) g->uf_cap = words;
            else { (void)hipGetLastError(); g->uf = nullptr; }
        }
        g->uf_on = g->uf != nullptr;
    }
    g->uf_list = g->uf_on ? g->uf + nodes : nullptr;
    g->uf_counts = g->uf_on ? g->uf_list + (size_t)nblk * segcap : nullptr;
    g->uf_segcap = segcap;
    return CG_OK;
}

int cg_strip_seed(cg_grid* g, int mat, int planes, const void* ghost_lower, const void* ghost_upper) {
    NEED(g);
    if (mat < 0 || mat > 255) return fail(g, CG_ERR_ARG, "material id %d outside 0..255", mat);
    g->pend.active = false;
    int rc = ensure_whole(g); if (rc) return rc;
    { int rcp = strip_prepare(g); if (rcp) return rcp; }
    const unsigned nblk = g->uf_nblk;
    const size_t segcap = g->uf_segcap;
    ProfScope ps(g, CG_K_OTHER);
    hipLaunchKernelGGL(k_strip_seed, dim3(nblk), dim3(256), 0, g->stream, view(g), make_ghost(g, planes, ghost_lower, ghost_upper), mat, g->cand,
                       g->uf_on ? g->uf : nullptr, g->uf_list, g->uf_counts, segcap);
    HIPCHK(g, hipGetLastError());
    return CG_OK;
}

int cg_strip_sweep(cg_grid* g, int mat, int planes, const void* ghost_lower, const void* ghost_upper, int* changed) {
    NEED(g);
    if (mat < 0 || mat > 255) return fail(g, CG_ERR_ARG, "material id %d outside 0..255", mat);
    if (g->split || !g->cand) return fail(g, CG_ERR_STATE, "cg_strip_sweep without cg_strip_seed");
    ProfScope ps(g, CG_K_OTHER);
    const dim3 grid(blocks_for(g->N)), blk(256);
    const Ghost gh = make_ghost(g, planes, ghost_lower, ghost_upper);
    int any = 0;
    if (g->uf_on) {
        // the pieces of the material are joined once (and the ones that touch a seed hung under the mark); each call
        // then hangs what touches a marked node of the ghost rows under the mark too and marks every node whose root
        // the mark is: two launches the first time, one or two afterwards
        const dim3 ufgrid(g->uf_nblk);                     // (the grid of the launch that made the lists)
        if (!g->uf_joined) {
            hipLaunchKernelGGL(k_strip_union, ufgrid, blk, 0, g->stream, view(g), mat, g->uf_list, g->uf_counts, g->uf_segcap, g->uf, 0);
            g->uf_joined = true;
        }
        if (gh.lo_len || gh.up_len)
            hipLaunchKernelGGL(k_strip_ghost, dim3((2 * (unsigned)g->ny + 255) / 256), blk, 0, g->stream, view(g), gh, mat, g->cand, g->uf);
        HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
        hipLaunchKernelGGL(k_strip_spread, ufgrid, blk, 0, g->stream, view(g), mat, g->cand, g->uf_list, g->uf_counts, g->uf_segcap, g->uf, g->d_flags, (int*)nullptr);
        HIPCHK(g, hipGetLastError());
        int fl[4]; int rc = read_flags(g, fl); if (rc) return rc;
        if (changed) *changed = fl[0] ? 1 : 0;
        return CG_OK;
    }
    // Relax to the local fixpoint.  Sweeps go out four at a time, up and down the columns in turn, each with a flag word
    // of its own; the words are read back together and the flood is over when the last sweep of a batch changed nothing
    // (the closure is a fixpoint of every sweep, whichever way it runs).
    for (int round = 0;; round++) {
        if (round > 1000000) return fail(g, CG_ERR_STATE, "strip_connected did not converge");
        HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
        for (int sweep = 0; sweep < 4; sweep++)
            hipLaunchKernelGGL(k_strip_flood, grid, blk, 0, g->stream, view(g), gh, mat, g->cand, sweep & 1, g->d_flags + sweep,
                               sweep > 0 ? g->d_flags + sweep - 1 : nullptr);
        HIPCHK(g, hipGetLastError());
        int fl[4]; int rc = read_flags(g, fl); if (rc) return rc;
        if (fl[0] || fl[1] || fl[2] || fl[3]) any = 1;
        if (!fl[3]) break;
    }
    if (changed) *changed = any;
    return CG_OK;
}

int cg_strip_finish(cg_grid* g) {
    NEED(g);
    if (g->split) return fail(g, CG_ERR_STATE, "cg_strip_finish without cg_strip_seed");
    ProfScope ps(g, CG_K_OTHER);
    HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
    hipLaunchKernelGGL(k_strip_compact, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g), g->d_flags, (const uint8_t*)nullptr);
    HIPCHK(g, hipGetLastError());
    int fl[4]; int rc = read_flags(g, fl); if (rc) return rc;
    g->has_gaps = fl[2] != 0;
    g->has_cavity = fl[1] != 0;
    return CG_OK;
}

int cg_strip_connected(cg_grid* g, int mat) {
    NEED(g);
    if (g->nx != g->nx_global)
        return fail(g, CG_ERR_UNSUPPORTED, "strip_connected on a slab crosses its edges: use cg_ghost_pack / cg_strip_seed / cg_strip_sweep / cg_strip_finish");
    int rc = cg_strip_seed(g, mat, 0, nullptr, nullptr); if (rc) return rc;
    rc = cg_strip_sweep(g, mat, 0, nullptr, nullptr, nullptr); if (rc) return rc;
    return cg_strip_finish(g);
}

int cg_cavity_state(cg_grid* g, int* has_gaps, int* has_cavity, int* max_len, uint32_t* top_q) {
    NEED(g);
    unsigned q = 0;
    if (max_len || top_q) { int rc = refresh_len_bound(g, &q); if (rc) return rc; }      // the flags alone cost nothing
    if (has_gaps) *has_gaps = g->has_gaps ? 1 : 0;
    if (has_cavity) *has_cavity = g->has_cavity ? 1 : 0;
    if (max_len) *max_len = g->len_bound;
    if (top_q) *top_q = q;
    return CG_OK;
}

int cg_cavity_fill(cg_grid* g, uint32_t hmax_q) {
    NEED(g);
    g->pend.active = false;
    int rc = ensure_whole(g); if (rc) return rc;
    rc = refresh_len_bound(g, nullptr); if (rc) return rc;
    if (g->len_bound >= g->lmax) { rc = grow_planes(g); if (rc) return rc; }       // room for the piece above every column
    {
        ProfScope ps(g, CG_K_OTHER);
        hipLaunchKernelGGL(k_cavity_fill, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g), (unsigned)hmax_q);
        HIPCHK(g, hipGetLastError());
    }
    if (g->len_bound < 255) g->len_bound += 1;
    g->has_cavity = true;
    return sync_vis(g);         // only past the point where the reference has overrun its own arrays (cg2:41-66)
}

int cg_identify_cavity(cg_grid* g) {
    NEED(g);
    // With no vertical gap and no cavity layer anywhere, every inserted cavity piece is a
    // column-top piece, hence exposed, hence stripped again (script_columngrid2.js:60-67,
    // script_columngrid.js:891), and nothing else of material 2 exists that the flood could
    // reach: the grid is left exactly as it was.
    if (!g->has_gaps && !g->has_cavity) return CG_OK;
    if (g->nx != g->nx_global)
        return fail(g, CG_ERR_UNSUPPORTED, "identify_cavity on a slab needs the whole domain: use cg_cavity_state / cg_cavity_fill and the cg_strip_* calls");
    unsigned qtop = 0;
    int rc = refresh_len_bound(g, &qtop); if (rc) return rc;
    const unsigned hmax_q = qtop > 0 ? qtop : (unsigned)js_round(1.0 / g->zscale) & 0xffffu;        // maxHeight() is 1 for a flat-zero grid (cg2:103)
    // The pass without the cavity layer that the reference lays on every column and strips again (k_cavity_scan): when
    // that layer could not make the visible Lmax grow, the flood has its forest, and no column tops out at height 0.
    static const bool no_scan = getenv("CG_CAVITY_FILL") != nullptr;
    if (!no_scan && qtop > 0 && g->len_bound + 1 <= g->lmax_vis && g->N < ((size_t)1 << 32) && (g->N & 3) == 0) {
        g->pend.active = false;
        rc = ensure_whole(g); if (rc) return rc;
        rc = strip_prepare(g, 4); if (rc) return rc;           // (k_cavity_scan: four columns per thread)
        if (g->uf_on) {
            const unsigned nblk = g->uf_nblk;
            int fl[4];
            {
                ProfScope ps(g, CG_K_OTHER);
                HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
                hipLaunchKernelGGL(k_cavity_scan, dim3(nblk), dim3(256), 0, g->stream, view(g), hmax_q, g->cand, g->uf, g->uf_list, g->uf_counts, g->uf_segcap, g->d_flags);
                HIPCHK(g, hipGetLastError());
            }
            rc = read_flags(g, fl); if (rc) return rc;
            if (!fl[0]) {
                ProfScope ps(g, CG_K_OTHER);
                const dim3 grid(nblk), blk(256);
                HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
                hipLaunchKernelGGL(k_strip_union, grid, blk, 0, g->stream, view(g), CG_MAT_CAVITY, g->uf_list, g->uf_counts, g->uf_segcap, g->uf, 1);
                // (flags: [0] a node was marked by the spread, [3] a cavity layer stays unmarked, [1] / [2] the compaction's)
                hipLaunchKernelGGL(k_strip_spread, grid, blk, 0, g->stream, view(g), CG_MAT_CAVITY, g->cand, g->uf_list, g->uf_counts, g->uf_segcap, g->uf, g->d_flags, g->d_flags + 3);
                hipLaunchKernelGGL(k_strip_compact, dim3(blocks_for(g->N / 4)), blk, 0, g->stream, view(g), g->d_flags, (const uint8_t*)g->cand);
                HIPCHK(g, hipGetLastError());
                rc = read_flags(g, fl); if (rc) return rc;
                g->has_gaps = fl[2] != 0;
                g->has_cavity = fl[1] != 0 || fl[3] != 0;
                return CG_OK;
            }
            // (a column whose top lies at height 0: the gaps are cavity layers by now and the exposed ones carry their marks,
            // which is what the passes below make of them too)
        }
    }
    rc = cg_cavity_fill(g, hmax_q); if (rc) return rc;
    return cg_strip_connected(g, CG_MAT_CAVITY);                                    // cg2:79
}

// ---- etch_wet (script_columngrid2.js:893 - script_columngrid3.js:308) --------------------------
static int wet_step_launch(cg_grid* g, int mat, int dt_q, int use_diagonal, unsigned Hq, const Ghost& gh) {
    const int dt_diag = std::max(1, (int)std::floor(dt_q * 0.7071067811865476));                 // cg3:92
    ProfScope ps(g, CG_K_OTHER);
    return run_out_of_place(g, 0, "etch_wet", [&](GridView dst) {
        hipLaunchKernelGGL(k_wet_step, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g), gh, dst, mat, dt_q, dt_diag,
                           use_diagonal ? 1 : 0, Hq, g->d_flags);
    });
}

int cg_etch_wet_step(cg_grid* g, int mat, int dt_q, int use_diagonal) {
    NEED(g);
    if (g->nx != g->nx_global)
        return fail(g, CG_ERR_UNSUPPORTED, "etch_wet on a slab needs its neighbours' boundary rows and the whole domain's height: use cg_etch_wet_step_slab");
    if (mat < 0 || mat > 255) return fail(g, CG_ERR_ARG, "material id %d outside 0..255", mat);
    if (dt_q < 1) return fail(g, CG_ERR_ARG, "etch_wet increment of %d height steps", dt_q);
    g->pend.active = false;
    unsigned qtop = 0;
    int rc = ensure_whole(g); if (rc) return rc;
    rc = refresh_len_bound(g, &qtop); if (rc) return rc;
    const Ghost none = make_ghost(g, 0, nullptr, nullptr);
    rc = wet_step_launch(g, mat, dt_q, use_diagonal, std::max(qtop, 1u), none); if (rc) return rc;      // cg3:8
    return cg_identify_cavity(g);                                                                // cg3:267
}

int cg_etch_wet_step_slab(cg_grid* g, int mat, int dt_q, int use_diagonal, uint32_t hmax_q, int planes,
                          const void* ghost_lower, const void* ghost_upper) {
    NEED(g);
    if (mat < 0 || mat > 255) return fail(g, CG_ERR_ARG, "material id %d outside 0..255", mat);
    if (dt_q < 1) return fail(g, CG_ERR_ARG, "etch_wet increment of %d height steps", dt_q);
    if (planes < 0 || planes > 256) return fail(g, CG_ERR_ARG, "bad ghost plane count %d", planes);
    g->pend.active = false;
    int rc = ensure_whole(g); if (rc) return rc;
    const Ghost gh = make_ghost(g, planes, ghost_lower, ghost_upper);
    return wet_step_launch(g, mat, dt_q, use_diagonal, std::max((unsigned)hmax_q, 1u), gh);
}

int cg_etch_wet(cg_grid* g, int mat, double total_thickness, double dt) {
    NEED(g);
    const double T = std::isnan(total_thickness) ? 0 : std::max(0.0, total_thickness);           // cg2:894-895
    if (T <= 0) return CG_OK;
    double base_dt = std::isnan(dt) ? std::min(g->dx, g->dy) : dt;                               // cg2:905
    if (!(base_dt > 1e-6)) base_dt = 1e-6;
    double R = 0, t = 0;
    int rc = CG_OK;
    while (t < T - 1e-9) {                                                                       // cg3:286-306
        R += 0.7071067811865476;
        int diag = 0;
        if (R > 1) { R -= 1; diag = 1; }
        const double dt_real = std::min(base_dt, T - t);
        int dt_q = js_to_int32(js_round(dt_real / g->zscale));
        if (dt_q < 1) dt_q = 1;
        rc = cg_etch_wet_step(g, mat, dt_q, diag); if (rc) break;
        t += (double)dt_q * g->zscale;
    }
    if (!rc) rc = cg_identify_cavity(g);                                                         // cg3:307
    cg_release_scratch(g);
    return rc;
}

// ---- deposit_ALD (script_columngrid2.js:111-868) -----------------------------------------------
namespace {
// collectALDSeeds (cg2:405-463).  A slab passes the height of the whole domain, one boundary row of each
// neighbour as ghost columns (their air decides the side faces of its own boundary rows), and
// the number of neighbour rows to leave room for in the seed arrays.
int collect_seeds(cg_grid* g, double total_thickness, const uint32_t* hmax_q, int planes, const void* ghost_lower, const void* ghost_upper,
                  int halo_rows, uint64_t* count) {
    if (!(total_thickness > 0)) return fail(g, CG_ERR_ARG, "deposit_ALD thickness %g", total_thickness);
    unsigned qtop = 0;
    int rc = ensure_whole(g); if (rc) return rc;
    rc = refresh_len_bound(g, &qtop); if (rc) return rc;
    if (hmax_q) qtop = *hmax_q;
    const double H0 = qtop > 0 ? (double)qtop * g->zscale : 1.0;
    const AldLayout L = make_bit_layout(g, H0 + total_thickness + g->zscale);                    // cg2:408
    size_t lds = 0;
    const int bt = ald_block(L, 1, &lds);
    if (!bt) return fail(g, CG_ERR_UNSUPPORTED, "deposit_ALD: %d height bins per column do not fit the bitsets", L.Zmax);
    drop_seeds(g);
    const size_t ny = (size_t)g->ny;
    // air: one row below and one above the slab; seeds: halo_rows rows on each side
    const ArrLayout A{g->N + (hmax_q ? 2 * ny : 0), hmax_q ? ny : 0, ghost_lower ? 1 : 0, ghost_upper ? 1 : 0};
    g->seed_rows = halo_rows; g->seed_lo = ghost_lower != nullptr; g->seed_hi = ghost_upper != nullptr;
    g->seed_stride = g->N + 2 * (size_t)halo_rows * ny; g->seed_off = (size_t)halo_rows * ny;
    const ArrLayout S = seed_layout(g);
    GridView ghosts[2]; size_t ghost_off[2]; int nghost = 0;
    for (int side = 0; side < 2; side++) {
        const void* gp = side ? ghost_upper : ghost_lower;
        if (!gp) continue;
        GridView v = view(g);
        v.nx = 1; v.N = ny; v.lmax = planes; v.split = 0;
        v.len = (uint8_t*)const_cast<void*>(gp); v.rec = (uint32_t*)((uint8_t*)const_cast<void*>(gp) + ghost_len_bytes(g));
        ghosts[nghost] = v; ghost_off[nghost] = side ? A.off + g->N : 0; nghost++;
    }
    ProfScope ps(g, CG_K_OTHER);
    uint16_t* air_n = nullptr; uint2* air = nullptr;
    auto cleanup = [&]() { if (air_n) (void)hipFree(air_n); if (air) (void)hipFree(air); };
    auto body = [&]() -> int {
        int fl[4];
        HIPCHK(g, hipMalloc(&air_n, std::max<size_t>(2 * A.stride, 4)));
        HIPCHK(g, hipMemsetAsync(air_n, 0, std::max<size_t>(2 * A.stride, 4), g->stream));
        HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
        hipLaunchKernelGGL(k_ald_air<false>, dim3(ald_blocks(g->N, bt)), dim3(bt), lds, g->stream, view(g), L, air_n, (uint2*)nullptr, 0, A.stride, A.off, g->d_flags);
        for (int k = 0; k < nghost; k++)
            hipLaunchKernelGGL(k_ald_air<false>, dim3(ald_blocks(ny, bt)), dim3(bt), lds, g->stream, ghosts[k], L, air_n, (uint2*)nullptr, 0, A.stride, ghost_off[k], g->d_flags);
        HIPCHK(g, hipGetLastError());
        int r = read_flags(g, fl); if (r) return r;
        const int air_planes = std::max(1, fl[3]);
        if (air_planes > 65535) return fail(g, CG_ERR_UNSUPPORTED, "deposit_ALD: more than 65535 air intervals in a column");
        HIPCHK(g, hipMalloc(&air, (size_t)air_planes * A.stride * sizeof(uint2)));
        hipLaunchKernelGGL(k_ald_air<true>, dim3(ald_blocks(g->N, bt)), dim3(bt), lds, g->stream, view(g), L, air_n, air, air_planes, A.stride, A.off, g->d_flags);
        for (int k = 0; k < nghost; k++)
            hipLaunchKernelGGL(k_ald_air<true>, dim3(ald_blocks(ny, bt)), dim3(bt), lds, g->stream, ghosts[k], L, air_n, air, air_planes, A.stride, ghost_off[k], g->d_flags);
        HIPCHK(g, hipGetLastError());
        // seeds: count, then fill
        const unsigned H0q = qtop > 0 ? qtop : 0xFFFFFFFFu;                                      // |z1 - maxHeight()| < eps (cg2:440)
        HIPCHK(g, hipMalloc(&g->seed_n, std::max<size_t>(2 * S.stride, 4)));
        HIPCHK(g, hipMemsetAsync(g->seed_n, 0, std::max<size_t>(2 * S.stride, 4), g->stream));    // neighbour rows: none until they arrive
        HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
        HIPCHK(g, hipMemsetAsync(g->d_sum, 0, sizeof(unsigned long long), g->stream));
        const dim3 grid(blocks_for(g->N)), blk(256);
        hipLaunchKernelGGL(k_ald_seeds<false>, grid, blk, 0, g->stream, view(g), H0q, air_n, air, A, g->seed_n, (uint2*)nullptr, S, 0, g->d_bbox, g->d_flags, g->d_sum);
        HIPCHK(g, hipGetLastError());
        r = read_flags(g, fl); if (r) return r;
        unsigned long long total = 0;
        HIPCHK(g, hipMemcpy(&total, g->d_sum, sizeof total, hipMemcpyDeviceToHost));
        if (fl[3] > 65535) return fail(g, CG_ERR_UNSUPPORTED, "deposit_ALD: more than 65535 seeds in a column");
        g->seed_planes = std::max(1, fl[3]);
        HIPCHK(g, hipMalloc(&g->seed, (size_t)g->seed_planes * S.stride * sizeof(uint2)));
        const int bb0[4] = {INT32_MAX, -1, INT32_MAX, -1};
        HIPCHK(g, hipMemcpyAsync(g->d_bbox, bb0, sizeof bb0, hipMemcpyHostToDevice, g->stream));
        hipLaunchKernelGGL(k_ald_seeds<true>, grid, blk, 0, g->stream, view(g), H0q, air_n, air, A, g->seed_n, g->seed, S, g->seed_planes, g->d_bbox, g->d_flags, g->d_sum);
        HIPCHK(g, hipGetLastError());
        HIPCHK(g, hipMemcpyAsync(g->seed_bbox, g->d_bbox, sizeof bb0, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
        g->seed_total = total; g->has_seeds = true;
        return CG_OK;
    };
    rc = body();
    (void)hipStreamSynchronize(g->stream);
    cleanup();
    if (rc) { drop_seeds(g); return rc; }
    if (count) *count = g->seed_total;
    return CG_OK;
}

// seeds of a neighbour's rows on the wire: counts u16[rows*ny] (padded to 8 bytes), then seeds uint2[planes][rows*ny]
size_t seed_halo_count_bytes(const cg_grid* g) { return (2 * (size_t)g->seed_rows * g->ny + 7) & ~(size_t)7; }
}  // namespace

int cg_ald_collect_seeds(cg_grid* g, double total_thickness, uint64_t* count) {
    NEED(g);
    if (g->nx != g->nx_global)
        return fail(g, CG_ERR_UNSUPPORTED, "deposit_ALD on a slab needs its neighbours' boundary rows and the whole domain's height: use cg_ald_collect_seeds_slab");
    return collect_seeds(g, total_thickness, nullptr, 0, nullptr, nullptr, 0, count);
}

int cg_ald_collect_seeds_slab(cg_grid* g, double total_thickness, uint32_t hmax_q, int planes, const void* ghost_lower, const void* ghost_upper,
                              int* seed_planes, uint64_t* count) {
    NEED(g);
    if (planes < 0 || planes > 256) return fail(g, CG_ERR_ARG, "bad ghost plane count %d", planes);
    if (!(total_thickness > 0)) return fail(g, CG_ERR_ARG, "deposit_ALD thickness %g", total_thickness);
    const int rows = js_to_int32(std::ceil(total_thickness / g->dx));                            // the widest coat's reach in rows (cg2:467)
    if ((ghost_lower || ghost_upper) && rows > g->nx)
        return fail(g, CG_ERR_UNSUPPORTED, "deposit_ALD: a coat of %g nm reaches %d rows, the slab has %d", total_thickness, rows, g->nx);
    int rc = collect_seeds(g, total_thickness, &hmax_q, planes, ghost_lower, ghost_upper, rows, count); if (rc) return rc;
    if (seed_planes) *seed_planes = g->seed_planes;
    return CG_OK;
}

int cg_ald_seed_halo_bytes(const cg_grid* g, int seed_planes, size_t* bytes) {
    if (!g || !bytes || seed_planes < 1 || !g->has_seeds) return CG_ERR_ARG;
    *bytes = seed_halo_count_bytes(g) + (size_t)seed_planes * g->seed_rows * g->ny * sizeof(uint2);
    return CG_OK;
}

int cg_ald_seed_halo_pack(cg_grid* g, int seed_planes, void* send_lower, void* send_upper) {
    NEED(g);
    if (!g->has_seeds) return fail(g, CG_ERR_STATE, "deposit_ALD: no seeds held (cg_ald_collect_seeds_slab first)");
    if (seed_planes < g->seed_planes || seed_planes > 65535 || !send_lower || !send_upper)
        return fail(g, CG_ERR_ARG, "bad seed exchange arguments (%d planes asked, %d held)", seed_planes, g->seed_planes);
    if (seed_planes > g->seed_planes) {                        // a neighbour's columns hold more seeds than any of ours
        uint2* bigger = nullptr;
        HIPCHK(g, hipMalloc(&bigger, (size_t)seed_planes * g->seed_stride * sizeof(uint2)));
        HIPCHK(g, hipMemsetAsync(bigger, 0, (size_t)seed_planes * g->seed_stride * sizeof(uint2), g->stream));
        HIPCHK(g, hipMemcpyAsync(bigger, g->seed, (size_t)g->seed_planes * g->seed_stride * sizeof(uint2), hipMemcpyDeviceToDevice, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
        (void)hipFree(g->seed);
        g->seed = bigger; g->seed_planes = seed_planes;
    }
    const size_t n = (size_t)g->seed_rows * g->ny, cb = seed_halo_count_bytes(g);
    if (n == 0) return CG_OK;
    const size_t first[2] = {g->seed_off, g->seed_off + g->N - n};       // our first rows go down, our last rows go up
    void* out[2] = {send_lower, send_upper};
    for (int side = 0; side < 2; side++) {
        HIPCHK(g, hipMemcpyAsync(out[side], g->seed_n + first[side], 2 * n, hipMemcpyDeviceToDevice, g->stream));
        HIPCHK(g, hipMemcpy2DAsync((uint8_t*)out[side] + cb, n * sizeof(uint2), g->seed + first[side], g->seed_stride * sizeof(uint2),
                                   n * sizeof(uint2), (size_t)seed_planes, hipMemcpyDeviceToDevice, g->stream));
    }
    return CG_OK;
}

int cg_ald_seed_halo_unpack(cg_grid* g, int seed_planes, const void* recv_lower, const void* recv_upper) {
    NEED(g);
    if (!g->has_seeds) return fail(g, CG_ERR_STATE, "deposit_ALD: no seeds held (cg_ald_collect_seeds_slab first)");
    if (seed_planes != g->seed_planes) return fail(g, CG_ERR_ARG, "seed exchange of %d planes, %d held (pack first)", seed_planes, g->seed_planes);
    if ((recv_lower != nullptr) != g->seed_lo || (recv_upper != nullptr) != g->seed_hi)
        return fail(g, CG_ERR_ARG, "seed rows must come from the sides the ghost columns came from");
    const size_t n = (size_t)g->seed_rows * g->ny, cb = seed_halo_count_bytes(g);
    const size_t first[2] = {0, g->seed_off + g->N};                     // the rows below row 0, the rows above the last
    const void* in[2] = {recv_lower, recv_upper};
    for (int side = 0; side < 2 && n > 0; side++) {
        if (!in[side]) continue;
        HIPCHK(g, hipMemcpyAsync(g->seed_n + first[side], in[side], 2 * n, hipMemcpyDeviceToDevice, g->stream));
        HIPCHK(g, hipMemcpy2DAsync(g->seed + first[side], g->seed_stride * sizeof(uint2), (const uint8_t*)in[side] + cb, n * sizeof(uint2),
                                   n * sizeof(uint2), (size_t)seed_planes, hipMemcpyDeviceToDevice, g->stream));
    }
    // a neighbour's seed may reach any of our columns: the coat looks at every column of the slab
    g->seed_bbox[0] = 0; g->seed_bbox[1] = g->nx - 1; g->seed_bbox[2] = 0; g->seed_bbox[3] = g->ny - 1;
    return CG_OK;
}

int cg_ald_seed_info(cg_grid* g, int* held, uint64_t* count) {
    if (!g) return fail(nullptr, CG_ERR_ARG, "null handle");
    if (held) *held = g->has_seeds ? 1 : 0;
    if (count) *count = g->has_seeds ? g->seed_total : 0;
    return CG_OK;
}

int cg_ald_drop_seeds(cg_grid* g) {
    NEED_H(g);
    HIPCHK(g, hipStreamSynchronize(g->stream));
    drop_seeds(g);
    return CG_OK;
}

int cg_ald_coat(cg_grid* g, int mat, double t) {
    NEED(g);
    if (g->nx != g->nx_global) return fail(g, CG_ERR_UNSUPPORTED, "deposit_ALD on a slab needs the whole domain's height: use cg_ald_coat_slab");
    if (mat < 0 || mat > 255) return fail(g, CG_ERR_ARG, "material id %d outside 0..255", mat);
    if (!(t > 0)) return fail(g, CG_ERR_ARG, "deposit_ALD coat thickness %g", t);
    g->pend.active = false;
    return ald_coat(g, mat, t);
}

int cg_ald_coat_slab(cg_grid* g, int mat, double t, uint32_t hmax_q) {
    NEED(g);
    if (mat < 0 || mat > 255) return fail(g, CG_ERR_ARG, "material id %d outside 0..255", mat);
    if (!(t > 0)) return fail(g, CG_ERR_ARG, "deposit_ALD coat thickness %g", t);
    g->pend.active = false;
    return ald_coat(g, mat, t, &hmax_q);
}

int cg_ald_increment(cg_grid* g, int mat, double t) {
    int rc = cg_ald_coat(g, mat, t); if (rc) return rc;                 // grid = base + coat(t); spare set 0 = base  (cg2:844-848)
    // layout of the merge: makeBitLayout(maxHeight(base) + zScale) (cg2:791); base's height was the grid's before the coat
    unsigned qtop_base = 0;
    {
        GridView b = alt_view(g, 0);
        HIPCHK(g, hipMemsetAsync(g->d_red, 0, 2 * sizeof(unsigned), g->stream));
        hipLaunchKernelGGL(k_max_top, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, b, g->d_red);
        HIPCHK(g, hipGetLastError());
        unsigned r[2];
        HIPCHK(g, hipMemcpyAsync(r, g->d_red, sizeof r, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
        qtop_base = r[0];
    }
    rc = cg_identify_cavity(g); if (rc) return rc;                      // cg2:851
    return cg_ald_merge_slab(g, qtop_base);
}

// mergeNewCavityFromTmp (cg2:766-814): the grid holds coat + cavities, spare set 0 the state before the coat,
// hbase_q the height of that state over the whole domain.
int cg_ald_merge_slab(cg_grid* g, uint32_t hbase_q) {
    NEED(g);
    if (!g->alt_rec[0]) return fail(g, CG_ERR_STATE, "deposit_ALD: no state from before a coat is held");
    int rc = ensure_whole(g); if (rc) return rc;
    const unsigned qtop_base = hbase_q;
    const double Hbase = qtop_base > 0 ? (double)qtop_base * g->zscale : 1.0;
    const AldLayout L = make_bit_layout(g, Hbase + g->zscale);
    size_t lds = 0;
    const int bt = ald_block(L, 1, &lds);
    if (!bt) return fail(g, CG_ERR_UNSUPPORTED, "deposit_ALD: %d height bins per column do not fit the bitsets", L.Zmax);
    // base + (cavities of the coated grid that base does not have) -> spare set 1 -> the grid  (cg2:854)
    ProfScope ps(g, CG_K_OTHER);
    const GridView base = alt_view(g, 0);
    for (int attempt = 0; attempt < 17; attempt++) {
        rc = ensure_alt(g, 1); if (rc) return rc;
        HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
        hipLaunchKernelGGL(k_ald_merge_cavity, dim3(ald_blocks(g->N, bt)), dim3(bt), lds, g->stream, view(g), base, alt_view(g, 1), L, g->d_flags);
        HIPCHK(g, hipGetLastError());
        int fl[4]; rc = read_flags(g, fl); if (rc) return rc;
        if (fl[3] > 255 || fl[0]) return fail(g, CG_ERR_UNSUPPORTED, "deposit_ALD: a column would need more than 255 intervals or layers that overlap");
        if (fl[3] <= g->lmax) {
            swap_alt(g, 1);
            g->len_bound = fl[3]; g->has_gaps = fl[2] != 0; g->has_cavity = fl[1] != 0;
            return sync_vis(g);
        }
        rc = grow_planes(g, ((fl[3] - g->lmax + 15) / 16) * 16); if (rc) return rc;
    }
    return fail(g, CG_ERR_STATE, "deposit_ALD: plane growth did not converge");
}

int cg_deposit_ald(cg_grid* g, int mat, double total_thickness, double dt) {
    NEED(g);
    const double T = std::isnan(total_thickness) ? 0 : std::max(0.0, total_thickness);           // cg2:112-113
    if (T <= 0) return CG_OK;
    double step = std::isnan(dt) ? std::max(std::max(g->dx, g->dy), std::min(g->dx, g->dy)) : dt;   // cg2:124-127
    if (!(step > 1e-3)) step = 1e-3;
    int rc = cg_ald_collect_seeds(g, T, nullptr);
    for (double t = step; !rc && t < T - 1e-9; t += step) rc = cg_ald_increment(g, mat, t);      // cg2:843-858
    if (!rc) rc = cg_ald_coat(g, mat, T);                                                        // cg2:861-862
    if (!rc) rc = cg_identify_cavity(g);                                                         // cg2:867
    cg_release_scratch(g);
    if (g) { (void)hipStreamSynchronize(g->stream); drop_seeds(g); }
    return rc;
}

int cg_release_scratch(cg_grid* g) {
    NEED_H(g);
    HIPCHK(g, hipStreamSynchronize(g->stream));
    free_alt(g, 0); free_alt(g, 1);
    if (g->uf) { (void)hipFree(g->uf); g->uf = nullptr; g->uf_cap = 0; g->uf_on = false; }
    if (g->mask_tab) { (void)hipFree(g->mask_tab); g->mask_tab = nullptr; g->mask_tab_cap = 0; }
    if (g->mask_list) { (void)hipFree(g->mask_list); g->mask_list = nullptr; g->mask_list_cap = 0; }
    return CG_OK;
}

int cg_max_height(cg_grid* g, double* z_nm, uint32_t* top_q) {
    NEED(g);
    unsigned q = 0;
    if (g->split && !g->len_valid) {             // the tops alone say it; no need to recount the lengths
        HIPCHK(g, hipMemsetAsync(g->d_red, 0, 2 * sizeof(unsigned), g->stream));
        { ProfScope ps(g, CG_K_REDUCE);
          hipLaunchKernelGGL(k_max_top_split, dim3(blocks_for(g->N)), dim3(256), 0, g->stream, view(g), g->d_red); }
        HIPCHK(g, hipGetLastError());
        HIPCHK(g, hipMemcpyAsync(&q, g->d_red, sizeof q, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
    } else { int rc = refresh_len_bound(g, &q); if (rc) return rc; }
    if (top_q) *top_q = q;
    if (z_nm) { double z = (double)q * g->zscale; *z_nm = z > 0 ? z : 1; }      // cg2:103
    return CG_OK;
}

// ---- single-column accessors (script_columngrid.js:196-218) ---------------------------------
static int edit_column(cg_grid* g, int i, int j, int op, int mat, unsigned z0q, unsigned z1q) {
    if (i < 0 || i >= g->nx || j < 0 || j >= g->ny) return fail(g, CG_ERR_ARG, "column (%d, %d) is outside the %d x %d grid", i, j, g->nx, g->ny);
    g->pend.active = false;
    int rc = ensure_whole(g); if (rc) return rc;
    const size_t c = (size_t)i * g->ny + j;
    for (int attempt = 0; attempt < 2; attempt++) {
        HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
        hipLaunchKernelGGL(k_edit_column, dim3(1), dim3(64), 0, g->stream, view(g), c, op, mat, z0q, z1q, g->d_flags);
        HIPCHK(g, hipGetLastError());
        int fl[4]; rc = read_flags(g, fl); if (rc) return rc;
        if (fl[1]) return fail(g, CG_ERR_UNSUPPORTED, "addLayer: the layer starts below the top of column (%d, %d)", i, j);
        if (fl[2]) g->has_gaps = true;
        if (!fl[0]) break;
        if (attempt || g->lmax >= 256) return fail(g, CG_ERR_UNSUPPORTED, "addLayer: column (%d, %d) is full", i, j);
        rc = grow_planes(g); if (rc) return rc;                 // script_columngrid.js:199
    }
    if (op == 0) { g->len_bound = std::min(255, g->len_bound + 2); if (mat == CG_MAT_CAVITY) g->has_cavity = true; return sync_vis(g); }
    return CG_OK;
}

int cg_add_layer(cg_grid* g, int i, int j, int mat, double z0_nm, double z1_nm) {
    NEED(g);
    if (mat < 0 || mat > 255) return fail(g, CG_ERR_ARG, "bad material id %d", mat);
    const unsigned q0 = (unsigned)js_to_int32(js_round(z0_nm / g->zscale)) & 0xffffu;       // _quantizeZ, stored as Uint16
    const unsigned q1 = (unsigned)js_to_int32(js_round(z1_nm / g->zscale)) & 0xffffu;
    return edit_column(g, i, j, 0, mat, q0, q1);
}

int cg_remove_top_layer(cg_grid* g, int i, int j) {
    NEED(g);
    return edit_column(g, i, j, 1, 0, 0, 0);
}

// ---- host exchange --------------------------------------------------------------------------
int cg_download(cg_grid* g, uint8_t* mat, uint16_t* zpair, uint8_t* len) {
    NEED(g);
    if (!mat || !zpair || !len) return fail(g, CG_ERR_ARG, "null output array");
    { int rc = ensure_len(g); if (rc) return rc; }
    const int L = g->lmax_vis;
    const size_t chunk = std::min<size_t>(g->N, (size_t)1 << 20);
    uint8_t *dm = nullptr, *dl = nullptr; uint16_t* dz = nullptr;
    HIPCHK(g, hipMalloc(&dm, chunk * L)); HIPCHK(g, hipMalloc(&dz, chunk * L * 4)); HIPCHK(g, hipMalloc(&dl, chunk));
    for (size_t c0 = 0; c0 < g->N; c0 += chunk) {
        size_t n = std::min(chunk, g->N - c0);
        hipLaunchKernelGGL(k_gather_columns, dim3(blocks_for(n)), dim3(256), 0, g->stream, view(g), c0, (size_t)1, n, L, 0, dm, dz, dl);
        HIPCHK(g, hipGetLastError());
        HIPCHK(g, hipMemcpyAsync(mat + c0 * L, dm, n * L, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipMemcpyAsync(zpair + 2 * c0 * L, dz, n * L * 4, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipMemcpyAsync(len + c0, dl, n, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
    }
    (void)hipFree(dm); (void)hipFree(dz); (void)hipFree(dl);
    return CG_OK;
}

int cg_upload(cg_grid* g, const uint8_t* mat, const uint16_t* zpair, const uint8_t* len, int lmax) {
    NEED_H0(g);
    if (!mat || !zpair || !len || lmax < 1 || lmax > 256) return fail(g, CG_ERR_ARG, "bad upload arguments (lmax=%d)", lmax);
    { int rcf = flush_vertical(g); if (rcf) return rcf; }       // (what is queued runs, although every column is replaced)
    g->pend.active = false;
    HIPCHK(g, hipStreamSynchronize(g->stream));
    if (lmax > g->lmax) { int rc = grow_planes(g, lmax - g->lmax); if (rc) return rc; }
    g->lmax_vis = lmax;
    g->pristine = false;                                    // every column is written below
    g->split = false;                                       // the scatter kernel writes whole stacks
    const int L = lmax;
    const size_t chunk = std::min<size_t>(g->N, (size_t)1 << 20);
    uint8_t *dm = nullptr, *dl = nullptr; uint16_t* dz = nullptr;
    HIPCHK(g, hipMalloc(&dm, chunk * L)); HIPCHK(g, hipMalloc(&dz, chunk * L * 4)); HIPCHK(g, hipMalloc(&dl, chunk));
    int fl[4] = {0, 0, 0, 0};
    for (int attempt = 0; attempt < 2; attempt++) {
        HIPCHK(g, hipMemsetAsync(g->d_flags, 0, 4 * sizeof(int), g->stream));
        for (size_t c0 = 0; c0 < g->N; c0 += chunk) {
            size_t n = std::min(chunk, g->N - c0);
            HIPCHK(g, hipMemcpyAsync(dm, mat + c0 * L, n * L, hipMemcpyHostToDevice, g->stream));
            HIPCHK(g, hipMemcpyAsync(dz, zpair + 2 * c0 * L, n * L * 4, hipMemcpyHostToDevice, g->stream));
            HIPCHK(g, hipMemcpyAsync(dl, len + c0, n, hipMemcpyHostToDevice, g->stream));
            hipLaunchKernelGGL(k_scatter_columns, dim3(blocks_for(n)), dim3(256), 0, g->stream, view(g), c0, n, L, dm, dz, dl, g->d_flags);
            HIPCHK(g, hipGetLastError());
            HIPCHK(g, hipStreamSynchronize(g->stream));
        }
        int rc = read_flags(g, fl); if (rc) return rc;
        if (fl[3] <= g->lmax) break;
        // gaps need intervals of their own: provision the planes they take and convert again
        rc = grow_planes(g, ((fl[3] - g->lmax + 15) / 16) * 16); if (rc) return rc;
    }
    (void)hipFree(dm); (void)hipFree(dz); (void)hipFree(dl);
    if (fl[1]) return fail(g, CG_ERR_UNSUPPORTED, "upload: a layer starts below the top of the layer under it");
    if (fl[3] > g->lmax) return fail(g, CG_ERR_UNSUPPORTED, "upload: a column needs %d intervals (layers + gaps), more than 256", fl[3]);
    g->has_gaps = fl[2] != 0;
    g->has_cavity = true;          // not inspected: assume the upload may contain cavity layers
    g->len_bound = 255;            // refreshed lazily when it matters
    return CG_OK;
}

int cg_snapshot_save(cg_grid* g, int slot) {
    NEED(g);
    if (slot < 0) return fail(g, CG_ERR_ARG, "bad snapshot slot %d", slot);
    cg_snapshot_drop(g, slot);
    // only the planes in use are copied: split form, the ones handed out so far; whole form, max(len)
    if (!g->split) { int rc = refresh_len_bound(g, nullptr); if (rc) return rc; }
    Snapshot s;
    s.planes = std::min(g->lmax, g->split ? g->nextp : g->len_bound);
    const size_t S = g->N * (size_t)s.planes;
    HIPCHK(g, hipMalloc(&s.len, std::max<size_t>(g->N, 4))); HIPCHK(g, hipMalloc(&s.rec, std::max<size_t>(4 * S, 4)));
    HIPCHK(g, hipMalloc(&s.top, std::max<size_t>(4 * g->N, 16)));
    s.nextp = g->nextp;
    HIPCHK(g, hipMemcpyAsync(s.len, g->len, g->N, hipMemcpyDeviceToDevice, g->stream));
    if (S) HIPCHK(g, hipMemcpyAsync(s.rec, g->rec, 4 * S, hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(g, hipMemcpyAsync(s.top, g->top, 4 * g->N, hipMemcpyDeviceToDevice, g->stream));
    s.split = g->split;
    s.len_valid = g->len_valid; s.lmax = g->lmax; s.lmax_vis = g->lmax_vis; s.len_bound = g->len_bound; s.has_gaps = g->has_gaps; s.has_cavity = g->has_cavity;
    g->snaps[slot] = s;
    return CG_OK;
}

int cg_snapshot_load(cg_grid* g, int slot) {
    NEED_H0(g);
    auto it = g->snaps.find(slot);
    if (it == g->snaps.end()) return fail(g, CG_ERR_ARG, "no snapshot in slot %d", slot);
    { int rcf = flush_vertical(g); if (rcf) return rcf; }       // (what is queued runs, although the grid is replaced)
    const Snapshot& s = it->second;
    g->pend.active = false;
    if (s.planes > g->lmax) { int rc = grow_planes(g, (s.planes - g->lmax + 15) / 16 * 16); if (rc) return rc; }
    const size_t S = g->N * (size_t)s.planes;
    HIPCHK(g, hipMemcpyAsync(g->len, s.len, g->N, hipMemcpyDeviceToDevice, g->stream));
    if (S) HIPCHK(g, hipMemcpyAsync(g->rec, s.rec, 4 * S, hipMemcpyDeviceToDevice, g->stream));
    HIPCHK(g, hipMemcpyAsync(g->top, s.top, 4 * g->N, hipMemcpyDeviceToDevice, g->stream));
    g->nextp = s.nextp;
    g->split = s.split;
    g->len_bound = s.len_bound; g->has_gaps = s.has_gaps; g->has_cavity = s.has_cavity;
    g->lmax_vis = s.lmax_vis; g->len_valid = s.len_valid; g->pristine = false;
    return CG_OK;
}

int cg_snapshot_drop(cg_grid* g, int slot) {
    NEED_H(g);
    auto it = g->snaps.find(slot);
    if (it == g->snaps.end()) return CG_OK;
    HIPCHK(g, hipStreamSynchronize(g->stream));
    (void)hipFree(it->second.len); (void)hipFree(it->second.rec); (void)hipFree(it->second.top);
    g->snaps.erase(it);
    return CG_OK;
}

int cg_slices_device(cg_grid* g, int axis, const int32_t* index, int count, void** d_mat, void** d_zpair, void** d_len) {
    NEED(g);
    if ((axis != 0 && axis != 1) || !index || count < 1) return fail(g, CG_ERR_ARG, "bad slice request");
    const size_t line = axis == 0 ? (size_t)g->ny : (size_t)g->nx;
    const int L = g->lmax_vis;
    const size_t need = line * count;
    for (int s = 0; s < count; s++) {
        const int lim = axis == 0 ? g->nx : g->ny;
        if (index[s] < 0 || index[s] >= lim) return fail(g, CG_ERR_ARG, "slice index %d outside [0,%d)", index[s], lim);
    }
    // two capacities: slots (columns x L) for mat / zpair, columns for len -- L can shrink (reset, snapshot
    // of a lower Lmax, upload) while the number of columns grows
    if (need * L > g->sl_cap || need > g->sl_cols_cap) {
        HIPCHK(g, hipStreamSynchronize(g->stream));
        if (g->sl_mat) (void)hipFree(g->sl_mat);
        if (g->sl_zp) (void)hipFree(g->sl_zp);
        if (g->sl_len) (void)hipFree(g->sl_len);
        const size_t slots = std::max(need * L, g->sl_cap), cols = std::max(need, g->sl_cols_cap);
        g->sl_mat = nullptr; g->sl_zp = nullptr; g->sl_len = nullptr; g->sl_cap = 0; g->sl_cols_cap = 0;
        HIPCHK(g, hipMalloc(&g->sl_mat, slots)); HIPCHK(g, hipMalloc(&g->sl_zp, slots * 4)); HIPCHK(g, hipMalloc(&g->sl_len, cols));
        g->sl_cap = slots; g->sl_cols_cap = cols;
    }
    if (count > g->sl_idx_cap) {
        if (g->sl_idx) (void)hipFree(g->sl_idx);
        g->sl_idx = nullptr; g->sl_idx_cap = 0;
        HIPCHK(g, hipMalloc(&g->sl_idx, (size_t)count * sizeof(int32_t)));
        g->sl_idx_cap = count; g->sl_idx_host.clear();
    }
    // the same lines are usually asked for step after step: their indices are uploaded once (the
    // copy from pageable memory would stall the caller until the stream has drained)
    if (g->sl_idx_host.size() != (size_t)count || memcmp(g->sl_idx_host.data(), index, (size_t)count * sizeof(int32_t)) != 0) {
        HIPCHK(g, hipMemcpyAsync(g->sl_idx, index, (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
        g->sl_idx_host.assign(index, index + count);
    }
    hipLaunchKernelGGL(k_gather_slices, dim3(blocks_for(need)), dim3(256), 0, g->stream, view(g), axis, g->sl_idx, count, line, L,
                       (g->split && !g->len_valid) ? 1 : 0, g->sl_mat, g->sl_zp, g->sl_len);
    HIPCHK(g, hipGetLastError());
    if (d_mat) *d_mat = g->sl_mat;
    if (d_zpair) *d_zpair = g->sl_zp;
    if (d_len) *d_len = g->sl_len;
    return CG_OK;
}

// ---- cross-sections without any allocation --------------------------------------------------------
// The index list of cg_slices_noalloc is staged in this array instead of a buffer the library
// allocates.  Every GPU holds its own copy at its own address until the process ends; all handles on
// one GPU share that copy.  In host code the name `scratch` is only a stand-in for the symbol: the
// device address is looked up once per device (scratch_resolve, at handle creation) and that pointer is what copies
// and kernels get.
static __device__ __align__(256) int32_t scratch[CG_SLICES_NOALLOC_MAX];

// Looks up the device address of `scratch` on the current device (once per device and process, under a lock: the
// array is shared by every handle and thread on that GPU).  Called from cg_create_slab, so that cg_slices_noalloc
// itself only reads the table.
static int scratch_resolve(cg_grid* g) {
    if (g->device < 0 || g->device >= MAX_DEVICES) return CG_OK;          // cg_slices_noalloc refuses such a device
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[g->device]) return CG_OK;
    int32_t* p = nullptr;
    HIPCHK(g, hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(g, CG_ERR_HIP, "hipGetSymbolAddress gave a null address");
    g_scratch_ptr[g->device] = p;
    return CG_OK;
}

int cg_slices_noalloc(cg_grid* g, int axis, const int32_t* index, int count, void* d_mat, void* d_zpair, void* d_len) {
    NEED(g);
    if ((axis != 0 && axis != 1) || !index || count < 1 || !d_mat || !d_zpair || !d_len) return fail(g, CG_ERR_ARG, "bad slice request");
    if (count > CG_SLICES_NOALLOC_MAX)
        return fail(g, CG_ERR_ARG, "cg_slices_noalloc takes at most %d lines per call, %d asked (split the request or use cg_slices_device)",
                    CG_SLICES_NOALLOC_MAX, count);
    const size_t line = axis == 0 ? (size_t)g->ny : (size_t)g->nx;
    const int lim = axis == 0 ? g->nx : g->ny;
    for (int s = 0; s < count; s++)
        if (index[s] < 0 || index[s] >= lim) return fail(g, CG_ERR_ARG, "slice index %d outside [0,%d)", index[s], lim);
    if (g->device < 0 || g->device >= MAX_DEVICES) return fail(g, CG_ERR_UNSUPPORTED, "device ordinal %d beyond %d", g->device, MAX_DEVICES);
    int32_t* const d_idx = g_scratch_ptr[g->device];                       // resolved when the handle was created
    if (!d_idx) return fail(g, CG_ERR_STATE, "the index array of this device was never resolved");
    // only the first `count` entries are rewritten; the kernel reads no others
    HIPCHK(g, hipMemcpyAsync(d_idx, index, (size_t)count * sizeof(int32_t), hipMemcpyHostToDevice, g->stream));
    hipLaunchKernelGGL(k_gather_slices, dim3(blocks_for(line * count)), dim3(256), 0, g->stream, view(g), axis, d_idx, count, line, g->lmax_vis,
                       (g->split && !g->len_valid) ? 1 : 0, (uint8_t*)d_mat, (uint16_t*)d_zpair, (uint8_t*)d_len);
    HIPCHK(g, hipGetLastError());
    return CG_OK;
}

int cg_slices(cg_grid* g, int axis, const int32_t* index, int count, uint8_t* mat, uint16_t* zpair, uint8_t* len) {
    if (!mat || !zpair || !len) return fail(g, CG_ERR_ARG, "null output array");
    int rc = cg_slices_device(g, axis, index, count, nullptr, nullptr, nullptr); if (rc) return rc;
    const size_t line = axis == 0 ? (size_t)g->ny : (size_t)g->nx;
    const size_t n = line * count; const int L = g->lmax_vis;
    HIPCHK(g, hipMemcpyAsync(mat, g->sl_mat, n * L, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipMemcpyAsync(zpair, g->sl_zp, n * L * 4, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipMemcpyAsync(len, g->sl_len, n, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    return CG_OK;
}

int cg_index_from_pos(const cg_grid* g, int axis, double pos, int32_t* index) {
    if (!g || !index || (axis != 0 && axis != 1)) return CG_ERR_ARG;
    const int n = axis == 0 ? g->nx_global : g->ny;
    const double pitch = axis == 0 ? g->dx : g->dy;
    const double off = -(n - 1) * pitch / 2;
    double r = js_round((pos - off) / pitch);
    if (r < 0) r = 0;
    if (r > n - 1) r = n - 1;
    if (std::isnan(r)) r = 0;
    *index = (int32_t)r;
    return CG_OK;
}

// ---- checks and measurement -----------------------------------------------------------------
int cg_checksum(cg_grid* g, uint64_t* sum) {
    NEED(g);
    if (!sum) return fail(g, CG_ERR_ARG, "null output");
    if (!g->split) { int rc = ensure_len(g); if (rc) return rc; }       // (the split form's chains say where a column ends)
    HIPCHK(g, hipMemsetAsync(g->d_sum, 0, sizeof(unsigned long long), g->stream));
    hipLaunchKernelGGL(k_checksum, dim3(blocks_for((g->N + CHECKSUM_U - 1) / CHECKSUM_U)), dim3(256), 0, g->stream, view(g), g->d_sum);
    HIPCHK(g, hipGetLastError());
    unsigned long long h = 0;
    HIPCHK(g, hipMemcpyAsync(&h, g->d_sum, sizeof h, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    *sum = h;
    return CG_OK;
}

int cg_step_stats(cg_grid* g, int enable, uint64_t stats[CG_NSTAT]) {
    NEED_H(g);
    if (stats) {
        unsigned long long h[CG_NSTAT];
        HIPCHK(g, hipMemcpyAsync(h, g->d_stats, sizeof h, hipMemcpyDeviceToHost, g->stream));
        HIPCHK(g, hipStreamSynchronize(g->stream));
        for (int q = 0; q < CG_NSTAT; q++) stats[q] = h[q];
        stats[0] = g->N;
    }
    g->stats_on = enable != 0;
    return CG_OK;
}

int cg_profile(cg_grid* g, int enable, double ms[CG_NKIND], uint64_t count[CG_NKIND]) {
    NEED_H(g);
    HIPCHK(g, hipStreamSynchronize(g->stream));
    for (auto& r : g->prof_recs) {
        float v = 0;
        if (hipEventElapsedTime(&v, r.a, r.b) == hipSuccess && r.kind >= 0 && r.kind < CG_NKIND) { g->prof_ms[r.kind] += v; g->prof_cnt[r.kind] += 1; }
        g->prof_pool.push_back(r.a); g->prof_pool.push_back(r.b);
    }
    g->prof_recs.clear();
    for (int k = 0; k < CG_NKIND; k++) {
        if (ms) ms[k] = g->prof_ms[k];
        if (count) count[k] = g->prof_cnt[k];
        g->prof_ms[k] = 0; g->prof_cnt[k] = 0;
    }
    g->prof_on = enable != 0;
    return CG_OK;
}

int cg_timer_start(cg_grid* g) { NEED_H(g); HIPCHK(g, hipEventRecord(g->ev0, g->stream)); return CG_OK; }

int cg_timer_stop(cg_grid* g, float* ms) {
    NEED_H(g);
    HIPCHK(g, hipEventRecord(g->ev1, g->stream));
    HIPCHK(g, hipEventSynchronize(g->ev1));
    float v = 0; HIPCHK(g, hipEventElapsedTime(&v, g->ev0, g->ev1));
    if (ms) *ms = v;
    return CG_OK;
}

}  // extern "C"
