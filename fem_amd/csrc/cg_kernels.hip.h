// Device code of libcolgrid: the per-column process kernels for gfx950 (MI355X).
//
// A column is a stack of contiguous intervals z1q | mat << 16 | flags << 24 (u32, see REC_*); interval k
// spans from the top of interval k-1 (0 for k = 0) to its own z1, so the bottom height is never
// stored in a record.  That is exactly what the reference's columns look like as long as they
// have no vertical gaps (every append starts at the old top, script_columngrid.js:419-423).  A
// gap -- left behind when strip_connected removes a buried layer -- is kept as an interval of
// its own with the HOLE flag; holes are not layers of the reference (they do not count in its
// len, cg_download skips them) and the top interval is never a hole.
//
// Two forms of the same columns (all arrays in HBM, owned by the handle; c = i*NY + j, j fastest):
//
//   whole form   len[c] u8, rec[k*N + c] = interval k, k < len      (plane-major)
//                what the kernels that restructure whole columns work on (strip, cavity, CMP,
//                wet etch, ALD, upload).
//
//   split form   top[c] u32 = the top interval (0: empty column); the six high bits of every
//                record name the plane that holds the interval below it (whose z1 is this one's
//                bottom; LNK_NONE if there is none), so a column is the chain top[c] ->
//                rec[lnk*N + c] -> ...  and an append moves 12 bytes: the old top is read, stored
//                as it is into the plane P of the launch, and the new top (link = P) is stored.
//                len[c] is NOT kept up to date by the 4-columns-per-thread kernels -- the chain says
//                everything they need -- and is recounted from the chain (k_relen) before anything
//                reads it.
//                What the vertical and lateral steps work on: they only ever look at the top, so
//                extend / partial etch / "material differs" touch one stream whose addresses do not
//                depend on the column length.  Every depositing launch takes a fresh plane, so that
//                its pushes fill whole 128-byte lines whatever the lengths of neighbouring columns
//                (measured: 4-byte pieces scattered over planes cost up to 3x, the memory side
//                read-modify-writes partial lines), and the removal of a whole layer follows the
//                links.  Six bits: planes 0..62, i.e. columns of up to 64 intervals; beyond that the
//                host keeps the grid in whole form.  Plane slots outside a column's chain hold
//                unspecified data.
//
// A thread owns 4 consecutive columns: HBM is only reached with >= 4 bytes per lane and access
// (measured on MI355X: 1 B/lane copies at 1.8 TB/s, 2 B/lane 3.0, 4 B/lane 5.4, 16 B/lane 5.7).
//
// Arithmetic follows the reference's JavaScript: heights are u16 steps of zscale nm,
// de-quantised as (double)q*zscale, re-quantised as Math.round(z/zscale) (halves toward
// +infinity) and stored modulo 2^16 (script_columngrid.js:117-122, 85).  This file must be
// compiled with -ffp-contract=off: the reference rounds after every operation.
//
// This header is included by several translation units (see cg_launch.h): kernels that are not templates have
// internal linkage (`static __global__`), so that every unit keeps only the ones it launches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CG_EPS 1e-9

namespace cgk {

struct GridView {
    uint8_t*  len;
    uint32_t* rec;
    uint32_t* top;      // split form: the top interval of every column (0 for an empty column)
    uint32_t* top_w;    // where updated tops are stored: top itself, or the other buffer in a lateral step
    int       P;        // split form: the plane this launch pushes old tops into (-1: it cannot append)
    int       split;    // 1: split form (see below); 0: whole form, planes hold intervals 0..len-1 by index
    size_t    N;        // columns of this handle
    int       nx, ny;
    int       lmax;     // planes provisioned
    int       i0;       // global row of local row 0
    int       tile_i0;  // lateral steps: first tile row of this launch (a launch may cover a band of tile rows)
    int       tile_in;  // lateral steps: tile rows of this launch
    int       sup_i, sup_j;   // lateral steps: tiles per super-tile (see tile_of_block)
    int       ti;       // lateral steps, streaming form: rows per tile
    double    zs;       // nm per step
    double    inv_zs;   // 1/zs, rounded once on the host
};

constexpr uint32_t REC_HOLE = 1u << 24;     // a vertical gap, not a layer
constexpr uint32_t REC_LAYER = 2u << 24;    // set in every layer record: a record is 0 only where there is none
constexpr uint32_t REC_MARK = 4u << 24;     // whole form only: scratch bit of the flood fill (strip_connected)
constexpr int LNK_SHIFT = 26;               // split form only: bits 26..31 = plane of the interval below
constexpr int LNK_NONE = 63;                // no interval below (so planes 0..62 can be linked)
constexpr uint32_t REC_BODY = (1u << LNK_SHIFT) - 1u;      // everything but the link
// whole form: a layer record (bits 26..31 clear)
__device__ __forceinline__ uint32_t rec_make(uint32_t z1q, int mat) { return (z1q & 0xffffu) | ((uint32_t)(mat & 255) << 16) | REC_LAYER; }
// split form: a layer record above the interval stored in plane lnk
__device__ __forceinline__ uint32_t rec_make_l(uint32_t z1q, int mat, int lnk) { return rec_make(z1q, mat) | ((uint32_t)lnk << LNK_SHIFT); }
__device__ __forceinline__ int rec_lnk(uint32_t r) { return (int)(r >> LNK_SHIFT); }
__device__ __forceinline__ uint32_t rec_z1(uint32_t r) { return r & 0xffffu; }
__device__ __forceinline__ int rec_mat(uint32_t r) { return (int)((r >> 16) & 255u); }
__device__ __forceinline__ bool rec_hole(uint32_t r) { return (r & REC_HOLE) != 0; }
__device__ __forceinline__ uint32_t rec_with_z1(uint32_t r, uint32_t z1q) { return (r & 0xffff0000u) | (z1q & 0xffffu); }    // link and flags stay

// statistics slots (cg_step_stats)
enum { ST_VISITED = 0, ST_OPEN = 1, ST_EXTEND = 2, ST_APPEND = 3, ST_ETCHED = 4, ST_REMOVED = 5, ST_FLOOR = 6, ST_N = 8 };

__device__ __forceinline__ double js_round(double x) {
    double r = floor(x);
    if (x - r >= 0.5) r += 1.0;
    return r;
}
// store of a Math.round result into a Uint16Array element
__device__ __forceinline__ uint16_t to_u16(double v) {
    if (!(fabs(v) < 9.0e18)) return 0;          // NaN / inf / absurd
    return (uint16_t)(long long)v;
}
// Math.round(z / zs) stored into a Uint16Array element -- the literal form: IEEE division,
// round half toward +infinity, conversion through 64-bit integers.  About 100 double-rate
// instructions; kept for the rare inputs the shortcut below cannot decide.
__device__ __noinline__ uint16_t quantz_exact(double z, double zs) { return to_u16(js_round(z / zs)); }

// The same value without the division.  y = z * (1/zs) differs from the correctly rounded
// quotient by at most 3.3e-16 * |y| (two roundings), i.e. < 4e-10 for |y| < 1e6.  Math.round
// of the quotient is the integer nearest to it unless the quotient sits within that distance
// of a half-integer; so when y is farther than 1e-8 from every half-integer, rint(y) is the
// answer, and otherwise (t = 7.25 nm at 0.1 nm steps, say) the literal form decides.
__device__ __forceinline__ uint16_t quantz(double z, double zs, double inv_zs) {
    const double y = z * inv_zs;
    const double r = rint(y);
    if (fabs(y - r) < 0.49999999 && fabs(y) < 1.0e6) return (uint16_t)(int)r;
    return quantz_exact(z, zs);
}
// Note: Math.round((q*zs)/zs) == q for every q < 2^16 (the two roundings move the quotient by
// < 2e-11), so the bottom of an appended layer, quantize(topZ) at script_columngrid.js:422, is
// the old top's z1q itself -- which is why the bottom need not be stored.

// ------------------------------------------------------------------------------------------
// per-column update primitives (scalar forms; the vector forms below do the same per lane)
// ------------------------------------------------------------------------------------------
// All of them work on the split form of a column (GridView::split): the top interval in
// top[c], intervals 0 .. len-2 in the planes.

// All l intervals of a column bottom-up into out[], whichever form the grid is in (read-only kernels).
__device__ __forceinline__ void load_column(const GridView& g, size_t c, int l, uint32_t* out) {
    if (!g.split) { for (int k = 0; k < l; k++) out[k] = g.rec[(size_t)k * g.N + c]; return; }
    if (l <= 0) return;
    uint32_t r = g.top[c];
    out[l - 1] = r & REC_BODY;                                 // links are not part of what a reader sees
    for (int k = l - 2; k >= 0; k--) {
        r = g.rec[(size_t)rec_lnk(r) * g.N + c];
        out[k] = r & REC_BODY;
    }
}

// split form: intervals in the chain of column c
__device__ __forceinline__ int chain_length(const GridView& g, size_t c) {
    uint32_t r = g.top[c];
    if (r == 0) return 0;
    int l = 1;
    for (int p = rec_lnk(r); p != LNK_NONE && l < 255; p = rec_lnk(r)) { r = g.rec[(size_t)p * g.N + c]; l++; }
    return l;
}

// The column has just lost its top; l intervals remain (l >= 1) and lk is the plane of the
// topmost of them.  Loads the new top (its link comes with it), drops it too if it is a gap (a gap
// is never on top), and looks up the new bottom.
__device__ __forceinline__ void reload_top(const GridView& g, size_t c, int& l, uint32_t& r, unsigned& z0q) {
    r = g.rec[(size_t)rec_lnk(r) * g.N + c];
    if (rec_hole(r)) {
        l -= 1;
        if (l == 0 || rec_lnk(r) == LNK_NONE) { l = 0; r = 0; z0q = 0; return; }
        r = g.rec[(size_t)rec_lnk(r) * g.N + c];
    }
    const int p2 = rec_lnk(r);
    z0q = p2 != LNK_NONE ? rec_z1(g.rec[(size_t)p2 * g.N + c]) : 0u;
}

// The one-column primitives below also take a grid in whole form (interval k in plane k, the top
// in plane len-1): the form the host keeps columns of more than LNK_NONE + 1 intervals in, which
// the links of the split form cannot chain.  Slower (the planes an append or a cut touches depend
// on the column length), and only reached by such stacks.

// top record of column c of length l, either form
__device__ __forceinline__ uint32_t top_record(const GridView& g, size_t c, int l) {
    if (g.split) return g.top[c];
    return l > 0 ? g.rec[(size_t)(l - 1) * g.N + c] : 0u;
}

// whole form: _depoAtTop_
__device__ __forceinline__ int depo_at_top_whole(const GridView& g, size_t c, int l, int matId, double dz) {
    const uint32_t r = l > 0 ? g.rec[(size_t)(l - 1) * g.N + c] : 0;
    const uint16_t q = quantz((double)rec_z1(r) * g.zs + dz, g.zs, g.inv_zs);
    if (l > 0 && rec_mat(r) == matId) { g.rec[(size_t)(l - 1) * g.N + c] = rec_with_z1(r, q); return 1; }
    if (l >= g.lmax) return 3;
    g.rec[(size_t)l * g.N + c] = rec_make(q, matId);
    g.len[c] = (uint8_t)(l + 1);
    return 2;
}

// whole form: _applyEtchAt; removed slots are cleared, a gap uncovered by a removal goes with it
__device__ __forceinline__ int apply_etch_at_whole(const GridView& g, size_t c, int l, int matId, double dz, bool isAll) {
    double remaining = dz;
    int removed = 0; bool touched = false;
    const int l0 = l;
    while (remaining > CG_EPS && l > 0) {
        const size_t at = (size_t)(l - 1) * g.N + c;
        const uint32_t r = g.rec[at];
        if (!isAll && rec_mat(r) != matId) break;
        const unsigned z0q = l > 1 ? rec_z1(g.rec[(size_t)(l - 2) * g.N + c]) : 0u;
        const double z1 = (double)rec_z1(r) * g.zs, z0 = (double)z0q * g.zs, thick = z1 - z0;
        touched = true;
        if (remaining < thick - CG_EPS) { g.rec[at] = rec_with_z1(r, quantz(z1 - remaining, g.zs, g.inv_zs)); break; }
        remaining -= thick; removed++;
        g.rec[at] = 0; l--;
        if (l > 0 && rec_hole(g.rec[(size_t)(l - 1) * g.N + c])) { g.rec[(size_t)(l - 1) * g.N + c] = 0; l--; }
    }
    if (l != l0) g.len[c] = (uint8_t)l;
    return (touched ? 0x10000 : 0) | removed;
}

// whole form: bottom of the top run of matId (the top layer, interval l-1, is known to be matId)
__device__ __forceinline__ double floor_of_top_run_whole(const GridView& g, size_t c, int l, int matId, unsigned& scanned) {
    scanned++;
    int k = l - 2;                 // topmost interval not known to belong to the run
    while (k >= 0) {
        const uint32_t b = g.rec[(size_t)k * g.N + c];
        int kl = k;                // the layer to compare with: interval k, or what lies under the gap k
        if (rec_hole(b)) { kl = k - 1; if (kl < 0) break; }
        if (rec_mat(g.rec[(size_t)kl * g.N + c]) != matId) break;
        scanned++;
        k = kl - 1;
    }
    return k >= 0 ? (double)rec_z1(g.rec[(size_t)k * g.N + c]) * g.zs : 0.0;
}

// _depoAtTop_ (script_columngrid.js:397-426) for one column.  `l` is the column's current
// length.  Returns 0 = nothing, 1 = top layer extended, 2 = layer appended, 3 = append needed
// but no plane left (column untouched; the host grows the planes and re-runs those columns).
__device__ __forceinline__ int depo_at_top(const GridView& g, size_t c, int l, int matId, double dz) {
    if (!(dz > 0)) return 0;
    if (!g.split) return depo_at_top_whole(g, c, l, matId, dz);
    const uint32_t r = l > 0 ? g.top[c] : 0;
    const uint32_t topq = rec_z1(r);
    const uint16_t q = quantz((double)topq * g.zs + dz, g.zs, g.inv_zs);
    if (l > 0 && rec_mat(r) == matId) { g.top_w[c] = rec_with_z1(r, q); return 1; }
    if (l >= g.lmax) return 3;
    if (l > 0) g.rec[(size_t)g.P * g.N + c] = r;               // the old top goes into this step's plane, link and all
    g.top_w[c] = rec_make_l(q, matId, l > 0 ? g.P : LNK_NONE);
    g.len[c] = (uint8_t)(l + 1);
    return 2;
}

// One round of _applyEtchAt's loop (script_columngrid.js:565-588) on a column held in
// registers: top record r, its bottom z0q, length l, thickness still to remove.
// Returns 0 = the loop is over, 1 = a whole layer went and the new top must be loaded.
__device__ __forceinline__ int etch_round(const GridView& g, uint32_t& r, unsigned z0q, int& l, double& remaining, int matId, bool isAll,
                                          bool& touched, int& removed) {
    if (!(remaining > CG_EPS) || l <= 0) return 0;
    if (!isAll && rec_mat(r) != matId) return 0;
    const double z1 = (double)rec_z1(r) * g.zs, z0 = (double)z0q * g.zs, thick = z1 - z0;
    touched = true;
    if (remaining < thick - CG_EPS) { r = rec_with_z1(r, quantz(z1 - remaining, g.zs, g.inv_zs)); remaining = 0; return 0; }
    remaining -= thick; l -= 1; removed++;
    return 1;
}

// _applyEtchAt (script_columngrid.js:553-589).  Returns number of whole layers removed in the
// low 16 bits and 0x10000 if the column was touched at all.
__device__ __forceinline__ int apply_etch_at(const GridView& g, size_t c, int l, int matId, double dz, bool isAll) {
    if (!(dz > 0) || l <= 0) return 0;
    if (!g.split) return apply_etch_at_whole(g, c, l, matId, dz, isAll);
    double remaining = dz;
    int removed = 0; bool touched = false;
    const int l0 = l;
    uint32_t r = g.top[c];
    unsigned z0q = rec_lnk(r) != LNK_NONE ? rec_z1(g.rec[(size_t)rec_lnk(r) * g.N + c]) : 0u;
    while (etch_round(g, r, z0q, l, remaining, matId, isAll, touched, removed)) {
        if (l > 0 && rec_lnk(r) != LNK_NONE) reload_top(g, c, l, r, z0q); else { l = 0; r = 0; z0q = 0; }
    }
    if (touched) {
        g.top_w[c] = r;
        if (l != l0) g.len[c] = (uint8_t)l;
    }
    return (touched ? 0x10000 : 0) | removed;
}

// bottom of the top run of matId (script_columngrid.js:646-660); the top layer (record r) is known
// to be matId and b is the interval below it (valid when r has a link).  Gaps do not interrupt the
// run (the reference only compares materials).  The usual case -- the layer below is another
// material -- is decided by b, which the caller needs anyway for the cut.
__device__ __forceinline__ double floor_of_top_run(const GridView& g, size_t c, int matId, uint32_t r, uint32_t b, unsigned& scanned) {
    scanned++;
    int p = rec_lnk(r);            // plane of the topmost interval not known to belong to the run
    if (p == LNK_NONE) return 0.0;
    uint32_t cur = b;              // the interval in plane p
    for (;;) {
        uint32_t lay = cur;        // the layer to compare with: cur, or what lies under the gap cur
        if (rec_hole(cur)) {
            if (rec_lnk(cur) == LNK_NONE) break;
            lay = g.rec[(size_t)rec_lnk(cur) * g.N + c];
        }
        if (rec_mat(lay) != matId) break;
        scanned++;
        p = rec_lnk(lay);
        if (p == LNK_NONE) return 0.0;
        cur = g.rec[(size_t)p * g.N + c];
    }
    return (double)rec_z1(cur) * g.zs;
}

// block-wide accumulation of statistics into global counters
template <int NS>
__device__ __forceinline__ void stats_commit(unsigned long long* gstats, const unsigned (&loc)[NS], const int (&slot)[NS]) {
    __shared__ unsigned s_acc[NS];
    if (threadIdx.x < NS) s_acc[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NS; q++) {
        unsigned v = loc[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_acc[q], v);
    }
    __syncthreads();
    if (threadIdx.x < NS && s_acc[threadIdx.x]) atomicAdd(&gstats[slot[threadIdx.x]], (unsigned long long)s_acc[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------
// vertical steps, one column per thread (any N; also the re-run after the planes were grown)
// ------------------------------------------------------------------------------------------
// only_len < 0: normal run.  only_len >= 0 (deposit): touch only the columns that could not
// append in the first pass (len == only_len and the top is another material).
template <bool ETCH, bool STATS>
__global__ __launch_bounds__(256) void k_vertical_1(GridView g, const uint8_t* __restrict__ mask, int matId, double t,
                                                    int only_len, int* __restrict__ flags, unsigned long long* gstats) {
    unsigned loc[3] = {0, 0, 0};
    const bool isAll = (matId == 255);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        if (mask && !mask[c]) continue;
        const int l = g.len[c];
        if (ETCH) {
            const int r = apply_etch_at(g, c, l, matId, t, isAll);
            if (STATS) { loc[0]++; loc[1] += (r >> 16); loc[2] += (r & 0xffff); }
        } else {
            if (only_len >= 0) {
                if (l != only_len) continue;
                if (l > 0 && rec_mat(g.split ? g.top_w[c] : top_record(g, c, l)) == matId) continue;     // done in the first pass
            }
            const int r = depo_at_top(g, c, l, matId, t);
            if (r == 3) flags[0] = 1;
            if (STATS) { loc[0]++; loc[1] += (r == 1); loc[2] += (r == 2); }
        }
    }
    if (STATS) {
        if (ETCH) { const int slot[3] = {ST_OPEN, ST_ETCHED, ST_REMOVED}; stats_commit<3>(gstats, loc, slot); }
        else { const int slot[3] = {ST_OPEN, ST_EXTEND, ST_APPEND}; stats_commit<3>(gstats, loc, slot); }
    }
}

// ------------------------------------------------------------------------------------------
// 4 columns per thread ("groups")
// ------------------------------------------------------------------------------------------
// A thread owns 4 consecutive columns: the mask moves as one dword, the 4 top records as one
// 16-byte vector -- streams whose addresses do not depend on the column lengths.  Only an append
// (old top -> the launch's plane), a cut (the top's bottom <- the plane its link names) and the
// removal of a whole layer (new top <- that plane) touch the planes; columns of a group whose
// links name the same plane share one 16-byte access.  N % 4 == 0.
typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned as_u32(u8x4 v) { union { u8x4 a; unsigned b; } u; u.a = v; return u.b; }
__device__ __forceinline__ u8x4 splat4(int x) { union { u8x4 a; unsigned b; } u; u.b = 0x01010101u * (unsigned)(x & 255); return u.a; }
__device__ __forceinline__ u8x4 ld_u8x4(const uint8_t* p) { return *(const u8x4*)p; }
__device__ __forceinline__ void st_u8x4(uint8_t* p, u8x4 v) { *(u8x4*)p = v; }
__device__ __forceinline__ u32x4 ld_rec4(const uint32_t* p) { return *(const u32x4*)p; }
__device__ __forceinline__ void st_rec4(uint32_t* p, u32x4 v) { *(u32x4*)p = v; }
__device__ __forceinline__ u16x4 ld_u16x4(const uint16_t* p) { return *(const u16x4*)p; }
__device__ __forceinline__ void st_u16x4(uint16_t* p, u16x4 v) { *(u16x4*)p = v; }
__device__ __forceinline__ unsigned open_bits(u8x4 mk) { return (mk[0] ? 1u : 0u) | (mk[1] ? 2u : 0u) | (mk[2] ? 4u : 0u) | (mk[3] ? 8u : 0u); }

// _depoAtTop_ on the open columns of a group: column j grows by dz[j] (> 0).  tv = the 4 tops.
// Returns whether the group's tops were stored.  The host launches this form only while every
// column has a plane to spare (the reference's growth rule needs the lengths).
template <bool STATS>
__device__ __forceinline__ bool depo_group4(const GridView& g, size_t c0, unsigned open, const double (&dz)[4], int matId,
                                            u32x4 tv, unsigned& n_ext, unsigned& n_app) {
    u32x4 nv = tv;
    unsigned ext = 0, app = 0, push = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!((open >> j) & 1)) continue;
        const uint32_t r = tv[j];                              // 0: empty column
        const uint32_t q = quantz((double)rec_z1(r) * g.zs + dz[j], g.zs, g.inv_zs);
        if (r != 0 && rec_mat(r) == matId) { nv[j] = rec_with_z1(r, q); ext |= 1u << j; }
        else { nv[j] = rec_make_l(q, matId, r != 0 ? g.P : LNK_NONE); app |= 1u << j; if (r != 0) push |= 1u << j; }
    }
    if (!(ext | app)) return false;
    if (push) {
        // the old tops go into this step's plane as they are -- each carries the link to what is
        // below it -- whatever the column lengths: one dense line per 32 columns
        const size_t b = (size_t)g.P * g.N + c0;
        st_rec4(g.rec + b, tv);              // (the plane is this launch's own: a slot no chain names may hold anything, so the group goes out whole)
    }
    if (STATS) n_app += __popc(app);
    st_rec4(g.top_w + c0, nv);
    if (STATS) n_ext += __popc(ext);
    return true;
}

// The intervals below the tops r[] of the columns in `want` (those that have one): their records
// b[] -- the top's bottom is rec_z1(b), and b, link included, is the new top should the whole top
// layer go.  Columns whose links name the same plane (they were pushed by the same step) share
// one vector load.
__device__ __forceinline__ void load_below4(const GridView& g, size_t c0, unsigned want, const uint32_t (&r)[4], uint32_t (&b)[4]) {
    unsigned todo = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) if (((want >> j) & 1) && rec_lnk(r[j]) != LNK_NONE) todo |= 1u << j;
    while (todo) {
        const int j0 = __ffs(todo) - 1;
        const int p = rec_lnk(r[j0]);
        unsigned sel = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) if (((todo >> j) & 1) && rec_lnk(r[j]) == p) sel |= 1u << j;
        todo &= ~sel;
        const size_t at = (size_t)p * g.N + c0;
        if (sel & (sel - 1)) {
            const u32x4 A = ld_rec4(g.rec + at);
#pragma unroll
            for (int j = 0; j < 4; j++) if ((sel >> j) & 1) b[j] = A[j];
        } else {
            b[j0] = g.rec[at + j0];
        }
    }
}

// The same records by four one-column loads that are issued back to back and nothing waits for (load_below4's
// loop consumes each load before it issues the next): the form for a kernel that has other work to do meanwhile.
// A column without an interval below reads plane 0 instead, and the value is not looked at.
__device__ __forceinline__ void load_below4_flat(const GridView& g, size_t c0, const u32x4 tv, uint32_t (&b)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int p = rec_lnk(tv[j]);
        b[j] = g.rec[(size_t)(p != LNK_NONE ? p : 0) * g.N + c0 + j];
    }
}

// _applyEtchAt on the open columns of a group: column j loses dz[j] (> 0) from the top.
// tv = the 4 tops; `have` names the columns whose interval below the top the caller has loaded
// already (b0, see load_below4).  A column that loses its whole top layer continues with the
// interval below it.
// Register form: r[] holds the 4 tops on entry and the new tops on return; returns the columns touched.
template <bool STATS>
__device__ __forceinline__ unsigned etch_group4_regs(const GridView& g, size_t c0, unsigned open, const double (&dz)[4], int matId,
                                                     bool isAll, uint32_t (&r)[4], unsigned have, const uint32_t (&b0)[4], unsigned& n_removed) {
    double rem[4] = {dz[0], dz[1], dz[2], dz[3]};
    uint32_t b[4] = {b0[0], b0[1], b0[2], b0[3]};
    unsigned dirty = 0, act = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (((open >> j) & 1) && r[j] != 0 && rem[j] > CG_EPS && (isAll || rec_mat(r[j]) == matId)) act |= 1u << j;   // cg1:565-569
    while (act) {
        load_below4(g, c0, act & ~have, r, b);
        have = 0;
        unsigned next = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!((act >> j) & 1)) continue;
            const bool has_below = rec_lnk(r[j]) != LNK_NONE;
            const uint32_t below = has_below ? b[j] : 0u;
            const double z1 = (double)rec_z1(r[j]) * g.zs, z0 = (double)rec_z1(below) * g.zs, thick = z1 - z0;
            dirty |= 1u << j;
            if (rem[j] < thick - CG_EPS) { r[j] = rec_with_z1(r[j], quantz(z1 - rem[j], g.zs, g.inv_zs)); rem[j] = 0; continue; }   // cg1:574-578
            rem[j] -= thick;                                                                                                        // cg1:579-587
            if (STATS) n_removed += 1;
            r[j] = below;                                      // 0 when nothing was below: the column is empty now
            if (rec_hole(r[j])) {                              // a gap under the removed layer goes with it
                if (rec_lnk(r[j]) == LNK_NONE) { r[j] = 0; continue; }
                r[j] = g.rec[(size_t)rec_lnk(r[j]) * g.N + c0 + j];
            }
            if (r[j] != 0 && rem[j] > CG_EPS && (isAll || rec_mat(r[j]) == matId)) next |= 1u << j;
        }
        act = next;
    }
    return dirty;
}

template <bool STATS>
__device__ __forceinline__ bool etch_group4(const GridView& g, size_t c0, unsigned open, const double (&dz)[4], int matId,
                                            bool isAll, u32x4 tv, unsigned have, const uint32_t (&b0)[4],
                                            unsigned& n_touched, unsigned& n_removed) {
    uint32_t r[4] = {tv[0], tv[1], tv[2], tv[3]};
    const unsigned dirty = etch_group4_regs<STATS>(g, c0, open, dz, matId, isAll, r, have, b0, n_removed);
    if (dirty) {
        u32x4 nv; nv[0] = r[0]; nv[1] = r[1]; nv[2] = r[2]; nv[3] = r[3];
        st_rec4(g.top_w + c0, nv);
    }
    if (STATS) n_touched += __popc(dirty);
    return dirty != 0;
}

// What the host has verified for one vertical step (thickness t the same for every column), each a licence to replace
// the double arithmetic of the reference by integer arithmetic on the height steps with the same result:
//   depo_k >= 0   Math.round((q*zs + t)/zs) == q + depo_k for every q with q + depo_k < 65536 (checked one by one)
//   etch_kc >= 0  with d = z1q - z0q of the top layer, thick = z1q*zs - z0q*zs in double (within 2e-12 of d*zs):
//                 d >= etch_kc  <=>  t < thick - 1e-9   (the cut stays inside the layer, cg1:574); the new top is then
//                                    z1q - etch_ke = Math.round((z1q*zs - t)/zs) (checked for every z1q >= etch_kc);
//                 d >  etch_kr  <=>  not (t - thick > 1e-9): the layer goes and nothing is left to take (cg1:565);
//                 d <= etch_kr: the layer goes and the etch continues below with what is left, in double as before.
//                 Both thresholds were found with a margin of 1e-10 around the reference's 1e-9, fifty times the
//                 rounding error of thick; a thickness that leaves less gets no licence.
struct VertOpts { int depo_k, etch_kc, etch_kr, etch_ke; };

__device__ __forceinline__ bool etchable(uint32_t r, int matId, bool isAll);

// depo_group4 for a step whose new heights are q + K (VertOpts::depo_k); a column too tall for that takes the double form.
__device__ __forceinline__ void depo_group4_int(const GridView& g, size_t c0, unsigned open, int K, double t, int matId, u32x4 tv) {
    u32x4 nv = tv;
    unsigned push = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!((open >> j) & 1)) continue;
        const uint32_t r = tv[j];
        uint32_t q = rec_z1(r) + (uint32_t)K;
        if (q > 65535u) q = quantz((double)rec_z1(r) * g.zs + t, g.zs, g.inv_zs);
        if (r != 0 && rec_mat(r) == matId) nv[j] = rec_with_z1(r, q);
        else { nv[j] = rec_make_l(q, matId, r != 0 ? g.P : LNK_NONE); if (r != 0) push |= 1u << j; }
    }
    if (push) {
        const size_t b = (size_t)g.P * g.N + c0;
        st_rec4(g.rec + b, tv);              // (the plane is this launch's own: a slot no chain names may hold anything, so the group goes out whole)
    }
    st_rec4(g.top_w + c0, nv);
}

// etch_group4 for a step with integer thresholds (VertOpts): a cut inside the top layer, or the top layer gone with
// nothing left to take, are decided on the height steps; columns that go on take the general form.
__device__ __forceinline__ void etch_group4_int(const GridView& g, size_t c0, unsigned open, const VertOpts& vo, const double (&dz)[4],
                                                int matId, bool isAll, u32x4 tv) {
    uint32_t r[4] = {tv[0], tv[1], tv[2], tv[3]}, b[4] = {0, 0, 0, 0};
    unsigned want = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) if (((open >> j) & 1) && etchable(r[j], matId, isAll)) want |= 1u << j;
    if (!want) return;
    load_below4(g, c0, want, r, b);
    unsigned general = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!((want >> j) & 1)) continue;
        const uint32_t below = rec_lnk(r[j]) != LNK_NONE ? b[j] : 0u;
        const int d = (int)rec_z1(r[j]) - (int)rec_z1(below);
        if (d >= vo.etch_kc) { r[j] = rec_with_z1(r[j], rec_z1(r[j]) - (uint32_t)vo.etch_ke); continue; }       // cg1:574-578
        if (d <= vo.etch_kr) { general |= 1u << j; continue; }
        r[j] = below;                                                                                         // cg1:579-587, and done
        if (rec_hole(r[j])) r[j] = rec_lnk(r[j]) == LNK_NONE ? 0u : g.rec[(size_t)rec_lnk(r[j]) * g.N + c0 + j];
    }
    if (general) { unsigned nrem = 0; etch_group4_regs<false>(g, c0, general, dz, matId, isAll, r, general, b, nrem); }
    u32x4 nv; nv[0] = r[0]; nv[1] = r[1]; nv[2] = r[2]; nv[3] = r[3];
    st_rec4(g.top_w + c0, nv);
}

// Vertical deposit / etch (fast paths, cg1:445-454 and 609-618).  A thread owns U groups that
// are 256 groups apart; all their loads are issued before anything is computed or stored.
// fresh (deposit without a mask only): every column is empty and the tops have not been cleared
// since the grid was reset -- nothing is loaded, and the launch leaves the stream defined.
template <bool ETCH, bool STATS, int U, bool INT = false>
__global__ __launch_bounds__(256) void k_vertical_g4(GridView g, const uint8_t* __restrict__ mask, int matId, double t, int fresh,
                                                     const VertOpts vo, int* __restrict__ flags, unsigned long long* gstats) {
    unsigned loc[3] = {0, 0, 0};
    const bool isAll = (matId == 255);
    const size_t ngroups = g.N >> 2;
    const size_t tile_groups = (size_t)256 * U;
    const size_t ntiles = (ngroups + tile_groups - 1) / tile_groups;
    const double dz[4] = {t, t, t, t};
    // Workgroup b runs on XCD b mod 8, and tile b of every image row covers the same column range;
    // a masked step whose open area covers some column ranges only would then load the XCDs
    // unevenly.  Rotating the tiles of image row r by r (and the sweeps of a capped grid by the
    // sweep number) spreads every column range over all XCDs.
    const size_t row_tiles = ((size_t)(g.ny >> 2) % tile_groups == 0) ? (size_t)(g.ny >> 2) / tile_groups : 0;
    const size_t nsweeps = (ntiles + gridDim.x - 1) / gridDim.x;
    for (size_t sweep = 0; sweep < nsweeps; sweep++) {
        size_t tile = sweep * gridDim.x + (blockIdx.x + sweep) % gridDim.x;
        if (tile >= ntiles) continue;
        if (row_tiles > 1) { const size_t r = tile / row_tiles, q = tile - r * row_tiles; tile = r * row_tiles + (q + r) % row_tiles; }
        const size_t gbase = tile * tile_groups + threadIdx.x;
        unsigned open[U]; u32x4 tv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t gi = gbase + (size_t)u * 256;
            open[u] = 0; tv[u] = (u32x4)(0);
            if (gi < ngroups) {
                const size_t c0 = gi << 2;
                open[u] = mask ? open_bits(ld_u8x4(mask + c0)) : 15u;
                if (open[u] && !(!ETCH && fresh)) tv[u] = ld_rec4(g.top + c0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!open[u]) continue;
            const size_t c0 = (gbase + (size_t)u * 256) << 2;
            if (STATS) loc[0] += __popc(open[u]);
            // (INT: the host holds the step's integer licence; never together with statistics)
            if (ETCH) {
                if (INT) etch_group4_int(g, c0, open[u], vo, dz, matId, isAll, tv[u]);
                else { const uint32_t nb[4] = {0, 0, 0, 0};
                       etch_group4<STATS>(g, c0, open[u], dz, matId, isAll, tv[u], 0u, nb, loc[1], loc[2]); }
            } else {
                if (INT) depo_group4_int(g, c0, open[u], vo.depo_k, t, matId, tv[u]);
                else depo_group4<STATS>(g, c0, open[u], dz, matId, tv[u], loc[1], loc[2]);
            }
        }
    }
    if (STATS) {
        if (ETCH) { const int slot[3] = {ST_OPEN, ST_ETCHED, ST_REMOVED}; stats_commit<3>(gstats, loc, slot); }
        else { const int slot[3] = {ST_OPEN, ST_EXTEND, ST_APPEND}; stats_commit<3>(gstats, loc, slot); }
    }
}

// ------------------------------------------------------------------------------------------
// A run of consecutive vertical steps in one pass over the columns
// ------------------------------------------------------------------------------------------
// Vertical steps touch a column and nothing around it, so a run of them commutes with the loop over the columns: the
// host queues such steps (nothing in between looks at the grid) and this kernel applies the whole run to the 4 columns
// of a lane in registers -- masks and top loaded once, every step's own arithmetic (the same integer licences or
// double expressions as k_vertical_g4, step by step, so the result is the one the single launches give), the final top
// stored once.  A deposit of the run pushes the old tops into its own fresh plane as always; while the run goes on
// they also stay in registers, so an etch later in the run that needs "the interval below" of a layer laid down by
// the run finds it there (a link that names one of the run's planes) and only goes to memory for older intervals.
constexpr int FUSE_MAX = 6;
struct FusedOp { double t; VertOpts vo; int etch, mat, P, pad_; };
// fresh: the run starts on a reset grid (every column empty, nothing to load).  Its first `nuni` steps are then deposits
// without a mask, which leave every column the same: the host has worked that column out (uni_top, and what each of
// those steps pushed, uni_pushed, 0 = nothing), and the kernel starts from it with step nuni.
constexpr int FUSE_UNI_MAX = 3;
// masks[k]: the mask of step k, null = none.  rt_shift: log2 of the tiles per image row when that is a power of two above 1, else -1.
struct FusedOps { FusedOp op[FUSE_MAX]; const uint8_t* masks[FUSE_MAX]; int n, fresh, rt_shift, nuni; uint32_t uni_top, uni_pushed[FUSE_MAX]; uint32_t pad_; };

// The intervals below the tops r[] of the columns in `want`: from the registers of the run's deposits before step k
// where the link names one of their planes, else from memory.
struct BelowFromMemory {
    __device__ __forceinline__ void operator()(const GridView& g, const size_t c0, const unsigned want, const uint32_t (&r)[4], uint32_t (&b)[4]) const {
        load_below4(g, c0, want, r, b);
    }
};
template <class BM>
__device__ __forceinline__ void fused_below4(const GridView& g, const FusedOps& F, const int k, const size_t c0, const unsigned want,
                                             const uint32_t (&r)[4], const uint32_t (&pushed)[FUSE_MAX][4], uint32_t (&b)[4], const BM& below_mem) {
    unsigned mem = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!((want >> j) & 1)) continue;
        const int p = rec_lnk(r[j]);
        if (p == LNK_NONE) { b[j] = 0u; continue; }
        bool found = false;
#pragma unroll
        for (int q = 0; q < FUSE_MAX; q++)
            if (q < k && !F.op[q].etch && F.op[q].P == p) { b[j] = pushed[q][j]; found = true; }
        if (!found) mem |= 1u << j;
    }
    if (mem) below_mem(g, c0, mem, r, b);
}

// The body of a run (cg_fused_steps.inc.h) as device functions: the FOLD form of k_stencil_lean calls these, while
// k_vertical_fused below includes the text -- each of the two is compiled noticeably better its own way.
// Steps KB .. F.n-1 of the run on the 4 columns at c0.
template <int KB, int KE = FUSE_MAX, class BM = BelowFromMemory>
__device__ __forceinline__ void fused_steps(const GridView& g, const FusedOps& F, const size_t c0, const unsigned (&open)[FUSE_MAX], uint32_t (&r)[4],
                                            uint32_t (&pushed)[FUSE_MAX][4], const BM& below_mem = BM()) {
#define CG_FS_KB KB
#define CG_FS_KE KE
#define CG_FS_PART 1
#include "cg_fused_steps.inc.h"
#undef CG_FS_PART
#undef CG_FS_KE
#undef CG_FS_KB
}
template <int KB, int KE = FUSE_MAX>
__device__ __forceinline__ void fused_store_planes(const GridView& g, const FusedOps& F, const size_t c0, const uint32_t (&pushed)[FUSE_MAX][4]) {
#define CG_FS_KB KB
#define CG_FS_KE KE
#define CG_FS_PART 2
#include "cg_fused_steps.inc.h"
#undef CG_FS_PART
#undef CG_FS_KE
#undef CG_FS_KB
}

// NUNI = FusedOps::nuni (0..3) as a compile-time number: what those steps pushed then stays in scalar registers.
template <int NUNI>
__global__ __launch_bounds__(256) void k_vertical_fused(GridView g, const FusedOps F) {
    const size_t ngroups = g.N >> 2;
    const size_t ntiles = (ngroups + 255) / 256;
    unsigned tile = blockIdx.x;
    if (tile >= ntiles) return;
    // (tiles of image row r rotated by r, as in k_vertical_g4)
    if (F.rt_shift > 0) { const unsigned rt = 1u << F.rt_shift, r = tile >> F.rt_shift, q = tile & (rt - 1u); tile = (r << F.rt_shift) + ((q + r) & (rt - 1u)); }
    const size_t gi = (size_t)tile * 256 + threadIdx.x;
    if (gi >= ngroups) return;
    const size_t c0 = gi << 2;
    // every load of the lane goes out before anything is looked at: the run's distinct masks (4 bytes each) and the tops
    unsigned open[FUSE_MAX], any_open = 0;               // the mask bytes of the 4 columns for step k (non-zero byte = open)
#pragma unroll
    for (int k = 0; k < FUSE_MAX; k++) {
        open[k] = k < F.n ? 0x01010101u : 0u;
        if (k < F.n && F.masks[k]) open[k] = *(const unsigned*)(F.masks[k] + c0);      // (a mask used twice comes from the cache the second time)
    }
    u32x4 tv = (u32x4)(0);
    if (!F.fresh) tv = ld_rec4(g.top + c0);
#pragma unroll
    for (int k = 0; k < FUSE_MAX; k++) any_open |= open[k];
    if (!any_open) return;
    if (NUNI > 0) { tv[0] = F.uni_top; tv[1] = F.uni_top; tv[2] = F.uni_top; tv[3] = F.uni_top; }
    uint32_t r[4] = {tv[0], tv[1], tv[2], tv[3]};
    uint32_t pushed[FUSE_MAX][4];                        // 0 = the column pushed nothing in that step (a pushed record never is 0)
#pragma unroll
    for (int k = 0; k < FUSE_MAX; k++) {
        const uint32_t u = k < NUNI ? F.uni_pushed[k] : 0u;
        pushed[k][0] = u; pushed[k][1] = u; pushed[k][2] = u; pushed[k][3] = u;
    }
    const BelowFromMemory below_mem;
#define CG_FS_KB NUNI
#define CG_FS_KE FUSE_MAX
#define CG_FS_PART 1
#include "cg_fused_steps.inc.h"
#undef CG_FS_PART
#undef CG_FS_KE
#undef CG_FS_KB
    // the old tops every deposit of the run pushed, each into its own plane; then the tops
#define CG_FS_KB 0
#define CG_FS_KE FUSE_MAX
#define CG_FS_PART 2
#include "cg_fused_steps.inc.h"
#undef CG_FS_PART
#undef CG_FS_KE
#undef CG_FS_KB
    if (F.fresh || r[0] != tv[0] || r[1] != tv[1] || r[2] != tv[2] || r[3] != tv[3]) {
        u32x4 nv; nv[0] = r[0]; nv[1] = r[1]; nv[2] = r[2]; nv[3] = r[3];
        st_rec4(g.top_w + c0, nv);
    }
}

// ------------------------------------------------------------------------------------------
// A run of vertical steps on a grid that has just been reset
// ------------------------------------------------------------------------------------------
// Every column of a reset grid is empty, and a vertical step changes a column from that column and its mask byte alone:
// after a run of such steps a column's state is a function of which of the run's masks are open there and of nothing
// else.  With NM distinct masks in the run there are 2^NM such classes; the host works the one column of each class out
// (the literal reference arithmetic on one column, see fresh_run_table in colgrid.hip) and this kernel only looks the
// class of every column up: mask bytes in, the top record and what each deposit of the run pushed into its plane out.
// The start of every rebuild -- createNewGrid, substrate, blanket films, the first patterning etches -- runs like this,
// at the rate of its stores.
constexpr int FRESH_MAX_STEPS = 16;        // steps of such a run (the queue of a reset grid may grow this long)
constexpr int FRESH_MAX_MASKS = 3;         // distinct masks in it
constexpr int FRESH_MAX_DEP = 12;          // deposits (each owns a plane)
struct FreshTab {
    uint32_t top[1 << FRESH_MAX_MASKS];                         // per class: the final top record (0: the column ends empty)
    uint32_t pushed[FRESH_MAX_DEP][1 << FRESH_MAX_MASKS];       // per deposit and class: the record stored into that deposit's plane, 0 = none
    int plane[FRESH_MAX_DEP];
    int ndep, nmask, rt_shift, pad_;
    const uint8_t* mask[FRESH_MAX_MASKS];
};

// A thread owns FRESH_U groups of 4 columns, 256 groups apart (a launch that mostly stores runs faster that way: measured
// +40 % for the write-only first deposit); all mask loads go out first.
constexpr int FRESH_U = 4;
template <int NM>
__global__ __launch_bounds__(256) void k_vertical_fresh(GridView g, const FreshTab T) {
    constexpr int NC = 1 << NM;
    __shared__ uint32_t s_tab[(1 + FRESH_MAX_DEP) * NC];          // [0..NC) tops, then NC entries per deposit
    for (int e = threadIdx.x; e < (1 + T.ndep) * NC; e += 256) s_tab[e] = e < NC ? T.top[e] : T.pushed[e / NC - 1][e % NC];
    __syncthreads();
    const size_t ngroups = g.N >> 2;
    const size_t gbase = (size_t)blockIdx.x * (256 * FRESH_U) + threadIdx.x;
    unsigned w[FRESH_U][NM > 0 ? NM : 1];
#pragma unroll
    for (int u = 0; u < FRESH_U; u++) {
        const size_t gi = gbase + (size_t)u * 256;
#pragma unroll
        for (int m = 0; m < NM; m++) w[u][m] = gi < ngroups ? *(const unsigned*)(T.mask[m] + (gi << 2)) : 0u;
    }
#pragma unroll
    for (int u = 0; u < FRESH_U; u++) {
        const size_t gi = gbase + (size_t)u * 256;
        if (gi >= ngroups) continue;
        const size_t c0 = gi << 2;
        unsigned cls[4] = {0, 0, 0, 0};
#pragma unroll
        for (int m = 0; m < NM; m++) {
#pragma unroll
            for (int j = 0; j < 4; j++) cls[j] |= ((w[u][m] >> (8 * j)) & 0xffu) ? (1u << m) : 0u;
        }
        u32x4 nv;
#pragma unroll
        for (int j = 0; j < 4; j++) nv[j] = s_tab[cls[j]];
        st_rec4(g.top_w + c0, nv);
        for (int d = 0; d < T.ndep; d++) {
            u32x4 pv;
#pragma unroll
            for (int j = 0; j < 4; j++) pv[j] = s_tab[(1 + d) * NC + cls[j]];
            uint32_t* at = g.rec + (size_t)T.plane[d] * g.N + c0;
            // (a slot of the deposit's own plane that no chain names may hold anything: one store for the group when any
            // of its columns pushed, zeros for the others)
            if (pv[0] | pv[1] | pv[2] | pv[3]) st_rec4(at, pv);
        }
    }
}

// ------------------------------------------------------------------------------------------
// lateral-stencil steps: dilation / erosion of the top-height map + per-column update
// ------------------------------------------------------------------------------------------
//
// A lateral step needs the OLD top height of every column within the kernel radius, and for
// etch the OLD "etchable here" flag E (cg1:637-643: open in the mask, not empty, top material
// matches).  Both come straight from the `top` stream; the updated tops go to the other of two
// buffers (GridView::top_w), which the host swaps in afterwards, so no snapshot pass is needed.
// Rows beyond the slab come from the neighbouring slab's message, or -- at a domain edge --
// are the clamp-replicated edge row (heights, cg1:373-387) with zero flags (cg1:671).
//
// Message of Rh rows: Rh*ny u16 heights, then Rh*ny u8 flags (zeros for deposit).
// send_lower = rows 0..Rh-1, send_upper = rows nx-Rh..nx-1.
// Workgroup b runs on XCD b mod 8, each XCD has its own L2, and neighbouring tiles share halo
// rows and columns (and the 128-B lines that straddle a tile edge).  Blocks that follow each other
// on ONE XCD (b, b+8, b+16, ...) therefore get the sup_i x sup_j tiles of one super-tile, so the
// shared lines are fetched from HBM once; super-tiles go round the XCDs, rotated by their row so
// that a mask open in some column ranges only does not load the XCDs unevenly.
// Blocks beyond the last super-tile or tile return false.
__device__ __forceinline__ bool tile_of_block(const GridView& g, int tiles_j, int& tile_i, int& tile_j) {
    const int per = g.sup_i * g.sup_j;
    const int b = blockIdx.x, x = b & 7, q = b >> 3;
    const int n_sj = (tiles_j + g.sup_j - 1) / g.sup_j, n_si = (g.tile_in + g.sup_i - 1) / g.sup_i;
    const int s = (q / per) * 8 + x, within = q % per;
    if (s >= n_sj * n_si) return false;
    const int s_i = s / n_sj, s_j = (s - s_i * n_sj + s_i) % n_sj;
    const int ti = s_i * g.sup_i + within / g.sup_j, tj = s_j * g.sup_j + within % g.sup_j;
    if (ti >= g.tile_in || tj >= tiles_j) return false;
    tile_i = g.tile_i0 + ti; tile_j = tj;
    return true;
}

__device__ __forceinline__ bool etchable(uint32_t r, int matId, bool isAll) { return r != 0 && (isAll || rec_mat(r) == matId); }

static __global__ __launch_bounds__(256) void k_halo_pack(GridView g, const uint8_t* __restrict__ mask, int matId, int etch, int Rh,
                                                   uint8_t* __restrict__ send_lower, uint8_t* __restrict__ send_upper) {
    const bool isAll = (matId == 255);
    const size_t n = (size_t)Rh * g.ny;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const size_t lo = e, hi = (size_t)(g.nx - Rh) * g.ny + e;
        const uint32_t rl = g.top[lo], rh = g.top[hi];
        ((uint16_t*)send_lower)[e] = (uint16_t)rec_z1(rl);
        ((uint16_t*)send_upper)[e] = (uint16_t)rec_z1(rh);
        send_lower[2 * n + e] = (etch && (!mask || mask[lo]) && etchable(rl, matId, isAll)) ? 1 : 0;
        send_upper[2 * n + e] = (etch && (!mask || mask[hi]) && etchable(rh, matId, isAll)) ? 1 : 0;
    }
}

// Where the heights of local row `il` (any integer) come from.
struct RowSrc { const uint8_t* msg; int row; bool edge; };     // msg != 0: message row `row`; else grid row `row`
__device__ __forceinline__ RowSrc row_source(const GridView& g, int il, int Rh, const uint8_t* halo_lo, const uint8_t* halo_up) {
    RowSrc s; s.msg = nullptr; s.row = il; s.edge = false;
    if (il < 0) { if (halo_lo) { s.msg = halo_lo; s.row = Rh + il; } else { s.row = 0; s.edge = true; } }
    else if (il >= g.nx) { if (halo_up) { s.msg = halo_up; s.row = il - g.nx; } else { s.row = g.nx - 1; s.edge = true; } }
    return s;
}

// One stencil tap: LDS offset relative to the centre (du*pitch + dv) and the float32 height
// increment of the ellipsoidal kernel (cg1:324-357).
struct Tap { int off; float add; };

// Geometry of the general stencil kernel: a block of 256 threads updates TI x TJ columns;
// thread (tx, ty) owns column j = tx and the TI/4 rows ty, ty+4, ...
constexpr int TJ = 64;
constexpr int TI = 16;
constexpr int ROWS_PER_THREAD = TI / 4;

__host__ __device__ inline size_t stencil_lds_bytes(int R) { return (size_t)(TI + 2 * R) * (TJ + 2 * R) * sizeof(float); }

// Why float32 max-plus is exact here: the reference widens the float32 height and the float32
// kernel increment to double, adds, compares, and rounds the running maximum to float32 once
// at the end (cg1:512-526).  Rounding is monotone, so that equals the maximum of the
// individually rounded sums -- provided each double sum is exact, which holds when the
// increment is a multiple of 2^-40 (heights are < 2^13 and multiples of 2^-40 as well, so the
// sum fits 53 bits).  The host sorts taps into `taps` (safe, done in float32) and `taps64`
// (increment too fine, done in double exactly like the reference).
//
// General form: any radius that fits the LDS, tap list at run time, one column at a time.
template <bool ETCH, bool STATS>
__global__ __launch_bounds__(256) void k_stencil_apply(GridView g, const uint8_t* __restrict__ mask, int matId, double t, int R,
                                                       const uint8_t* __restrict__ halo_lo, const uint8_t* __restrict__ halo_up,
                                                       const Tap* __restrict__ taps, int ntaps,
                                                       const Tap* __restrict__ taps64, int ntaps64,
                                                       int only_len, int* __restrict__ flags, unsigned long long* gstats) {
    extern __shared__ __align__(16) float s_h[];
    const int pitch = TJ + 2 * R;
    const int tile_rows = TI + 2 * R;
    const int tiles_j = (g.ny + TJ - 1) / TJ;
    int tile_i, tile_j;
    if (!tile_of_block(g, tiles_j, tile_i, tile_j)) return;
    const int i_base = tile_i * TI, j_base = tile_j * TJ;

    // ---- stage heights (as float32 of q*zscale, cg1:465) with halo; flags fold into +inf
    const bool isAll = (matId == 255);
    const size_t msg_n = (size_t)R * g.ny;
    for (int e = threadIdx.x; e < tile_rows * pitch; e += 256) {
        const int r = e / pitch, cidx = e - r * pitch;
        const int j = j_base + cidx - R;
        const bool outside = (j < 0 || j >= g.ny);
        const int jc = j < 0 ? 0 : (j >= g.ny ? g.ny - 1 : j);            // clamp-replicate in j (cg1:374-380)
        const RowSrc src = row_source(g, i_base + r - R, R, halo_lo, halo_up);
        unsigned q = 0; bool ok = false;
        if (src.msg) {
            if (src.row >= 0 && src.row < R) {
                const size_t p = (size_t)src.row * g.ny + jc;
                q = ((const uint16_t*)src.msg)[p];
                ok = src.msg[2 * msg_n + p] != 0;
            }
        } else {
            const size_t p = (size_t)src.row * g.ny + jc;
            const uint32_t rr = g.top[p];
            q = rec_z1(rr);
            if (ETCH) ok = !src.edge && (!mask || mask[p]) && etchable(rr, matId, isAll);
        }
        float h = (float)((double)q * g.zs);
        if (ETCH && (outside || !ok)) h = __builtin_inff();               // not etchable: never a minimum (cg1:718)
        s_h[e] = h;
    }
    __syncthreads();

    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = j_base + tx;
    unsigned loc[4] = {0, 0, 0, 0};

    if (j < g.ny) {
        float m[ROWS_PER_THREAD];
        float own[ROWS_PER_THREAD];
        bool act[ROWS_PER_THREAD];
        int lbase[ROWS_PER_THREAD];
#pragma unroll
        for (int r = 0; r < ROWS_PER_THREAD; r++) {
            int il = i_base + ty + 4 * r;
            act[r] = false; m[r] = 0.f; own[r] = 0.f;
            lbase[r] = (ty + 4 * r + R) * pitch + (tx + R);
            if (il < g.nx) {
                size_t c = (size_t)il * g.ny + j;
                const uint32_t rr = g.top[c];
                float h = (float)((double)rec_z1(rr) * g.zs);
                own[r] = h;
                if (g.split && only_len < 0 && g.top_w != g.top) g.top_w[c] = rr;    // columns that stay as they are
                if (ETCH) {
                    act[r] = (!mask || mask[c]) && etchable(rr, matId, isAll);
                    m[r] = (float)((double)h - t);                     // cg1:702
                } else {
                    act[r] = (!mask || mask[c]);
                    m[r] = (float)((double)h + t);                     // cg1:499
                }
            }
        }
        if (act[0] | act[1] | act[2] | act[3]) {
            for (int q = 0; q < ntaps; q++) {
                const int off = taps[q].off; const float add = taps[q].add;
#pragma unroll
                for (int r = 0; r < ROWS_PER_THREAD; r++) {
                    float hv = s_h[lbase[r] + off];
                    if (ETCH) m[r] = fminf(m[r], hv - add); else m[r] = fmaxf(m[r], hv + add);
                }
            }
            if (ntaps64 > 0) {
                double md[ROWS_PER_THREAD];
#pragma unroll
                for (int r = 0; r < ROWS_PER_THREAD; r++) md[r] = (double)m[r];
                for (int q = 0; q < ntaps64; q++) {
                    const int off = taps64[q].off; const double add = (double)taps64[q].add;
#pragma unroll
                    for (int r = 0; r < ROWS_PER_THREAD; r++) {
                        double hv = (double)s_h[lbase[r] + off];
                        double cand = ETCH ? hv - add : hv + add;
                        if (ETCH) { if (cand < md[r]) md[r] = cand; } else { if (cand > md[r]) md[r] = cand; }
                    }
                }
#pragma unroll
                for (int r = 0; r < ROWS_PER_THREAD; r++) m[r] = (float)md[r];
            }
#pragma unroll
            for (int r = 0; r < ROWS_PER_THREAD; r++) {
                if (!act[r]) continue;
                int il = i_base + ty + 4 * r;
                size_t c = (size_t)il * g.ny + j;
                int l = g.len[c];
                if (ETCH) {
                    float F = 0.f;
                    if (!isAll && l > 0) {
                        if (g.split) {
                            const uint32_t rt = g.top[c];
                            const uint32_t below = rec_lnk(rt) != LNK_NONE ? g.rec[(size_t)rec_lnk(rt) * g.N + c] : 0u;
                            F = (float)fmax(0.0, floor_of_top_run(g, c, matId, rt, below, loc[3]));
                        } else F = (float)fmax(0.0, floor_of_top_run_whole(g, c, l, matId, loc[3]));
                    }
                    float hp = fmaxf(0.f, fmaxf(F, m[r]));             // cg1:725
                    double dz = (double)own[r] - (double)hp;
                    if (dz > CG_EPS) {
                        int rr = apply_etch_at(g, c, l, matId, dz, isAll);
                        if (STATS) { loc[1] += (rr >> 16); loc[2] += (rr & 0xffff); }
                    }
                    if (STATS) loc[0]++;
                } else {
                    if (only_len >= 0) {
                        if (l != only_len) continue;
                        if (l > 0 && rec_mat(g.split ? g.top_w[c] : top_record(g, c, l)) == matId) continue;     // done in the first pass
                    }
                    double dz = (double)m[r] - (double)own[r];         // cg1:534-536
                    if (STATS) loc[0]++;
                    if (dz <= CG_EPS) continue;
                    int rr = depo_at_top(g, c, l, matId, dz);
                    if (rr == 3) flags[0] = 1;
                    if (STATS) { loc[1] += (rr == 1); loc[2] += (rr == 2); }
                }
            }
        }
    }
    if (STATS) {
        if (ETCH) { const int slot[4] = {ST_OPEN, ST_ETCHED, ST_REMOVED, ST_FLOOR}; stats_commit<4>(gstats, loc, slot); }
        else { const int slot[4] = {ST_OPEN, ST_EXTEND, ST_APPEND, ST_FLOOR}; stats_commit<4>(gstats, loc, slot); }
    }
}

// ------------------------------------------------------------------------------------------
// lateral-stencil steps, 4 columns per thread, radius known at compile time
// ------------------------------------------------------------------------------------------
// The ellipsoidal kernel is symmetric in +-du and +-dv (script_columngrid.js:337-347 computes
// add from du*du and dv*dv), so one quadrant table of (R+1)^2 increments describes it; entries
// outside the ellipse, and entries that need double evaluation, hold -infinity so that they
// never win.  max(h1,h2,h3,h4)+a == max(h1+a,...,h4+a) exactly because rounding is monotone:
// one add and one max serve four taps.
// lim[du] = largest dv whose entry of row du is a float32 tap (-1: none); the ellipse is convex, so
// the entries 0..lim[du] of a row are the valid ones and the loops stop there (uniform branches).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int R> struct AddTab { float a[(R + 1) * (R + 1)]; int lim[R + 1]; };

// Tile of the symmetric kernel: TJS columns x (1024 / TJS) rows per block of 256 threads.
// A block stages one tile and updates it in PASSES passes of 1024 columns each.  The halo rows are
// staged (de-quantised, flags folded in) once per tile, so a taller tile spends less on staging and
// halo reads per updated row; measured: 5 % faster at R <= 2, slower from R = 3 on (40 % at R = 7).
constexpr int sym_passes(int R) { return R <= 2 ? 2 : 1; }
template <int R, int TJS> struct SymGeom {
    static constexpr int PASSES = sym_passes(R);
    static constexpr int TIP = 1024 / TJS;           // rows per pass
    static constexpr int TIS = TIP * PASSES;         // rows per block
    static constexpr int GX = TJS / 4;               // groups (threads) per row
    static constexpr int Rp = (R + 3) & ~3;          // halo columns rounded up to whole groups
    static constexpr int PITCH = TJS + 2 * Rp;       // floats per LDS row
    static constexpr int ROWS = TIS + 2 * R;
    static constexpr int W = 4 + 2 * Rp;             // floats a thread reads per row
    static constexpr int GPR = PITCH / 4;            // groups per LDS row
};

// Thread (gx, ry) updates columns j0..j0+3 of row ry of the block's tile.
// Rimg = halo rows of the snapshot image (the cached kernel's Rx, >= R).
template <int R, int TJS, bool ETCH, bool STATS>
__global__ __launch_bounds__(256) void k_stencil_sym(GridView g, const uint8_t* __restrict__ mask, int matId, double t, int Rimg,
                                                     const uint8_t* __restrict__ halo_lo, const uint8_t* __restrict__ halo_up,
                                                     const AddTab<R> tab, const Tap* __restrict__ taps64, int ntaps64,
                                                     int* __restrict__ flags, unsigned long long* gstats) {
    using G = SymGeom<R, TJS>;
    constexpr int Rp = G::Rp, PITCH = G::PITCH, ROWS = G::ROWS, W = G::W, GPR = G::GPR, TIS = G::TIS, TIP = G::TIP, GX = G::GX;
    __shared__ __align__(16) float s_h[ROWS * PITCH];
    const int tiles_j = (g.ny + TJS - 1) / TJS;
    int tile_i, tile_j;
    if (!tile_of_block(g, tiles_j, tile_i, tile_j)) return;
    const int i_base = tile_i * TIS, j_base = tile_j * TJS;
    const float INF = __builtin_inff();

    // ---- stage float32(q*zscale) (cg1:465) with halo; not-etchable cells become +inf (cg1:718)
    const bool isAll = (matId == 255);
    const size_t msg_n = (size_t)Rimg * g.ny;
    for (int e = threadIdx.x; e < ROWS * GPR; e += 256) {
        const int r = e / GPR, gq = e - r * GPR;
        const int j4 = j_base - Rp + 4 * gq;
        const bool left = j4 < 0, right = j4 >= g.ny;
        const int jc = left ? 0 : (right ? g.ny - 4 : j4);
        const RowSrc src = row_source(g, i_base + r - R, Rimg, halo_lo, halo_up);
        u16x4 q = (u16x4)(0); u8x4 e4 = splat4(0);
        if (src.msg) {
            if (src.row >= 0 && src.row < Rimg) {
                const size_t p = (size_t)src.row * g.ny + jc;
                q = *(const u16x4*)((const uint16_t*)src.msg + p);
                if (ETCH) e4 = ld_u8x4(src.msg + 2 * msg_n + p);
            }
        } else {
            const size_t p = (size_t)src.row * g.ny + jc;
            const u32x4 tv = ld_rec4(g.top + p);
            q[0] = (unsigned short)rec_z1(tv[0]); q[1] = (unsigned short)rec_z1(tv[1]);
            q[2] = (unsigned short)rec_z1(tv[2]); q[3] = (unsigned short)rec_z1(tv[3]);
            if (ETCH && !src.edge) {
                const u8x4 mk = mask ? ld_u8x4(mask + p) : splat4(1);
#pragma unroll
                for (int c = 0; c < 4; c++) e4[c] = (mk[c] && etchable(tv[c], matId, isAll)) ? 1 : 0;
            }
        }
        if (left) q = (u16x4)(q[0]);                   // clamp-replicate in j (cg1:374-380)
        if (right) q = (u16x4)(q[3]);
        float4 h;
        h.x = (float)((double)q[0] * g.zs); h.y = (float)((double)q[1] * g.zs);
        h.z = (float)((double)q[2] * g.zs); h.w = (float)((double)q[3] * g.zs);
        if (ETCH) {
            if (left || right) h = make_float4(INF, INF, INF, INF);          // flags are zero outside the domain (cg1:671)
            else { if (!e4[0]) h.x = INF; if (!e4[1]) h.y = INF; if (!e4[2]) h.z = INF; if (!e4[3]) h.w = INF; }
        }
        *(float4*)(s_h + r * PITCH + 4 * gq) = h;
    }
    __syncthreads();

    const int gx = threadIdx.x % GX;
    const int j0 = j_base + 4 * gx;
    unsigned loc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int pass = 0; pass < G::PASSES; pass++) {
    const int ry = threadIdx.x / GX + pass * TIP;
    const int il = i_base + ry;
    if (il < g.nx && j0 < g.ny) {
        const size_t c0 = (size_t)il * g.ny + j0;
        const u32x4 tv = ld_rec4(g.top + c0);
        bool stored = false;
        unsigned act = mask ? open_bits(ld_u8x4(mask + c0)) : 15u;
        if (ETCH) {
#pragma unroll
            for (int c = 0; c < 4; c++) if (!etchable(tv[c], matId, isAll)) act &= ~(1u << c);
        }
        if (act) {
            const float* rowc = s_h + (ry + R) * PITCH + 4 * gx;
            float hc[W];
#pragma unroll
            for (int k = 0; k < W / 4; k++) { const float4 v = *(const float4*)(rowc + 4 * k); hc[4 * k] = v.x; hc[4 * k + 1] = v.y; hc[4 * k + 2] = v.z; hc[4 * k + 3] = v.w; }
            float own[4], m[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                own[c] = hc[Rp + c];
                // for etch, own of a non-etchable column is +inf; it is never used (act bit clear)
                m[c] = ETCH ? (float)((double)own[c] - t) : (float)((double)own[c] + t);     // cg1:702 / cg1:499
            }
// Deposit: max of non-negative floats (and -inf) is the max of their bit patterns as signed
// integers.  Etch: the candidates are >= 0, +inf, or negative; every negative result is clamped
// to 0 afterwards (cg1:725), so it only has to compare below the non-negative ones -- which a
// negative float does as a signed integer too.  No NaN can occur, and the integer form needs
// no canonicalisation of loaded values (and fuses into v_max3/v_min3).
#define CG_OP(a, b) __int_as_float(ETCH ? min(__float_as_int(a), __float_as_int(b)) : max(__float_as_int(a), __float_as_int(b)))
#define CG_ADD(x, a) (ETCH ? ((x) - (a)) : ((x) + (a)))
// four columns' tap at once: the two adds of a pair of columns are one packed instruction
#define CG_TAP4(src, dv, a)                                                                                   \
            do {                                                                                              \
                f32x2 x01 = {CG_OP(src[Rp + 0 + (dv)], src[Rp + 0 - (dv)]), CG_OP(src[Rp + 1 + (dv)], src[Rp + 1 - (dv)])}; \
                f32x2 x23 = {CG_OP(src[Rp + 2 + (dv)], src[Rp + 2 - (dv)]), CG_OP(src[Rp + 3 + (dv)], src[Rp + 3 - (dv)])}; \
                const f32x2 aa = {(a), (a)};                                                                  \
                x01 = ETCH ? x01 - aa : x01 + aa; x23 = ETCH ? x23 - aa : x23 + aa;                           \
                m[0] = CG_OP(m[0], x01[0]); m[1] = CG_OP(m[1], x01[1]); m[2] = CG_OP(m[2], x23[0]); m[3] = CG_OP(m[3], x23[1]); \
            } while (0)
            // centre row
#pragma unroll
            for (int c = 0; c < 4; c++) m[c] = CG_OP(m[c], CG_ADD(hc[Rp + c], tab.a[0]));
#pragma unroll
            for (int dv = 1; dv <= R; dv++) {
                if (dv > tab.lim[0]) break;
                const float a = tab.a[dv];
                CG_TAP4(hc, dv, a);
            }
            // row pairs +-du: v[k] = op(row+du[k], row-du[k]) is shared by every dv
#pragma unroll
            for (int du = 1; du <= R; du++) {
                if (tab.lim[du] < 0) continue;
                float v[W];
#pragma unroll
                for (int k = 0; k < W / 4; k++) {
                    const float4 p = *(const float4*)(rowc + du * PITCH + 4 * k);
                    const float4 n = *(const float4*)(rowc - du * PITCH + 4 * k);
                    v[4 * k] = CG_OP(p.x, n.x); v[4 * k + 1] = CG_OP(p.y, n.y); v[4 * k + 2] = CG_OP(p.z, n.z); v[4 * k + 3] = CG_OP(p.w, n.w);
                }
                const float a0 = tab.a[du * (R + 1)];
#pragma unroll
                for (int c = 0; c < 4; c++) m[c] = CG_OP(m[c], CG_ADD(v[Rp + c], a0));
#pragma unroll
                for (int dv = 1; dv <= R; dv++) {
                    if (dv > tab.lim[du]) break;
                    const float a = tab.a[du * (R + 1) + dv];
                    CG_TAP4(v, dv, a);
                }
            }
            // taps whose increment is too fine for float32: double, like the reference
            if (ntaps64 > 0) {
                double md[4];
#pragma unroll
                for (int c = 0; c < 4; c++) md[c] = (double)m[c];
                const float* ctr = s_h + (ry + R) * PITCH + Rp + 4 * gx;
                for (int q = 0; q < ntaps64; q++) {
                    const int off = taps64[q].off; const double add = (double)taps64[q].add;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const double hv = (double)ctr[c + off];
                        const double cand = ETCH ? hv - add : hv + add;
                        if (ETCH) { if (cand < md[c]) md[c] = cand; } else { if (cand > md[c]) md[c] = cand; }
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; c++) m[c] = (float)md[c];
            }
#undef CG_TAP4
#undef CG_OP
#undef CG_ADD
            // ---- per-column update
            double dz[4] = {0, 0, 0, 0};
            unsigned open = 0;
            if (STATS) loc[0] += __popc(act);
            if (!ETCH) {
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (!((act >> c) & 1)) continue;
                    const double d = (double)m[c] - (double)own[c];          // cg1:534-537
                    if (d > CG_EPS) { dz[c] = d; open |= 1u << c; }
                }
                if (open) stored = depo_group4<STATS>(g, c0, open, dz, matId, tv, loc[1], loc[2]);
            } else {
                // every etchable column needs the bottom of its top layer: for the floor (when a
                // material is named) and for the cut itself
                const uint32_t r4[4] = {tv[0], tv[1], tv[2], tv[3]};
                uint32_t b4[4] = {0, 0, 0, 0};
                load_below4(g, c0, act, r4, b4);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (!((act >> c) & 1)) continue;
                    // floor = bottom of the top run of matId (cg1:646-660)
                    float F = 0.f;
                    if (!isAll) F = (float)fmax(0.0, floor_of_top_run(g, c0 + c, matId, tv[c], b4[c], loc[3]));
                    const float hpv = fmaxf(0.f, fmaxf(F, m[c]));                     // cg1:725
                    const double d = (double)own[c] - (double)hpv;
                    if (d > CG_EPS) { dz[c] = d; open |= 1u << c; }
                }
                if (open) stored = etch_group4<STATS>(g, c0, open, dz, matId, isAll, tv, act, b4, loc[1], loc[2]);
            }
        }
        if (!stored && g.top_w != g.top) st_rec4(g.top_w + c0, tv);       // columns that stay as they are
    }
    }
    if (STATS) {
        if (ETCH) { const int slot[4] = {ST_OPEN, ST_ETCHED, ST_REMOVED, ST_FLOOR}; stats_commit<4>(gstats, loc, slot); }
        else { const int slot[4] = {ST_OPEN, ST_EXTEND, ST_APPEND, ST_FLOOR}; stats_commit<4>(gstats, loc, slot); }
    }
}

// ------------------------------------------------------------------------------------------
// lateral-stencil steps, row-streaming form (radius known at compile time)
// ------------------------------------------------------------------------------------------
// One wave owns a strip of STRIP columns (4 per lane) and walks down the rows of its tile.  Every
// input row is staged once (one LDS row, read back with its halo columns) and SCATTERS into the
// 2R+1 output rows it reaches, instead of every output row gathering 2R+1 input rows:
//   p[dv]   = op(h[j+dv], h[j-dv])                       shared by every du          (R ops per column)
//   D[du]   = op over dv of (p[dv] + a(du, dv))          serves output rows r-du and r+du
//   A[k]    = op(A[k+1], D[|R-k|]),  A[2R] = D[R]        the running results of the output rows centred
//                                                        on input rows r-R .. r+R; the shift through the
//                                                        registers rides on the three-operand op
// After input row r the output row centred on r-R is complete (A[0]) and is updated like in
// k_stencil_sym.  About 200 vector instructions per input row and 4 columns at R = 5, against
// 340 for the gathering form, one LDS row write and W/4 reads per row instead of (2R+1) W/4 reads,
// and halo rows are staged once per ti rows.  Same arithmetic (float adds, integer op on the bit
// patterns, see k_stencil_sym), only the order of the maxima differs, which cannot change them.
constexpr int STRIP = 256;
template <int R> struct StreamGeom {
    static constexpr int Rp = (R + 3) & ~3;          // halo columns rounded up to whole groups
    static constexpr int PITCH = STRIP + 2 * Rp;     // floats per staged row
    static constexpr int W = 4 + 2 * Rp;             // floats a lane reads per row
    static constexpr int HG = Rp / 4;                // halo groups on each side of the strip
};

// 4 consecutive columns of a staged row: top records and mask bytes.  A row of a neighbour's message
// (u16 heights + flag bytes) is turned into records of the step's material on arrival, so that one
// code path stages both; such rows are few (the R rows next to a slab edge).
struct RawGroup { u32x4 a; unsigned m; };

// MSG = false: the tile reaches no neighbour's rows; rows beyond the slab repeat its edge row
// (cg1:381-387) with flags zero (cg1:671).  No branch: every lane always issues the same two loads.
template <bool ETCH, bool MSG, bool MASKED>
__device__ __forceinline__ RawGroup stream_load(const GridView& g, const uint8_t* __restrict__ mask, int il, int Rimg, size_t msg_n, int jc, int matId,
                                                const uint8_t* __restrict__ halo_lo, const uint8_t* __restrict__ halo_up) {
    RawGroup o; o.a = (u32x4)(0); o.m = 0;
    if (MSG) {
        const RowSrc src = row_source(g, il, Rimg, halo_lo, halo_up);
        if (src.msg) {
            if (src.row >= 0 && src.row < Rimg) {
                const size_t p = (size_t)src.row * g.ny + jc;
                const uint2 h = *(const uint2*)((const uint16_t*)src.msg + p);
                o.a[0] = rec_make(h.x & 0xffffu, matId); o.a[1] = rec_make(h.x >> 16, matId);
                o.a[2] = rec_make(h.y & 0xffffu, matId); o.a[3] = rec_make(h.y >> 16, matId);
                if (ETCH) o.m = *(const unsigned*)(src.msg + 2 * msg_n + p);
            }
            return o;
        }
    }
    const bool edge = il < 0 || il >= g.nx;
    const size_t p = (size_t)(il < 0 ? 0 : (il >= g.nx ? g.nx - 1 : il)) * g.ny + jc;
    o.a = ld_rec4(g.top + p);
    if (ETCH) {
        if (MASKED) { const unsigned w = *(const unsigned*)(mask + p); o.m = edge ? 0u : w; }      // (loaded either way: no branch)
        else o.m = edge ? 0u : 0x01010101u;
    }
    return o;
}

// ONE column of a staged row, same rules: the halo columns left and right of a strip are staged one per lane (2 Rp lanes
// busy), because whatever those few lanes execute the whole wave pays for: as groups of 4 they cost as many
// instructions per row as the staging of the strip itself.
struct RawCol { uint32_t a; unsigned m; };
template <bool ETCH, bool MSG, bool MASKED>
__device__ __forceinline__ RawCol stream_load1(const GridView& g, const uint8_t* __restrict__ mask, int il, int Rimg, size_t msg_n, int jc, int matId,
                                               const uint8_t* __restrict__ halo_lo, const uint8_t* __restrict__ halo_up) {
    RawCol o; o.a = 0; o.m = 0;
    if (MSG) {
        const RowSrc src = row_source(g, il, Rimg, halo_lo, halo_up);
        if (src.msg) {
            if (src.row >= 0 && src.row < Rimg) {
                const size_t p = (size_t)src.row * g.ny + jc;
                o.a = rec_make(((const uint16_t*)src.msg)[p], matId);
                if (ETCH) o.m = src.msg[2 * msg_n + p];
            }
            return o;
        }
    }
    const bool edge = il < 0 || il >= g.nx;
    const size_t p = (size_t)(il < 0 ? 0 : (il >= g.nx ? g.nx - 1 : il)) * g.ny + jc;
    o.a = g.top[p];
    if (ETCH) {
        if (MASKED) { const unsigned w = mask[p]; o.m = edge ? 0u : w; }
        else o.m = edge ? 0u : 1u;
    }
    return o;
}

// What the host has verified for this handle and step, each a licence to replace double arithmetic
// by float arithmetic with the same result:
//   deq_f32    fmaf(q, zs_hi, q*zs_lo) == float32(double(q)*zscale) for every q < 65536 (checked one by one)
//   skip_init  the step's thickness is a float32 number and equals the centre increment of the table: the
//              start value float32(H + t) (cg1:499) is then the centre tap's candidate and need not be made
//   grow_f32   zscale is large enough that two different float32 heights differ by more than the
//              reference's 1e-9: "Hp - H > 1e-9" (cg1:537) is then "Hp > H" on the float32 values
//   a00        the centre increment of the table (a float32-exact tap, or this kernel is not used)
//   fast_rows  etch: the rows of a tile are first tried without their neighbours (stream_etch_fast_rows)
// sink: 4 MiB nobody reads, for stores that must be issued but have nothing to say (see k_stencil_lean's deposit update).
struct StreamOpts { float zs_hi, zs_lo, izf; int deq_f32, skip_init, grow_f32; float a00; int fast_rows; uint32_t* sink; };

__device__ __forceinline__ float stream_deq(const GridView& g, const StreamOpts& o, unsigned q) {
    if (o.deq_f32) { const float qf = (float)q; return __builtin_fmaf(qf, o.zs_hi, qf * o.zs_lo); }
    return (float)((double)q * g.zs);
}

// float32(q*zscale) (cg1:465) of a raw group; left / right: the group lies beyond the first / last
// column and repeats the edge column (cg1:374-380); not-etchable cells become +inf (cg1:718)
template <bool ETCH>
__device__ __forceinline__ float4 stream_heights(const GridView& g, const StreamOpts& so, const RawGroup& w, bool left, bool right, int matId, bool isAll) {
    unsigned q[4], e = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        q[c] = rec_z1(w.a[c]);
        if (ETCH && ((w.m >> (8 * c)) & 0xffu) && etchable(w.a[c], matId, isAll)) e |= 1u << c;
    }
    if (left) { q[1] = q[0]; q[2] = q[0]; q[3] = q[0]; }
    if (right) { q[0] = q[3]; q[1] = q[3]; q[2] = q[3]; }
    float4 h;
    h.x = stream_deq(g, so, q[0]); h.y = stream_deq(g, so, q[1]); h.z = stream_deq(g, so, q[2]); h.w = stream_deq(g, so, q[3]);
    if (ETCH) {
        const float INF = __builtin_inff();
        if (left || right) e = 0;
        if (!(e & 1u)) h.x = INF; if (!(e & 2u)) h.y = INF; if (!(e & 4u)) h.z = INF; if (!(e & 8u)) h.w = INF;
    }
    return h;
}

// the same for one column (beyond: it lies past the first / last column and repeats the edge column, flags zero)
template <bool ETCH>
__device__ __forceinline__ float stream_height1(const GridView& g, const StreamOpts& so, const RawCol& w, bool beyond, int matId, bool isAll) {
    float h = stream_deq(g, so, rec_z1(w.a));
    if (ETCH && !(!beyond && w.m && etchable(w.a, matId, isAll))) h = __builtin_inff();
    return h;
}

// min of two floats as integers on their bit patterns: right for non-negative numbers and +inf; with a negative operand it
// returns SOME negative number, which is all its users need (they clamp at 0 right after: cg1:725).
__device__ __forceinline__ float stream_min_bits(float a, float b) { return __int_as_float(min(__float_as_int(a), __float_as_int(b))); }

// The isotropic etch's update of 4 columns of an output row: A0 = the eroded height from the taps (cg1:704-722), tv = the
// columns' top records, b4 = the records below them, act = the columns that are open and etchable.
// (stream_etch_decide works the new tops out into rn[] and returns the columns that changed; stream_etch_update stores them.)
__device__ __forceinline__ unsigned stream_etch_decide(const GridView& g, const StreamOpts& so, const size_t c0, const u32x4 tv, const unsigned act,
                                                       const uint32_t (&b4)[4], const float (&A0)[4], const int matId, const bool isAll, const double t,
                                                       uint32_t (&rn)[4], unsigned (&loc)[4]) {
    // _applyEtchAt (cg1:553-589) by dz = H - Hp (cg1:730-739) on 4 columns.  The usual outcomes -- a cut inside the
    // top layer, or the top layer going as a whole with nothing more to take below -- are decided here, in float
    // arithmetic where that provably gives the reference's numbers and otherwise by the reference's own double
    // expressions; a column that goes on into further layers, rests on a gap, or whose floor lies below the
    // layer under it takes the general form (etch_group4_regs).
    rn[0] = tv[0]; rn[1] = tv[1]; rn[2] = tv[2]; rn[3] = tv[3];
    double dzs[4] = {0, 0, 0, 0};
    unsigned walk = 0, general = 0, changed = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (!((act >> c) & 1)) continue;
        const uint32_t rc = tv[c], b = b4[c];
        const bool has_b = rec_lnk(rc) != LNK_NONE;
        const unsigned q1 = rec_z1(rc), bq = has_b ? rec_z1(b) : 0u;
        if (has_b && (rec_hole(b) || (!isAll && rec_mat(b) == matId))) { walk |= 1u << c; continue; }     // floor and cut need a look further down
        const float own = stream_deq(g, so, q1), bottom = stream_deq(g, so, bq);
        float m = A0[c];
        if (!so.skip_init) m = stream_min_bits(m, (float)((double)own - t));                                       // cg1:702
        const float hpv = fmaxf(0.f, fmaxf(isAll ? 0.f : bottom, m));                                    // cg1:725 (the floor is the layer's own bottom)
        const bool cut = so.grow_f32 ? own > hpv : (double)own - (double)hpv > CG_EPS;                    // cg1:731-733
        if (!cut) continue;
        // cg1:574: the cut stays inside the top layer iff remaining < thick - 1e-9, remaining = H - Hp, thick = z1 - z0.
        // H and `bottom` are the float32 of z1 and z0 (off by <= 2^-24 of them), so remaining - thick = -(Hp - bottom) + e,
        // |e| < 6e-8 (H + bottom): a gap of twice that and 1e-8 decides it; closer calls use the double expressions.
        bool partial = (hpv - bottom) > ((own + bottom) * 1.3e-7f + 1.0e-8f) * 1.01f;
        double rem = 0;
        if (!partial) {
            rem = (double)own - (double)hpv;
            const double thick = (double)q1 * g.zs - (double)bq * g.zs;
            partial = rem < thick - CG_EPS;
            if (!partial) {                                                                               // the layer goes (cg1:579-587)
                const double rest = rem - thick;
                if (has_b && rest > CG_EPS && (isAll || rec_mat(b) == matId)) { dzs[c] = rem; general |= 1u << c; continue; }
                rn[c] = has_b ? b : 0u; changed |= 1u << c;
                continue;
            }
        }
        // new top of the cut layer: Math.round((z1 - remaining)/zs) = round((Hp - e1)/zs), as in the deposit update
        const float y = hpv * so.izf, rr = __builtin_rintf(y);
        unsigned qn = (unsigned)(int)rr;
        if (!(__builtin_fabsf(y - rr) < 0.5f - (y * 1.4e-7f + (float)q1 * 6.2e-8f + 2e-6f)) || !(y < 1.0e6f))
            qn = quantz((double)q1 * g.zs - ((double)own - (double)hpv), g.zs, g.inv_zs);
        rn[c] = rec_with_z1(rc, qn); changed |= 1u << c;
    }
    if (walk) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (!((walk >> c) & 1)) continue;
            const float own = stream_deq(g, so, rec_z1(tv[c]));
            const float m = stream_min_bits(A0[c], (float)((double)own - t));
            float F = 0.f;                                           // floor = bottom of the top run of matId (cg1:646-660)
            if (!isAll) F = (float)fmax(0.0, floor_of_top_run(g, c0 + c, matId, tv[c], b4[c], loc[3]));
            const float hpv = fmaxf(0.f, fmaxf(F, m));
            const double d = (double)own - (double)hpv;
            if (d > CG_EPS) { dzs[c] = d; general |= 1u << c; }
        }
    }
    if (general) changed |= etch_group4_regs<false>(g, c0, general, dzs, matId, isAll, rn, act, b4, loc[2]);
    return changed;
}

__device__ __forceinline__ void stream_etch_update(const GridView& g, const StreamOpts& so, const size_t c0, const u32x4 tv, const unsigned act,
                                                   const uint32_t (&b4)[4], const float (&A0)[4], const int matId, const bool isAll, const double t,
                                                   const bool m_right, unsigned (&loc)[4]) {
    uint32_t rn[4];
    const unsigned changed = stream_etch_decide(g, so, c0, tv, act, b4, A0, matId, isAll, t, rn, loc);
    if (!m_right && (changed || g.top_w != g.top)) {
        u32x4 nv; nv[0] = rn[0]; nv[1] = rn[1]; nv[2] = rn[2]; nv[3] = rn[3];
        st_rec4(g.top_w + c0, nv);
    }
}

// An etchable column whose own candidate H - t (cg1:702, and the centre tap) already lies at or below its floor ends at
// max(0, floor) whatever its neighbours are: cg1:725 takes the maximum of floor and eroded height, and neighbours can only
// lower the latter.  That is the usual case wherever a step takes off a whole film with some over-etch.  A tile
// therefore starts without the stencil: its output rows are updated from their own columns alone, one after the other, for
// as long as no open column of the row could end above its floor; from the first row where one could, the streaming form
// takes over.  Returns the number of rows done.  Three rows are in flight: the tops of row o+2 and the records below the
// tops of row o+1 are requested before row o is looked at (rows past the tile's last repeat it).  (Chunks of four rows
// with two chunks in flight were no faster, and their registers cost the small radii a wave per SIMD.)
template <bool MASKED>
__device__ __forceinline__ int stream_etch_fast_rows(const GridView& g, const uint8_t* __restrict__ mask, const StreamOpts& so, const int i_base,
                                                     const int rows_out, const int m_jc, const bool m_right, const int matId, const bool isAll,
                                                     const double t, unsigned (&loc)[4]) {
    const int last = rows_out - 1;
    const auto col0 = [&](int o) { return (size_t)(i_base + min(o, last)) * g.ny + m_jc; };
    u32x4 tvA = ld_rec4(g.top + col0(0)), tvB = ld_rec4(g.top + col0(1));
    unsigned mkA = 0x01010101u, mkB = 0x01010101u;
    if (MASKED) { mkA = *(const unsigned*)(mask + col0(0)); mkB = *(const unsigned*)(mask + col0(1)); }
    uint32_t bA[4], bB[4];
    load_below4_flat(g, col0(0), tvA, bA);
    int o = 0;
    for (; o < rows_out; o++) {
        const u32x4 tvC = ld_rec4(g.top + col0(o + 2));
        unsigned mkC = 0x01010101u;
        if (MASKED) mkC = *(const unsigned*)(mask + col0(o + 2));
        load_below4_flat(g, col0(o + 1), tvB, bB);
        // ---- row o
        unsigned act = ((mkA & 0xffu) ? 1u : 0u) | ((mkA & 0xff00u) ? 2u : 0u) | ((mkA & 0xff0000u) ? 4u : 0u) | ((mkA & 0xff000000u) ? 8u : 0u);
#pragma unroll
        for (int c = 0; c < 4; c++) if (!etchable(tvA[c], matId, isAll)) act &= ~(1u << c);
        if (m_right) act = 0;
        float A0[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned need = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (!((act >> c) & 1)) continue;
            const uint32_t rc = tvA[c], b = bA[c];
            const bool has_b = rec_lnk(rc) != LNK_NONE;
            if (has_b && (rec_hole(b) || (!isAll && rec_mat(b) == matId))) { need |= 1u << c; continue; }      // the floor lies further down
            const float own = stream_deq(g, so, rec_z1(rc)), bottom = stream_deq(g, so, has_b ? rec_z1(b) : 0u);
            float m = own - so.a00;                                                                            // the centre tap (cg1:719)
            A0[c] = m;
            if (!so.skip_init) m = stream_min_bits(m, (float)((double)own - t));                               // cg1:702
            if (m > fmaxf(0.f, isAll ? 0.f : bottom)) need |= 1u << c;
        }
        if (__builtin_amdgcn_ballot_w64(need != 0) != 0) break;
        stream_etch_update(g, so, col0(o), tvA, act, bA, A0, matId, isAll, t, m_right, loc);
        tvA = tvB; mkA = mkB; tvB = tvC; mkB = mkC;
#pragma unroll
        for (int c = 0; c < 4; c++) bA[c] = bB[c];
    }
    return o;
}

// Could entry (du, dv) of the quadrant table lie inside the ellipse of a kernel of radius R?  The
// ellipse is a circle of radius rho in cells, R <= rho < R+1.  Entries that cannot are skipped at
// compile time; the others are all evaluated (those outside a particular kernel hold -inf, the
// identity of the op), which keeps the loop body free of branches.
constexpr bool stream_entry(int R, int du, int dv) { return du * du + dv * dv < (R + 1) * (R + 1); }
constexpr int STREAM_TAB_PITCH = 12;                // floats per table row in LDS (R + 1 <= 9, rounded up to whole float4s)

template <bool B> struct BoolTag { static constexpr bool value = B; };

// Block = one wave.  Tile = g.ti rows x STRIP columns; tile_of_block names it.
// quad: the (R+1) x (R+1) quadrant table in device memory (symmetric, since dx == dy).
// LIC: the host has found every licence of StreamOpts that depends on zscale alone to hold (deq_f32, grow_f32), the usual
// case; they are then compile-time facts here instead of tests in the row loop (there was a branch in front of every
// de-quantisation).
template <int R, bool ETCH, bool MSG, bool MASKED, bool LIC = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3))) void k_stencil_stream(GridView g, const uint8_t* __restrict__ mask, int matId, double t, int Rimg,
                                                       const uint8_t* __restrict__ halo_lo, const uint8_t* __restrict__ halo_up,
                                                       const float* __restrict__ quad, const StreamOpts so_arg, int* __restrict__ flags, unsigned long long* gstats) {
    StreamOpts so = so_arg;
    if (LIC) { so.deq_f32 = 1; so.grow_f32 = 1; }
    using G = StreamGeom<R>;
    constexpr int Rp = G::Rp, PITCH = G::PITCH, W = G::W, TP = STREAM_TAB_PITCH;
    __shared__ __align__(16) float s_row[2][PITCH];
    __shared__ __align__(16) float s_tab[(R + 1) * TP];
    // The top records (etch: and mask bytes) of the last R + 2 input rows, as loaded for staging: input row r - R is the
    // output row that input row r completes, and its records come from here instead of a second trip to memory (which
    // missed the L2 more often than not: 2R rows of every resident wave lie between the two uses).
    // (Deposits up to R = 6.  Beyond, the ring would cost a resident wave per compute unit, and those kernels wait on
    // arithmetic; the etch, whose update starts with loads that depend on these records, measured 6 % slower with it.)
    constexpr bool RINGED = !ETCH && R <= 6;
    constexpr int RING = RINGED ? R + 2 : 1;
    __shared__ __align__(16) uint32_t s_top[RING][64 * 4];
    __shared__ unsigned s_msk[ETCH ? RING : 1][64];
    const int tiles_j = (g.ny + STRIP - 1) / STRIP;
    int tile_i, tile_j;
    if (!tile_of_block(g, tiles_j, tile_i, tile_j)) return;
    int i_base = tile_i * g.ti, rows_out = min(g.ti, g.nx - i_base);
    const int j_base = tile_j * STRIP;
    const int lane = threadIdx.x;
    const int j0 = j_base + 4 * lane;
    const bool isAll = (matId == 255);
    const size_t msg_n = (size_t)Rimg * g.ny;
    for (int e = lane; e < (R + 1) * TP; e += 64) {
        const int dv = e / TP, du = e - dv * TP;
        s_tab[e] = du <= R ? quad[dv * (R + 1) + du] : -__builtin_inff();
    }
    // this lane's own group (beyond the last column it repeats the edge: it is halo of the lanes before it) ...
    const bool m_right = j0 >= g.ny;
    const int m_jc = m_right ? g.ny - 4 : j0;
    // ... and, for the first 2 Rp lanes, one halo column left or right of the strip (the other lanes load the last of
    // those columns along with them and drop it: a load inside a branch costs a wait for it right there)
    const bool has_halo = lane < 2 * Rp;
    const int hl = min(lane, 2 * Rp - 1);
    const bool h_upper = hl >= Rp;
    const int hj = h_upper ? j_base + STRIP + (hl - Rp) : j_base - Rp + hl;
    const bool h_beyond = hj < 0 || hj >= g.ny;
    const int h_jc = hj < 0 ? 0 : (hj >= g.ny ? g.ny - 1 : hj);
    const int h_at = h_upper ? STRIP + hl : hl;
    unsigned loc[4] = {0, 0, 0, 0};      // (statistics are not kept by this form: with them enabled the host runs k_stencil_sym)

    if (ETCH && so.fast_rows) {          // the rows at the head of the tile that need no neighbours
        const int done = stream_etch_fast_rows<MASKED>(g, mask, so, i_base, rows_out, m_jc, m_right, matId, isAll, t, loc);
        if (done == rows_out) return;
        i_base += done; rows_out -= done;
    }
    const int rows_in = rows_out + 2 * R;

    float A[2 * R + 1][4];
#pragma unroll
    for (int k = 0; k <= 2 * R; k++) { A[k][0] = 0.f; A[k][1] = 0.f; A[k][2] = 0.f; A[k][3] = 0.f; }

    // Software pipeline.  While the taps of input row r are evaluated (the long part), the loads of row r+2
    // and of the columns of the next output row are in flight, and the stores of the output row before
    // are draining: the one wait on vector memory in an iteration comes after the taps.
    RawGroup in_m = stream_load<ETCH, MSG, MASKED>(g, mask, i_base - R, Rimg, msg_n, m_jc, matId, halo_lo, halo_up);
    RawCol in_h = stream_load1<ETCH, MSG, MASKED>(g, mask, i_base - R, Rimg, msg_n, h_jc, matId, halo_lo, halo_up);
    *(float4*)(s_row[0] + Rp + 4 * lane) = stream_heights<ETCH>(g, so, in_m, false, m_right, matId, isAll);
    if (has_halo) s_row[0][h_at] = stream_height1<ETCH>(g, so, in_h, h_beyond, matId, isAll);
    if (RINGED) { *(u32x4*)(s_top[0] + 4 * lane) = in_m.a; if (ETCH) s_msk[0][lane] = in_m.m; }
    if (1 < rows_in) {
        in_m = stream_load<ETCH, MSG, MASKED>(g, mask, i_base - R + 1, Rimg, msg_n, m_jc, matId, halo_lo, halo_up);
        in_h = stream_load1<ETCH, MSG, MASKED>(g, mask, i_base - R + 1, Rimg, msg_n, h_jc, matId, halo_lo, halo_up);
    }

#define CG_OP(a, b) __int_as_float(ETCH ? min(__float_as_int(a), __float_as_int(b)) : max(__float_as_int(a), __float_as_int(b)))
    // One input row.  FIN: the row completes an output row (every row from the 2R-th on), which is then updated.
    // The two cases are separate loops so that in the second one every path consumes what it loads: a load whose
    // use sits in a branch makes the compiler wait for ALL vector memory traffic -- the stores of the row
    // before included -- at the loop head.
    auto step = [&](const int r, auto fin_tag) __attribute__((always_inline)) {
        constexpr bool FIN = decltype(fin_tag)::value;
        const float* buf = s_row[r & 1];
        // The wave's LDS accesses execute in order, and the compiler keeps stores and loads of the one array in
        // program order: no fence is needed between the store of a row and its loads by other lanes of the wave.
        __builtin_amdgcn_wave_barrier();
        float hc[W];
        {
            const float* rowc = buf + 4 * lane;
#pragma unroll
            for (int k = 0; k < W / 4; k++) { const float4 v = *(const float4*)(rowc + 4 * k); hc[4 * k] = v.x; hc[4 * k + 1] = v.y; hc[4 * k + 2] = v.z; hc[4 * k + 3] = v.w; }
        }
        // the columns of the output row that this input row completes: needed after the taps (lanes beyond the
        // last column load a copy of the edge group: every lane takes part, only their stores are left out)
        const size_t c0 = (size_t)(i_base + (FIN ? r - 2 * R : 0)) * g.ny + m_jc;
        u32x4 tv = (u32x4)(0); unsigned mk = 0x01010101u;
        if (FIN && !RINGED) tv = ld_rec4(g.top + c0);
        if (FIN && MASKED && !(ETCH && RINGED)) mk = *(const unsigned*)(mask + c0);
        // ---- D[du] = op over dv of (p[dv] + a(du, dv)); row dv of the (symmetric) table lists the du's
        float D[R + 1][4];
#pragma unroll
        for (int dv = 0; dv <= R; dv++) {
            float p[4];
#pragma unroll
            for (int c = 0; c < 4; c++) p[c] = dv == 0 ? hc[Rp + c] : CG_OP(hc[Rp + c + dv], hc[Rp + c - dv]);
            float a[TP];
#pragma unroll
            for (int k = 0; k < (R + 4) / 4; k++) { const float4 v = *(const float4*)(s_tab + dv * TP + 4 * k); a[4 * k] = v.x; a[4 * k + 1] = v.y; a[4 * k + 2] = v.z; a[4 * k + 3] = v.w; }
#pragma unroll
            for (int du = 0; du <= R; du++) {
                if (!stream_entry(R, du, dv)) continue;
                f32x2 x01 = {p[0], p[1]}, x23 = {p[2], p[3]};
                const f32x2 aa = {a[du], a[du]};
                x01 = ETCH ? x01 - aa : x01 + aa; x23 = ETCH ? x23 - aa : x23 + aa;
                if (dv == 0) { D[du][0] = x01[0]; D[du][1] = x01[1]; D[du][2] = x23[0]; D[du][3] = x23[1]; }
                else { D[du][0] = CG_OP(D[du][0], x01[0]); D[du][1] = CG_OP(D[du][1], x01[1]); D[du][2] = CG_OP(D[du][2], x23[0]); D[du][3] = CG_OP(D[du][3], x23[1]); }
            }
        }
        // ---- every output row in reach takes its D; slot k moves to slot k-1 on the way
#pragma unroll
        for (int k = 0; k < 2 * R; k++) {
            const int du = k < R ? R - k : k - R;
#pragma unroll
            for (int c = 0; c < 4; c++) A[k][c] = CG_OP(A[k + 1][c], D[du][c]);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) A[2 * R][c] = D[R][c];
        // (the taps stay here, before the wait on vector memory below: without this the compiler sinks them past it)
#pragma unroll
        for (int k = 0; k <= 2 * R; k++) { asm volatile("" : "+v"(A[k][0]), "+v"(A[k][1]), "+v"(A[k][2]), "+v"(A[k][3])); }
        // ---- the output row centred on input row r - R is complete in A[0]: which of its columns take part
        // (The output row's columns arrive here on every path: a use that sat only inside the branches below would
        // leave them "maybe pending" at the loop head and cost a wait for everything there.)
        if (FIN) {
            if (RINGED) {
                const int slot = (r - R) % RING;
                tv = *(const u32x4*)(s_top[slot] + 4 * lane);
                if (ETCH) mk = s_msk[slot][lane];
            }
            asm volatile("" : "+v"(tv[0]), "+v"(tv[1]), "+v"(tv[2]), "+v"(tv[3]), "+v"(mk));
        }
        unsigned act = 0;
        uint32_t b4[4] = {0, 0, 0, 0};
        if (FIN) {
            act = ((mk & 0xffu) ? 1u : 0u) | ((mk & 0xff00u) ? 2u : 0u) | ((mk & 0xff0000u) ? 4u : 0u) | ((mk & 0xff000000u) ? 8u : 0u);
            if (ETCH) {
#pragma unroll
                for (int c = 0; c < 4; c++) if (!etchable(tv[c], matId, isAll)) act &= ~(1u << c);
            }
            if (m_right) act = 0;
            // etch: every etchable column needs the interval below its top (floor and cut); the loads travel while row r+1 is staged
            if (ETCH && act) load_below4_flat(g, c0, tv, b4);
        }
        // ---- stage input row r+1
        if (r + 1 < rows_in) {
            float* nxt = s_row[(r + 1) & 1];
            *(float4*)(nxt + Rp + 4 * lane) = stream_heights<ETCH>(g, so, in_m, false, m_right, matId, isAll);
            asm volatile("" : "+v"(in_h.a), "+v"(in_h.m));      // (looked at from here on: the compiler otherwise starts on it right behind its load, and waits there)
            if (has_halo) nxt[h_at] = stream_height1<ETCH>(g, so, in_h, h_beyond, matId, isAll);
            if (RINGED) {
                const int slot = (r + 1) % RING;
                *(u32x4*)(s_top[slot] + 4 * lane) = in_m.a;
                if (ETCH) s_msk[slot][lane] = in_m.m;
            }
        }
        // (the records loaded above are looked at from here on, not before the staging)
        if (ETCH && FIN) asm volatile("" : "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
        // ---- start the loads of row r+2.  Deposit: now, the update below loads nothing.  Etch: after the update, whose
        // waits for its own loads would otherwise wait for these as well.
        if (!(ETCH && FIN) && r + 2 < rows_in) {
            in_m = stream_load<ETCH, MSG, MASKED>(g, mask, i_base - R + r + 2, Rimg, msg_n, m_jc, matId, halo_lo, halo_up);
            in_h = stream_load1<ETCH, MSG, MASKED>(g, mask, i_base - R + r + 2, Rimg, msg_n, h_jc, matId, halo_lo, halo_up);
        }
        // ---- the update of the output row
        if (FIN && !ETCH) {
            // _depoAtTop_ (cg1:397-426) by dz = Hp - H (cg1:534-540) on 4 columns, without double arithmetic
            // where float arithmetic provably gives the same numbers (StreamOpts, and the quantiser below)
            unsigned q[4]; float own[4], m[4];
#pragma unroll
            for (int c = 0; c < 4; c++) { q[c] = rec_z1(tv[c]); own[c] = stream_deq(g, so, q[c]); m[c] = A[0][c]; }
            if (!so.skip_init) {
#pragma unroll
                for (int c = 0; c < 4; c++) m[c] = CG_OP(m[c], (float)((double)own[c] + t));                 // cg1:499
            }
            unsigned grow = 0;                                  // columns with Hp - H > 1e-9 (cg1:537)
            if (so.grow_f32) {
#pragma unroll
                for (int c = 0; c < 4; c++) if (m[c] > own[c]) grow |= 1u << c;
            } else {
#pragma unroll
                for (int c = 0; c < 4; c++) if ((double)m[c] - (double)own[c] > CG_EPS) grow |= 1u << c;
            }
            grow &= act;
            bool stored = false;
            if (grow) {
                // New top height: Math.round((q*zs + (Hp - H)) / zs) (cg1:407-411, 117-119).  In real numbers the
                // argument is (Hp - e)/zs with e = H - q*zs, |e| <= 2^-24 q zs (H is the float32 of q*zs), and the
                // double operations move it by < 1e-10.  y = Hp * float32(1/zs) is within 1.2e-7 y of Hp/zs.  So when y
                // is farther than y*1.4e-7 + q*6.2e-8 + 2e-6 from a half-integer, the answer is the integer nearest y;
                // the few columns where it is not take the literal double form.
                unsigned qn[4], slow = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float y = m[c] * so.izf, rr = __builtin_rintf(y);
                    const float margin = y * 1.4e-7f + (float)q[c] * 6.2e-8f + 2e-6f;
                    qn[c] = (unsigned)(int)rr;
                    if (!(__builtin_fabsf(y - rr) < 0.5f - margin) || !(y < 1.0e6f)) slow |= 1u << c;
                }
                slow &= grow;
                if (slow) {
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        if ((slow >> c) & 1) qn[c] = quantz((double)q[c] * g.zs + ((double)m[c] - (double)own[c]), g.zs, g.inv_zs);
                }
                u32x4 nv = tv;
                unsigned push = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (!((grow >> c) & 1)) continue;
                    const uint32_t rc = tv[c];
                    if (rc != 0 && rec_mat(rc) == matId) nv[c] = rec_with_z1(rc, qn[c]);
                    else { nv[c] = rec_make_l(qn[c], matId, rc != 0 ? g.P : LNK_NONE); if (rc != 0) push |= 1u << c; }
                }
                if (push) {                                     // the old tops go into this step's plane as they are (see depo_group4)
                    const size_t b = (size_t)g.P * g.N + c0;
                    st_rec4(g.rec + b, tv);          // (the group goes out whole, see depo_group4)
                }
                st_rec4(g.top_w + c0, nv);
                stored = true;
            }
            if (!stored && !m_right && g.top_w != g.top) st_rec4(g.top_w + c0, tv);          // columns that stay as they are
        }
        if (FIN && ETCH) stream_etch_update(g, so, c0, tv, act, b4, A[0], matId, isAll, t, m_right, loc);
        if (ETCH && FIN && r + 2 < rows_in) {
            in_m = stream_load<ETCH, MSG, MASKED>(g, mask, i_base - R + r + 2, Rimg, msg_n, m_jc, matId, halo_lo, halo_up);
            in_h = stream_load1<ETCH, MSG, MASKED>(g, mask, i_base - R + r + 2, Rimg, msg_n, h_jc, matId, halo_lo, halo_up);
        }
    };
    for (int r = 0; r < 2 * R; r++) step(r, BoolTag<false>{});
    for (int r = 2 * R; r < rows_in; r++) step(r, BoolTag<true>{});
#undef CG_OP
}

// ------------------------------------------------------------------------------------------
// mask rasteriser: _isPointBlocked / _pointInsideShape / _pointInPolygon (rt:54-141)
// ------------------------------------------------------------------------------------------
struct Shape {          // == cg_shape
    int32_t type, positive, rotated, pt_off, pt_cnt, _pad;
    double left, top, w, h, cosr, sinr, scale_x, scale_y, a, b;
};

__device__ __forceinline__ bool point_in_polygon(double x, double y, const double* __restrict__ pts, int n) {
    bool inside = false;
    for (int i = 0, j = n - 1; i < n; j = i++) {
        double xi = pts[2 * i], yi = pts[2 * i + 1], xj = pts[2 * j], yj = pts[2 * j + 1];
        bool hit = ((yi > y) != (yj > y)) && (x < (xj - xi) * (y - yi) / (yj - yi + 1e-9) + xi);
        if (hit) inside = !inside;
    }
    return inside;
}

__device__ __forceinline__ bool point_inside_shape(double x, double y, const Shape& s, const double* __restrict__ pts) {
    double dx = x - s.left, dy = y - s.top;
    double lx = dx, ly = dy;
    if (s.rotated) {
        lx = dx * s.cosr - dy * s.sinr;
        ly = dx * s.sinr + dy * s.cosr;
    }
    lx /= s.scale_x; ly /= s.scale_y;
    switch (s.type) {
    case 0: return lx >= 0 && lx <= s.w && ly >= 0 && ly <= s.h;
    case 1: { double cx = s.w / 2, cy = s.h / 2; return (lx - cx) * (lx - cx) + (ly - cy) * (ly - cy) <= s.a * s.a; }
    case 2: { double ex = lx - s.w / 2, ey = ly - s.h / 2; return (ex * ex) / (s.a * s.a) + (ey * ey) / (s.b * s.b) <= 1; }
    case 3: return point_in_polygon(lx, ly, pts + 2 * (size_t)s.pt_off, s.pt_cnt);
    default: return false;
    }
}

static __global__ __launch_bounds__(256) void k_mask_shapes(int nx, int ny, int i0, double dx, double dy, const Shape* __restrict__ shapes,
                                                     int nshapes, const double* __restrict__ pts, uint8_t* __restrict__ raster) {
    const size_t N = (size_t)nx * ny;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += stride) {
        int i = (int)(c / ny), j = (int)(c - (size_t)i * ny);
        double x = ((double)(i0 + i) + 0.5) * dx, y = ((double)j + 0.5) * dy;
        int open = 1;
        for (int s = nshapes - 1; s >= 0; s--) {
            if (point_inside_shape(x, y, shapes[s], pts)) { open = !shapes[s].positive; break; }
        }
        raster[c] = (uint8_t)open;
    }
}

// tiles of the binned rasteriser (cg_mask_bin.hip.h)
constexpr int MASK_TILE = 32;

// out[0] = 1 if any byte of the raster is non-zero (n bytes; 4 at a time, then the tail)
static __global__ __launch_bounds__(256) void k_mask_any(const uint8_t* __restrict__ m, size_t n, unsigned* __restrict__ out) {
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned any = 0;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) any |= ((const unsigned*)m)[q];
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) any |= m[4 * n4 + threadIdx.x];
    if (any) out[0] = 1;
}

static __global__ __launch_bounds__(256) void k_mask_normalise(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += stride) out[c] = in[c] ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// reductions and checks
// ------------------------------------------------------------------------------------------
// Monotone device-wide maximum: one same-address atomic per wave serialises (a million of them cost
// a few tenths of a millisecond), so a wave first looks at the current value and only joins in when it
// would change it.
__device__ __forceinline__ void lazy_umax(unsigned* p, unsigned v) {
    if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v);
}

// out[0] = max top z1q, out[1] = max len (len must be current, see k_relen)
static __global__ __launch_bounds__(256) void k_max_top(GridView g, unsigned* __restrict__ out) {
    unsigned mq = 0, ml = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        int l = g.len[c];
        if (l > 0) {
            const unsigned q = rec_z1(g.split ? g.top[c] : g.rec[(size_t)(l - 1) * g.N + c]);
            mq = q > mq ? q : mq; ml = (unsigned)l > ml ? (unsigned)l : ml;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        unsigned a = __shfl_down(mq, o, 64), b = __shfl_down(ml, o, 64);
        mq = a > mq ? a : mq; ml = b > ml ? b : ml;
    }
    if ((threadIdx.x & 63) == 0) { lazy_umax(&out[0], mq); lazy_umax(&out[1], ml); }
}

// split form: out[0] = max top z1q from the top stream alone (an empty column's top is 0)
static __global__ __launch_bounds__(256) void k_max_top_split(GridView g, unsigned* __restrict__ out) {
    unsigned mq = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        const uint32_t r = g.top[c];
        const unsigned q = r ? rec_z1(r) : 0u;
        mq = q > mq ? q : mq;
    }
    for (int o = 32; o > 0; o >>= 1) { unsigned a = __shfl_down(mq, o, 64); mq = a > mq ? a : mq; }
    if ((threadIdx.x & 63) == 0) lazy_umax(&out[0], mq);
}

// whole form: top[c] = the column's top interval (0: empty), so that a lateral step can stage the OLD
// heights from the top stream while it rewrites the planes in place
static __global__ __launch_bounds__(256) void k_snapshot_tops(GridView g) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        const int l = g.len[c];
        g.top[c] = l > 0 ? (g.rec[(size_t)(l - 1) * g.N + c] & REC_BODY & ~REC_MARK) : 0u;
    }
}

// split form: len[c] recounted from the chain (the 4-columns-per-thread kernels do not maintain it)
static __global__ __launch_bounds__(256) void k_relen(GridView g) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        g.len[c] = (uint8_t)chain_length(g, c);
    }
}

// whole form: out[0] = max number of layers (gap intervals not counted), out[1] = max len
static __global__ __launch_bounds__(256) void k_max_layers(GridView g, unsigned* __restrict__ out) {
    unsigned mw = 0, ml = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        const int l = g.len[c];
        unsigned w = 0;
        for (int k = 0; k < l; k++) w += rec_hole(g.rec[(size_t)k * g.N + c]) ? 0u : 1u;
        mw = w > mw ? w : mw; ml = (unsigned)l > ml ? (unsigned)l : ml;
    }
    for (int o = 32; o > 0; o >>= 1) {
        unsigned a = __shfl_down(mw, o, 64), b = __shfl_down(ml, o, 64);
        mw = a > mw ? a : mw; ml = b > ml ? b : ml;
    }
    if ((threadIdx.x & 63) == 0) { lazy_umax(&out[0], mw); lazy_umax(&out[1], ml); }
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Checksum over the reference's view of the columns.  With hg = mix(global column index), column c contributes
//     mix(hg + layers + sum over its layers of term(hg, mat | z0q << 8 | z1q << 24)),
// gaps not counted.  The inner sum lets the intervals come in any order -- bottom-up through the planes of the whole
// form, top-down along the chain of the split form -- with every record loaded once and nothing kept per column;
// term() is two 32-bit multiplies (one-to-one in the interval for a given hg), the 64-bit mixing is spent once per column.
__device__ __forceinline__ unsigned long long checksum_term(unsigned long long hg, uint32_t r, unsigned z0q) {
    const uint32_t lo = rec_mat(r) | (z0q << 8) | (rec_z1(r) << 24), hi = rec_z1(r) >> 8;
    const uint32_t a = (lo ^ (uint32_t)hg) * 0x9E3779B1u;
    const uint32_t b = (hi ^ (uint32_t)(hg >> 32) ^ (a >> 16)) * 0x85EBCA77u;
    return ((unsigned long long)a << 32) | b;
}
// A lane walks CHECKSUM_U columns side by side (a block takes CHECKSUM_U runs of 256 neighbouring columns), so that
// CHECKSUM_U dependent loads are in flight per lane instead of one; every load of a wave is of neighbouring columns.
constexpr int CHECKSUM_U = 8;
static __global__ __launch_bounds__(256) void k_checksum(GridView g, unsigned long long* __restrict__ out) {
    constexpr int U = CHECKSUM_U;
    unsigned long long sum = 0;
    const size_t span = (size_t)U * blockDim.x;
    for (size_t base = (size_t)blockIdx.x * span; base < g.N; base += (size_t)gridDim.x * span) {
        size_t c[U]; unsigned long long hg[U], h[U]; bool in[U];
        #pragma unroll
        for (int q = 0; q < U; q++) {
            c[q] = base + (size_t)q * blockDim.x + threadIdx.x; in[q] = c[q] < g.N;
            if (!in[q]) c[q] = 0;
            hg[q] = splitmix64((unsigned long long)g.i0 * g.ny + c[q]); h[q] = hg[q];
        }
        if (!g.split) {
            int l[U], lm = 0; unsigned prev[U];
            #pragma unroll
            for (int q = 0; q < U; q++) { l[q] = in[q] ? (int)g.len[c[q]] : 0; lm = max(lm, l[q]); prev[q] = 0; }
            for (int k = 0; k < lm; k++) {
                uint32_t r[U];
                #pragma unroll
                for (int q = 0; q < U; q++) r[q] = k < l[q] ? g.rec[(size_t)k * g.N + c[q]] : 0u;
                #pragma unroll
                for (int q = 0; q < U; q++) {
                    if (k >= l[q]) continue;
                    if (!rec_hole(r[q])) h[q] += checksum_term(hg[q], r[q], prev[q]) + 1;
                    prev[q] = rec_z1(r[q]);
                }
            }
        } else {
            uint32_t r[U];
            #pragma unroll
            for (int q = 0; q < U; q++) r[q] = in[q] ? g.top[c[q]] : 0u;
            for (int guard = 0; guard < 256; guard++) {
                bool any = false;
                #pragma unroll
                for (int q = 0; q < U; q++) any |= r[q] != 0;
                if (!any) break;
                uint32_t below[U];
                #pragma unroll
                for (int q = 0; q < U; q++) {
                    const int p = rec_lnk(r[q]);
                    below[q] = (r[q] != 0 && p != LNK_NONE) ? g.rec[(size_t)p * g.N + c[q]] : 0u;
                }
                #pragma unroll
                for (int q = 0; q < U; q++) {
                    if (r[q] != 0 && !rec_hole(r[q])) h[q] += checksum_term(hg[q], r[q], rec_z1(below[q])) + 1;
                    r[q] = below[q];
                }
            }
        }
        #pragma unroll
        for (int q = 0; q < U; q++) if (in[q]) sum += splitmix64(h[q]);
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, sum);
}

// ------------------------------------------------------------------------------------------
// CMP (script_columngrid.js:747-859)
// ------------------------------------------------------------------------------------------
// out[0] = max z1q over every layer of material stopMat (0 when there is none), out[1] = 1 if any
static __global__ __launch_bounds__(256) void k_cmp_stopper_top(GridView g, int stopMat, unsigned* __restrict__ out) {
    unsigned mq = 0, any = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        const int l = min((int)g.len[c], g.lmax);
        for (int k = 0; k < l; k++) {
            const uint32_t r = g.rec[(size_t)k * g.N + c];
            if (!rec_hole(r) && rec_mat(r) == stopMat) { any = 1; mq = max(mq, rec_z1(r)); }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { mq = max(mq, (unsigned)__shfl_down(mq, o, 64)); any |= __shfl_down(any, o, 64); }
    if ((threadIdx.x & 63) == 0 && any) { lazy_umax(&out[0], mq); lazy_umax(&out[1], 1u); }
}

// per-column truncation to limit = max(targetZ, z1 of the column's lowest stopper layer)
static __global__ __launch_bounds__(256) void k_cmp_apply(GridView g, double targetZ, int stopMat) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        int l = g.len[c];
        if (l == 0) continue;
        double stopZ = -__builtin_inf();
        if (stopMat != 0)
            for (int k = 0; k < l; k++) {
                const uint32_t r = g.rec[(size_t)k * g.N + c];
                if (!rec_hole(r) && rec_mat(r) == stopMat) { stopZ = (double)rec_z1(r) * g.zs; break; }      // cg1:801-806
            }
        const double limitZ = fmax(targetZ, stopZ);
        const int l0 = l;
        while (l > 0) {                                                                                      // cg1:813-834
            const size_t b = (size_t)(l - 1) * g.N + c;
            const uint32_t r = g.rec[b];
            if (rec_hole(r)) { g.rec[b] = 0; l--; continue; }          // a gap uncovered by the removal above it
            const uint32_t below = l > 1 ? g.rec[(size_t)(l - 2) * g.N + c] : 0;
            const double z1 = (double)rec_z1(r) * g.zs, z0 = (double)rec_z1(below) * g.zs;
            if (z1 <= limitZ + CG_EPS) break;
            if (z0 < limitZ - CG_EPS) { g.rec[b] = rec_with_z1(r, quantz(limitZ, g.zs, g.inv_zs)); break; }
            g.rec[b] = 0; l--;
        }
        if (l > 0) {                                                                                         // cg1:840-853
            const size_t b = (size_t)(l - 1) * g.N + c;
            const uint32_t r = g.rec[b];
            const uint32_t below = l > 1 ? g.rec[(size_t)(l - 2) * g.N + c] : 0;
            if (rec_z1(r) <= rec_z1(below)) {
                g.rec[b] = 0; l--;
                if (l > 0 && rec_hole(g.rec[(size_t)(l - 1) * g.N + c])) { g.rec[(size_t)(l - 1) * g.N + c] = 0; l--; }
            }
        }
        if (l != l0) g.len[c] = (uint8_t)l;
    }
}

// ------------------------------------------------------------------------------------------
// strip_connected (script_columngrid.js:867 - script_columngrid2.js:1) and identify_cavity
// (script_columngrid2.js:9-83)
// ------------------------------------------------------------------------------------------
// Nodes are the layers of material matId; a node is a seed when it is exposed to air, and the
// marked set is the closure of the seeds under "4-neighbour column, same material, z ranges
// overlap".  The mark lives in the record's flag byte.  All height tests are done on the
// quantised integers: the reference's eps (1e-9 nm) is far below one step, so
// a < b - eps on de-quantised values is a < b on the integers.
__device__ __forceinline__ unsigned z0_of(const GridView& g, size_t c, int k) { return k > 0 ? rec_z1(g.rec[(size_t)(k - 1) * g.N + c]) : 0u; }

// A slab of a decomposed domain sees one row of its lower and of its upper neighbour as
// "ghost" columns: len[ny] and rec[p*ny + j] for the planes exchanged (same record format, the
// MARK bits included).  Null pointers = no neighbour on that side (the domain ends there).
struct Ghost { const uint8_t* lo_len; const uint32_t* lo_rec; const uint8_t* up_len; const uint32_t* up_rec; int planes; };

// Column (ni, nj) as seen from this slab: ok == false beyond the domain.
struct ColRef { const uint32_t* rec; size_t stride; int len; bool ok; };
__device__ __forceinline__ ColRef column_ref(const GridView& g, const Ghost& gh, int ni, int nj) {
    ColRef r; r.rec = nullptr; r.stride = 0; r.len = 0; r.ok = false;
    if (nj < 0 || nj >= g.ny) return r;
    if (ni >= 0 && ni < g.nx) {
        const size_t n = (size_t)ni * g.ny + nj;
        r.rec = g.rec + n; r.stride = g.N; r.len = g.len[n]; r.ok = true;
    } else if (ni == -1 && gh.lo_len) {
        r.rec = gh.lo_rec + nj; r.stride = (size_t)g.ny; r.len = min((int)gh.lo_len[nj], gh.planes); r.ok = true;
    } else if (ni == g.nx && gh.up_len) {
        r.rec = gh.up_rec + nj; r.stride = (size_t)g.ny; r.len = min((int)gh.up_len[nj], gh.planes); r.ok = true;
    }
    return r;
}

// isExposed (cg1:882-905) for layer [z0q, z1q] of column (i, j)
__device__ __forceinline__ bool layer_exposed(const GridView& g, const Ghost& gh, int i, int j, unsigned z0q, unsigned z1q, unsigned topq) {
    if (z0q == 0) return false;                       // resting on the floor: not exposed
    if (z1q == topq) return true;                     // top of its column
    const int di[4] = {1, -1, 0, 0}, dj[4] = {0, 0, 1, -1};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const ColRef n = column_ref(g, gh, i + di[d], j + dj[d]);
        if (!n.ok) return false;                      // the domain edge ends the scan (cg1:896)
        if (n.len == 0) return true;
        if (rec_z1(n.rec[(size_t)(n.len - 1) * n.stride]) < z1q) return true;
    }
    return false;
}

// cand[c]: bit 0 while column c still has an unmarked node of matId (the passes that spread marks skip the others
// after reading one byte), bit 1 when it has any node of matId at all.  parent (may be null): the forest of
// k_strip_union, every unmarked node of matId its own root to begin with; with it, block b also lists the columns it
// found an unmarked node in, list[b*segcap ..] (segcap = the number of columns a block visits), counts[b] of them.
static __global__ __launch_bounds__(256) void k_strip_seed(GridView g, Ghost gh, int matId, uint8_t* __restrict__ cand, uint32_t* __restrict__ parent,
                                                    uint32_t* __restrict__ list, uint32_t* __restrict__ counts, size_t segcap) {
    __shared__ unsigned s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    // Runs of 256 columns are dealt to the blocks round by round, one place further (times 5) along every round: with a
    // fixed deal a block would meet the same stretch of j in every row it visits, and patterns that run along i would
    // give some blocks' lists many times the entries of others'.
    const size_t chunks = (g.N + 255) / 256, rounds = (chunks + gridDim.x - 1) / gridDim.x;
    for (size_t k = 0; k < rounds; k++) {
        const size_t q = k * gridDim.x + (blockIdx.x + 5 * k) % gridDim.x;
        const size_t c = q * 256 + threadIdx.x;
        if (c >= g.N) continue;
        const int l = g.len[c];
        if (l == 0) { cand[c] = 0; continue; }
        bool open_node = false, node = false;
        const int i = (int)(c / g.ny), j = (int)(c - (size_t)i * g.ny);
        const unsigned topq = rec_z1(g.rec[(size_t)(l - 1) * g.N + c]);
        unsigned prev = 0;
        for (int k = 0; k < l; k++) {
            const uint32_t r = g.rec[(size_t)k * g.N + c];
            if (!rec_hole(r) && rec_mat(r) == matId) {
                node = true;
                if (layer_exposed(g, gh, i, j, prev, rec_z1(r), topq)) g.rec[(size_t)k * g.N + c] = r | REC_MARK;
                else { open_node = true; if (parent) parent[(size_t)k * g.N + c] = (uint32_t)((size_t)k * g.N + c); }
            }
            prev = rec_z1(r);
        }
        cand[c] = (open_node ? 1 : 0) | (node ? 2 : 0);
        if (open_node && parent) list[(size_t)blockIdx.x * segcap + atomicAdd(&s_n, 1u)] = (uint32_t)c;
    }
    if (parent) {
        __syncthreads();
        if (threadIdx.x == 0) counts[blockIdx.x] = s_n;
    }
}

// identify_cavity (cg2:9-83) without the pass that lays a cavity layer on top of every column: k_cavity_fill followed by
// k_strip_seed(matId = 2) writes such a layer into nearly every column, marks it (the top of a column is exposed,
// cg1:889), and k_strip_compact takes it out again.  All that those layers do in between is to stand next to the real
// nodes -- the gaps and cavity layers inside the columns -- as marked neighbours; k_strip_union's air_above knows them
// without their being there.  This kernel is the two passes for the real nodes alone: every gap becomes a cavity layer
// (cg2:23-40), and every cavity layer is a node, marked when exposed.  With every column that is not empty reaching
// hmax_q after the filling pass, isExposed (cg1:882-905) comes down to: resting on the floor -- no; ending at hmax_q --
// yes; then the four neighbours in the reference's order: beyond the domain -- no, an empty column -- yes, anything else
// reaches hmax_q and decides nothing.  The layer the filling pass would lay on a column whose top is at height 0 rests
// on the floor and is NOT exposed: flags[0] reports such a column and the host takes the passes as they are written.
// cand / parent / list / counts as in k_strip_seed.
// A thread takes four neighbouring columns (N a multiple of 4): their lengths as one word, their intervals plane by plane
// as one 16-byte load; a block visits runs of 1024 columns, dealt as k_strip_seed deals its runs of 256.
static __global__ __launch_bounds__(256) void k_cavity_scan(GridView g, unsigned hmax_q, uint8_t* __restrict__ cand, uint32_t* __restrict__ parent,
                                                     uint32_t* __restrict__ list, uint32_t* __restrict__ counts, size_t segcap, int* __restrict__ flags) {
    __shared__ unsigned s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    int floor_top = 0;
    const size_t ngroups = g.N >> 2;
    const size_t chunks = (ngroups + 255) / 256, rounds = (chunks + gridDim.x - 1) / gridDim.x;
    for (size_t k = 0; k < rounds; k++) {
        const size_t q = k * gridDim.x + (blockIdx.x + 5 * k) % gridDim.x;
        const size_t gi = q * 256 + threadIdx.x;
        if (gi >= ngroups) continue;
        const size_t c0 = gi << 2;
        const unsigned lw = *(const unsigned*)(g.len + c0);
        if (lw == 0) { *(unsigned*)(cand + c0) = 0u; continue; }
        int l[4]; int lmx = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { l[j] = (int)((lw >> (8 * j)) & 0xffu); lmx = max(lmx, l[j]); }
        unsigned prev[4] = {0, 0, 0, 0}, z1[4] = {0, 0, 0, 0}, cw = 0;
        int by_nb[4] = {-1, -1, -1, -1};                       // exposed through a neighbour: not looked at yet
        for (int kk = 0; kk < lmx; kk++) {
            const u32x4 v = ld_rec4(g.rec + (size_t)kk * g.N + c0);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (kk >= l[j]) continue;
                uint32_t r = v[j];
                z1[j] = rec_z1(r);
                const bool hole = rec_hole(r);
                if (hole || rec_mat(r) == 2) {
                    const size_t c = c0 + j;
                    cw |= 2u << (8 * j);
                    if (hole) r = rec_make(z1[j], 2);          // cg2:23-40
                    bool ex = false;
                    if (prev[j] != 0) {
                        if (z1[j] == hmax_q) ex = true;
                        else {
                            if (by_nb[j] < 0) {
                                const int i = (int)(c / g.ny), jj = (int)(c - (size_t)i * g.ny);
                                const int di[4] = {1, -1, 0, 0}, dj[4] = {0, 0, 1, -1};
                                by_nb[j] = 0;
#pragma unroll
                                for (int d = 0; d < 4; d++) {
                                    const int ni = i + di[d], nj = jj + dj[d];
                                    if (ni < 0 || ni >= g.nx || nj < 0 || nj >= g.ny) break;                     // the domain edge ends the scan (cg1:896)
                                    if (g.len[(size_t)ni * g.ny + nj] == 0) { by_nb[j] = 1; break; }
                                }
                            }
                            ex = by_nb[j] != 0;
                        }
                    }
                    if (ex) { r |= REC_MARK; cw |= 4u << (8 * j); }
                    else { cw |= 1u << (8 * j); parent[(size_t)kk * g.N + c] = (uint32_t)((size_t)kk * g.N + c); }
                    if (hole || ex) g.rec[(size_t)kk * g.N + c] = r;
                }
                prev[j] = z1[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (l[j] > 0 && z1[j] == 0) floor_top = 1;         // (z1: the column's top by now)
            if (cw & (1u << (8 * j))) list[(size_t)blockIdx.x * segcap + atomicAdd(&s_n, 1u)] = (uint32_t)(c0 + j);
        }
        *(unsigned*)(cand + c0) = cw;
    }
    if (floor_top) flags[0] = 1;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = s_n;
}

// One relaxation sweep: an unmarked node next to an overlapping marked node becomes marked, seeing whatever marks of
// this same sweep have already become visible.  Marks only ever get added and the closure is unique, so the order in
// which they spread does not matter; blocks start roughly in index order, so a sweep carries marks a long way in the
// direction it runs, and the host alternates the direction (reverse: from the last column down) until a sweep changes
// nothing.  (Relaxing a tile to its own fixpoint inside one launch was tried: lanes of a wave run in step, a mark moves
// one column per pass along a row, and the tiles that do change cost up to a tile's width in passes: 2x slower overall.)
// before: the flag word of the sweep launched just before this one on the same stream, or null.  A sweep that changed
// nothing has found the fixpoint, so the sweeps queued behind it in the same batch return at once.
static __global__ __launch_bounds__(256) void k_strip_flood(GridView g, Ghost gh, int matId, uint8_t* __restrict__ cand, int reverse, int* __restrict__ changed,
                                                     const int* __restrict__ before) {
    if (before && __hip_atomic_load(before, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int any = 0;
    for (size_t cc = (size_t)blockIdx.x * blockDim.x + threadIdx.x; cc < g.N; cc += stride) {
        const size_t c = reverse ? g.N - 1 - cc : cc;
        if (!(cand[c] & 1)) continue;
        const int l = g.len[c];
        bool open_node = false;
        const int i = (int)(c / g.ny), j = (int)(c - (size_t)i * g.ny);
        unsigned prev = 0;
        for (int k = 0; k < l; k++) {
            const uint32_t r = g.rec[(size_t)k * g.N + c];
            const unsigned a0 = prev, a1 = rec_z1(r);
            prev = a1;
            if (rec_hole(r) || (r & REC_MARK) || rec_mat(r) != matId) continue;
            bool hit = false;
            const int di[4] = {1, -1, 0, 0}, dj[4] = {0, 0, 1, -1};
#pragma unroll
            for (int d = 0; d < 4 && !hit; d++) {
                const ColRef n = column_ref(g, gh, i + di[d], j + dj[d]);
                if (!n.ok) continue;
                unsigned b0 = 0;
                for (int kk = 0; kk < n.len; kk++) {
                    const uint32_t rb = n.rec[(size_t)kk * n.stride];
                    const unsigned b1 = rec_z1(rb);
                    if ((rb & REC_MARK) && !rec_hole(rb) && rec_mat(rb) == matId && a0 < b1 && b0 < a1) { hit = true; break; }   // cg1:879
                    b0 = b1;
                    if (b0 >= a1) break;              // everything above starts at or over our top
                }
            }
            if (hit) { g.rec[(size_t)k * g.N + c] = r | REC_MARK; any = 1; }
            else open_node = true;
        }
        if (!open_node) cand[c] = 2;
    }
    if (any) *changed = 1;
}

// ---- the same closure by connected components: a fixed number of launches, whatever the shape of the material ----
// Node (c, k) -- interval k of column c -- has the number k*N + c (the host checks that these stay below 2^32 - 1), and
// parent[] is a union-find forest over the unmarked nodes of matId in which a link always points to a higher number,
// so there are no cycles whatever the interleaving.  The highest number of all, UF_MARKED, stands for "marked": it has
// no slot in parent[] and is a root by definition.  k_strip_union joins every pair of overlapping unmarked nodes in
// neighbouring columns (each pair once: towards i+1 and j+1) and hangs a node that overlaps a marked one under
// UF_MARKED; afterwards the tree of UF_MARKED holds exactly the closure, and k_strip_spread marks every node whose root
// it is.  On a decomposed domain the forest is built once; after every exchange k_strip_ghost hangs the border nodes
// that touch a marked node of a neighbour's ghost row under UF_MARKED and k_strip_spread runs again.
constexpr uint32_t UF_MARKED = 0xffffffffu;
__device__ __forceinline__ uint32_t uf_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t uf_find(uint32_t* parent, uint32_t x) {
    if (x == UF_MARKED) return x;
    uint32_t p = uf_load(parent + x);
    while (p != x) {                                   // halve the path on the way up: x is not a root and never becomes one again
        if (p == UF_MARKED) return p;
        const uint32_t gp = uf_load(parent + p);
        if (gp != p) __hip_atomic_store(parent + x, gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p; p = gp;
    }
    return x;
}
__device__ __forceinline__ void uf_union(uint32_t* parent, uint32_t a, uint32_t b) {
    a = uf_find(parent, a); b = uf_find(parent, b);
    while (a != b) {
        if (a > b) { const uint32_t t = a; a = b; b = t; }                 // the lower root goes under the higher one
        const uint32_t old = atomicCAS(parent + a, a, b);                   // ... if it still is a root
        if (old == a) return;
        a = uf_find(parent, old); b = uf_find(parent, b);
    }
}

// Launched with the grid of k_strip_seed: block b works through the columns block b listed there, so every lane has a
// column with an unmarked node to look at (such columns are few, and a pass over all columns would spend its time on
// waves with two or three busy lanes).  A pair of unmarked nodes is met from both sides; the second meeting finds them
// joined.  A marked node needs no visit of its own: the unmarked nodes around it come to it.
// air_above: identify_cavity without its filling pass (k_cavity_scan): every column that is not empty counts as carrying
// a marked node of matId from its top up to the highest column of the domain, as k_cavity_fill + k_strip_seed would have
// left it there.  An unmarked node (a0, a1] overlaps that piece of a neighbour exactly when the neighbour's top lies
// below a1 (a1 never exceeds the highest column).
static __global__ __launch_bounds__(256) void k_strip_union(GridView g, int matId, const uint32_t* __restrict__ list, const uint32_t* __restrict__ counts, size_t segcap,
                                                     uint32_t* __restrict__ parent, int air_above) {
    const unsigned n = counts[blockIdx.x];
    const uint32_t* mine = list + (size_t)blockIdx.x * segcap;
    for (unsigned idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const size_t c = mine[idx];
        const int i = (int)(c / g.ny), j = (int)(c - (size_t)i * g.ny);
        const bool okn[4] = {i + 1 < g.nx, i > 0, j + 1 < g.ny, j > 0};
        const size_t nb[4] = {c + g.ny, c - g.ny, c + 1, c - 1};
        const int l = g.len[c];
        unsigned prev = 0;
        for (int k = 0; k < l; k++) {
            const uint32_t r = g.rec[(size_t)k * g.N + c];
            const unsigned a0 = prev, a1 = rec_z1(r);
            prev = a1;
            if (rec_hole(r) || (r & REC_MARK) || rec_mat(r) != matId) continue;
            const uint32_t ida = (uint32_t)((size_t)k * g.N + c);
            bool done = false;                           // under the mark: whoever else touches this node finds that out from its own side
            for (int d = 0; d < 4 && !done; d++) {
                if (!okn[d]) continue;
                const size_t nn = nb[d];
                const int ln = g.len[nn];
                unsigned b0 = 0;
                for (int kk = 0; kk < ln; kk++) {
                    const uint32_t rb = g.rec[(size_t)kk * g.N + nn];
                    const unsigned b1 = rec_z1(rb);
                    if (!rec_hole(rb) && rec_mat(rb) == matId && a0 < b1 && b0 < a1) {                                  // cg1:879
                        if (rb & REC_MARK) { uf_union(parent, ida, UF_MARKED); done = true; break; }
                        uf_union(parent, ida, (uint32_t)((size_t)kk * g.N + nn));
                    }
                    b0 = b1;
                    if (b0 >= a1) break;
                }
                if (air_above && !done && ln > 0 && rec_z1(g.rec[(size_t)(ln - 1) * g.N + nn]) < a1) { uf_union(parent, ida, UF_MARKED); done = true; }
            }
        }
    }
}

// border rows only: one thread per column of row 0 and one per column of row nx-1 (a slab of one row is seen by both)
static __global__ __launch_bounds__(256) void k_strip_ghost(GridView g, Ghost gh, int matId, const uint8_t* __restrict__ cand, uint32_t* __restrict__ parent) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * g.ny) return;
    const int side = e >= g.ny, j = e - side * g.ny;
    if (side ? !gh.up_len : !gh.lo_len) return;
    const int i = side ? g.nx - 1 : 0;
    const size_t c = (size_t)i * g.ny + j;
    if (!(cand[c] & 1)) return;
    const ColRef n = column_ref(g, gh, side ? g.nx : -1, j);
    if (!n.ok) return;
    const int l = g.len[c];
    unsigned prev = 0;
    for (int k = 0; k < l; k++) {
        const size_t id = (size_t)k * g.N + c;
        const uint32_t r = g.rec[id];
        const unsigned a0 = prev, a1 = rec_z1(r);
        prev = a1;
        if (rec_hole(r) || (r & REC_MARK) || rec_mat(r) != matId) continue;
        unsigned b0 = 0;
        for (int kk = 0; kk < n.len; kk++) {
            const uint32_t rb = n.rec[(size_t)kk * n.stride];
            const unsigned b1 = rec_z1(rb);
            if ((rb & REC_MARK) && !rec_hole(rb) && rec_mat(rb) == matId && a0 < b1 && b0 < a1) { uf_union(parent, (uint32_t)id, UF_MARKED); break; }
            b0 = b1;
            if (b0 >= a1) break;
        }
    }
}

// (the grid of k_strip_seed again)
// cand bit 2 (k_cavity_scan's floods): the column holds a marked node; left (may be null): set when a node stays unmarked.
static __global__ __launch_bounds__(256) void k_strip_spread(GridView g, int matId, uint8_t* __restrict__ cand, const uint32_t* __restrict__ list,
                                                      const uint32_t* __restrict__ counts, size_t segcap, uint32_t* __restrict__ parent, int* __restrict__ changed,
                                                      int* __restrict__ left) {
    const unsigned n = counts[blockIdx.x];
    const uint32_t* mine = list + (size_t)blockIdx.x * segcap;
    int any = 0, stays = 0;
    for (unsigned idx = threadIdx.x; idx < n; idx += blockDim.x) {
        const size_t c = mine[idx];
        const unsigned cb = cand[c];
        if (!(cb & 1)) continue;
        const int l = g.len[c];
        bool open_node = false, marked = false;
        for (int k = 0; k < l; k++) {
            const size_t id = (size_t)k * g.N + c;
            const uint32_t r = g.rec[id];
            if (rec_hole(r) || (r & REC_MARK) || rec_mat(r) != matId) continue;
            if (uf_find(parent, (uint32_t)id) == UF_MARKED) { g.rec[id] = r | REC_MARK; any = 1; marked = true; }
            else open_node = true;
        }
        if (!open_node || marked) cand[c] = (uint8_t)((open_node ? 1u : 0u) | 2u | (cb & 4u) | (marked ? 4u : 0u));
        if (open_node) stays = 1;
    }
    if (any) *changed = 1;
    if (stays && left) *left = 1;
}

// Boundary rows of a slab as ghost messages: len[ny] (padded to a multiple of 4 bytes), then
// rec[planes][ny].  send_lower = row 0, send_upper = row nx-1.
static __global__ __launch_bounds__(256) void k_ghost_pack(GridView g, int planes, uint8_t* __restrict__ send_lower, uint8_t* __restrict__ send_upper) {
    const size_t lenb = ((size_t)g.ny + 3) & ~(size_t)3;
    const size_t n = (size_t)g.ny * (planes + 1);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int p = (int)(e / g.ny) - 1, j = (int)(e % g.ny);
        const size_t lo = j, hi = (size_t)(g.nx - 1) * g.ny + j;
        if (p < 0) { send_lower[j] = g.len[lo]; send_upper[j] = g.len[hi]; }
        else {
            ((uint32_t*)(send_lower + lenb))[(size_t)p * g.ny + j] = p < (int)g.len[lo] ? g.rec[(size_t)p * g.N + lo] : 0u;
            ((uint32_t*)(send_upper + lenb))[(size_t)p * g.ny + j] = p < (int)g.len[hi] ? g.rec[(size_t)p * g.N + hi] : 0u;
        }
    }
}

// removal + compaction (cg1:970-999): marked layers go; one in the middle of a column leaves a
// gap interval (adjacent gaps merge, a gap on top goes too).  flags[2] = a gap remains somewhere,
// flags[1] = a cavity layer (material 2) remains somewhere.
// nodes (may be null): cand[] of a flood whose nodes were ALL the gaps and cavity layers of the grid (k_cavity_scan): a
// column without bit 2 holds neither a mark nor a gap, and is passed over after that one byte (flags[1] then only
// speaks for the columns looked at: the caller knows from k_strip_spread whether a cavity layer stays elsewhere).
__device__ __forceinline__ void strip_compact_column(const GridView& g, const size_t c, int& gap, int& cav);
static __global__ __launch_bounds__(256) void k_strip_compact(GridView g, int* __restrict__ flags, const uint8_t* __restrict__ nodes) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int gap = 0, cav = 0;
    if (nodes && (g.N & 3) == 0) {
        // four columns' bytes at a time (the host sizes the grid for N / 4 threads); only a column with a marked node
        // changes (bit 2), and the gaps of the others were all turned into cavity layers by the scan
        for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < (g.N >> 2); q += stride) {
            const unsigned w = *(const unsigned*)(nodes + 4 * q) & 0x04040404u;
            if (!w) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) if (w & (4u << (8 * j))) strip_compact_column(g, 4 * q + j, gap, cav);
        }
    } else {
        for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
            if (nodes && !(nodes[c] & 4)) continue;
            strip_compact_column(g, c, gap, cav);
        }
    }
    if (gap) flags[2] = 1;
    if (cav) flags[1] = 1;
}
__device__ __forceinline__ void strip_compact_column(const GridView& g, const size_t c, int& gap, int& cav) {
    {
        const int l = g.len[c];
        int w = 0, nh = 0, nc = 0;                 // intervals kept, gaps and cavity layers among them
        bool dirty = false, below_hole = false;    // below_hole: the interval at w - 1 is a gap
        for (int k = 0; k < l; k++) {
            uint32_t r = g.rec[(size_t)k * g.N + c];
            if (r & REC_MARK) { r = rec_z1(r) | REC_HOLE; dirty = true; }
            const bool hole = rec_hole(r);
            if (hole && below_hole) { w--; nh--; dirty = true; }              // merge with the gap below
            if (dirty || w != k) g.rec[(size_t)w * g.N + c] = r;
            if (hole) nh++; else if (rec_mat(r) == 2) nc++;
            below_hole = hole;
            w++;
        }
        if (below_hole) { w--; nh--; dirty = true; }                          // no gap on top (gaps never neighbour each other here)
        if (dirty) {
            for (int k = w; k < l; k++) g.rec[(size_t)k * g.N + c] = 0;
            g.len[c] = (uint8_t)w;
        }
        if (nh > 0) gap = 1;
        if (nc > 0) cav = 1;
    }
}

// identify_cavity step 1 (cg2:23-71): every gap becomes a cavity layer (material 2) and every
// column lower than the highest one gets a cavity layer up to that height.  The host has made
// sure that a plane is free above every column.
static __global__ __launch_bounds__(256) void k_cavity_fill(GridView g, unsigned hmax_q) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        const int l = g.len[c];
        if (l == 0) continue;
        for (int k = 0; k < l; k++) {
            const uint32_t r = g.rec[(size_t)k * g.N + c];
            if (rec_hole(r)) g.rec[(size_t)k * g.N + c] = rec_make(rec_z1(r), 2);
        }
        const unsigned topq = rec_z1(g.rec[(size_t)(l - 1) * g.N + c]);
        if (hmax_q > topq && l < g.lmax) {
            g.rec[(size_t)l * g.N + c] = rec_make(hmax_q, 2);
            g.len[c] = (uint8_t)(l + 1);
        }
    }
}

// addLayer / removeTopLayer (script_columngrid.js:196-218) on one column of a grid in whole form.
// op 0 appends (mat, z0q, z1q): a start above the column top takes a gap interval of its own,
// a start below it cannot be held (flags[1]); flags[0] asks the host for more planes.
// op 1 drops the top layer, and with it a gap it was resting on.
static __global__ void k_edit_column(GridView g, size_t c, int op, int mat, unsigned z0q, unsigned z1q, int* flags) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int l = g.len[c];
    if (op == 0) {
        const unsigned prev = l > 0 ? rec_z1(g.rec[(size_t)(l - 1) * g.N + c]) : 0u;
        if (z0q < prev) { flags[1] = 1; return; }
        const int need = z0q > prev ? 2 : 1;
        if (l + need > g.lmax) { flags[0] = 1; return; }
        if (z0q > prev) { g.rec[(size_t)l * g.N + c] = (z0q & 0xffffu) | REC_HOLE; l++; flags[2] = 1; }
        g.rec[(size_t)l * g.N + c] = rec_make(z1q, mat); l++;
        g.len[c] = (uint8_t)l;
    } else {
        if (l == 0) return;
        l--; g.rec[(size_t)l * g.N + c] = 0;
        if (l > 0 && rec_hole(g.rec[(size_t)(l - 1) * g.N + c])) { l--; g.rec[(size_t)l * g.N + c] = 0; }
        g.len[c] = (uint8_t)l;
    }
}

// ------------------------------------------------------------------------------------------
// layout conversion: device intervals <-> the reference's column-major host layout
// ------------------------------------------------------------------------------------------
// One column (l device intervals, interval k = at(k), asked for in rising k, each once) as a row of the reference layout: L slots of mat and of
// (z0, z1), gaps skipped, slots past the last layer zero.  Returns the number of layers.  When L
// is a multiple of 4 the row goes out as 4-byte / 16-byte stores (rows of neighbouring threads are
// adjacent, so a wave fills whole lines); scalar stores otherwise.
template <class At>
__device__ __forceinline__ int store_reference_row(At at, int l, int L, uint8_t* __restrict__ mrow, uint16_t* __restrict__ zrow) {
    int w = 0; unsigned prev = 0;
    if ((L & 3) == 0) {
        int k = 0;
        for (int base = 0; base < L; base += 4) {
            uint32_t m4 = 0; u16x4 za = (u16x4)(0), zb = (u16x4)(0);      // slots base..base+1 and base+2..base+3
            for (int q = 0; q < 4; q++) {
                uint32_t r = 0; bool have = false;
                while (k < l) {
                    r = at(k++);
                    if (!rec_hole(r)) { have = true; break; }
                    prev = rec_z1(r);
                }
                if (!have) break;
                m4 |= (uint32_t)rec_mat(r) << (8 * q);
                if (q < 2) { za[2 * q] = (unsigned short)prev; za[2 * q + 1] = (unsigned short)rec_z1(r); }
                else { zb[2 * (q - 2)] = (unsigned short)prev; zb[2 * (q - 2) + 1] = (unsigned short)rec_z1(r); }
                prev = rec_z1(r); w++;
            }
            *(uint32_t*)(mrow + base) = m4;
            *(u16x4*)(zrow + 2 * base) = za;
            *(u16x4*)(zrow + 2 * base + 4) = zb;
        }
        return w;
    }
    for (int k = 0; k < l; k++) {
        const uint32_t r = at(k);
        if (!rec_hole(r) && w < L) { mrow[w] = (uint8_t)rec_mat(r); zrow[2 * w] = (uint16_t)prev; zrow[2 * w + 1] = (uint16_t)rec_z1(r); w++; }
        prev = rec_z1(r);
    }
    for (int k = w; k < L; k++) { mrow[k] = 0; zrow[2 * k] = 0; zrow[2 * k + 1] = 0; }
    return w;
}

// One column into a row of the reference layout.  The whole form is read plane by plane as the row is written; the
// split form's chain runs top-down, so its intervals (at most LNK_NONE + 1) are first collected in a small per-lane array.
__device__ __forceinline__ int gather_column(const GridView& g, size_t c, int count_chain, int L, uint8_t* __restrict__ mrow, uint16_t* __restrict__ zrow) {
    if (g.split) {
        // one walk down the chain: the intervals are collected top first and handed out bottom first (the chain says how
        // long it is, so neither len[] nor a walk of its own to count is needed)
        uint32_t col[LNK_NONE + 1];
        int n = 0;
        uint32_t x = g.top[c];
        while (x != 0 && n <= LNK_NONE) {
            col[n++] = x & REC_BODY;
            const int p = rec_lnk(x);
            x = p == LNK_NONE ? 0u : g.rec[(size_t)p * g.N + c];
        }
        return store_reference_row([&](int k) { return col[n - 1 - k]; }, n, L, mrow, zrow);
    }
    const uint32_t* __restrict__ rc = g.rec + c;
    const size_t N = g.N;
    return store_reference_row([=](int k) { return rc[(size_t)k * N]; }, (int)g.len[c], L, mrow, zrow);
}

// Gather `ncols` columns (column s of the output is device column col0 + s*col_stride) into
// reference layout with stride L: omat[s*L+k], ozp[2*(s*L+k)+{0,1}], olen[s]; gaps are skipped,
// slots at and above the layer count are zero.
// count_chain: the grid is in split form and len[] is behind (see k_relen): count the chain instead.
static __global__ __launch_bounds__(256) void k_gather_columns(GridView g, size_t col0, size_t col_stride, size_t ncols, int L, int count_chain,
                                                        uint8_t* __restrict__ omat, uint16_t* __restrict__ ozp, uint8_t* __restrict__ olen) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < ncols; s += stride) {
        const size_t c = col0 + s * col_stride;
        olen[s] = (uint8_t)gather_column(g, c, count_chain, L, omat + s * (size_t)L, ozp + 2 * s * (size_t)L);
    }
}

// `count` cross-sections in one launch: slice s is row index[s] (axis 0: columns index[s]*ny + p) or
// column index[s] (axis 1: columns p*ny + index[s]), p < line; output as in k_gather_columns, slice
// after slice.
static __global__ __launch_bounds__(256) void k_gather_slices(GridView g, int axis, const int32_t* __restrict__ index, int count, size_t line, int L,
                                                       int count_chain, uint8_t* __restrict__ omat, uint16_t* __restrict__ ozp, uint8_t* __restrict__ olen) {
    const size_t total = (size_t)count * line;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < total; s += stride) {
        const size_t q = s / line, p = s - q * line;
        const size_t c = axis == 0 ? (size_t)index[q] * g.ny + p : p * g.ny + (size_t)index[q];
        olen[s] = (uint8_t)gather_column(g, c, count_chain, L, omat + s * (size_t)L, ozp + 2 * s * (size_t)L);
    }
}

// Inverse of k_gather_columns for a contiguous column range.  A layer that starts above the
// top of the one below becomes a gap interval + the layer.  flags[2]: some column has a gap;
// flags[3]: a column needs more planes than provisioned (the largest need); flags[1]: a layer
// starts below the top of the one under it (cannot be represented).
static __global__ __launch_bounds__(256) void k_scatter_columns(GridView g, size_t col0, size_t ncols, int L,
                                                         const uint8_t* __restrict__ imat, const uint16_t* __restrict__ izp,
                                                         const uint8_t* __restrict__ ilen, int* __restrict__ flags) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < ncols; s += stride) {
        const size_t c = col0 + s;
        const int l = min((int)ilen[s], L);
        int w = 0; unsigned prev = 0;
        for (int k = 0; k < l; k++) {
            const size_t p = s * (size_t)L + k;
            const unsigned z0 = izp[2 * p], z1 = izp[2 * p + 1];
            if (z0 > prev) {
                if (w < g.lmax) g.rec[(size_t)w * g.N + c] = (z0 & 0xffffu) | REC_HOLE;
                w++; flags[2] = 1;
            } else if (z0 < prev) flags[1] = 1;
            if (w < g.lmax) g.rec[(size_t)w * g.N + c] = rec_make(z1, imat[p]);
            w++;
            prev = z1;
        }
        if (w > g.lmax) atomicMax(&flags[3], w);
        g.len[c] = (uint8_t)min(w, g.lmax);
        for (int k = min(w, g.lmax); k < g.lmax; k++) g.rec[(size_t)k * g.N + c] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// the two forms of a column
// ------------------------------------------------------------------------------------------
// split -> whole: the chain that starts at the top is laid out by index in the planes, links cleared.  A
// column's planes are its own, so it is read completely before anything of it is written.
static __global__ __launch_bounds__(256) void k_demote(GridView g) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        const int l = min((int)g.len[c], LNK_NONE + 1);     // (the split form holds no taller column)
        if (l == 0) continue;
        uint32_t col[LNK_NONE + 1];
        load_column(g, c, l, col);
        for (int k = 0; k < l; k++) g.rec[(size_t)k * g.N + c] = col[k];
    }
}

// whole -> split: interval k stays in plane k and gets the link to plane k-1; the top interval is
// copied to the top stream.  The host has checked that no column is taller than LNK_NONE + 1.
static __global__ __launch_bounds__(256) void k_promote(GridView g) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        const int l = g.len[c];
        uint32_t r = 0;
        for (int k = 0; k < l; k++) {
            const size_t at = (size_t)k * g.N + c;
            const uint32_t old = g.rec[at];
            r = (old & REC_BODY & ~REC_MARK) | ((uint32_t)(k > 0 ? k - 1 : LNK_NONE) << LNK_SHIFT);
            if (r != old && k < l - 1) g.rec[at] = r;
        }
        g.top[c] = r;
    }
}

}  // namespace cgk
