// Device code of libcolgrid: the per-column process kernels for gfx950 (MI355X).
//
// Data layout (all in HBM, owned by the handle):
//   len[c]        u8   layers in column c                      c = i*NY + j, j fastest
//   mat[k*N + c]  u8   material of layer k                     "plane-major": the 64 lanes of a
//   z0 [k*N + c]  u16  quantised bottom of layer k             wavefront that touch layer k of 64
//   z1 [k*N + c]  u16  quantised top of layer k                neighbouring columns read one line
//
// Arithmetic follows the reference's JavaScript: heights are u16 steps of zscale nm,
// de-quantised as (double)q*zscale, re-quantised as Math.round(z/zscale) (halves toward
// +infinity) and stored modulo 2^16 (template/script_columngrid.js:117-122, 85).  This file
// must be compiled with -ffp-contract=off: the reference rounds after every operation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CG_EPS 1e-9

namespace cgk {

struct GridView {
    uint8_t*  len;
    uint8_t*  mat;
    uint16_t* z0;
    uint16_t* z1;
    size_t    N;        // columns of this handle
    int       nx, ny;
    int       lmax;     // planes provisioned
    int       i0;       // global row of local row 0
    double    zs;       // nm per step
};

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
__device__ __forceinline__ uint16_t quantz(double z, double zs) { return to_u16(js_round(z / zs)); }

// ------------------------------------------------------------------------------------------
// per-column update primitives
// ------------------------------------------------------------------------------------------

// _depoAtTop_ (script_columngrid.js:397-426) for one column.  `l` is the column's current
// length.  Returns 0 = nothing, 1 = top layer extended, 2 = layer appended, 3 = append needed
// but no plane left (column untouched; the host grows the planes and re-runs those columns).
__device__ __forceinline__ int depo_at_top(const GridView& g, size_t c, int l, int matId, double dz) {
    if (!(dz > 0)) return 0;
    double topZ = 0; int topMat = 0;
    size_t top = 0;
    if (l > 0) {
        top = (size_t)(l - 1) * g.N + c;
        topZ = (double)g.z1[top] * g.zs;
        topMat = g.mat[top];
    }
    double newTop = topZ + dz;
    if (l > 0 && topMat == matId) {
        g.z1[top] = quantz(newTop, g.zs);
        return 1;
    }
    if (l >= g.lmax) return 3;
    size_t b = (size_t)l * g.N + c;
    g.mat[b] = (uint8_t)matId;
    g.z0[b] = quantz(topZ, g.zs);
    g.z1[b] = quantz(newTop, g.zs);
    g.len[c] = (uint8_t)(l + 1);
    return 2;
}

// _applyEtchAt (script_columngrid.js:553-589).  Returns number of whole layers removed in the
// low 16 bits and 0x10000 if the column was touched at all.
__device__ __forceinline__ int apply_etch_at(const GridView& g, size_t c, int l, int matId, double dz, bool isAll) {
    if (!(dz > 0)) return 0;
    double remaining = dz;
    int removed = 0, touched = 0;
    const int l0 = l;
    while (remaining > CG_EPS && l > 0) {
        size_t b = (size_t)(l - 1) * g.N + c;
        int curMat = g.mat[b];
        if (!isAll && curMat != matId) break;
        double z0 = (double)g.z0[b] * g.zs, z1 = (double)g.z1[b] * g.zs;
        double thick = z1 - z0;
        touched = 0x10000;
        if (remaining < thick - CG_EPS) {
            g.z1[b] = quantz(z1 - remaining, g.zs);
            remaining = 0;
        } else {
            remaining -= thick;
            l -= 1;
            g.mat[b] = 0; g.z0[b] = 0; g.z1[b] = 0;      // removed slots are cleared (cg1:584-586)
            removed++;
        }
    }
    if (l != l0) g.len[c] = (uint8_t)l;
    return touched | removed;
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
// vertical steps (fast paths of deposit_general / etch_general, cg1:445-454 and 609-618)
// ------------------------------------------------------------------------------------------

// only_len < 0: normal run.  only_len >= 0: second pass after the planes were grown -- touch
// only the columns that could not append in the first pass (len == only_len, top differs).
template <bool STATS>
__global__ __launch_bounds__(256) void k_depo_vertical(GridView g, const uint8_t* __restrict__ mask, int matId, double t,
                                                       int only_len, int* __restrict__ flags, unsigned long long* gstats) {
    unsigned loc[3] = {0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        if (mask && !mask[c]) continue;
        int l = g.len[c];
        if (only_len >= 0) {
            if (l != only_len) continue;
            if (l > 0 && g.mat[(size_t)(l - 1) * g.N + c] == matId) continue;
        }
        int r = depo_at_top(g, c, l, matId, t);
        if (r == 3) flags[0] = 1;
        if (STATS) { loc[0]++; loc[1] += (r == 1); loc[2] += (r == 2); }
    }
    if (STATS) { const int slot[3] = {ST_OPEN, ST_EXTEND, ST_APPEND}; stats_commit<3>(gstats, loc, slot); }
}

template <bool STATS>
__global__ __launch_bounds__(256) void k_etch_vertical(GridView g, const uint8_t* __restrict__ mask, int matId, double t,
                                                       unsigned long long* gstats) {
    unsigned loc[3] = {0, 0, 0};
    const bool isAll = (matId == 255);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        if (mask && !mask[c]) continue;
        int l = g.len[c];
        int r = apply_etch_at(g, c, l, matId, t, isAll);
        if (STATS) { loc[0]++; loc[1] += (r >> 16); loc[2] += (r & 0xffff); }
    }
    if (STATS) { const int slot[3] = {ST_OPEN, ST_ETCHED, ST_REMOVED}; stats_commit<3>(gstats, loc, slot); }
}

// ------------------------------------------------------------------------------------------
// lateral-stencil steps: snapshot of the tops, halo handling, dilation / erosion + update
// ------------------------------------------------------------------------------------------
//
// The snapshot is a (nx + 2R) x ny image of u16 top heights (rows R .. R+nx-1 are this
// handle's rows) and, for etch, a u8 image of "etchable here" flags E (cg1:637-643).
// The R rows below and above come from the neighbouring slab, or are the clamp-replicated
// edge row (heights, cg1:373-387) / zero (flags, cg1:671) at a domain edge.

template <bool ETCH>
__global__ __launch_bounds__(256) void k_snapshot_tops(GridView g, const uint8_t* __restrict__ mask, int matId, int R,
                                                       uint16_t* __restrict__ topq, uint8_t* __restrict__ eflag,
                                                       int* __restrict__ flags) {
    const bool isAll = (matId == 255);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t off = (size_t)R * g.ny;
    int any = 0;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        int l = g.len[c];
        uint16_t q = 0; int ok = 0;
        if (l > 0) {
            size_t top = (size_t)(l - 1) * g.N + c;
            q = g.z1[top];
            if (ETCH) ok = isAll || (g.mat[top] == matId);
        }
        topq[off + c] = q;
        if (ETCH) {
            int e = ((!mask || mask[c]) && ok) ? 1 : 0;
            eflag[off + c] = (uint8_t)e;
            any |= e;
        }
    }
    if (ETCH && any) flags[1] = 1;          // "some column is etchable" (cg1:692)
}

// rows to send: first R rows -> lower neighbour, last R rows -> upper neighbour.
// message layout: R*ny u16 heights, then R*ny u8 flags (zeros for deposit).
__global__ __launch_bounds__(256) void k_halo_pack(const uint16_t* __restrict__ topq, const uint8_t* __restrict__ eflag, int nx, int ny,
                                                   int R, uint8_t* __restrict__ send_lower, uint8_t* __restrict__ send_upper) {
    const size_t n = (size_t)R * ny;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        size_t lo = (size_t)R * ny + e;                     // rows R .. 2R-1 of the image = local rows 0..R-1
        size_t hi = (size_t)nx * ny + e;                    // rows nx .. nx+R-1 of the image = local rows nx-R..nx-1
        ((uint16_t*)send_lower)[e] = topq[lo];
        ((uint16_t*)send_upper)[e] = topq[hi];
        send_lower[2 * n + e] = eflag ? eflag[lo] : 0;
        send_upper[2 * n + e] = eflag ? eflag[hi] : 0;
    }
}

// fill the halo rows of the image from the received messages, or from the edge rule
__global__ __launch_bounds__(256) void k_halo_unpack(uint16_t* __restrict__ topq, uint8_t* __restrict__ eflag, int nx, int ny, int R,
                                                     const uint8_t* __restrict__ recv_lower, const uint8_t* __restrict__ recv_upper) {
    const size_t n = (size_t)R * ny;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        int j = (int)(e % ny);
        size_t lo = e;                                      // rows 0 .. R-1
        size_t hi = (size_t)(nx + R) * ny + e;              // rows nx+R .. nx+2R-1
        if (recv_lower) {
            topq[lo] = ((const uint16_t*)recv_lower)[e];
            if (eflag) eflag[lo] = recv_lower[2 * n + e];
        } else {
            topq[lo] = topq[(size_t)R * ny + j];            // replicate local row 0
            if (eflag) eflag[lo] = 0;
        }
        if (recv_upper) {
            topq[hi] = ((const uint16_t*)recv_upper)[e];
            if (eflag) eflag[hi] = recv_upper[2 * n + e];
        } else {
            topq[hi] = topq[(size_t)(nx + R - 1) * ny + j]; // replicate local row nx-1
            if (eflag) eflag[hi] = 0;
        }
    }
}

// One stencil tap: LDS offset relative to the centre (du*pitch + dv) and the float32 height
// increment of the ellipsoidal kernel (cg1:324-357).
struct Tap { int off; float add; };

// Stencil tile geometry: a block of 256 threads updates TI x TJ columns; thread (tx, ty)
// owns column j = tx and the TI/4 rows ty, ty+4, ...
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
template <bool ETCH, bool STATS>
__global__ __launch_bounds__(256) void k_stencil_apply(GridView g, const uint8_t* __restrict__ mask, int matId, double t, int R,
                                                       const uint16_t* __restrict__ topq, const uint8_t* __restrict__ eflag,
                                                       const Tap* __restrict__ taps, int ntaps,
                                                       const Tap* __restrict__ taps64, int ntaps64,
                                                       int only_len, int* __restrict__ flags, unsigned long long* gstats) {
    extern __shared__ __align__(16) float s_h[];
    const int pitch = TJ + 2 * R;
    const int tile_rows = TI + 2 * R;
    const int tiles_j = (g.ny + TJ - 1) / TJ;
    const int tile_i = blockIdx.x / tiles_j, tile_j = blockIdx.x % tiles_j;
    const int i_base = tile_i * TI, j_base = tile_j * TJ;
    const float zsf_dummy = 0.f; (void)zsf_dummy;

    // ---- stage heights (as float32 of q*zscale, cg1:465) with halo; flags fold into +inf
    for (int e = threadIdx.x; e < tile_rows * pitch; e += 256) {
        int r = e / pitch, cidx = e - r * pitch;
        int irow = i_base + r;                         // row in the (nx+2R) image
        int j = j_base + cidx - R;
        float h;
        if (irow >= g.nx + 2 * R) { h = ETCH ? __builtin_inff() : 0.f; }
        else {
            bool outside = (j < 0 || j >= g.ny);
            int jc = j < 0 ? 0 : (j >= g.ny ? g.ny - 1 : j);        // clamp-replicate in j (cg1:374-380)
            size_t p = (size_t)irow * g.ny + jc;
            h = (float)((double)topq[p] * g.zs);
            if (ETCH && (outside || !eflag[p])) h = __builtin_inff();   // not etchable: never a minimum (cg1:718)
        }
        s_h[e] = h;
    }
    __syncthreads();

    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = j_base + tx;
    unsigned loc[4] = {0, 0, 0, 0};
    const bool isAll = (matId == 255);

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
                size_t p = (size_t)(il + R) * g.ny + j;
                float h = (float)((double)topq[p] * g.zs);
                own[r] = h;
                if (ETCH) {
                    act[r] = eflag[p] != 0;
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
                    // floor: bottom of the top run of matId (cg1:646-660); 0 for ALL
                    float F = 0.f;
                    if (!isAll && l > 0) {
                        double floorZ = (double)g.z0[(size_t)(l - 1) * g.N + c] * g.zs;
                        if (STATS) loc[3]++;
                        for (int s = l - 2; s >= 0; s--) {
                            size_t b = (size_t)s * g.N + c;
                            if (g.mat[b] != matId) break;
                            floorZ = (double)g.z0[b] * g.zs;
                            if (STATS) loc[3]++;
                        }
                        F = (float)fmax(0.0, floorZ);
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
                        if (l > 0 && g.mat[(size_t)(l - 1) * g.N + c] == matId) continue;
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

__global__ __launch_bounds__(256) void k_mask_shapes(int nx, int ny, int i0, double dx, double dy, const Shape* __restrict__ shapes,
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

__global__ __launch_bounds__(256) void k_mask_normalise(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += stride) out[c] = in[c] ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// reductions and checks
// ------------------------------------------------------------------------------------------
// out[0] = max top z1q, out[1] = max len
__global__ __launch_bounds__(256) void k_max_top(GridView g, unsigned* __restrict__ out) {
    unsigned mq = 0, ml = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        int l = g.len[c];
        if (l > 0) { unsigned q = g.z1[(size_t)(l - 1) * g.N + c]; mq = q > mq ? q : mq; ml = (unsigned)l > ml ? (unsigned)l : ml; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        unsigned a = __shfl_down(mq, o, 64), b = __shfl_down(ml, o, 64);
        mq = a > mq ? a : mq; ml = b > ml ? b : ml;
    }
    if ((threadIdx.x & 63) == 0) { if (mq) atomicMax(&out[0], mq); if (ml) atomicMax(&out[1], ml); }
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void k_checksum(GridView g, unsigned long long* __restrict__ out) {
    unsigned long long sum = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        unsigned long long gc = (unsigned long long)g.i0 * g.ny + c;
        int l = g.len[c];
        unsigned long long h = splitmix64(gc);
        h = splitmix64(h ^ (unsigned long long)l);
        for (int k = 0; k < l && k < g.lmax; k++) {
            size_t b = (size_t)k * g.N + c;
            unsigned long long v = (unsigned long long)g.mat[b] | ((unsigned long long)g.z0[b] << 8) | ((unsigned long long)g.z1[b] << 24);
            h = splitmix64(h ^ v);
        }
        sum += h;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, sum);
}

// ------------------------------------------------------------------------------------------
// layout conversion: plane-major device planes <-> the reference's column-major host layout
// ------------------------------------------------------------------------------------------
// Gather `ncols` columns (column s of the output is device column col0 + s*col_stride) into
// reference layout with stride L: omat[s*L+k], ozp[2*(s*L+k)+{0,1}], olen[s].
__global__ __launch_bounds__(256) void k_gather_columns(GridView g, size_t col0, size_t col_stride, size_t ncols, int L,
                                                        uint8_t* __restrict__ omat, uint16_t* __restrict__ ozp, uint8_t* __restrict__ olen) {
    // thread -> (column s, layer k) with k fastest so that the writes are contiguous
    const size_t total = ncols * (size_t)L;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        size_t s = e / L; int k = (int)(e - s * L);
        size_t c = col0 + s * col_stride;
        uint8_t m = 0; uint16_t a = 0, b = 0;
        if (k < g.lmax) { size_t p = (size_t)k * g.N + c; m = g.mat[p]; a = g.z0[p]; b = g.z1[p]; }
        omat[e] = m; ozp[2 * e] = a; ozp[2 * e + 1] = b;
        if (k == 0) olen[s] = g.len[c];
    }
}

// inverse of k_gather_columns for a contiguous column range; also reports whether any column
// has a vertical gap (z0 of a layer above the z1 below it, or z0[0] > 0), flags[2]
__global__ __launch_bounds__(256) void k_scatter_columns(GridView g, size_t col0, size_t ncols, int L,
                                                         const uint8_t* __restrict__ imat, const uint16_t* __restrict__ izp,
                                                         const uint8_t* __restrict__ ilen, int* __restrict__ flags) {
    const size_t total = ncols * (size_t)g.lmax;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        size_t s = e / g.lmax; int k = (int)(e - s * g.lmax);
        size_t c = col0 + s;
        uint8_t m = 0; uint16_t a = 0, b = 0;
        if (k < L) { size_t p = s * (size_t)L + k; m = imat[p]; a = izp[2 * p]; b = izp[2 * p + 1]; }
        size_t p = (size_t)k * g.N + c;
        g.mat[p] = m; g.z0[p] = a; g.z1[p] = b;
        int l = ilen[s];
        if (k == 0) g.len[c] = (uint8_t)l;
        if (k < l && k < L) {
            unsigned below = (k == 0) ? 0u : izp[2 * (s * (size_t)L + k - 1) + 1];
            if ((double)a * g.zs > (double)below * g.zs + CG_EPS) flags[2] = 1;
        }
    }
}

}  // namespace cgk
