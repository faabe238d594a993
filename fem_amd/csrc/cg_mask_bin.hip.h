// Binning of a mask's shapes by tile on the device, for the rasteriser (k_mask_shapes_tiles below).
//
// The reference decides "does the step apply at (x, y)" by walking the mask's objects from the last to the first and
// letting the first containing shape decide (template/script_runtime.js:54-71): the answer at a point is the polarity of
// the containing shape with the HIGHEST index.  With hundreds of thousands of shapes (the benchmark's periodic masks) a
// cell must not walk them all: shapes are binned by MASK_TILE x MASK_TILE-cell tile through a padded world-space box
// outside which a shape contains no point, and a cell looks only at its tile's shapes -- in any order, keeping the
// highest index that contains it -- plus the few shapes that reach a large part of the grid ("wide").  Count, scan, fill:
// three small kernels over the shape table, which is on the device anyway; the host reads back one number (the length
// of the lists) to size them.
#pragma once
#include "cg_kernels.hip.h"

namespace cgk {

struct TileRange { int i0, i1, j0, j1; };                 // tiles [i0, i1] x [j0, j1]; i1 < i0: none

// World-space box [x0, x1] x [y0, y1] outside which shape s contains no point (rt:73-129), padded; false when no finite
// box can be vouched for.  An empty shape gets an empty box (x1 < x0).
__device__ inline bool shape_box_dev(const Shape& s, const double* __restrict__ pts, double& x0, double& x1, double& y0, double& y1) {
    double lx0, lx1, ly0, ly1;                              // the box in the shape's own frame
    switch (s.type) {
    case 0: lx0 = fmin(0.0, s.w); lx1 = fmax(0.0, s.w); ly0 = fmin(0.0, s.h); ly1 = fmax(0.0, s.h); break;
    case 1: lx0 = s.w / 2 - fabs(s.a); lx1 = s.w / 2 + fabs(s.a); ly0 = s.h / 2 - fabs(s.a); ly1 = s.h / 2 + fabs(s.a); break;
    case 2: lx0 = s.w / 2 - fabs(s.a); lx1 = s.w / 2 + fabs(s.a); ly0 = s.h / 2 - fabs(s.b); ly1 = s.h / 2 + fabs(s.b); break;
    case 3: {
        if (s.pt_cnt <= 0) { x0 = y0 = 1; x1 = y1 = 0; return true; }
        const double* p = pts + 2 * (size_t)s.pt_off;
        lx0 = lx1 = p[0]; ly0 = ly1 = p[1];
        for (int k = 0, j = s.pt_cnt - 1; k < s.pt_cnt; j = k++) {
            lx0 = fmin(lx0, p[2 * k]); lx1 = fmax(lx1, p[2 * k]); ly0 = fmin(ly0, p[2 * k + 1]); ly1 = fmax(ly1, p[2 * k + 1]);
            // the crossing test divides by (yj - yi + 1e-9) (rt:137): on an edge flatter than that guard a crossing can
            // land anywhere, so no box is vouched for
            const double ey = p[2 * j + 1] - p[2 * k + 1];
            if (ey != 0 && fabs(ey) < 1e-6) return false;
        }
        break;
    }
    default: x0 = y0 = 1; x1 = y1 = 0; return true;         // never contains a point (rt:126)
    }
    double c = 1, sn = 0;
    if (s.rotated) { c = s.cosr; sn = s.sinr; }
    const double det = c * c + sn * sn;
    if (!(det > 1e-12) || !isfinite(det)) return false;
    double bx0 = __builtin_inf(), bx1 = -__builtin_inf(), by0 = __builtin_inf(), by1 = -__builtin_inf();
    for (int k = 0; k < 4; k++) {
        const double u = ((k & 1) ? lx1 : lx0) * s.scale_x, v = ((k & 2) ? ly1 : ly0) * s.scale_y;      // undo lx /= scaleX (rt:92)
        const double wx = s.left + (c * u + sn * v) / det, wy = s.top + (-sn * u + c * v) / det;       // undo the rotation (rt:88-90)
        bx0 = fmin(bx0, wx); bx1 = fmax(bx1, wx); by0 = fmin(by0, wy); by1 = fmax(by1, wy);
    }
    if (!isfinite(bx0) || !isfinite(bx1) || !isfinite(by0) || !isfinite(by1)) return false;
    const double pad = 1e-6 + 1e-9 * fmax(fmax(fabs(bx0), fabs(bx1)), fmax(fabs(by0), fabs(by1)));
    x0 = bx0 - pad; x1 = bx1 + pad; y0 = by0 - pad; y1 = by1 + pad;
    return true;
}

// Pass 1: the tile range of every shape; counts[tile] += 1 for the tiles of the narrow ones; the wide ones (more than
// wide_area tiles, or no box) go to `wide` (counter[0] = how many; beyond wide_cap they are only counted).
// counter[1] = total length of the tile lists (may exceed what fits an int: saturates at 2^31 - 1).
// counter[3] != 0: a polygon names points outside [0, npts) (the host refuses the list).
static __global__ __launch_bounds__(256) void k_shape_ranges(int nx, int ny, int i0g, double dx, double dy, const Shape* __restrict__ shapes, int nshapes,
                                                             const double* __restrict__ pts, int tn_i, int tn_j, unsigned wide_area, TileRange* __restrict__ rg,
                                                             int* __restrict__ counts, int* __restrict__ wide, int wide_cap, unsigned* __restrict__ counter, int npts) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nshapes) return;
    double x0, x1, y0, y1;
    TileRange r{0, tn_i - 1, 0, tn_j - 1};
    // a polygon whose points lie outside the table: reported (counter[3] = 1 + the largest such shape), never looked at
    if (shapes[s].type == CG_SHAPE_POLYGON && (shapes[s].pt_off < 0 || shapes[s].pt_cnt < 0 || (long long)shapes[s].pt_off + shapes[s].pt_cnt > npts)) {
        atomicMax(&counter[3], (unsigned)s + 1u);
        rg[s] = TileRange{0, -1, 0, -1};
        return;
    }
    if (shape_box_dev(shapes[s], pts, x0, x1, y0, y1)) {
        if (x1 < x0 || y1 < y0) { rg[s] = TileRange{0, -1, 0, -1}; return; }
        // cell centres are (i + 0.5) * dx: one cell of slack on each side
        const double ci0 = floor(x0 / dx - 0.5) - 1 - i0g, ci1 = ceil(x1 / dx - 0.5) + 1 - i0g;
        const double cj0 = floor(y0 / dy - 0.5) - 1, cj1 = ceil(y1 / dy - 0.5) + 1;
        if (ci1 < 0 || cj1 < 0 || ci0 > nx - 1 || cj0 > ny - 1) { rg[s] = TileRange{0, -1, 0, -1}; return; }
        r.i0 = (int)fmax(0.0, ci0) / MASK_TILE; r.i1 = (int)fmin((double)nx - 1, ci1) / MASK_TILE;
        r.j0 = (int)fmax(0.0, cj0) / MASK_TILE; r.j1 = (int)fmin((double)ny - 1, cj1) / MASK_TILE;
    }
    const unsigned long long area = (unsigned long long)(r.i1 - r.i0 + 1) * (unsigned long long)(r.j1 - r.j0 + 1);
    if (area > wide_area) {
        const unsigned at = atomicAdd(&counter[0], 1u);
        if ((int)at < wide_cap) wide[at] = s;
        rg[s] = TileRange{0, -1, 0, -1};
        return;
    }
    rg[s] = r;
    const unsigned before = atomicAdd(&counter[1], (unsigned)area);
    if (before + (unsigned)area < before) counter[2] = 1;          // the total wrapped: far too many, the host gives up on binning
    for (int ti = r.i0; ti <= r.i1; ti++) for (int tj = r.j0; tj <= r.j1; tj++) atomicAdd(&counts[(size_t)ti * tn_j + tj], 1);
}

// Pass 2: exclusive prefix sum of counts[0..n) into off[0..n], one workgroup of 1024 (n is a few hundred thousand at most)
static __global__ __launch_bounds__(1024) void k_tile_scan(const int* __restrict__ counts, int n, int* __restrict__ off) {
    __shared__ int s_part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = min(n, t * per), e = min(n, b + per);
    int sum = 0;
    for (int k = b; k < e; k++) sum += counts[k];
    s_part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                    // inclusive scan of the partial sums
        const int v = t >= d ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = s_part[t] - sum;
    for (int k = b; k < e; k++) { off[k] = run; run += counts[k]; }
    if (t == 1023) off[n] = s_part[1023];
}

// Pass 3: the lists.  cursor[] starts as a copy of off[0..n); within a tile the order is whatever the atomics give.
static __global__ __launch_bounds__(256) void k_shape_fill(const TileRange* __restrict__ rg, int nshapes, int tn_j, int* __restrict__ cursor, int* __restrict__ list) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nshapes) return;
    const TileRange r = rg[s];
    for (int ti = r.i0; ti <= r.i1; ti++) for (int tj = r.j0; tj <= r.j1; tj++) list[atomicAdd(&cursor[(size_t)ti * tn_j + tj], 1)] = s;
}

// The rasteriser over the binned shapes: a cell keeps the highest index among the shapes of its tile's list and of the
// wide list that contain its centre (the lists are in no particular order), and that shape's polarity decides (rt:58-66).
static __global__ __launch_bounds__(256) void k_mask_shapes_tiles(int nx, int ny, int i0, double dx, double dy, const Shape* __restrict__ shapes,
                                                                  const double* __restrict__ pts, int tiles_j, const int* __restrict__ tile_off,
                                                                  const int* __restrict__ tile_list, const int* __restrict__ wide, int nwide,
                                                                  uint8_t* __restrict__ raster) {
    const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j;
    const int b = tile_off[blockIdx.x], e = tile_off[blockIdx.x + 1];
    for (int k = threadIdx.x; k < MASK_TILE * MASK_TILE; k += blockDim.x) {
        const int i = ti * MASK_TILE + k / MASK_TILE, j = tj * MASK_TILE + k % MASK_TILE;
        if (i >= nx || j >= ny) continue;
        const double x = ((double)(i0 + i) + 0.5) * dx, y = ((double)j + 0.5) * dy;
        int best = -1;
        for (int a = b; a < e; a++) { const int s = tile_list[a]; if (s > best && point_inside_shape(x, y, shapes[s], pts)) best = s; }
        for (int w = 0; w < nwide; w++) { const int s = wide[w]; if (s > best && point_inside_shape(x, y, shapes[s], pts)) best = s; }
        raster[(size_t)i * ny + j] = (uint8_t)(best < 0 ? 1 : !shapes[best].positive);
    }
}

}  // namespace cgk
