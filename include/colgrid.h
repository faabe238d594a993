/*
 * colgrid.h -- C ABI of libcolgrid, the MI355X (gfx950) column-grid process engine.
 *
 * The library keeps one process-emulation grid per handle resident in HBM and runs
 * every process step as hand-written HIP kernels.  The grid is a 2-D array of columns
 * (index c = i*NY + j, j fastest); every column is a stack of (material, z0, z1) layers.
 * On the device a column is its top interval top[c] (u32: z1 | material << 16 | flags << 24 |
 * plane of the interval below << 26) and, plane-major, the intervals below it rec[p*N + c] (u32,
 * same format: every record names the plane of the one below it), so that a wavefront touching the
 * tops -- or the intervals one process step laid down -- of neighbouring columns reads contiguous
 * bytes; a layer's bottom is the top of the interval below it.
 * cg_download / cg_upload convert to and from the reference's host layout
 *     mat[c*Lmax + k],  zpair[2*(c*Lmax + k) + {0,1}],  len[c]
 * (reference: template/script_columngrid.js:84-86, 111-112).
 *
 * Every entry point is extern "C", takes plain pointers and sizes, returns an int status
 * (0 = ok) and never throws.  A handle is not re-entrant: use it from one thread at a time.
 * All work of a handle is enqueued on one HIP stream (its own, or the one given to
 * cg_set_stream); entry points that return data to the host synchronise that stream.
 * Vertical steps (cg_deposit_general / cg_etch_general / cg_stencil_begin with t*S < min(dx, dy))
 * are queued in the handle, up to six (sixteen on a grid that has just been reset), and launched
 * together by the next entry point that is not such a step (cg_sync included): same results, one
 * pass over the columns instead of one per step.  A lateral deposit (on an undivided grid or on a slab)
 * may wait in the same way for the vertical steps that follow it (up to five), which then run inside
 * its launch.
 * Consequences for a caller that brings its own stream (cg_set_stream): when such a call returns,
 * nothing of it may be on the stream yet -- an event recorded, or dependent work enqueued, right
 * after it does not wait for the step.  cg_sync, cg_get_stream or any entry point other than a
 * vertical step puts the queued steps on the stream first (the calls that complete an open lateral
 * step -- cg_halo_buffers, cg_stencil_interior, cg_stencil_finish -- may leave them waiting); a
 * launch error of a queued step is reported by that later call.  cg_reset, cg_clear, cg_upload and cg_snapshot_load replace every
 * column; steps still queued are launched first all the same.
 *
 * The comment above each declaration names the reference interface it stands in for.
 * "cg1/cg2/rt/sl" = template/script_columngrid.js, script_columngrid2.js,
 * script_runtime.js, script_slice2d.js of the reference.
 */
#ifndef COLGRID_H
#define COLGRID_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------- */
#define CG_OK               0
#define CG_ERR_ARG          1   /* bad argument (null handle, unknown mask id, bad size ...)   */
#define CG_ERR_HIP          2   /* a HIP runtime call failed; see cg_last_error               */
#define CG_ERR_UNSUPPORTED  3   /* input outside what the engine reproduces bit-exactly        */
#define CG_ERR_NOMEM        4
#define CG_ERR_STATE        5   /* call sequence error (e.g. stencil_finish without begin)    */

/* ---- material ids (cg1:7-8) --------------------------------------------------------- */
#define CG_MAT_EMPTY   0
#define CG_MAT_AIR     1
#define CG_MAT_CAVITY  2
#define CG_MAT_ALL     255

#define CG_MASK_NONE  (-1)      /* "no mask": the step applies to every column               */

typedef struct cg_grid cg_grid;   /* opaque */

/* Domain and layout facts of a grid (fields of the reference's ColumnGrid object that the
 * runtime and the consumers read: cg1:49-60, script_gridrenderer.js:216-218). */
typedef struct cg_info {
    int32_t nx, ny;            /* columns owned by this handle (a slab when nx < nx_global)  */
    int32_t nx_global;         /* NX of the whole domain                                      */
    int32_t i0;                /* global i index of local row 0                               */
    int32_t lmax;              /* the reference's Lmax for this history: += 16 each time an append
                                * finds its column full (cg1:129-168, 418); stride of cg_download.
                                * The planes the device holds may be more (gaps take intervals). */
    int32_t device;            /* HIP device ordinal                                          */
    double  dx, dy, zscale;
    double  offset_x, offset_y, lx_eff, ly_eff;   /* of the whole domain                      */
} cg_info;

/* One mask shape, already reduced to numbers (rt:73-129).  The host computes cos/sin of
 * rad = (-angle*PI)/180 itself so that only ONE libm (the caller's) is ever involved.
 * Invisible objects (rt:62) are dropped by the host and never reach the library. */
#define CG_SHAPE_RECT     0
#define CG_SHAPE_CIRCLE   1
#define CG_SHAPE_ELLIPSE  2
#define CG_SHAPE_POLYGON  3   /* also 'path': its M/L vertices (rt:119-124)                  */
#define CG_SHAPE_NEVER    4   /* any other fabric type: never contains a point (rt:126)      */
typedef struct cg_shape {
    int32_t type;
    int32_t positive;          /* polarity 'positive' (default): a hit BLOCKS the step (rt:65) */
    int32_t rotated;           /* angle !== 0 (rt:87)                                          */
    int32_t pt_off, pt_cnt;    /* polygon: range in the points array (pairs x,y)               */
    int32_t _pad;
    double  left, top, w, h;   /* obj.left||0, obj.top||0, obj.width||0, obj.height||0         */
    double  cosr, sinr;        /* Math.cos(rad), Math.sin(rad)                                 */
    double  scale_x, scale_y;  /* obj.scaleX||1, obj.scaleY||1                                 */
    double  a, b;              /* circle: a = radius||w/2.  ellipse: a = rx||w/2, b = ry||h/2  */
} cg_shape;

/* ---- lifetime ------------------------------------------------------------------------ */

/* new ColumnGrid(LXnm, LYnm, dx, dy, LmaxInit=32, zScale=0.1) (cg1:21-38, 49-60):
 * NX = max(1, round(LX/dx)), NY likewise; Lmax = nextPowerOf2(LmaxInit) (cg1:302-304);
 * all columns empty.  The grid lives on the current HIP device. */
int cg_create(double lx_nm, double ly_nm, double dx, double dy,
              int lmax_init, double zscale, cg_grid** out);

/* A slab of a larger domain: rows i in [i0, i0+nx_local) of an nx_global x ny grid.
 * No counterpart in the reference (it is single-threaded); used for the 1-D domain
 * decomposition over GPUs.  Mask rasters use global cell centres (cg1:448). */
int cg_create_slab(int nx_global, int ny, int i0, int nx_local, double dx, double dy,
                   int lmax_init, double zscale, cg_grid** out);

int cg_destroy(cg_grid* g);

/* createNewGrid(LmaxInit = this.Lmax) (cg1:40-43): re-provision at nextPowerOf2(lmax_init)
 * layers (pass 0 to keep the current Lmax) and empty every column. */
int cg_reset(cg_grid* g, int lmax_init);

/* Most planes (intervals per column kept below the tops) the handle may back with memory
 * before it compacts the columns instead of mapping more.  Every depositing step takes a
 * plane; by default more are mapped in place while device memory allows (room for one copy
 * of the planes in use is always left), which keeps long flows free of compaction passes.
 * No counterpart in the reference; the Lmax a caller sees (cg_info.lmax) does not depend on it. */
int cg_set_plane_budget(cg_grid* g, int max_planes);

/* The reference's Lmax belongs to the whole grid: += 16 whenever an append finds its column full, wherever that
 * column is (cg1:129-168, 418).  Every slab of a decomposed domain follows that rule for its own columns; the
 * driver raises all of them to the largest after a step (never lowers: a smaller value is ignored).  No counterpart
 * in the reference (one array, one Lmax). */
int cg_raise_lmax(cg_grid* g, int lmax);

/* clearAll() (cg1:223-227): empty every column, keep Lmax. */
int cg_clear(cg_grid* g);

int cg_get_info(const cg_grid* g, cg_info* out);

/* Text of the last error on this handle (or, for g == NULL, of the last failed create on
 * the calling thread).  Valid until the next call on the handle. */
const char* cg_last_error(const cg_grid* g);

/* Use the caller's HIP stream (hipStream_t passed as void*) for all work of this handle;
 * NULL switches back to the handle's own stream. */
int cg_set_stream(cg_grid* g, void* hip_stream);
/* The stream the handle's work is enqueued on right now (hipStream_t as void*). */
int cg_get_stream(cg_grid* g, void** hip_stream);
int cg_sync(cg_grid* g);

/* ---- masks (rt:45-141) --------------------------------------------------------------- */

/* A mask is a per-column predicate "the step applies here", stored on the device as one
 * byte per column (1 = open), order c = i*NY + j, evaluated at the cell centre
 * x = (i+0.5)*dx, y = (j+0.5)*dy (cg1:448).  Ids are small non-negative integers. */

/* From a host raster the caller filled by evaluating maskFn at every centre of THIS
 * handle's columns (nx*ny bytes, non-zero = open). */
int cg_mask_from_raster(cg_grid* g, const uint8_t* host_raster, size_t n, int* mask_id);

/* Same, from a raster already in device memory (copied). */
int cg_mask_from_device_raster(cg_grid* g, const void* dev_raster, size_t n, int* mask_id);

/* From the shape list of a mask record: _isPointBlocked(x, y, maskData) (rt:54-71) run on
 * the device.  shapes[] are the VISIBLE objects in their original order; the walk is from
 * the last to the first and the first containing shape decides.  pts = polygon vertices
 * (x0,y0,x1,y1,...), npts = number of doubles / 2.  A polygon that names points outside
 * [0, npts) is refused (CG_ERR_ARG).  The device copies of the shape table and the per-tile
 * lists made from it stay with the handle for the next call (about 80 bytes per shape of the
 * longest list so far; cg_release_scratch frees them). */
int cg_mask_from_shapes(cg_grid* g, const cg_shape* shapes, int nshapes,
                        const double* pts, int npts, int* mask_id);

/* Does the mask open any column of THIS handle (CG_MASK_NONE: yes)?  deposit_general's lateral path
 * returns untouched -- no identify_cavity either -- when no cell of the domain is open (cg1:474-485);
 * an undivided grid checks that itself, a slab driver asks every slab and combines the answers. */
int cg_mask_any_open(cg_grid* g, int mask_id, int* any);

/* _getConformalKernelArray's cache (cg1:324-357, 32 entries, FIFO, keyed by t, S, dx, dy truncated to 1e-3) gets
 * its entry for a lateral step of this thickness and conformality, as deposit_general creates it BEFORE it finds
 * that the mask opens no cell and returns (cg1:470 vs 485).  For a slab driver that skips such a step on every
 * slab: a later step whose key collides must find the same cached kernel as on an undivided grid. */
int cg_kernel_cache_touch(cg_grid* g, double thickness, double conformality);

int cg_mask_download(cg_grid* g, int mask_id, uint8_t* host_out, size_t n);
int cg_mask_destroy(cg_grid* g, int mask_id);

/* ---- process steps ------------------------------------------------------------------- */

/* deposit_general(maskFn, matId, thickness, conformality) (cg1:435-544).
 * t<=0: no-op.  S clamped to [0,1].  t*S < min(dx,dy): vertical deposit on open columns
 * (cg1:445-454).  Otherwise conformal growth by an ellipsoidal max-plus dilation of the
 * top-height map (cg1:457-540) followed by identify_cavity (cg1:543).
 * Cells that are not square: the reference pads the height map by Rx = ceil(t/dx) cells in both
 * directions (cg1:471, 666); a lateral step (deposit or etch) whose taps reach further along y,
 * ceil(t/dy) > Rx, reads outside that pad there and is refused with CG_ERR_UNSUPPORTED. */
int cg_deposit_general(cg_grid* g, int mask_id, int mat, double thickness, double conformality);

/* etch_general(maskFn, matId, thickness, conformality) (cg1:598-740); mat 255 = ALL. */
int cg_etch_general(cg_grid* g, int mask_id, int mat, double thickness, double conformality);

/* cmp(depth, stopMatId) (cg1:747-859). */
int cg_cmp(cg_grid* g, double depth, int stop_mat);

/* strip_connected(matId) (cg1:867 - cg2:1). */
int cg_strip_connected(cg_grid* g, int mat);

/* identify_cavity() (cg2:9-83). */
int cg_identify_cavity(cg_grid* g);

/* ---- cmp / strip_connected / identify_cavity in pieces, for a domain cut into slabs -------
 * No counterpart in the reference (it is one array).  These steps need whole-domain scalars
 * and a flood fill that crosses slab edges, so the driver (fem_amd/slabs.py) runs them as
 *   cmp:              cg_cmp_scan on every slab, max-reduce (top_q, stopper_q, found),
 *                     targetZ as cg1:781, cg_cmp_apply, then identify_cavity as below;
 *   identify_cavity:  cg_cavity_state, OR-reduce the flags (nothing to do when both are clear
 *                     everywhere), max-reduce top_q, cg_cavity_fill(hmax_q), strip of material 2;
 *   strip_connected:  planes = max-reduced max_len; cg_ghost_pack + swap of one boundary row
 *                     with each neighbour, cg_strip_seed, pack + swap again, then repeat
 *                     { cg_strip_sweep; stop when no slab changed; pack + swap }, cg_strip_finish.
 * A ghost message is len[ny] (padded to a multiple of 4 bytes) followed by rec[planes][ny]
 * (u32 records, flood marks included); cg_ghost_bytes gives its size.  ghost_lower / ghost_upper
 * are the messages received from the slab below / above, NULL where the domain ends.
 * cg_cmp, cg_strip_connected and cg_identify_cavity are these pieces run on an undivided grid. */
int cg_cmp_scan(cg_grid* g, int stop_mat, uint32_t* top_q, uint32_t* stopper_q, int* stopper_found);   /* cg1:755-781 */
int cg_cmp_apply(cg_grid* g, double target_z, int stop_mat);                                          /* cg1:786-853 */
int cg_cavity_state(cg_grid* g, int* has_gaps, int* has_cavity, int* max_len, uint32_t* top_q);
int cg_cavity_fill(cg_grid* g, uint32_t hmax_q);                                                       /* cg2:23-71 */
int cg_ghost_bytes(const cg_grid* g, int planes, size_t* bytes);
int cg_ghost_pack(cg_grid* g, int planes, void* send_lower, void* send_upper);
int cg_strip_seed(cg_grid* g, int mat, int planes, const void* ghost_lower, const void* ghost_upper);  /* cg1:910-929 */
int cg_strip_sweep(cg_grid* g, int mat, int planes, const void* ghost_lower, const void* ghost_upper, int* changed);   /* cg1:934-965 */
int cg_strip_finish(cg_grid* g);                                                                       /* cg1:970-999 */

/* etch_wet(matId, totalThk, {dt}) without the slider cache (cg2:893 - cg3:308): isotropic
 * removal of `mat` (255 = every solid, id >= 3) in increments of dt nm, each followed by
 * identify_cavity.  total_thickness <= 0: no-op.  dt = NaN: the reference's default
 * min(dx, dy).  Frees the engine's scratch planes before it returns. */
int cg_etch_wet(cg_grid* g, int mat, double total_thickness, double dt);

/* One increment of etch_wet: stepOnce(dtQ, useDiagonal) (cg3:250-268), identify_cavity
 * included.  dt_q >= 1 is the increment in height steps.  A host that keeps the reference's
 * anchors (sliderCache.wetetchCache, cg3:272-306) drives the loop itself with this call,
 * cg_snapshot_save/load and cg_release_scratch. */
int cg_etch_wet_step(cg_grid* g, int mat, int dt_q, int use_diagonal);

/* The same increment on a slab of a decomposed domain, WITHOUT the cavity pass: the caller has
 * packed and swapped one ghost row per side (cg_ghost_pack with `planes` = the largest column
 * of any slab), passes the whole domain's quantised maximum height, and afterwards runs the
 * cavity pass over all slabs (cg_cavity_state / cg_cavity_fill / cg_strip_*).  Null ghost
 * pointers = the domain ends there.  No counterpart in the reference (it is single-threaded). */
int cg_etch_wet_step_slab(cg_grid* g, int mat, int dt_q, int use_diagonal, uint32_t hmax_q, int planes,
                          const void* ghost_lower, const void* ghost_upper);

/* deposit_ALD(matId, totalThk, {dt}) without the slider cache (cg2:111-868): seeds on every
 * exposed face, a coat of thickness t grown from them for t = dt, 2dt, ... < total (only the
 * cavities it pinches off are kept), then the final coat and identify_cavity.
 * dt = NaN: the reference's default max(dx, dy).  Frees the scratch planes and the seeds. */
int cg_deposit_ald(cg_grid* g, int mat, double total_thickness, double dt);

/* The pieces of deposit_ALD for a host that keeps the reference's cache
 * (sliderCache.aldCache, cg2:131-136, 821-865):
 *   cg_ald_collect_seeds  collectALDSeeds (cg2:405-463) for the current grid; the seeds stay
 *                         in the handle until replaced or dropped.  count may be NULL.
 *   cg_ald_seed_info      are seeds held, and how many.
 *   cg_ald_increment      one pass of the loop body cg2:844-854 at thickness t.
 *   cg_ald_coat           computeCoatBitsSingleT + applyCoatFromBits at thickness t
 *                         (cg2:861-862); the caller snapshots, then calls cg_identify_cavity. */
int cg_ald_collect_seeds(cg_grid* g, double total_thickness, uint64_t* count);
int cg_ald_seed_info(cg_grid* g, int* held, uint64_t* count);
int cg_ald_drop_seeds(cg_grid* g);
int cg_ald_increment(cg_grid* g, int mat, double t);
int cg_ald_coat(cg_grid* g, int mat, double t);

/* deposit_ALD on a slab of a decomposed domain (no counterpart in the reference, which is
 * single-threaded).  The seeds are collected once (cg2:826), so the slabs talk once:
 *   cg_ald_collect_seeds_slab  collectALDSeeds for the slab's own rows.  The caller has packed and
 *                         swapped one ghost row per side (cg_ghost_pack, `planes` = the largest
 *                         column of any slab; null = the domain ends there) and passes the whole
 *                         domain's quantised maximum height.  *seed_planes = most seeds in one of
 *                         the slab's columns; the caller takes the maximum over the slabs.
 *   cg_ald_seed_halo_*    the seeds of the slab's first / last ceil(total/dx) rows as messages for
 *                         the lower / upper neighbour (counts u16[rows*ny] padded to 8 bytes, then
 *                         (zA, zB) u32 pairs [seed_planes][rows*ny]), and the neighbours' messages
 *                         into the handle.  Device pointers; pack, swap, unpack.
 *   cg_ald_coat_slab      cg_ald_coat against the whole domain's height (cg2:501).
 *   cg_ald_merge_slab     mergeNewCavityFromTmp (cg2:766-814) after the cavity pass over all slabs
 *                         (cg_cavity_state / cg_cavity_fill / cg_strip_*); hbase_q = the whole
 *                         domain's height before the coat, i.e. what cg_ald_coat_slab was given. */
int cg_ald_collect_seeds_slab(cg_grid* g, double total_thickness, uint32_t hmax_q, int planes,
                              const void* ghost_lower, const void* ghost_upper, int* seed_planes, uint64_t* count);
int cg_ald_seed_halo_bytes(const cg_grid* g, int seed_planes, size_t* bytes);
int cg_ald_seed_halo_pack(cg_grid* g, int seed_planes, void* send_lower, void* send_upper);
int cg_ald_seed_halo_unpack(cg_grid* g, int seed_planes, const void* recv_lower, const void* recv_upper);
int cg_ald_coat_slab(cg_grid* g, int mat, double t, uint32_t hmax_q);
int cg_ald_merge_slab(cg_grid* g, uint32_t hbase_q);

/* Frees the second and third set of planes that etch_wet / deposit_ALD steps work in
 * (they are as large as the grid itself) and the tables cg_mask_from_shapes keeps.  No
 * counterpart in the reference. */
int cg_release_scratch(cg_grid* g);

/* addLayer(i, j, matId, z0, z1) (cg1:196-211): appends one layer to column (i, j), heights
 * quantised like _quantizeZ and stored as Uint16; grows the planes when the column is full
 * (cg1:199).  A layer that starts above the column top leaves a gap below it, as in the
 * reference; one that starts BELOW the top cannot be held by the engine's gapless intervals
 * and is refused with CG_ERR_UNSUPPORTED (no reference step produces such a column).
 * removeTopLayer(i, j) (cg1:213-218): drops the top layer; no-op on an empty column.  The freed
 * slot reads back as zeros (the reference leaves its old bytes there until it is reused). */
int cg_add_layer(cg_grid* g, int i, int j, int mat, double z0_nm, double z1_nm);
int cg_remove_top_layer(cg_grid* g, int i, int j);

/* maxHeight() (cg2:85-104): max top z over this handle's columns in nm, 1 if that is 0.
 * top_q receives the quantised maximum (0 for an empty grid); either pointer may be NULL. */
int cg_max_height(cg_grid* g, double* z_nm, uint32_t* top_q);

/* ---- lateral-stencil steps split in two for the multi-GPU halo exchange --------------
 * No counterpart in the reference.  cg_deposit_general / cg_etch_general on a slab whose
 * neighbours matter are run as:
 *   cg_stencil_begin  -> snapshot of the top heights (+ etchable flags) of the slab,
 *                        returns the stencil radius R in rows (0: vertical fast path or
 *                        no-op, nothing to exchange, the step is already complete);
 *   cg_halo_buffers   -> device pointers of the rows to send to / receive from the lower
 *                        (i-1) and upper (i+1) neighbour, each R*ny*3 bytes
 *                        (u16 top height rows followed by u8 etchable-flag rows);
 *   (caller moves the bytes, e.g. with RCCL send/recv on the handle's stream)
 *   cg_stencil_interior (optional, while the bytes are in flight) -> the same update for
 *                        the tile rows whose stencil stays inside the slab; they read no
 *                        halo message, so the exchange overlaps with them;
 *   cg_stencil_finish -> dilation / erosion and the per-column update (of the rows
 *                        cg_stencil_interior has not done).
 * A deposit without a mask that finds a run of vertical steps waiting on a grid that has just been reset
 * takes that run along as well: cg_stencil_begin then applies the run to the rows near the slab's edges
 * only (the rows to send are packed from them) and the run stays queued while the step is open --
 * cg_halo_buffers, cg_stencil_interior and cg_stencil_finish do not launch it, any other call does.
 * A deposit that waits for the vertical steps behind it (see the top of this file) launches nothing
 * in cg_stencil_interior nor in cg_stencil_finish: both parts go out with the queued steps, and the
 * received halo rows are read then -- they must stay untouched until the handle's next call that is
 * not a vertical step (the next cg_stencil_begin of a lateral step packs only after it has launched
 * them, so a driver that exchanges after every cg_stencil_begin needs no extra care).
 * has_lower / has_upper = 0 means domain edge: heights are clamp-replicated and flags are
 * zero there (cg1:373-387, 671).
 */
#define CG_STENCIL_DEPOSIT 0
#define CG_STENCIL_ETCH    1
int cg_stencil_begin(cg_grid* g, int kind, int mask_id, int mat, double thickness,
                     double conformality, int has_lower, int has_upper, int* radius_rows);
int cg_stencil_interior(cg_grid* g);
int cg_halo_buffers(cg_grid* g, void** send_lower, void** send_upper,
                    void** recv_lower, void** recv_upper, size_t* bytes_each);
int cg_stencil_finish(cg_grid* g);
/* Let the caller own the four halo buffers (device memory, bytes_each >= R*ny*3 for every
 * step to come), e.g. tensors of the framework that does the communication.  Four NULLs
 * return to library-owned buffers. */
int cg_set_halo_buffers(cg_grid* g, void* send_lower, void* send_upper,
                        void* recv_lower, void* recv_upper, size_t bytes_each);

/* ---- data exchange with the host (the `cols` contract, script_gridrenderer.js:216-263) */

/* Copy the grid into caller-owned arrays in the reference layout with stride lmax
 * (cg_info.lmax): mat[n*lmax], zpair[2*n*lmax], len[n], n = nx*ny.  Dead slots are 0. */
int cg_download(cg_grid* g, uint8_t* mat, uint16_t* zpair, uint8_t* len);

/* Replace the grid by host arrays in the reference layout with the given stride
 * (loadCache's restore, cg1:276-278).  lmax becomes the handle's Lmax. */
int cg_upload(cg_grid* g, const uint8_t* mat, const uint16_t* zpair, const uint8_t* len, int lmax);

/* Device-side snapshots: saveCache / loadCache keep the grid in HBM (cg1:247-282); the
 * use-count / eviction policy stays with the host.  Slots are caller-chosen ints >= 0. */
int cg_snapshot_save(cg_grid* g, int slot);
int cg_snapshot_load(cg_grid* g, int slot);
int cg_snapshot_drop(cg_grid* g, int slot);

/* Cross-sections (sl:18-40): copy `count` whole grid lines into host arrays in the reference
 * layout.  axis 0 = 'YZ' (fixed i = index[s], ny columns each), axis 1 = 'XZ' (fixed j,
 * nx columns each).  Output s starts at s*line_cols*lmax (mat), 2*that (zpair),
 * s*line_cols (len). */
int cg_slices(cg_grid* g, int axis, const int32_t* index, int count,
              uint8_t* mat, uint16_t* zpair, uint8_t* len);
/* Same but leaves the result in device memory owned by the handle (valid until the next
 * cg_slices* call); for consumers that stay on the GPU. */
int cg_slices_device(cg_grid* g, int axis, const int32_t* index, int count,
                     void** d_mat, void** d_zpair, void** d_len);

/* The same cross-sections for callers that cannot allocate device memory: tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference.  The caller owns the
 * outputs, all in DEVICE memory: d_mat count*line_cols*lmax bytes (4-byte aligned), d_zpair 4 times
 * that (16-byte aligned), d_len count*line_cols bytes (line_cols = ny for axis 0, nx for axis 1;
 * lmax = cg_info.lmax); hipMalloc'ed buffers satisfy the alignments.  index is a host array.  The
 * call calls no hipMalloc / hipFree: the index list is staged in a fixed array of
 * CG_SLICES_NOALLOC_MAX entries (16 KiB) that lives in the library's device image, one copy per GPU,
 * shared by every handle on that GPU (its address is looked up when a handle is created).  count
 * beyond the maximum is refused with CG_ERR_ARG.  Work is enqueued on the handle's stream and not
 * waited for, but the call is NOT meant for stream capture: it first launches queued vertical steps
 * and copies the index list from the caller's (pageable) host memory.  Two calls that overlap in time
 * on one GPU (two handles on different streams, two threads) race on the shared array. */
#define CG_SLICES_NOALLOC_MAX 4096
int cg_slices_noalloc(cg_grid* g, int axis, const int32_t* index, int count,
                      void* d_mat, void* d_zpair, void* d_len);

/* indexFromX / indexFromY (sl:15-16): clamp(round((pos - offset)/pitch), 0, n-1). */
int cg_index_from_pos(const cg_grid* g, int axis, double pos_nm, int32_t* index);

/* ---- a grid cut into slabs that the library drives itself --------------------------------------------
 * No counterpart in the reference (it is single-threaded, one array).  A cg_group is ONE grid of nx_global x ny
 * columns whose rows are dealt out over `nslabs` slab handles, slab k on HIP device devices[k] (several slabs may
 * name the same device); the calling process drives all of them.  Every cg_group_* call below is the handle call of
 * the same name applied to the whole domain -- same arguments, same results as on an undivided grid, bit for bit:
 * lateral steps swap R halo rows between neighbouring slabs (the rows that need none are updated while those
 * travel), strip_connected / identify_cavity flood across the slab edges, cmp / maxHeight / etch_wet / deposit_ALD
 * combine the slabs' scalars, cg_group_download / _upload / _slices use the reference layout of the WHOLE grid, and
 * cg_info.lmax is the reference's one Lmax.  Rows travel device to device over RCCL (ncclSend / ncclRecv) when every
 * slab has a device of its own and librccl can be loaded, by hipMemcpyPeerAsync otherwise; the environment variable
 * CG_GROUP_TRANSPORT=copy|rccl forces one.  Mask ids are the group's own.  A group is not re-entrant.
 * This is the one-process form of the decomposition; fem_amd/slabs.py drives the same handle calls with one process
 * per GPU over torch.distributed. */
typedef struct cg_group cg_group;
int cg_group_create(int nx_global, int ny, double dx, double dy, int lmax_init, double zscale,
                    const int* devices, int nslabs, cg_group** out);
int cg_group_destroy(cg_group* g);
const char* cg_group_last_error(const cg_group* g);           /* g == NULL: the last failed create on this thread */
int cg_group_slab_count(const cg_group* g, int* nslabs);
int cg_group_slab(cg_group* g, int k, cg_grid** slab);        /* the handle of slab k (owned by the group) */
/* which transport the group uses (1 = RCCL), and how many exchanges / bytes it has moved so far */
int cg_group_transport(const cg_group* g, int* rccl, uint64_t* exchanges, uint64_t* bytes_moved);
int cg_group_get_info(cg_group* g, cg_info* out);             /* nx = nx_global, i0 = 0, lmax = the domain's */
int cg_group_sync(cg_group* g);
int cg_group_reset(cg_group* g, int lmax_init);                                                   /* cg_reset */
int cg_group_clear(cg_group* g);                                                                  /* cg_clear */
int cg_group_set_plane_budget(cg_group* g, int max_planes);
int cg_group_mask_from_shapes(cg_group* g, const cg_shape* shapes, int nshapes, const double* pts, int npts, int* mask_id);
int cg_group_mask_from_raster(cg_group* g, const uint8_t* host_raster, size_t n, int* mask_id);  /* n = nx_global*ny */
int cg_group_mask_download(cg_group* g, int mask_id, uint8_t* host_out, size_t n);
int cg_group_mask_destroy(cg_group* g, int mask_id);
int cg_group_deposit_general(cg_group* g, int mask_id, int mat, double thickness, double conformality);   /* cg1:435-544 */
int cg_group_etch_general(cg_group* g, int mask_id, int mat, double thickness, double conformality);      /* cg1:598-740 */
int cg_group_cmp(cg_group* g, double depth, int stop_mat);                                               /* cg1:747-859 */
int cg_group_strip_connected(cg_group* g, int mat);                                                      /* cg1:867 - cg2:1 */
int cg_group_identify_cavity(cg_group* g);                                                               /* cg2:9-83 */
int cg_group_etch_wet(cg_group* g, int mat, double total_thickness, double dt);                           /* cg_etch_wet */
int cg_group_etch_wet_step(cg_group* g, int mat, int dt_q, int use_diagonal);                             /* cg_etch_wet_step */
int cg_group_deposit_ald(cg_group* g, int mat, double total_thickness, double dt);                        /* cg_deposit_ald */
int cg_group_ald_collect_seeds(cg_group* g, double total_thickness, uint64_t* count);                     /* cg_ald_collect_seeds */
int cg_group_ald_drop_seeds(cg_group* g);
int cg_group_ald_increment(cg_group* g, int mat, double t);                                               /* cg_ald_increment */
int cg_group_ald_coat(cg_group* g, int mat, double t);                                                    /* cg_ald_coat */
int cg_group_release_scratch(cg_group* g);
int cg_group_add_layer(cg_group* g, int i, int j, int mat, double z0_nm, double z1_nm);                   /* global i */
int cg_group_remove_top_layer(cg_group* g, int i, int j);
int cg_group_max_height(cg_group* g, double* z_nm, uint32_t* top_q);                                      /* cg2:85-104 */
int cg_group_checksum(cg_group* g, uint64_t* sum);
int cg_group_download(cg_group* g, uint8_t* mat, uint16_t* zpair, uint8_t* len);
int cg_group_upload(cg_group* g, const uint8_t* mat, const uint16_t* zpair, const uint8_t* len, int lmax);
int cg_group_snapshot_save(cg_group* g, int slot);
int cg_group_snapshot_load(cg_group* g, int slot);
int cg_group_snapshot_drop(cg_group* g, int slot);
int cg_group_slices(cg_group* g, int axis, const int32_t* index, int count, uint8_t* mat, uint16_t* zpair, uint8_t* len);
int cg_group_index_from_pos(const cg_group* g, int axis, double pos_nm, int32_t* index);

/* ---- checks and measurement ---------------------------------------------------------- */

/* Order-independent 64-bit checksum of the live layers: sum over columns of a per-column
 * hash of (global column index, len, mat/z0q/z1q of each live layer), mod 2^64.  The sum
 * over the slabs of a decomposed grid equals the checksum of the whole grid. */
int cg_checksum(cg_grid* g, uint64_t* sum);

/* Column counts per case of the last step, for the algorithmic-byte model:
 * [0] columns visited, [1] open (mask) columns, [2] top layer extended, [3] new layer
 * appended, [4] columns etched, [5] whole layers removed, [6] layers scanned for the floor. */
#define CG_NSTAT 8
int cg_step_stats(cg_grid* g, int enable, uint64_t stats[CG_NSTAT]);

/* Per-kernel timing.  While enabled, every kernel launch of this handle is bracketed by a
 * pair of HIP events on the handle's stream (no host synchronisation at launch time).
 * A call synchronises the stream, returns the milliseconds and launch counts accumulated
 * per kernel kind since the previous call, clears them, and sets the enabled state. */
#define CG_K_DEPO_VERTICAL   0   /* k_vertical_g4<deposit> (k_vertical_1 when N % 4 != 0)       */
#define CG_K_ETCH_VERTICAL   1   /* k_vertical_g4<etch>                                          */
#define CG_K_VERTICAL_FUSED  2   /* k_vertical_fused: a run of consecutive vertical steps in one pass over the columns */
#define CG_K_HALO            3   /* k_halo_pack, k_ghost_pack                                  */
#define CG_K_STENCIL_DEPOSIT 4   /* k_stencil_stream / k_stencil_sym <deposit> (k_stencil_apply for radius > 8) */
#define CG_K_STENCIL_ETCH    5   /* k_stencil_stream / k_stencil_sym <etch>                        */
#define CG_K_MASK            6   /* k_mask_shapes                                              */
#define CG_K_CLEAR           7   /* memsets of cg_reset / cg_clear                             */
#define CG_K_REDUCE          8   /* k_max_top                                                  */
#define CG_K_OTHER           9
#define CG_K_STENCIL_DEPOSIT_STEPS 10 /* k_stencil_lean<deposit> with the vertical steps that follow it applied inside the launch */
#define CG_NKIND 12
int cg_profile(cg_grid* g, int enable, double ms[CG_NKIND], uint64_t count[CG_NKIND]);

/* HIP-event timing on the handle's stream: cg_timer_start records an event, cg_timer_stop
 * records a second one, waits for it and returns the elapsed milliseconds in between. */
int cg_timer_start(cg_grid* g);
int cg_timer_stop(cg_grid* g, float* ms);

/* Library version string, e.g. "colgrid 0.1 gfx950". */
const char* cg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* COLGRID_H */
