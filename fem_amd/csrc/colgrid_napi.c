/*
 * colgrid_napi.c -- Node.js N-API glue over the C ABI of libcolgrid (include/colgrid.h).
 *
 * Plain C N-API (node_api.h); one JavaScript function per C entry point, same argument
 * order.  A grid handle is a JavaScript external whose finalizer calls cg_destroy; a handle made by
 * createGroup stands for a cg_group -- one grid cut into slabs on several devices, driven by the library --
 * and every function below takes either kind: the call goes to cg_xxx or to cg_group_xxx.  Typed
 * arrays passed in are only read during the call (the library never keeps host pointers);
 * arrays handed back are freshly allocated ArrayBuffers owned by JavaScript.  A non-zero
 * status becomes a thrown Error carrying cg_last_error.
 *
 * The JavaScript classes with the reference's ColumnGrid / ProcessRuntime call surface
 * (fem_amd/js/columngrid.js, runtime.js) are written against this module only.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/colgrid.h"

#define NAPI_OK(env, call)                                                        \
    do {                                                                          \
        if ((call) != napi_ok) {                                                  \
            napi_throw_error((env), NULL, "N-API call failed: " #call);          \
            return NULL;                                                          \
        }                                                                         \
    } while (0)

#define MAXARGS 8
#define ARGS(env, info, n)                                                        \
    size_t argc = MAXARGS; napi_value argv[MAXARGS];                              \
    NAPI_OK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));           \
    if (argc < (size_t)(n)) { napi_throw_type_error(env, NULL, "too few arguments"); return NULL; }

/* what a JavaScript handle points at: an undivided grid (or one slab), or a group of slabs */
typedef struct H { int group; void* p; } H;
#define GRID(h)  ((cg_grid*)(h)->p)
#define GROUP(h) ((cg_group*)(h)->p)
/* cg_xxx(handle, ...) or cg_group_xxx(group, ...): the two families share names and arguments */
#define CALL0(h, op)      ((h)->group ? cg_group_##op(GROUP(h)) : cg_##op(GRID(h)))
#define CALL(h, op, ...)  ((h)->group ? cg_group_##op(GROUP(h), __VA_ARGS__) : cg_##op(GRID(h), __VA_ARGS__))

static napi_value throw_cg(napi_env env, const H* h, int rc) {
    char buf[700];
    const char* msg = !h ? cg_last_error(NULL) : (h->group ? cg_group_last_error(GROUP(h)) : cg_last_error(GRID(h)));
    snprintf(buf, sizeof buf, "libcolgrid error %d: %s", rc, msg ? msg : "");
    napi_throw_error(env, NULL, buf);
    return NULL;
}

static void release(H* h) {
    if (h->p) { if (h->group) cg_group_destroy(GROUP(h)); else cg_destroy(GRID(h)); }
    h->p = NULL;
}

static void finalize_grid(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    H* box = (H*)data;
    if (box) { release(box); free(box); }
}

static H* grid_of(napi_env env, napi_value v) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !((H*)p)->p) {
        napi_throw_type_error(env, NULL, "not a live colgrid handle");
        return NULL;
    }
    return (H*)p;
}

static int get_double(napi_env env, napi_value v, double* out) {
    napi_value num;
    if (napi_coerce_to_number(env, v, &num) != napi_ok || napi_get_value_double(env, num, out) != napi_ok) {
        napi_throw_type_error(env, NULL, "number expected"); return 0;
    }
    return 1;
}
static int get_int(napi_env env, napi_value v, int32_t* out) {
    double d; if (!get_double(env, v, &d)) return 0;
    *out = (int32_t)d; return 1;
}

static napi_value wrap(napi_env env, void* g, int group) {
    H* box = (H*)malloc(sizeof *box);
    box->group = group; box->p = g;
    napi_value ext;
    if (napi_create_external(env, box, finalize_grid, NULL, &ext) != napi_ok) { release(box); free(box); napi_throw_error(env, NULL, "external"); return NULL; }
    return ext;
}
static napi_value wrap_grid(napi_env env, cg_grid* g) { return wrap(env, g, 0); }

static napi_value undefined(napi_env env) { napi_value u; napi_get_undefined(env, &u); return u; }

static napi_value new_typed(napi_env env, napi_typedarray_type type, size_t count, size_t elem, void** data) {
    napi_value ab, ta;
    if (napi_create_arraybuffer(env, count * elem, data, &ab) != napi_ok) { napi_throw_error(env, NULL, "arraybuffer"); return NULL; }
    if (napi_create_typedarray(env, type, count, ab, 0, &ta) != napi_ok) { napi_throw_error(env, NULL, "typedarray"); return NULL; }
    return ta;
}

static int typed_in(napi_env env, napi_value v, napi_typedarray_type want, void** data, size_t* count) {
    napi_typedarray_type t; napi_value ab; size_t off;
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is ||
        napi_get_typedarray_info(env, v, &t, count, data, &ab, &off) != napi_ok || t != want) {
        napi_throw_type_error(env, NULL, "typed array of the wrong kind"); return 0;
    }
    return 1;
}

/* ---- lifetime -------------------------------------------------------------------------- */
static napi_value js_create(napi_env env, napi_callback_info info) {          /* cg_create */
    ARGS(env, info, 6);
    double lx, ly, dx, dy, zs; int32_t lmax;
    if (!get_double(env, argv[0], &lx) || !get_double(env, argv[1], &ly) || !get_double(env, argv[2], &dx) ||
        !get_double(env, argv[3], &dy) || !get_int(env, argv[4], &lmax) || !get_double(env, argv[5], &zs)) return NULL;
    cg_grid* g = NULL;
    int rc = cg_create(lx, ly, dx, dy, lmax, zs, &g);
    if (rc) return throw_cg(env, NULL, rc);
    return wrap_grid(env, g);
}

static napi_value js_create_slab(napi_env env, napi_callback_info info) {     /* cg_create_slab */
    ARGS(env, info, 8);
    int32_t nxg, ny, i0, nxl, lmax; double dx, dy, zs;
    if (!get_int(env, argv[0], &nxg) || !get_int(env, argv[1], &ny) || !get_int(env, argv[2], &i0) || !get_int(env, argv[3], &nxl) ||
        !get_double(env, argv[4], &dx) || !get_double(env, argv[5], &dy) || !get_int(env, argv[6], &lmax) || !get_double(env, argv[7], &zs)) return NULL;
    cg_grid* g = NULL;
    int rc = cg_create_slab(nxg, ny, i0, nxl, dx, dy, lmax, zs, &g);
    if (rc) return throw_cg(env, NULL, rc);
    return wrap_grid(env, g);
}

/* createGroup(LXnm, LYnm, dx, dy, LmaxInit, zScale, Int32Array devices): cg_group_create over setDomain's NX, NY (cg1:49-60) */
static napi_value js_create_group(napi_env env, napi_callback_info info) {
    ARGS(env, info, 7);
    double lx, ly, dx, dy, zs; int32_t lmax; void* devs; size_t ndev;
    if (!get_double(env, argv[0], &lx) || !get_double(env, argv[1], &ly) || !get_double(env, argv[2], &dx) ||
        !get_double(env, argv[3], &dy) || !get_int(env, argv[4], &lmax) || !get_double(env, argv[5], &zs) ||
        !typed_in(env, argv[6], napi_int32_array, &devs, &ndev)) return NULL;
    if (!(dx > 0) || !(dy > 0)) { napi_throw_range_error(env, NULL, "bad pitch"); return NULL; }
    double nx = lx / dx, ny = ly / dy;                      /* Math.round: halves toward +infinity */
    nx = (nx - (double)(long long)nx >= 0.5) ? (double)(long long)nx + 1 : (double)(long long)nx;
    ny = (ny - (double)(long long)ny >= 0.5) ? (double)(long long)ny + 1 : (double)(long long)ny;
    cg_group* g = NULL;
    int rc = cg_group_create(nx < 1 ? 1 : (int)nx, ny < 1 ? 1 : (int)ny, dx, dy, lmax, zs, (const int*)devs, (int)ndev, &g);
    if (rc) {
        char buf[700]; snprintf(buf, sizeof buf, "libcolgrid error %d: %s", rc, cg_group_last_error(NULL));
        napi_throw_error(env, NULL, buf); return NULL;
    }
    return wrap(env, g, 1);
}

/* groupTransport(handle) -> { rccl, exchanges, bytes } (a plain grid: all zero) */
static napi_value js_group_transport(napi_env env, napi_callback_info info) {
    ARGS(env, info, 1);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int rccl = 0; uint64_t ex = 0, by = 0;
    if (g->group) { int rc = cg_group_transport(GROUP(g), &rccl, &ex, &by); if (rc) return throw_cg(env, g, rc); }
    napi_value o, v; NAPI_OK(env, napi_create_object(env, &o));
    napi_get_boolean(env, rccl != 0, &v); napi_set_named_property(env, o, "rccl", v);
    napi_create_double(env, (double)ex, &v); napi_set_named_property(env, o, "exchanges", v);
    napi_create_double(env, (double)by, &v); napi_set_named_property(env, o, "bytes", v);
    return o;
}

static napi_value js_destroy(napi_env env, napi_callback_info info) {         /* cg_destroy */
    ARGS(env, info, 1);
    void* p = NULL;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) release((H*)p);
    return undefined(env);
}

#define SIMPLE0(NAME, OP)                                                     \
    static napi_value NAME(napi_env env, napi_callback_info info) {           \
        ARGS(env, info, 1);                                                   \
        H* g = grid_of(env, argv[0]); if (!g) return NULL;                    \
        int rc = CALL0(g, OP); if (rc) return throw_cg(env, g, rc);           \
        return undefined(env);                                                \
    }
SIMPLE0(js_clear, clear)
SIMPLE0(js_identify_cavity, identify_cavity)
SIMPLE0(js_sync, sync)

static napi_value js_reset(napi_env env, napi_callback_info info) {           /* cg_reset */
    ARGS(env, info, 2);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t lmax; if (!get_int(env, argv[1], &lmax)) return NULL;
    int rc = CALL(g, reset, lmax); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}

static napi_value js_info(napi_env env, napi_callback_info info) {            /* cg_get_info */
    ARGS(env, info, 1);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    cg_info inf; int rc = CALL(g, get_info, &inf); if (rc) return throw_cg(env, g, rc);
    napi_value o, v; NAPI_OK(env, napi_create_object(env, &o));
#define SETI(name, val) do { napi_create_int32(env, (val), &v); napi_set_named_property(env, o, name, v); } while (0)
#define SETD(name, val) do { napi_create_double(env, (val), &v); napi_set_named_property(env, o, name, v); } while (0)
    SETI("nx", inf.nx); SETI("ny", inf.ny); SETI("nxGlobal", inf.nx_global); SETI("i0", inf.i0); SETI("lmax", inf.lmax); SETI("device", inf.device);
    SETD("dx", inf.dx); SETD("dy", inf.dy); SETD("zscale", inf.zscale); SETD("offsetX", inf.offset_x); SETD("offsetY", inf.offset_y);
    SETD("lxEff", inf.lx_eff); SETD("lyEff", inf.ly_eff);
    return o;
}

/* ---- masks ------------------------------------------------------------------------------- */
static napi_value js_mask_from_raster(napi_env env, napi_callback_info info) {   /* cg_mask_from_raster */
    ARGS(env, info, 2);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    void* data; size_t n; if (!typed_in(env, argv[1], napi_uint8_array, &data, &n)) return NULL;
    int id = -1; int rc = CALL(g, mask_from_raster, (const uint8_t*)data, n, &id); if (rc) return throw_cg(env, g, rc);
    napi_value v; napi_create_int32(env, id, &v); return v;
}

static napi_value js_mask_from_shapes(napi_env env, napi_callback_info info) {   /* cg_mask_from_shapes */
    ARGS(env, info, 4);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    void *sh, *pts; size_t nb, np; int32_t nshapes;
    if (!typed_in(env, argv[1], napi_uint8_array, &sh, &nb) || !get_int(env, argv[2], &nshapes) ||
        !typed_in(env, argv[3], napi_float64_array, &pts, &np)) return NULL;
    if (nshapes < 0 || (size_t)nshapes * sizeof(cg_shape) > nb) { napi_throw_range_error(env, NULL, "shape buffer too small"); return NULL; }
    /* the ArrayBuffer may not be 8-byte aligned for the doubles inside cg_shape: copy */
    cg_shape* copy = (cg_shape*)malloc((nshapes > 0 ? nshapes : 1) * sizeof(cg_shape));
    if (nshapes > 0) memcpy(copy, sh, (size_t)nshapes * sizeof(cg_shape));
    int id = -1; int rc = CALL(g, mask_from_shapes, copy, nshapes, (const double*)pts, (int)(np / 2), &id);
    free(copy);
    if (rc) return throw_cg(env, g, rc);
    napi_value v; napi_create_int32(env, id, &v); return v;
}

static napi_value js_mask_download(napi_env env, napi_callback_info info) {      /* cg_mask_download */
    ARGS(env, info, 2);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t id; if (!get_int(env, argv[1], &id)) return NULL;
    cg_info inf; CALL(g, get_info, &inf);
    size_t n = (size_t)inf.nx * inf.ny; void* data;
    napi_value ta = new_typed(env, napi_uint8_array, n, 1, &data); if (!ta) return NULL;
    int rc = CALL(g, mask_download, id, (uint8_t*)data, n); if (rc) return throw_cg(env, g, rc);
    return ta;
}

static napi_value js_mask_destroy(napi_env env, napi_callback_info info) {       /* cg_mask_destroy */
    ARGS(env, info, 2);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t id; if (!get_int(env, argv[1], &id)) return NULL;
    int rc = CALL(g, mask_destroy, id); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}

/* ---- steps --------------------------------------------------------------------------------- */
static napi_value step4(napi_env env, napi_callback_info info, int (*fn)(cg_grid*, int, int, double, double),
                        int (*gfn)(cg_group*, int, int, double, double)) {
    ARGS(env, info, 5);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t mask, mat; double t, S;
    if (!get_int(env, argv[1], &mask) || !get_int(env, argv[2], &mat) || !get_double(env, argv[3], &t) || !get_double(env, argv[4], &S)) return NULL;
    int rc = g->group ? gfn(GROUP(g), mask, mat, t, S) : fn(GRID(g), mask, mat, t, S); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}
static napi_value js_deposit_general(napi_env env, napi_callback_info info) { return step4(env, info, cg_deposit_general, cg_group_deposit_general); }
static napi_value js_etch_general(napi_env env, napi_callback_info info) { return step4(env, info, cg_etch_general, cg_group_etch_general); }

static napi_value js_cmp(napi_env env, napi_callback_info info) {                /* cg_cmp */
    ARGS(env, info, 3);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    double depth; int32_t stop; if (!get_double(env, argv[1], &depth) || !get_int(env, argv[2], &stop)) return NULL;
    int rc = CALL(g, cmp, depth, stop); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}

static napi_value js_strip_connected(napi_env env, napi_callback_info info) {    /* cg_strip_connected */
    ARGS(env, info, 2);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t mat; if (!get_int(env, argv[1], &mat)) return NULL;
    int rc = CALL(g, strip_connected, mat); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}

static napi_value js_max_height(napi_env env, napi_callback_info info) {         /* cg_max_height */
    ARGS(env, info, 1);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    double z; int rc = CALL(g, max_height, &z, NULL); if (rc) return throw_cg(env, g, rc);
    napi_value v; napi_create_double(env, z, &v); return v;
}

/* ---- etch_wet / deposit_ALD ------------------------------------------------------------------ */
static napi_value js_etch_wet_step(napi_env env, napi_callback_info info) {      /* cg_etch_wet_step */
    ARGS(env, info, 4);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t mat, dtq, diag;
    if (!get_int(env, argv[1], &mat) || !get_int(env, argv[2], &dtq) || !get_int(env, argv[3], &diag)) return NULL;
    int rc = CALL(g, etch_wet_step, mat, dtq, diag); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}
static napi_value whole_run(napi_env env, napi_callback_info info, int (*fn)(cg_grid*, int, double, double), int (*gfn)(cg_group*, int, double, double)) {
    ARGS(env, info, 4);                                                          /* (handle, mat, total, dt or NaN) */
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t mat; double total, dt;
    if (!get_int(env, argv[1], &mat) || !get_double(env, argv[2], &total) || !get_double(env, argv[3], &dt)) return NULL;
    int rc = g->group ? gfn(GROUP(g), mat, total, dt) : fn(GRID(g), mat, total, dt); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}
static napi_value js_etch_wet(napi_env env, napi_callback_info info) { return whole_run(env, info, cg_etch_wet, cg_group_etch_wet); }        /* cg_etch_wet */
static napi_value js_deposit_ald(napi_env env, napi_callback_info info) { return whole_run(env, info, cg_deposit_ald, cg_group_deposit_ald); }  /* cg_deposit_ald */

static napi_value js_ald_collect_seeds(napi_env env, napi_callback_info info) {  /* cg_ald_collect_seeds -> seed count */
    ARGS(env, info, 2);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    double total; if (!get_double(env, argv[1], &total)) return NULL;
    uint64_t n = 0;
    int rc = CALL(g, ald_collect_seeds, total, &n); if (rc) return throw_cg(env, g, rc);
    napi_value v; napi_create_double(env, (double)n, &v); return v;
}
static napi_value js_ald_drop_seeds(napi_env env, napi_callback_info info) {     /* cg_ald_drop_seeds */
    ARGS(env, info, 1);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int rc = CALL0(g, ald_drop_seeds); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}
static napi_value ald_at(napi_env env, napi_callback_info info, int (*fn)(cg_grid*, int, double), int (*gfn)(cg_group*, int, double)) {
    ARGS(env, info, 3);                                                          /* (handle, mat, t) */
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t mat; double t;
    if (!get_int(env, argv[1], &mat) || !get_double(env, argv[2], &t)) return NULL;
    int rc = g->group ? gfn(GROUP(g), mat, t) : fn(GRID(g), mat, t); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}
static napi_value js_ald_increment(napi_env env, napi_callback_info info) { return ald_at(env, info, cg_ald_increment, cg_group_ald_increment); }   /* cg_ald_increment */
static napi_value js_ald_coat(napi_env env, napi_callback_info info) { return ald_at(env, info, cg_ald_coat, cg_group_ald_coat); }             /* cg_ald_coat */
static napi_value js_release_scratch(napi_env env, napi_callback_info info) {    /* cg_release_scratch */
    ARGS(env, info, 1);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int rc = CALL0(g, release_scratch); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}

static napi_value js_set_plane_budget(napi_env env, napi_callback_info info) {   /* cg_set_plane_budget */
    ARGS(env, info, 2);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t n; if (!get_int(env, argv[1], &n)) return NULL;
    int rc = CALL(g, set_plane_budget, n); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}

/* ---- single-column accessors ------------------------------------------------------------------ */
static napi_value js_add_layer(napi_env env, napi_callback_info info) {          /* cg_add_layer */
    ARGS(env, info, 6);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t i, j, mat; double z0, z1;
    if (!get_int(env, argv[1], &i) || !get_int(env, argv[2], &j) || !get_int(env, argv[3], &mat) ||
        !get_double(env, argv[4], &z0) || !get_double(env, argv[5], &z1)) return NULL;
    int rc = CALL(g, add_layer, i, j, mat, z0, z1); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}
static napi_value js_remove_top_layer(napi_env env, napi_callback_info info) {   /* cg_remove_top_layer */
    ARGS(env, info, 3);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t i, j;
    if (!get_int(env, argv[1], &i) || !get_int(env, argv[2], &j)) return NULL;
    int rc = CALL(g, remove_top_layer, i, j); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}

/* ---- data exchange ------------------------------------------------------------------------- */
static napi_value js_download(napi_env env, napi_callback_info info) {           /* cg_download */
    ARGS(env, info, 1);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    cg_info inf; CALL(g, get_info, &inf);
    size_t n = (size_t)inf.nx * inf.ny, L = (size_t)inf.lmax;
    void *m, *z, *l;
    napi_value tm = new_typed(env, napi_uint8_array, n * L, 1, &m); if (!tm) return NULL;
    napi_value tz = new_typed(env, napi_uint16_array, 2 * n * L, 2, &z); if (!tz) return NULL;
    napi_value tl = new_typed(env, napi_uint8_array, n, 1, &l); if (!tl) return NULL;
    int rc = CALL(g, download, (uint8_t*)m, (uint16_t*)z, (uint8_t*)l); if (rc) return throw_cg(env, g, rc);
    napi_value o, v; NAPI_OK(env, napi_create_object(env, &o));
    napi_set_named_property(env, o, "mat", tm); napi_set_named_property(env, o, "zpair", tz); napi_set_named_property(env, o, "len", tl);
    napi_create_int32(env, inf.lmax, &v); napi_set_named_property(env, o, "lmax", v);
    return o;
}

static napi_value js_upload(napi_env env, napi_callback_info info) {             /* cg_upload */
    ARGS(env, info, 5);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    void *m, *z, *l; size_t nm, nz, nl; int32_t lmax;
    if (!typed_in(env, argv[1], napi_uint8_array, &m, &nm) || !typed_in(env, argv[2], napi_uint16_array, &z, &nz) ||
        !typed_in(env, argv[3], napi_uint8_array, &l, &nl) || !get_int(env, argv[4], &lmax)) return NULL;
    cg_info inf; CALL(g, get_info, &inf);
    size_t n = (size_t)inf.nx * inf.ny;
    if (lmax < 1 || nl != n || nm != n * (size_t)lmax || nz != 2 * n * (size_t)lmax) { napi_throw_range_error(env, NULL, "array sizes do not match the grid"); return NULL; }
    int rc = CALL(g, upload, (const uint8_t*)m, (const uint16_t*)z, (const uint8_t*)l, lmax); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}

static napi_value snap(napi_env env, napi_callback_info info, int (*fn)(cg_grid*, int), int (*gfn)(cg_group*, int)) {
    ARGS(env, info, 2);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t slot; if (!get_int(env, argv[1], &slot)) return NULL;
    int rc = g->group ? gfn(GROUP(g), slot) : fn(GRID(g), slot); if (rc) return throw_cg(env, g, rc);
    return undefined(env);
}
static napi_value js_snapshot_save(napi_env env, napi_callback_info info) { return snap(env, info, cg_snapshot_save, cg_group_snapshot_save); }
static napi_value js_snapshot_load(napi_env env, napi_callback_info info) { return snap(env, info, cg_snapshot_load, cg_group_snapshot_load); }
static napi_value js_snapshot_drop(napi_env env, napi_callback_info info) { return snap(env, info, cg_snapshot_drop, cg_group_snapshot_drop); }

static napi_value js_slices(napi_env env, napi_callback_info info) {             /* cg_slices */
    ARGS(env, info, 3);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t axis; void* idx; size_t cnt;
    if (!get_int(env, argv[1], &axis) || !typed_in(env, argv[2], napi_int32_array, &idx, &cnt)) return NULL;
    cg_info inf; CALL(g, get_info, &inf);
    size_t line = axis == 0 ? (size_t)inf.ny : (size_t)inf.nx, n = line * cnt, L = (size_t)inf.lmax;
    void *m, *z, *l;
    napi_value tm = new_typed(env, napi_uint8_array, n * L, 1, &m); if (!tm) return NULL;
    napi_value tz = new_typed(env, napi_uint16_array, 2 * n * L, 2, &z); if (!tz) return NULL;
    napi_value tl = new_typed(env, napi_uint8_array, n, 1, &l); if (!tl) return NULL;
    int rc = CALL(g, slices, axis, (const int32_t*)idx, (int)cnt, (uint8_t*)m, (uint16_t*)z, (uint8_t*)l); if (rc) return throw_cg(env, g, rc);
    napi_value o, v; NAPI_OK(env, napi_create_object(env, &o));
    napi_set_named_property(env, o, "mat", tm); napi_set_named_property(env, o, "zpair", tz); napi_set_named_property(env, o, "len", tl);
    napi_create_int32(env, inf.lmax, &v); napi_set_named_property(env, o, "lmax", v);
    napi_create_int32(env, (int32_t)line, &v); napi_set_named_property(env, o, "line", v);
    return o;
}

static napi_value js_index_from_pos(napi_env env, napi_callback_info info) {     /* cg_index_from_pos */
    ARGS(env, info, 3);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    int32_t axis; double pos; if (!get_int(env, argv[1], &axis) || !get_double(env, argv[2], &pos)) return NULL;
    int32_t out = 0; int rc = CALL(g, index_from_pos, axis, pos, &out); if (rc) return throw_cg(env, g, rc);
    napi_value v; napi_create_int32(env, out, &v); return v;
}

static napi_value js_checksum(napi_env env, napi_callback_info info) {           /* cg_checksum -> 16 hex digits */
    ARGS(env, info, 1);
    H* g = grid_of(env, argv[0]); if (!g) return NULL;
    uint64_t s = 0; int rc = CALL(g, checksum, &s); if (rc) return throw_cg(env, g, rc);
    char buf[32]; snprintf(buf, sizeof buf, "%016llx", (unsigned long long)s);
    napi_value v; napi_create_string_utf8(env, buf, NAPI_AUTO_LENGTH, &v); return v;
}

static napi_value js_version(napi_env env, napi_callback_info info) {            /* cg_version */
    (void)info; napi_value v; napi_create_string_utf8(env, cg_version(), NAPI_AUTO_LENGTH, &v); return v;
}

static napi_value js_shape_bytes(napi_env env, napi_callback_info info) {        /* sizeof(cg_shape), for the packing code */
    (void)info; napi_value v; napi_create_int32(env, (int32_t)sizeof(cg_shape), &v); return v;
}

static napi_value init(napi_env env, napi_value exports) {
    static const struct { const char* name; napi_callback fn; } fns[] = {
        {"create", js_create}, {"createSlab", js_create_slab}, {"createGroup", js_create_group}, {"groupTransport", js_group_transport},
        {"destroy", js_destroy}, {"reset", js_reset}, {"clear", js_clear},
        {"info", js_info}, {"sync", js_sync},
        {"maskFromRaster", js_mask_from_raster}, {"maskFromShapes", js_mask_from_shapes}, {"maskDownload", js_mask_download},
        {"maskDestroy", js_mask_destroy},
        {"depositGeneral", js_deposit_general}, {"etchGeneral", js_etch_general}, {"cmp", js_cmp},
        {"stripConnected", js_strip_connected}, {"identifyCavity", js_identify_cavity}, {"maxHeight", js_max_height},
        {"download", js_download}, {"upload", js_upload},
        {"snapshotSave", js_snapshot_save}, {"snapshotLoad", js_snapshot_load}, {"snapshotDrop", js_snapshot_drop},
        {"slices", js_slices}, {"indexFromPos", js_index_from_pos}, {"checksum", js_checksum}, {"version", js_version},
        {"shapeBytes", js_shape_bytes},
        {"etchWetStep", js_etch_wet_step}, {"etchWet", js_etch_wet}, {"depositAld", js_deposit_ald},
        {"aldCollectSeeds", js_ald_collect_seeds}, {"aldDropSeeds", js_ald_drop_seeds}, {"aldIncrement", js_ald_increment},
        {"aldCoat", js_ald_coat}, {"releaseScratch", js_release_scratch},
        {"addLayer", js_add_layer}, {"removeTopLayer", js_remove_top_layer}, {"setPlaneBudget", js_set_plane_budget},
    };
    for (size_t k = 0; k < sizeof fns / sizeof fns[0]; k++) {
        napi_value f;
        if (napi_create_function(env, fns[k].name, NAPI_AUTO_LENGTH, fns[k].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[k].name, f) != napi_ok) return NULL;
    }
    return exports;
}

NAPI_MODULE(colgrid_napi, init)
