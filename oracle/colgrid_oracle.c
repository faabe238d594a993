/*
 * colgrid_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the reference's column-grid algorithms, kept in
 * the reference's own host layout (mat[c*Lmax+k], zpair[2*(c*Lmax+k)+{0,1}], len[c]) and
 * written loop for loop after the JavaScript so that it can be read side by side with it.
 * It exists to CHECK the HIP engine: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (libcolgrid) never links or calls it.
 *
 * Pinning: the reference holds no tests or golden vectors.  This restatement is pinned by
 * outputs of the reference itself, produced in the build container by oracle/gen_golden.js
 * (Node running the reference's files in place) and committed under tests/golden/;
 * tests/test_oracle_golden.py replays every fixture through this file and requires equality
 * of the complete len / mat / zpair arrays (dead slots included).
 *
 * Citations: cg1 = template/script_columngrid.js, cg2 = template/script_columngrid2.js,
 * rt = template/script_runtime.js of the reference.
 *
 * JavaScript arithmetic that is reproduced on purpose:
 *   - Math.round rounds halves toward +infinity;
 *   - stores into Uint16Array / Uint8Array wrap modulo 2^16 / 2^8;
 *   - Float32Array stores round to float, every +,-,compare is done in double on the
 *     widened value (build with -ffp-contract=off so that no FMA is formed).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/colgrid.h"   /* cg_shape, material ids */

#define OG_EPS 1e-9

typedef struct og_kernel {
    int key[4];
    int Rx, Ry;
    double baseAdd;
    int *row_off;      /* 2*Rx+2 offsets into taps */
    float *dv;         /* per tap: dv, stored as float like the Float32Array rows */
    float *add;
} og_kernel;

typedef struct og_grid {
    int NX, NY;
    int i0, NXglobal;        /* slab support (test extension); i0 = 0, NXglobal = NX normally */
    double dx, dy, zScale;
    int Lmax, expandStep;
    uint8_t *mat; uint16_t *zpair; uint8_t *len;
    /* conformal kernel cache, FIFO of at most 32 entries (cg1:324-357) */
    og_kernel *kc[40]; int nkc;
    /* statistics of the last stencil/vertical step (for the byte model) */
    uint64_t stats[CG_NSTAT];
    /* sliderCache of the reference (cg1:34): filled by deposit_ALD / etch_wet */
    struct og_slider *slider;
} og_grid;

void og_slider_reset(og_grid *g);

/* ---- JS number helpers ---------------------------------------------------------------- */
static double js_round(double x) {           /* Math.round */
    double r = floor(x);
    if (x - r >= 0.5) r += 1.0;
    return r;
}
static uint16_t to_u16(double v) {           /* ToUint16 */
    if (!isfinite(v)) return 0;
    double m = fmod(trunc(v), 65536.0);
    if (m < 0) m += 65536.0;
    return (uint16_t)m;
}
static int next_pow2(int n) {                /* cg1:302-304 */
    int v = n - 1; if (v < 1) v = 1;
    int lz = __builtin_clz((unsigned)v);
    return 1 << (32 - lz);
}
static double quantZ(const og_grid *g, double z) { return js_round(z / g->zScale); }   /* cg1:117-119 */
static double deqZ(const og_grid *g, unsigned q) { return (double)q * g->zScale; }      /* cg1:120-122 */

/* ---- lifetime (cg1:21-95) ------------------------------------------------------------- */
static int og_alloc(og_grid *g) {
    size_t N = (size_t)g->NX * g->NY, S = N * g->Lmax;
    free(g->mat); free(g->zpair); free(g->len);
    g->mat = calloc(S ? S : 1, 1);
    g->zpair = calloc(S ? 2 * S : 1, 2);
    g->len = calloc(N ? N : 1, 1);
    return (g->mat && g->zpair && g->len) ? 0 : 1;
}

og_grid *og_create_slab(int nx_global, int ny, int i0, int nx_local, double dx, double dy,
                        int lmax_init, double zscale) {
    og_grid *g = calloc(1, sizeof *g);
    if (!g) return NULL;
    g->NX = nx_local; g->NY = ny; g->i0 = i0; g->NXglobal = nx_global;
    g->dx = dx; g->dy = dy; g->zScale = zscale;
    g->Lmax = next_pow2(lmax_init); g->expandStep = 16;
    if (og_alloc(g)) { free(g); return NULL; }
    return g;
}

og_grid *og_create(double lx, double ly, double dx, double dy, int lmax_init, double zscale) {
    /* setDomain, cg1:49-60 */
    double nx = js_round(lx / dx), ny = js_round(ly / dy);
    int NX = nx < 1 ? 1 : (int)nx, NY = ny < 1 ? 1 : (int)ny;
    return og_create_slab(NX, NY, 0, NX, dx, dy, lmax_init, zscale);
}

void og_destroy(og_grid *g) {
    if (!g) return;
    og_slider_reset(g);
    for (int i = 0; i < g->nkc; i++) {
        free(g->kc[i]->row_off); free(g->kc[i]->dv); free(g->kc[i]->add); free(g->kc[i]);
    }
    free(g->mat); free(g->zpair); free(g->len); free(g);
}

int og_reset(og_grid *g, int lmax_init) {    /* createNewGrid, cg1:40-43 */
    g->Lmax = next_pow2(lmax_init > 0 ? lmax_init : g->Lmax);
    return og_alloc(g);
}

int og_nx(const og_grid *g) { return g->NX; }
int og_ny(const og_grid *g) { return g->NY; }
int og_lmax(const og_grid *g) { return g->Lmax; }
uint8_t *og_mat(og_grid *g) { return g->mat; }
uint16_t *og_zpair(og_grid *g) { return g->zpair; }
uint8_t *og_len(og_grid *g) { return g->len; }
void og_stats(const og_grid *g, uint64_t out[CG_NSTAT]) { memcpy(out, g->stats, sizeof g->stats); }

/* loadCache's restore (cg1:276-278) with an explicit stride */
int og_upload(og_grid *g, const uint8_t *mat, const uint16_t *zpair, const uint8_t *len, int lmax) {
    g->Lmax = lmax;
    if (og_alloc(g)) return 1;
    size_t N = (size_t)g->NX * g->NY, S = N * lmax;
    memcpy(g->mat, mat, S); memcpy(g->zpair, zpair, 4 * S); memcpy(g->len, len, N);
    return 0;
}

/* _expandBuffers, cg1:129-168 */
static void og_expand(og_grid *g) {
    int oldL = g->Lmax, newL = oldL + g->expandStep;
    size_t N = (size_t)g->NX * g->NY;
    uint8_t *nm = calloc(N * newL, 1);
    uint16_t *nz = calloc(N * newL * 2, 2);
    for (size_t c = 0; c < N; c++) {
        int l = g->len[c];
        memcpy(nm + c * newL, g->mat + c * oldL, (size_t)l);
        memcpy(nz + c * newL * 2, g->zpair + c * oldL * 2, (size_t)l * 4);
    }
    free(g->mat); free(g->zpair);
    g->mat = nm; g->zpair = nz; g->Lmax = newL;
}

/* ---- addLayer / removeTopLayer, cg1:196-218 ---------------------------------------------- */
int og_add_layer(og_grid *g, int i, int j, int matId, double z0, double z1) {
    size_t cidx = (size_t)i * g->NY + j;
    int l = g->len[cidx];
    if (l >= g->Lmax) og_expand(g);
    size_t base = cidx * g->Lmax + l;
    g->mat[base] = (uint8_t)matId;
    g->zpair[base * 2] = to_u16(quantZ(g, z0));
    g->zpair[base * 2 + 1] = to_u16(quantZ(g, z1));
    g->len[cidx] = (uint8_t)(l + 1);
    return 0;
}

int og_remove_top_layer(og_grid *g, int i, int j) {          /* the slot keeps its bytes (cg1:213-218) */
    size_t cidx = (size_t)i * g->NY + j;
    int l = g->len[cidx];
    if (l == 0) return 0;
    g->len[cidx] = (uint8_t)(l - 1);
    return 0;
}

/* ---- _depoAtTop_, cg1:397-426 --------------------------------------------------------- */
static void depo_at_top(og_grid *g, size_t cidx, int matId, double dz) {
    if (dz <= 0) return;
    int l = g->len[cidx];
    double topZ = 0; int topMat = 0;
    if (l > 0) {
        size_t baseTop = cidx * g->Lmax + (l - 1);
        topZ = deqZ(g, g->zpair[baseTop * 2 + 1]);
        topMat = g->mat[baseTop];
    }
    double newTop = topZ + dz;
    if (l > 0 && topMat == matId) {
        size_t baseTop = cidx * g->Lmax + (l - 1);
        g->zpair[baseTop * 2 + 1] = to_u16(quantZ(g, newTop));
        g->stats[2]++;
    } else {
        if (l >= g->Lmax) og_expand(g);
        size_t base = cidx * g->Lmax + l;
        g->mat[base] = (uint8_t)matId;
        g->zpair[base * 2] = to_u16(quantZ(g, topZ));
        g->zpair[base * 2 + 1] = to_u16(quantZ(g, newTop));
        g->len[cidx] = (uint8_t)(l + 1);
        g->stats[3]++;
    }
}

/* ---- _applyEtchAt, cg1:553-589 -------------------------------------------------------- */
static void apply_etch_at(og_grid *g, size_t cidx, int matId, double dz, int isAll) {
    if (dz <= 0) return;
    double remaining = dz;
    int touched = 0;
    while (remaining > OG_EPS && g->len[cidx] > 0) {
        int topIdx = g->len[cidx] - 1;
        size_t base = cidx * g->Lmax + topIdx;
        int curMat = g->mat[base];
        if (!isAll && curMat != matId) break;
        double z0 = deqZ(g, g->zpair[base * 2]);
        double z1 = deqZ(g, g->zpair[base * 2 + 1]);
        double thick = z1 - z0;
        touched = 1;
        if (remaining < thick - OG_EPS) {
            g->zpair[base * 2 + 1] = to_u16(quantZ(g, z1 - remaining));
            remaining = 0;
        } else {
            remaining -= thick;
            g->len[cidx] = (uint8_t)topIdx;
            g->mat[base] = 0; g->zpair[base * 2] = 0; g->zpair[base * 2 + 1] = 0;
            g->stats[5]++;
        }
    }
    if (touched) g->stats[4]++;
}

/* ---- _getConformalKernelArray, cg1:324-357 -------------------------------------------- */
static int js_to_int32(double v) {           /* (v | 0) */
    if (!isfinite(v)) return 0;
    double m = fmod(trunc(v), 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int)(uint32_t)m;
}

static og_kernel *get_kernel(og_grid *g, double t, double S, double dx, double dy) {
    int key[4] = { js_to_int32(t * 1000), js_to_int32(S * 1000), js_to_int32(dx * 1000), js_to_int32(dy * 1000) };
    for (int i = 0; i < g->nkc; i++)
        if (!memcmp(g->kc[i]->key, key, sizeof key)) return g->kc[i];

    og_kernel *k = calloc(1, sizeof *k);
    memcpy(k->key, key, sizeof key);
    int Rx = (int)ceil(t / dx), Ry = (int)ceil(t / dy);
    double t2 = t * t, Sinv = 1 / fmax(S, 1e-6), dx2 = dx * dx, dy2 = dy * dy;
    size_t cap = (size_t)(2 * Rx + 1) * (2 * Ry + 1);
    k->Rx = Rx; k->Ry = Ry; k->baseAdd = t;
    k->row_off = calloc(2 * Rx + 2, sizeof(int));
    k->dv = malloc(cap * sizeof(float)); k->add = malloc(cap * sizeof(float));
    int n = 0;
    for (int du = -Rx; du <= Rx; du++) {
        k->row_off[du + Rx] = n;
        double du2dx2 = (double)du * du * dx2;
        for (int dv = -Ry; dv <= Ry; dv++) {
            double r2 = du2dx2 + (double)dv * dv * dy2;
            double r_lat = sqrt(r2) * Sinv;
            if (r_lat <= t + 1e-9) {
                double add = sqrt(fmax(0, t2 - r_lat * r_lat));
                k->dv[n] = (float)dv; k->add[n] = (float)add; n++;
            }
        }
    }
    k->row_off[2 * Rx + 1] = n;
    g->kc[g->nkc++] = k;
    if (g->nkc > 32) {                       /* FIFO eviction of the oldest key */
        og_kernel *o = g->kc[0];
        free(o->row_off); free(o->dv); free(o->add); free(o);
        memmove(g->kc, g->kc + 1, (size_t)(g->nkc - 1) * sizeof g->kc[0]);
        g->nkc--;
    }
    return k;
}

/* number of taps / radius of the kernel a step would use (for tests) */
int og_kernel_info(og_grid *g, double t, double S, int *Rx, int *ntaps) {
    og_kernel *k = get_kernel(g, t, S, g->dx, g->dy);
    *Rx = k->Rx; *ntaps = k->row_off[2 * k->Rx + 1];
    return 0;
}
int og_kernel_taps(og_grid *g, double t, double S, int *du, int *dv, float *add) {
    og_kernel *k = get_kernel(g, t, S, g->dx, g->dy);
    int n = 0;
    for (int r = 0; r <= 2 * k->Rx; r++)
        for (int q = k->row_off[r]; q < k->row_off[r + 1]; q++) {
            du[n] = r - k->Rx; dv[n] = (int)k->dv[q]; add[n] = k->add[q]; n++;
        }
    return n;
}

/* ---- _buildPaddedH, cg1:363-389 (callers pass Ry := Rx) -------------------------------
 * Test extension for slab decomposition: when halo rows are given (top heights of the
 * R rows beyond each slab edge, as floats), they replace the clamp-replicated rows. */
typedef struct og_halo {
    const float *lowerH, *upperH;        /* R*NY floats each, row-major, nearest row LAST for lower */
    const uint8_t *lowerE, *upperE;      /* R*NY flags each (etch only) */
} og_halo;

static float *build_padded_H(const float *H, int NX, int NY, int Rx, int Ry, const og_halo *halo) {
    int W = NX + 2 * Rx, Hh = NY + 2 * Ry;
    float *pad = malloc((size_t)W * Hh * sizeof(float));
    for (int i = 0; i < NX; i++)
        memcpy(pad + (size_t)(i + Rx) * Hh + Ry, H + (size_t)i * NY, (size_t)NY * sizeof(float));
    for (int i = Rx; i < Rx + NX; i++) {
        size_t base = (size_t)i * Hh;
        float first = pad[base + Ry];
        for (int j = 0; j < Ry; j++) pad[base + j] = first;
        float last = pad[base + Ry + NY - 1];
        for (int j = Ry + NY; j < Ry + NY + Ry; j++) pad[base + j] = last;
    }
    for (int j = 0; j < Hh; j++) {
        float L = pad[(size_t)Rx * Hh + j], R = pad[(size_t)(Rx + NX - 1) * Hh + j];
        for (int i = 0; i < Rx; i++) pad[(size_t)i * Hh + j] = L;
        for (int i = Rx + NX; i < Rx + NX + Rx; i++) pad[(size_t)i * Hh + j] = R;
    }
    if (halo) {
        for (int side = 0; side < 2; side++) {
            const float *src = side ? halo->upperH : halo->lowerH;
            if (!src) continue;
            for (int r = 0; r < Rx; r++) {
                int prow = side ? (Rx + NX + r) : r;
                size_t base = (size_t)prow * Hh;
                memcpy(pad + base + Ry, src + (size_t)r * NY, (size_t)NY * sizeof(float));
                float first = pad[base + Ry], last = pad[base + Ry + NY - 1];
                for (int j = 0; j < Ry; j++) pad[base + j] = first;
                for (int j = Ry + NY; j < Hh; j++) pad[base + j] = last;
            }
        }
    }
    return pad;
}

/* top-height map H (cg1:458-467): float32 of the dequantised top z1 */
static void top_height_map(const og_grid *g, float *H) {
    size_t N = (size_t)g->NX * g->NY;
    for (size_t idx = 0; idx < N; idx++) {
        int l = g->len[idx];
        if (l == 0) { H[idx] = 0; continue; }
        size_t baseTop = idx * g->Lmax + (l - 1);
        H[idx] = (float)deqZ(g, g->zpair[baseTop * 2 + 1]);
    }
}

void og_top_heights(const og_grid *g, float *H) { top_height_map(g, H); }

static void og_identify_cavity_impl(og_grid *g, double Hmax_override, int use_override);

/* ---- deposit_general, cg1:435-544 ------------------------------------------------------
 * mask: NULL = "maskFn is null"; else one byte per column, non-zero = maskFn(x,y) truthy. */
static int deposit_general_impl(og_grid *g, const uint8_t *mask, int matId, double thickness,
                                double conformality, const og_halo *halo, int run_cavity) {
    memset(g->stats, 0, sizeof g->stats);
    double t = fmax(0, thickness);
    if (!(t > 0)) return 0;
    int NX = g->NX, NY = g->NY; double dx = g->dx, dy = g->dy;
    double S = fmax(0, fmin(1, conformality));
    double latRange = t * S;
    size_t N = (size_t)NX * NY;
    g->stats[0] = N;

    if (latRange < fmin(dx, dy)) {                       /* cg1:445-454 */
        for (int i = 0; i < NX; i++)
            for (int j = 0; j < NY; j++) {
                size_t c = (size_t)i * NY + j;
                if (mask && !mask[c]) continue;
                g->stats[1]++;
                depo_at_top(g, c, matId, t);
            }
        return 0;
    }

    float *H = malloc(N * sizeof(float));
    top_height_map(g, H);
    og_kernel *K = get_kernel(g, t, S, dx, dy);
    int Rx = K->Rx; double baseAdd = K->baseAdd;
    float *HPad = build_padded_H(H, NX, NY, Rx, Rx, halo);
    int PH = NY + 2 * Rx;

    int iMin = 0, iMax = NX - 1, jMin = 0, jMax = NY - 1;
    if (mask) {                                          /* cg1:475-490 */
        int found = 0, a = NX, b = -1, c2 = NY, d = -1;
        for (int i = 0; i < NX; i++) for (int j = 0; j < NY; j++)
            if (mask[(size_t)i * NY + j]) {
                found = 1;
                if (i < a) a = i; if (i > b) b = i;
                if (j < c2) c2 = j; if (j > d) d = j;
            }
        if (!found) { free(H); free(HPad); return 0; }
        iMin = a - Rx < 0 ? 0 : a - Rx; iMax = b + Rx > NX - 1 ? NX - 1 : b + Rx;
        jMin = c2 - Rx < 0 ? 0 : c2 - Rx; jMax = d + Rx > NY - 1 ? NY - 1 : d + Rx;
    }

    float *Hp = malloc(N * sizeof(float));               /* cg1:493-504 */
    for (size_t idx = 0; idx < N; idx++)
        Hp[idx] = (!mask || mask[idx]) ? (float)((double)H[idx] + baseAdd) : H[idx];

    for (int i = iMin; i <= iMax; i++) {                 /* cg1:506-528 */
        size_t ii = (size_t)i * NY; int ip = i + Rx;
        for (int j = jMin; j <= jMax; j++) {
            if (mask && !mask[ii + j]) continue;
            g->stats[1]++;
            double m = Hp[ii + j];
            int jp = j + Rx;
            for (int du = -Rx; du <= Rx; du++) {
                int q0 = K->row_off[du + Rx], q1 = K->row_off[du + Rx + 1];
                if (q0 == q1) continue;
                size_t center = (size_t)(ip + du) * PH + jp;
                for (int q = q0; q < q1; q++) {
                    int dv = (int)K->dv[q];
                    double cand = (double)HPad[(ptrdiff_t)center + dv] + (double)K->add[q];
                    if (cand > m) m = cand;
                }
            }
            Hp[ii + j] = (float)m;
        }
    }

    for (int i = iMin; i <= iMax; i++) {                 /* cg1:531-540 */
        size_t ii = (size_t)i * NY;
        for (int j = jMin; j <= jMax; j++) {
            double dz = (double)Hp[ii + j] - (double)H[ii + j];
            if (dz <= OG_EPS) continue;
            depo_at_top(g, ii + j, matId, dz);
        }
    }
    free(H); free(HPad); free(Hp);
    if (run_cavity) og_identify_cavity_impl(g, 0, 0);    /* cg1:543 */
    return 0;
}

int og_deposit_general(og_grid *g, const uint8_t *mask, int matId, double t, double S) {
    return deposit_general_impl(g, mask, matId, t, S, NULL, 1);
}

/* ---- etch_general, cg1:598-740 --------------------------------------------------------- */
static int etch_general_impl(og_grid *g, const uint8_t *mask, int matId, double thickness,
                             double conformality, const og_halo *halo) {
    memset(g->stats, 0, sizeof g->stats);
    double t = fmax(0, thickness);
    if (!(t > 0)) return 0;
    int NX = g->NX, NY = g->NY; double dx = g->dx, dy = g->dy;
    double S = fmax(0, fmin(1, conformality));
    double latRange = t * S;
    int isAll = (matId == 255);
    size_t N = (size_t)NX * NY;
    g->stats[0] = N;

    if (latRange < fmin(dx, dy)) {                       /* cg1:609-618 */
        for (int i = 0; i < NX; i++)
            for (int j = 0; j < NY; j++) {
                size_t c = (size_t)i * NY + j;
                if (mask && !mask[c]) continue;
                g->stats[1]++;
                apply_etch_at(g, c, matId, t, isAll);
            }
        return 0;
    }

    float *H = malloc(N * sizeof(float)), *F = malloc(N * sizeof(float));
    uint8_t *E = malloc(N);
    for (size_t cidx = 0; cidx < N; cidx++) {            /* cg1:625-662 */
        int L = g->len[cidx];
        double topZ = 0; int etchTopOK = 0;
        if (L > 0) {
            size_t topBase = cidx * g->Lmax + (L - 1);
            int topMat = g->mat[topBase];
            topZ = deqZ(g, g->zpair[topBase * 2 + 1]);
            etchTopOK = isAll || (topMat == matId);
        }
        int inMask = !mask || mask[cidx];
        H[cidx] = (float)topZ;
        E[cidx] = (inMask && etchTopOK) ? 1 : 0;
        if (inMask) g->stats[1]++;
        if (L == 0 || isAll || !etchTopOK) F[cidx] = 0;
        else {
            double floorZ = deqZ(g, g->zpair[(cidx * g->Lmax + (L - 1)) * 2]);
            g->stats[6]++;
            for (int s = L - 2; s >= 0; s--) {
                size_t baseS = cidx * g->Lmax + s;
                if (g->mat[baseS] != matId) break;
                floorZ = deqZ(g, g->zpair[baseS * 2]);
                g->stats[6]++;
            }
            F[cidx] = (float)fmax(0, floorZ);
        }
    }

    og_kernel *K = get_kernel(g, t, S, dx, dy);
    int Rx = K->Rx;
    float *HPad = build_padded_H(H, NX, NY, Rx, Rx, halo);
    int PH = NY + 2 * Rx;
    size_t EPadLen = (size_t)(NX + 2 * Rx) * (NY + 2 * Rx);
    uint8_t *EPad = calloc(EPadLen, 1);                  /* cg1:669-676 */
    for (int i = 0; i < NX; i++)
        memcpy(EPad + (size_t)(i + Rx) * PH + Rx, E + (size_t)i * NY, (size_t)NY);
    if (halo) {
        for (int side = 0; side < 2; side++) {
            const uint8_t *src = side ? halo->upperE : halo->lowerE;
            if (!src) continue;
            for (int r = 0; r < Rx; r++) {
                int prow = side ? (Rx + NX + r) : r;
                memcpy(EPad + (size_t)prow * PH + Rx, src + (size_t)r * NY, (size_t)NY);
            }
        }
    }

    int found = 0;                                       /* cg1:679-697; the box only matters when empty */
    for (size_t idx = 0; idx < N; idx++) if (E[idx] == 1) { found = 1; break; }
    if (!found) { free(H); free(F); free(E); free(HPad); free(EPad); return 0; }

    float *Hp = malloc(N * sizeof(float));               /* cg1:700-702 */
    for (size_t idx = 0; idx < N; idx++)
        Hp[idx] = (E[idx] == 1) ? (float)((double)H[idx] - t) : H[idx];

    for (int i = 0; i < NX; i++) {                       /* cg1:704-727 */
        size_t ii = (size_t)i * NY; int ip = i + Rx;
        for (int j = 0; j < NY; j++) {
            if (E[ii + j] != 1) continue;
            double m = Hp[ii + j];
            int jp = j + Rx;
            for (int du = -Rx; du <= Rx; du++) {
                int q0 = K->row_off[du + Rx], q1 = K->row_off[du + Rx + 1];
                if (q0 == q1) continue;
                size_t center = (size_t)(ip + du) * PH + jp;
                for (int q = q0; q < q1; q++) {
                    int dv = (int)K->dv[q];
                    if (EPad[(ptrdiff_t)center + dv] != 1) continue;
                    double cand = (double)HPad[(ptrdiff_t)center + dv] - (double)K->add[q];
                    if (cand < m) m = cand;
                }
            }
            double floorZ = F[ii + j];
            Hp[ii + j] = (float)fmax(0, fmax(floorZ, m));
        }
    }

    for (size_t idx = 0; idx < N; idx++) {               /* cg1:730-739 */
        if (E[idx] != 1) continue;
        double dz = (double)H[idx] - (double)Hp[idx];
        if (dz > OG_EPS) apply_etch_at(g, idx, matId, dz, isAll);
    }
    free(H); free(F); free(E); free(HPad); free(EPad); free(Hp);
    return 0;
}

int og_etch_general(og_grid *g, const uint8_t *mask, int matId, double t, double S) {
    return etch_general_impl(g, mask, matId, t, S, NULL);
}

/* Slab variants (test extension): the caller supplies the neighbours' rows; the trailing
 * identify_cavity of the conformal deposit is left to the caller (og_cavity_fill + a strip),
 * because its Hmax is a whole-domain maximum. */
int og_deposit_general_halo(og_grid *g, const uint8_t *mask, int matId, double t, double S,
                            const float *lowerH, const float *upperH) {
    og_halo h = { lowerH, upperH, NULL, NULL };
    return deposit_general_impl(g, mask, matId, t, S, &h, 0);
}
int og_etch_general_halo(og_grid *g, const uint8_t *mask, int matId, double t, double S,
                         const float *lowerH, const float *upperH,
                         const uint8_t *lowerE, const uint8_t *upperE) {
    og_halo h = { lowerH, upperH, lowerE, upperE };
    return etch_general_impl(g, mask, matId, t, S, &h);
}
/* etchable flags E (cg1:637-643) of every column, for the halo of a slab neighbour */
void og_etch_flags(const og_grid *g, const uint8_t *mask, int matId, uint8_t *E) {
    size_t N = (size_t)g->NX * g->NY; int isAll = matId == 255;
    for (size_t c = 0; c < N; c++) {
        int L = g->len[c], ok = 0;
        if (L > 0) ok = isAll || g->mat[c * g->Lmax + (L - 1)] == matId;
        E[c] = ((!mask || mask[c]) && ok) ? 1 : 0;
    }
}

/* ---- maxHeight, cg2:85-104 ------------------------------------------------------------- */
double og_max_height(const og_grid *g) {
    double maxZ = 0; size_t N = (size_t)g->NX * g->NY;
    for (size_t c = 0; c < N; c++) {
        int L = g->len[c];
        if (L == 0) continue;
        double zTop = deqZ(g, g->zpair[(c * g->Lmax + (L - 1)) * 2 + 1]);
        if (zTop > maxZ) maxZ = zTop;
    }
    return maxZ > 0 ? maxZ : 1;
}

/* ---- strip_connected, cg1:867 - cg2:1 -------------------------------------------------- */
static int overlap(double a0, double a1, double b0, double b1) {
    return (a0 < b1 - OG_EPS) && (b0 < a1 - OG_EPS);
}

int og_strip_connected(og_grid *g, int matId) {
    int NX = g->NX, NY = g->NY, Lmax = g->Lmax;
    size_t Ncols = (size_t)NX * NY, numSeg = Ncols * Lmax;
    static const int dirs[4][2] = { {1, 0}, {-1, 0}, {0, 1}, {0, -1} };
    uint8_t *visited = calloc(numSeg ? numSeg : 1, 1), *mark = calloc(numSeg ? numSeg : 1, 1);
    size_t qcap = 1024, qn = 0;
    int (*q)[3] = malloc(qcap * sizeof *q);
#define QPUSH(a, b, c3) do { if (qn == qcap) { qcap *= 2; q = realloc(q, qcap * sizeof *q); } \
        q[qn][0] = (a); q[qn][1] = (b); q[qn][2] = (c3); qn++; } while (0)

    /* seeds, cg1:910-929; typed-array reads past the end yield undefined -> no match */
    for (int i = 0; i < NX; i++) for (int j = 0; j < NY; j++) {
        size_t cidx = (size_t)i * NY + j; int L = g->len[cidx];
        for (int k = 0; k < L; k++) {
            size_t base = cidx * Lmax + k;
            if (base >= numSeg || g->mat[base] != matId) continue;
            double z0 = deqZ(g, g->zpair[base * 2]), z1 = deqZ(g, g->zpair[base * 2 + 1]);
            /* isExposed, cg1:882-905 */
            int exposed = 0;
            if (!(z0 <= OG_EPS)) {
                size_t baseTop = cidx * Lmax + (L - 1);
                double zTop = baseTop < numSeg ? deqZ(g, g->zpair[baseTop * 2 + 1]) : NAN;
                if (fabs(z1 - zTop) < OG_EPS) exposed = 1;
                else for (int d = 0; d < 4; d++) {
                    int ni = i + dirs[d][0], nj = j + dirs[d][1];
                    if (ni < 0 || ni >= NX || nj < 0 || nj >= NY) { exposed = 0; break; }
                    size_t nidx = (size_t)ni * NY + nj; int nL = g->len[nidx];
                    if (nL == 0) { exposed = 1; break; }
                    size_t nTopBase = nidx * Lmax + (nL - 1);
                    double nTop = nTopBase < numSeg ? deqZ(g, g->zpair[nTopBase * 2 + 1]) : NAN;
                    if (nTop < z1 - OG_EPS) { exposed = 1; break; }
                }
            }
            if (!exposed) continue;
            visited[base] = 1; mark[base] = 1;
            QPUSH(i, j, k);
        }
    }
    /* flood fill, cg1:934-965 */
    while (qn) {
        qn--;
        int i = q[qn][0], j = q[qn][1], k = q[qn][2];
        size_t base = ((size_t)i * NY + j) * Lmax + k;
        double z0 = deqZ(g, g->zpair[base * 2]), z1 = deqZ(g, g->zpair[base * 2 + 1]);
        for (int d = 0; d < 4; d++) {
            int ni = i + dirs[d][0], nj = j + dirs[d][1];
            if (ni < 0 || ni >= NX || nj < 0 || nj >= NY) continue;
            size_t nidx = (size_t)ni * NY + nj; int nL = g->len[nidx];
            for (int kk = 0; kk < nL; kk++) {
                size_t nbase = nidx * Lmax + kk;
                if (nbase >= numSeg) continue;
                if (visited[nbase]) continue;
                if (g->mat[nbase] != matId) continue;
                double z0b = deqZ(g, g->zpair[nbase * 2]), z1b = deqZ(g, g->zpair[nbase * 2 + 1]);
                if (!overlap(z0, z1, z0b, z1b)) continue;
                visited[nbase] = 1; mark[nbase] = 1;
                QPUSH(ni, nj, kk);
            }
        }
    }
    /* removal + compaction, cg1:970-999 */
    for (size_t cidx = 0; cidx < Ncols; cidx++) {
        int L = g->len[cidx];
        if (L == 0) continue;
        int writePtr = 0;
        for (int k = 0; k < L; k++) {
            size_t base = cidx * Lmax + k;
            if (base < numSeg && mark[base]) continue;
            if (writePtr != k) {
                size_t dst = cidx * Lmax + writePtr;
                if (dst < numSeg) {
                    g->mat[dst] = base < numSeg ? g->mat[base] : 0;
                    g->zpair[dst * 2] = base < numSeg ? g->zpair[base * 2] : 0;
                    g->zpair[dst * 2 + 1] = base < numSeg ? g->zpair[base * 2 + 1] : 0;
                }
            }
            writePtr++;
        }
        g->len[cidx] = (uint8_t)writePtr;
        for (int k = writePtr; k < L; k++) {
            size_t base = cidx * Lmax + k;
            if (base >= numSeg) continue;
            g->mat[base] = 0; g->zpair[base * 2] = 0; g->zpair[base * 2 + 1] = 0;
        }
    }
#undef QPUSH
    free(visited); free(mark); free(q);
    return 0;
}

/* ---- identify_cavity, cg2:9-83 --------------------------------------------------------- */
static void og_identify_cavity_impl(og_grid *g, double Hmax_override, int use_override) {
    int Lmax = g->Lmax;
    size_t Ncols = (size_t)g->NX * g->NY, numSeg = Ncols * Lmax;
    double Hmax = use_override ? Hmax_override : og_max_height(g);
    uint8_t *newMat = calloc(numSeg ? numSeg : 1, 1);
    uint16_t *newZ = calloc(numSeg ? 2 * numSeg : 1, 2);
    uint8_t *newLen = calloc(Ncols ? Ncols : 1, 1);
    for (size_t cidx = 0; cidx < Ncols; cidx++) {
        int len = g->len[cidx];
        if (len == 0) continue;
        double lastZ = 0; int writePtr = 0;
        for (int k = 0; k < len; k++) {
            size_t base = cidx * Lmax + k;
            double z0 = deqZ(g, g->zpair[base * 2]), z1 = deqZ(g, g->zpair[base * 2 + 1]);
            int mat = g->mat[base];
            if (z0 > lastZ + OG_EPS) {
                size_t wbase = cidx * Lmax + writePtr;
                if (wbase < numSeg) {                    /* stores past the end are dropped */
                    newMat[wbase] = CG_MAT_CAVITY;
                    newZ[wbase * 2] = to_u16(quantZ(g, lastZ));
                    newZ[wbase * 2 + 1] = to_u16(quantZ(g, z0));
                }
                writePtr++;
            }
            size_t wbase2 = cidx * Lmax + writePtr;
            if (wbase2 < numSeg) {
                newMat[wbase2] = (uint8_t)mat;
                newZ[wbase2 * 2] = g->zpair[base * 2];
                newZ[wbase2 * 2 + 1] = g->zpair[base * 2 + 1];
            }
            writePtr++;
            lastZ = z1;
        }
        if (Hmax > lastZ + OG_EPS) {
            size_t wbase3 = cidx * Lmax + writePtr;
            if (wbase3 < numSeg) {
                newMat[wbase3] = CG_MAT_CAVITY;
                newZ[wbase3 * 2] = to_u16(quantZ(g, lastZ));
                newZ[wbase3 * 2 + 1] = to_u16(quantZ(g, Hmax));
            }
            writePtr++;
        }
        newLen[cidx] = (uint8_t)writePtr;
    }
    free(g->mat); free(g->zpair); free(g->len);
    g->mat = newMat; g->zpair = newZ; g->len = newLen;
    if (use_override != 2) og_strip_connected(g, CG_MAT_CAVITY);
}

/* slab extension (tests of the multi-GPU driver): step 1 of identify_cavity alone, with the
 * whole domain's maximum height given */
int og_cavity_fill(og_grid *g, double Hmax) { og_identify_cavity_impl(g, Hmax, 2); return 0; }

int og_identify_cavity(og_grid *g) { og_identify_cavity_impl(g, 0, 0); return 0; }

/* ---- cmp, cg1:747-859 ------------------------------------------------------------------ */
int og_cmp_apply(og_grid *g, double targetZ, int stopMatId);
int og_cmp(og_grid *g, double depth, int stopMatId) {
    if (depth <= 0) return 0;
    int Lmax = g->Lmax; size_t Ncols = (size_t)g->NX * g->NY;
    double zTop = 0, stopperTop = -INFINITY;
    for (size_t cidx = 0; cidx < Ncols; cidx++) {
        int len = g->len[cidx];
        if (len == 0) continue;
        double z1 = deqZ(g, g->zpair[(cidx * Lmax + (len - 1)) * 2 + 1]);
        zTop = fmax(zTop, z1);
        if (stopMatId != 0)
            for (int k = 0; k < len; k++) {
                size_t base = cidx * Lmax + k;
                if (g->mat[base] == stopMatId)
                    stopperTop = fmax(stopperTop, deqZ(g, g->zpair[base * 2 + 1]));
            }
    }
    double targetZ = (stopMatId == 0 || stopperTop == -INFINITY) ? zTop - depth
                                                                  : fmax(zTop - depth, stopperTop);
    og_cmp_apply(g, targetZ, stopMatId);
    og_identify_cavity_impl(g, 0, 0);
    return 0;
}

/* the per-column part of cmp (cg1:786-853) for a given target height (slab extension) */
int og_cmp_apply(og_grid *g, double targetZ, int stopMatId) {
    int Lmax = g->Lmax; size_t Ncols = (size_t)g->NX * g->NY;
    for (size_t cidx = 0; cidx < Ncols; cidx++) {
        int len = g->len[cidx];
        if (len == 0) continue;
        double stopZ = -INFINITY;
        if (stopMatId != 0)
            for (int k = 0; k < len; k++) {
                size_t base = cidx * Lmax + k;
                if (g->mat[base] == stopMatId) { stopZ = deqZ(g, g->zpair[base * 2 + 1]); break; }
            }
        double limitZ = fmax(targetZ, stopZ);
        while (len > 0) {
            size_t base = cidx * Lmax + (len - 1);
            double z0 = deqZ(g, g->zpair[base * 2]), z1 = deqZ(g, g->zpair[base * 2 + 1]);
            if (z1 <= limitZ + OG_EPS) break;
            if (z0 < limitZ - OG_EPS) {
                g->zpair[base * 2 + 1] = to_u16(quantZ(g, limitZ));
                break;
            } else {
                len -= 1;
                g->mat[base] = 0; g->zpair[base * 2] = 0; g->zpair[base * 2 + 1] = 0;
            }
        }
        g->len[cidx] = (uint8_t)len;
        if (len > 0) {
            size_t base = cidx * Lmax + (len - 1);
            if (g->zpair[base * 2 + 1] <= g->zpair[base * 2]) {
                g->len[cidx] = (uint8_t)(len - 1);
                g->mat[base] = 0; g->zpair[base * 2] = 0; g->zpair[base * 2 + 1] = 0;
            }
        }
    }
    return 0;
}

/* ---- mask predicate, rt:54-141 ---------------------------------------------------------
 * Shapes are the VISIBLE objects in original order, reduced to numbers by the caller
 * (cos/sin included, so the caller's libm is the only one involved). */
static int point_in_polygon(double x, double y, const double *pts, int n) {   /* rt:131-141 */
    int inside = 0;
    for (int i = 0, j = n - 1; i < n; j = i++) {
        double xi = pts[2 * i], yi = pts[2 * i + 1], xj = pts[2 * j], yj = pts[2 * j + 1];
        int intersect = ((yi > y) != (yj > y)) &&
                        (x < (xj - xi) * (y - yi) / (yj - yi + 1e-9) + xi);
        if (intersect) inside = !inside;
    }
    return inside;
}

static int point_inside_shape(double x, double y, const cg_shape *s, const double *pts) {  /* rt:73-129 */
    double dx = x - s->left, dy = y - s->top;
    double lx = dx, ly = dy;
    if (s->rotated) {
        lx = dx * s->cosr - dy * s->sinr;
        ly = dx * s->sinr + dy * s->cosr;
    }
    lx /= s->scale_x; ly /= s->scale_y;
    switch (s->type) {
    case CG_SHAPE_RECT:
        return lx >= 0 && lx <= s->w && ly >= 0 && ly <= s->h;
    case CG_SHAPE_CIRCLE: {
        double cx = s->w / 2, cy = s->h / 2;
        return (lx - cx) * (lx - cx) + (ly - cy) * (ly - cy) <= s->a * s->a;
    }
    case CG_SHAPE_ELLIPSE: {
        double ex = lx - s->w / 2, ey = ly - s->h / 2;
        return (ex * ex) / (s->a * s->a) + (ey * ey) / (s->b * s->b) <= 1;
    }
    case CG_SHAPE_POLYGON:
        return point_in_polygon(lx, ly, pts + 2 * (size_t)s->pt_off, s->pt_cnt);
    default:
        return 0;
    }
}

/* raster[c] = 1 where the step applies (_isPointBlocked returns true), rt:54-71 */
void og_mask_raster(const og_grid *g, const cg_shape *shapes, int nshapes, const double *pts,
                    uint8_t *raster) {
    for (int i = 0; i < g->NX; i++)
        for (int j = 0; j < g->NY; j++) {
            double x = ((g->i0 + i) + 0.5) * g->dx, y = (j + 0.5) * g->dy;
            int open = 1;
            for (int s = nshapes - 1; s >= 0; s--)
                if (point_inside_shape(x, y, &shapes[s], pts)) { open = !shapes[s].positive; break; }
            raster[(size_t)i * g->NY + j] = (uint8_t)open;
        }
}

/* ---- checksums ------------------------------------------------------------------------- */
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* same definition as cg_checksum of the engine: with hg = mix(global column index), column c contributes
 *   mix(hg + len + sum over its layers of term(hg, mat | z0q << 8 | z1q << 24)) */
static uint64_t checksum_term(uint64_t hg, uint32_t mat, uint32_t z0, uint32_t z1) {
    uint32_t lo = mat | (z0 << 8) | (z1 << 24), hi = z1 >> 8;
    uint32_t a = (lo ^ (uint32_t)hg) * 0x9E3779B1u;
    uint32_t b = (hi ^ (uint32_t)(hg >> 32) ^ (a >> 16)) * 0x85EBCA77u;
    return ((uint64_t)a << 32) | b;
}
uint64_t og_checksum(const og_grid *g) {
    uint64_t sum = 0; size_t N = (size_t)g->NX * g->NY;
    for (size_t c = 0; c < N; c++) {
        uint64_t hg = splitmix64((uint64_t)g->i0 * g->NY + c);
        int l = g->len[c];
        uint64_t h = hg + (uint64_t)l;
        for (int k = 0; k < l; k++) {
            size_t b = c * g->Lmax + k;
            h += checksum_term(hg, g->mat[b], g->zpair[2 * b], g->zpair[2 * b + 1]);
        }
        sum += splitmix64(h);
    }
    return sum;
}

/* FNV-1a/32 over, per column in order: len, then (mat, z0q, z1q) of each live layer */
uint32_t og_fnv(const og_grid *g) {
    uint32_t h = 2166136261u; size_t N = (size_t)g->NX * g->NY;
#define MIX(v) (h = (h ^ (uint32_t)(v)) * 16777619u)
    for (size_t c = 0; c < N; c++) {
        int l = g->len[c]; MIX(l);
        for (int k = 0; k < l; k++) {
            size_t b = c * g->Lmax + k;
            MIX(g->mat[b]); MIX(g->zpair[2 * b]); MIX(g->zpair[2 * b + 1]);
        }
    }
#undef MIX
    return h;
}

/* slice extraction semantics (script_slice2d.js:15-16): clamp(round((pos-offset)/pitch), 0, n-1) */
int og_index_from_pos(const og_grid *g, int axis, double pos) {
    int n = axis == 0 ? g->NXglobal : g->NY;
    double pitch = axis == 0 ? g->dx : g->dy;
    double off = -(n - 1) * pitch / 2;
    double r = js_round((pos - off) / pitch);
    if (r < 0) r = 0; if (r > n - 1) r = n - 1;
    return (int)r;
}

/* ---- deposit_ALD and etch_wet (cg2:111 - cg3:308) --------------------------------------- */
#include "colgrid_oracle_ald_wet.inc"
