// Row-streaming lateral step, lean form: the same algorithm and the same arithmetic as k_stencil_stream
// (cg_kernels.hip.h), for the case the host has vouched for -- every float-arithmetic licence of StreamOpts holds,
// the tile reads no neighbour's message, the grid is in split form -- with those facts known at compile time.
// k_stencil_stream spends more vector instructions around its taps than in them (410 per row and wave at R = 5 of
// which 186 are taps; 720 / 370 for the R = 7 etch) and is bound by their issue; what this form does about it:
//   - no run-time licence tests in the row loop (there was a branch in front of every de-quantisation);
//   - addresses are a uniform 64-bit row base plus a 32-bit lane offset (one scalar add per row instead of 64-bit
//     vector arithmetic per access);
//   - deposit: an open column always grows (its own height + t is among the candidates, and the host checks that t
//     exceeds the spacing of float32 heights), so the old height is never de-quantised in the update; the quantiser's
//     safety margin is one fused multiply-add;
//   - etch: the top records (and mask bytes) of the last R + 1 staged rows stay in an LDS ring like the deposit's, so the
//     output row's records are not fetched a second time, and the records BELOW them -- loads whose addresses depend on
//     those records -- are requested one row early, from the ring, instead of at the head of the update;
//   - nothing in the row loop depends on the row number: rows past the tile's last are staged like any other.
#pragma once
#include "cg_kernels.hip.h"


namespace cgk {

// float32(q * zscale) under the deq_f32 licence (see StreamOpts)
__device__ __forceinline__ float lean_deq(const StreamOpts& o, unsigned q) {
    const float qf = (float)q;
    return __builtin_fmaf(qf, o.zs_hi, qf * o.zs_lo);
}

// Math.round(h / zscale) for a float32 height h >= the column's old height, decided in float arithmetic when h * (1/zs)
// is clear of every half-integer by more than the error of that product and of h itself (see the update of
// k_stencil_stream: y*1.4e-7 + q*6.2e-8 + 2e-6 with q <= y*(1 + 1e-7)); returns false when the literal form must decide.
__device__ __forceinline__ bool lean_quant(const StreamOpts& o, float h, unsigned& qn) {
    const float y = h * o.izf, rr = __builtin_rintf(y);
    const float margin = __builtin_fmaf(y, 2.1e-7f, 2e-6f);
    qn = (unsigned)(int)rr;
    return (__builtin_fabsf(y - rr) < 0.5f - margin) && (y < 1.0e6f);
}

struct LeanRow { u32x4 a; uint32_t h; unsigned m, mh; };     // a staged row in flight: 4 tops, one halo column's top, their mask bytes
struct LeanFresh { unsigned m[FRESH_MAX_MASKS], h[FRESH_MAX_MASKS]; };

// FRESH + FOLD: what the run in front of the deposit pushed is stored by this very launch, so a step behind the deposit
// that digs down to those intervals must not look for them in memory: they come from the class table (tab: the LDS copy,
// tops first, then one row of classes per deposit of the run; cls: the classes of the lane's 4 columns, 3 bits each).
struct BelowFromFreshRun {
    const uint32_t* tab; const int* plane; int ndep; unsigned cls;
    __device__ __forceinline__ void operator()(const GridView& g, const size_t c0, const unsigned want, const uint32_t (&r)[4], uint32_t (&b)[4]) const {
        unsigned mem = want;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!((want >> j) & 1)) continue;
            const int p = rec_lnk(r[j]);
            for (int d = 0; d < ndep; d++)
                if (plane[d] == p) { b[j] = tab[(1 + d) * (1 << FRESH_MAX_MASKS) + ((cls >> (3 * j)) & 7u)]; mem &= ~(1u << j); }
        }
        if (mem) load_below4(g, c0, mem, r, b);          // (cannot happen on a grid that was empty before the run; kept for safety)
    }
};      // FRESH: the bytes of the run's masks at those columns instead of tops

// Waves per SIMD the register allocation aims at.  Measured at 16384^2: R = 5 at 4 waves spills two registers inside the row
// loop and is slower than at 3; the etch's update holds more values than the deposit's and spills at 5 waves already at R = 3.
constexpr int lean_waves(int R, bool etch, bool fold) { return R <= 3 ? ((etch || fold) ? 4 : 5) : 3; }      // (the etch's update holds more: at 5 waves it spills)

// FOLD (deposits): the vertical steps that follow this one in the flow (F.op[1 .. F.n-1]; op[0] stands for this step, a
// deposit into plane g.P) are applied to every output row right behind the step's own update, in registers, as
// k_vertical_fused would apply them in a launch of its own: that launch's pass over the tops is saved.
// FRESH (deposits): the grid has just been reset and a run of vertical steps is waiting in front of this step.  The top
// of a column after that run is one of 2^masks records (FreshTab, see k_vertical_fresh): every row this tile reads is
// put together from the run's mask bytes instead of being loaded, and the tile stores what the run's deposits pushed for
// the rows it owns -- the run never gets a launch of its own, and its tops are never written or read.
template <int R, bool ETCH, bool MASKED, int NF = 0, bool FRESH = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(lean_waves(R, ETCH, NF > 1)))) void k_stencil_lean(GridView g, const uint8_t* __restrict__ mask, int matId, double t,
                                                                                           const float* __restrict__ quad, const StreamOpts so, const FusedOps F, const FreshTab T) {
    // NF: how many steps behind the deposit the launch can apply (0: none).  A compile-time number, so that the code and
    // the scalar registers of steps that cannot be there cost nothing: NF = 1 and 2 for the short tails, FUSE_MAX - 1 for any.
    constexpr bool FOLD = NF > 0;
    constexpr int KE = 1 + NF;
    static_assert(NF >= 0 && NF < FUSE_MAX, "a launch holds FUSE_MAX steps, its own deposit included");
    static_assert(!((FOLD || FRESH) && ETCH), "only deposits take vertical steps along");
    constexpr int NC = 1 << FRESH_MAX_MASKS;
    __shared__ uint32_t s_ftab[FRESH ? (1 + FRESH_MAX_DEP) * NC : 1];       // [0..NC) tops per class, then NC entries per deposit of the run
    __shared__ unsigned s_cls[FRESH ? R + 1 : 1][64];                          // the classes of the 4 columns of a staged row (3 bits each)
    using G = StreamGeom<R>;
    constexpr int Rp = G::Rp, PITCH = G::PITCH, W = G::W, TP = STREAM_TAB_PITCH;
    constexpr int RING = R + 1;         // the tops of staged rows r - R .. r: the slot of row r - R is read before row r + 1 takes it
    __shared__ __align__(16) float s_row[2][PITCH];
    __shared__ __align__(16) float s_tab[(R + 1) * TP];
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
    for (int e = lane; e < (R + 1) * TP; e += 64) {
        const int dv = e / TP, du = e - dv * TP;
        s_tab[e] = du <= R ? quad[dv * (R + 1) + du] : -__builtin_inff();
    }
    if (FRESH) {
        for (int e = lane; e < (1 + T.ndep) * NC; e += 64) s_ftab[e] = e < NC ? T.top[e] : T.pushed[e / NC - 1][e % NC];
        __builtin_amdgcn_s_barrier();
    }
    const bool m_right = j0 >= g.ny;                       // lanes beyond the last column repeat the edge group; only their stores are left out
    const unsigned m_off = 4u * (unsigned)(m_right ? g.ny - 4 : j0);          // byte offset of the lane's group inside a row of tops
    const bool has_halo = lane < 2 * Rp;                   // the first 2 Rp lanes also stage one halo column each
    const int hl = min(lane, 2 * Rp - 1);
    const bool h_upper = hl >= Rp;
    const int hj = h_upper ? j_base + STRIP + (hl - Rp) : j_base - Rp + hl;
    const bool h_beyond = hj < 0 || hj >= g.ny;
    const unsigned h_off = 4u * (unsigned)(hj < 0 ? 0 : (hj >= g.ny ? g.ny - 1 : hj));
    const int h_at = h_upper ? STRIP + hl : hl;
    unsigned loc[4] = {0, 0, 0, 0};

    if (ETCH && so.fast_rows) {                            // the rows at the head of the tile that need no neighbours
        const int done = stream_etch_fast_rows<MASKED>(g, mask, so, i_base, rows_out, (int)(m_off >> 2), m_right, matId, isAll, t, loc);
        if (done == rows_out) return;
        i_base += done; rows_out -= done;
    }
    const int rows_in = rows_out + 2 * R;
    const size_t row_bytes = 4 * (size_t)g.ny;
    const int last_row = g.nx - 1;

    // input row `il` (any integer; rows beyond the slab repeat its edge row with every flag clear, cg1:381-387, 671)
    LeanFresh wf;                                          // (FRESH) filled by load_row next to the row it returns, looked at by stage
    const auto load_row = [&](int il) __attribute__((always_inline)) {
        LeanRow w;
        const bool edge = il < 0 || il > last_row;
        const int ic = il < 0 ? 0 : (il > last_row ? last_row : il);
        const char* rowp = (const char*)g.top + (size_t)ic * row_bytes;                        // uniform
        if (!FRESH) {
            w.a = *(const u32x4*)(rowp + m_off);
            w.h = *(const uint32_t*)(rowp + h_off);
        } else {
            w.a = (u32x4)(0); w.h = 0;
#pragma unroll
            for (int q = 0; q < FRESH_MAX_MASKS; q++) {
                wf.m[q] = 0; wf.h[q] = 0;
                if (q < T.nmask) {
                    const uint8_t* mrow = T.mask[q] + (size_t)ic * (size_t)g.ny;               // uniform
                    wf.m[q] = *(const unsigned*)(mrow + (m_off >> 2));
                    wf.h[q] = mrow[h_off >> 2];
                }
            }
        }
        w.m = 0x01010101u; w.mh = 1u;
        if (ETCH) {
            if (MASKED) {
                const uint8_t* mrow = mask + (size_t)ic * (size_t)g.ny;                        // uniform
                w.m = *(const unsigned*)(mrow + (m_off >> 2));
                w.mh = mrow[h_off >> 2];
            }
            if (edge) { w.m = 0u; w.mh = 0u; }
        }
        return w;
    };
    // float32 heights of a loaded row into s_row[slot]; its tops (etch: and mask bytes) into ring slot `rs`
    const auto stage = [&](const LeanRow& w_in, int slot, int rs) __attribute__((always_inline)) {
        LeanRow w = w_in;
        if (FRESH) {
            unsigned cl[4] = {0, 0, 0, 0}, clh = 0;
#pragma unroll
            for (int q = 0; q < FRESH_MAX_MASKS; q++) {
#pragma unroll
                for (int c = 0; c < 4; c++) cl[c] |= ((wf.m[q] >> (8 * c)) & 0xffu) ? (1u << q) : 0u;
                clh |= wf.h[q] ? (1u << q) : 0u;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) w.a[c] = s_ftab[cl[c]];
            w.h = s_ftab[clh];
            s_cls[rs][lane] = cl[0] | (cl[1] << 3) | (cl[2] << 6) | (cl[3] << 9);
        }
        float4 h;
        unsigned q[4];
#pragma unroll
        for (int c = 0; c < 4; c++) q[c] = rec_z1(w.a[c]);
        if (m_right) { q[0] = q[3]; q[1] = q[3]; q[2] = q[3]; }                                  // clamp-replicate in j (cg1:374-380)
        h.x = lean_deq(so, q[0]); h.y = lean_deq(so, q[1]); h.z = lean_deq(so, q[2]); h.w = lean_deq(so, q[3]);
        float hh = lean_deq(so, rec_z1(w.h));
        unsigned e = 0;                                                                          // etch: the columns that are open and etchable (cg1:637-643)
        if (ETCH) {
            const float INF = __builtin_inff();
            const bool e0 = (w.m & 0xffu) && etchable(w.a[0], matId, isAll), e1 = (w.m & 0xff00u) && etchable(w.a[1], matId, isAll);
            const bool e2 = (w.m & 0xff0000u) && etchable(w.a[2], matId, isAll), e3 = (w.m & 0xff000000u) && etchable(w.a[3], matId, isAll);
            e = (e0 ? 1u : 0u) | (e1 ? 2u : 0u) | (e2 ? 4u : 0u) | (e3 ? 8u : 0u);
            if (m_right) e = 0;                                                                  // flags are zero outside the domain (cg1:671)
            if (!(e & 1u)) h.x = INF; if (!(e & 2u)) h.y = INF; if (!(e & 4u)) h.z = INF; if (!(e & 8u)) h.w = INF;
            if (h_beyond || !w.mh || !etchable(w.h, matId, isAll)) hh = INF;
        }
        *(float4*)(s_row[slot] + Rp + 4 * lane) = h;
        if (has_halo) s_row[slot][h_at] = hh;
        *(u32x4*)(s_top[rs] + 4 * lane) = w.a;
        if (ETCH) s_msk[rs][lane] = e;                                                           // what the update of this row will go by
    };

    float A[2 * R + 1][4];
#pragma unroll
    for (int k = 0; k <= 2 * R; k++) { A[k][0] = 0.f; A[k][1] = 0.f; A[k][2] = 0.f; A[k][3] = 0.f; }

    LeanRow in = load_row(i_base - R);
    stage(in, 0, 0);
    in = load_row(i_base - R + 1);
    // Ring slots: staged row s (input row i_base - R + s) lives in slot s mod RING.  Input row r completes the output row
    // whose records are staged row r - R; they are read before row r + 1 is staged, into the slot they leave.
    int ring_w = 1 % RING;                                 // slot the next staged row goes to
    int ring_o = R % RING;                                 // slot of the first output row's records (staged row R)

    // etch: the intervals below the tops of an output row are requested at the END of the iteration before the one that
    // updates it (their addresses depend on the tops, which come from the ring), and looked at a whole row of taps later.
    uint32_t bn[4] = {0, 0, 0, 0};
    const auto request_below = [&](int o, int slot) __attribute__((always_inline)) {
        const u32x4 tn = *(const u32x4*)(s_top[slot] + 4 * lane);
        const int oc = min(o, rows_out - 1);               // (past the tile's last row: that row again, the values are not looked at)
        const uint32_t* base = g.rec + (size_t)(i_base + oc) * (size_t)g.ny;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int p = rec_lnk(tn[c]);
            bn[c] = base[(size_t)(p != LNK_NONE ? p : 0) * g.N + (m_off >> 2) + c];
        }
    };

#define CG_OP(a, b) __int_as_float(ETCH ? min(__float_as_int(a), __float_as_int(b)) : max(__float_as_int(a), __float_as_int(b)))
    // One input row.  FIN: it completes an output row (every row from the 2R-th on), which is then updated.  Nothing in
    // the row loop is conditional on the row number: rows past the tile's last are staged and loaded like any other (they
    // repeat a real row) and never looked at, so that every load has its use on every path.
    auto step = [&](const int r, auto fin_tag) __attribute__((always_inline)) {
        constexpr bool FIN = decltype(fin_tag)::value;
        const float* buf = s_row[r & 1];
        __builtin_amdgcn_wave_barrier();
        float hc[W];
        {
            const float* rowc = buf + 4 * lane;
#pragma unroll
            for (int k = 0; k < W / 4; k++) { const float4 v = *(const float4*)(rowc + 4 * k); hc[4 * k] = v.x; hc[4 * k + 1] = v.y; hc[4 * k + 2] = v.z; hc[4 * k + 3] = v.w; }
        }
        const int o = r - 2 * R;                           // output row (relative to i_base) this input row completes
        const size_t orow = (size_t)(i_base + (FIN ? o : 0)) * (size_t)g.ny;                    // uniform
        unsigned mk = 0x01010101u;                         // deposit: the mask bytes of the output row (the etch keeps them in the ring)
        if (FIN && MASKED && !ETCH) mk = *(const unsigned*)(mask + orow + (m_off >> 2));
        unsigned fopen[FUSE_MAX];                          // FOLD: the mask bytes of the steps taken along, for the output row
        if constexpr (FOLD) {
#pragma unroll
            for (int k = 0; k < FUSE_MAX; k++) {
                fopen[k] = k < F.n ? 0x01010101u : 0u;
                if (FIN && k >= 1 && k < KE && k < F.n && F.masks[k]) fopen[k] = *(const unsigned*)(F.masks[k] + orow + (m_off >> 2));
            }
        }
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
        // ---- the output row's records, from the ring (read before row r + 1 takes their slot)
        u32x4 tv = (u32x4)(0);
        unsigned fcl = 0;                                  // FRESH: the classes of the output row's columns
        if (FIN) {
            tv = *(const u32x4*)(s_top[ring_o] + 4 * lane);
            if (ETCH) mk = s_msk[ring_o][lane];
            if (FRESH) fcl = s_cls[ring_o][lane];
        }
        // ---- stage input row r + 1, start the loads of row r + 2
        // (the loaded row is looked at from here on: the compiler otherwise starts on it right behind its load, and waits there)
        if (!FRESH) asm volatile("" : "+v"(in.a[0]), "+v"(in.a[1]), "+v"(in.a[2]), "+v"(in.a[3]), "+v"(in.h), "+v"(in.m), "+v"(in.mh));
        else asm volatile("" : "+v"(wf.m[0]), "+v"(wf.m[1]), "+v"(wf.m[2]), "+v"(wf.h[0]), "+v"(wf.h[1]), "+v"(wf.h[2]));
        stage(in, (r + 1) & 1, ring_w);
        ring_w = ring_w + 1 >= RING ? 0 : ring_w + 1;
        in = load_row(i_base - R + r + 2);
        if (!FIN) {
            if (ETCH && r == 2 * R - 1) request_below(0, ring_o);       // for the first output row, updated by the next input row
            return;
        }
        ring_o = ring_o + 1 >= RING ? 0 : ring_o + 1;
        // ---- the update of the output row
        const unsigned act0 = ETCH ? mk : (((mk & 0xffu) ? 1u : 0u) | ((mk & 0xff00u) ? 2u : 0u) | ((mk & 0xff0000u) ? 4u : 0u) | ((mk & 0xff000000u) ? 8u : 0u));
        if (!ETCH) {
            // _depoAtTop_ (cg1:397-426) by dz = Hp - H (cg1:534-540) on the open columns: every one of them grows (see the head)
            float m[4] = {A[0][0], A[0][1], A[0][2], A[0][3]};
            if (!so.skip_init) {
#pragma unroll
                for (int c = 0; c < 4; c++) m[c] = CG_OP(m[c], (float)((double)lean_deq(so, rec_z1(tv[c])) + t));      // cg1:499
            }
            unsigned qn[4], slow = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) if (!lean_quant(so, m[c], qn[c])) slow |= 1u << c;
            slow &= act0;
            if (slow) {
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if ((slow >> c) & 1) {
                        const unsigned q = rec_z1(tv[c]);
                        qn[c] = quantz((double)q * g.zs + ((double)m[c] - (double)lean_deq(so, q)), g.zs, g.inv_zs);       // cg1:407-411, 117-119
                    }
            }
            u32x4 nv = tv;
            unsigned push = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t rc = tv[c];
                const bool on = ((act0 >> c) & 1) != 0, some = rc != 0, same = some & (rec_mat(rc) == matId);
                const uint32_t ext = rec_with_z1(rc, qn[c]), app = rec_make_l(qn[c], matId, some ? g.P : LNK_NONE);
                nv[c] = on ? (same ? ext : app) : rc;
                push |= (on & some & !same) ? (1u << c) : 0u;
            }
            uint32_t pushed[FUSE_MAX][4];
            if constexpr (FOLD) {
                uint32_t r4[4] = {nv[0], nv[1], nv[2], nv[3]};
#pragma unroll
                for (int k = 0; k < FUSE_MAX; k++) {
#pragma unroll
                    for (int c = 0; c < 4; c++) pushed[k][c] = (k == 0 && ((push >> c) & 1)) ? tv[c] : 0u;
                }
                if constexpr (FRESH) fused_steps<1, KE>(g, F, orow + (m_off >> 2), fopen, r4, pushed, BelowFromFreshRun{s_ftab, T.plane, T.ndep, fcl});
                else fused_steps<1, KE>(g, F, orow + (m_off >> 2), fopen, r4, pushed);
                nv[0] = r4[0]; nv[1] = r4[1]; nv[2] = r4[2]; nv[3] = r4[3];
            }
            {
                // Exactly two full stores per row and lane, whatever the columns do (no branches, no single-record
                // stores next to a material boundary).  The old tops go into this step's plane when any column of the wave
                // pushes (a slot no chain names may hold anything); a wave without a push, and the lanes that repeat the
                // edge group, store into the sink, which stays in the cache.  Measured at 16384^2, R = 5 where half the
                // columns extend their top layer: 0.78 -> 0.72 ms.
                const bool anyp = __ballot(push != 0) != 0ull;
                char* sink = (char*)so.sink + ((blockIdx.x & 4095u) << 10) + 16 * lane;
                char* pb = (char*)(g.rec + (size_t)g.P * g.N + orow) + m_off;
                st_rec4((uint32_t*)((anyp && !m_right) ? pb : sink), tv);
                st_rec4((uint32_t*)(m_right ? sink : (char*)(g.top_w + orow) + m_off), nv);
            }
            if (!m_right) {
                if constexpr (FOLD) fused_store_planes<1, KE>(g, F, orow + (m_off >> 2), pushed);
                if constexpr (FRESH) {                          // what the deposits of the run in front pushed (as k_vertical_fresh stores it)
                    for (int d = 0; d < T.ndep; d++) {
                        u32x4 pv;
#pragma unroll
                        for (int c = 0; c < 4; c++) pv[c] = s_ftab[(1 + d) * NC + ((fcl >> (3 * c)) & 7u)];
                        uint32_t* at = (uint32_t*)((char*)(g.rec + (size_t)T.plane[d] * g.N + orow) + m_off);
                        if (__ballot((pv[0] | pv[1] | pv[2] | pv[3]) != 0) != 0ull) st_rec4(at, pv);      // (whole waves: zeros where nothing was pushed)
                    }
                }
            }
        } else {
            // _applyEtchAt (cg1:553-589) by dz = H - Hp (cg1:730-739) on the open, etchable columns (act0, worked out when
            // the row was staged).  The one outcome that needs nothing but float arithmetic -- a cut that stays inside the
            // top layer, the layer under it being another material -- is computed for all four columns without a branch;
            // every other outcome (the layer goes, the floor lies further down, a close call) takes stream_etch_decide,
            // the same code k_stencil_stream runs, for those columns only.
            uint32_t rn[4] = {tv[0], tv[1], tv[2], tv[3]};
            unsigned slow = 0, close = 0;
            float m4[4] = {A[0][0], A[0][1], A[0][2], A[0][3]};
            if (!so.skip_init) {
#pragma unroll
                for (int c = 0; c < 4; c++) m4[c] = stream_min_bits(m4[c], (float)((double)lean_deq(so, rec_z1(tv[c])) - t));        // cg1:702
            }
            float own4[4], hp4[4]; unsigned q14[4], bq4[4], qr4[4], qok4 = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t rc = tv[c], b = bn[c];
                const bool on = ((act0 >> c) & 1) != 0;
                const bool has_b = rec_lnk(rc) != LNK_NONE;
                const unsigned q1 = rec_z1(rc), bq = has_b ? rec_z1(b) : 0u;
                const bool walk = has_b & (rec_hole(b) | (!isAll & (rec_mat(b) == matId)));
                const float own = lean_deq(so, q1), bottom = lean_deq(so, bq);
                const float hpv = fmaxf(0.f, fmaxf(isAll ? 0.f : bottom, m4[c]));                 // cg1:725
                const bool cut = own > hpv;                                                      // cg1:731-733 (grow_f32)
                // cg1:574, as in stream_etch_decide: the cut stays inside the top layer when Hp clears its bottom by this margin
                const bool inside = (hpv - bottom) > ((own + bottom) * 1.3e-7f + 1.0e-8f) * 1.01f;
                const float y = hpv * so.izf, rr = __builtin_rintf(y);
                const bool qok = (__builtin_fabsf(y - rr) < 0.5f - (y * 1.4e-7f + (float)q1 * 6.2e-8f + 2e-6f)) & (y < 1.0e6f);
                const bool fast = !walk & cut & inside & qok, same = !walk & !cut;
                rn[c] = (on & fast) ? rec_with_z1(rc, (unsigned)(int)rr) : rc;
                slow |= (on & !fast & !same & (walk | inside)) ? (1u << c) : 0u;                 // the floor lies further down, or the float quantiser would not vouch
                close |= (on & !walk & cut & !inside) ? (1u << c) : 0u;                          // the cut ends at (or below) the layer's bottom
                own4[c] = own; hp4[c] = hpv; q14[c] = q1; bq4[c] = bq; qr4[c] = (unsigned)(int)rr; qok4 |= qok ? (1u << c) : 0u;
            }
            if (close) {
                // The cut ends within rounding of the layer's bottom -- the usual end of a lateral etch next to a step, and of
                // every column when a film is taken off by exactly its thickness: whether the layer stays (as a layer of no
                // thickness) or goes is the reference's double comparison, remaining < thick - 1e-9 (cg1:574), literally.
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t rc = tv[c], b = bn[c];
                    const bool cl = ((close >> c) & 1) != 0;
                    const bool has_b = rec_lnk(rc) != LNK_NONE;
                    const double rem = (double)own4[c] - (double)hp4[c];
                    const double thick = (double)q14[c] * g.zs - (double)bq4[c] * g.zs;
                    const bool partial = rem < thick - CG_EPS;
                    const double rest = rem - thick;
                    const bool goes_on = !partial & has_b & (rest > CG_EPS) & (isAll | (rec_mat(b) == matId));     // cg1:579-587, then on into the next layer
                    const bool q_ok = ((qok4 >> c) & 1) != 0;
                    const uint32_t gone = has_b ? b : 0u;
                    const uint32_t r2 = partial ? rec_with_z1(rc, qr4[c]) : gone;
                    const bool hard = goes_on | (partial & !q_ok);
                    rn[c] = (cl & !hard) ? r2 : rn[c];
                    slow |= (cl & hard) ? (1u << c) : 0u;
                }
            }
            if (slow) {
                uint32_t rs[4];
                const float A0s[4] = {A[0][0], A[0][1], A[0][2], A[0][3]};
                stream_etch_decide(g, so, orow + (m_off >> 2), tv, slow, bn, A0s, matId, isAll, t, rs, loc);
#pragma unroll
                for (int c = 0; c < 4; c++) if ((slow >> c) & 1) rn[c] = rs[c];
            }
            if (!m_right) {
                u32x4 nv; nv[0] = rn[0]; nv[1] = rn[1]; nv[2] = rn[2]; nv[3] = rn[3];
                st_rec4((uint32_t*)((char*)(g.top_w + orow) + m_off), nv);
            }
            request_below(o + 1, ring_o);                  // (ring_o names the next output row's slot by now)
        }
    };
    for (int r = 0; r < 2 * R; r++) step(r, BoolTag<false>{});
    for (int r = 2 * R; r < rows_in; r++) step(r, BoolTag<true>{});
#undef CG_OP
}

}  // namespace cgk
