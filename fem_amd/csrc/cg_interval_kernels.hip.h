// Device code of libcolgrid, part 2: the two incremental process models of the reference,
//   etch_wet     script_columngrid2.js:893 - script_columngrid3.js:308   (integer intervals)
//   deposit_ALD  script_columngrid2.js:111-868                           (bitsets of height bins)
//
// Both read the OLD state of neighbouring columns while a column is rewritten, and both can
// make a column longer or shorter anywhere in its stack, so every kernel here reads one set of
// planes and writes another (the host swaps them afterwards).  One thread owns one column.
//
// etch_wet works on the quantised integers, exactly like the reference.  deposit_ALD is stated
// by the reference on 32-bit words of per-column bitsets, and two of its routines do not do
// what their comments say:
//   * markRangeQ (cg2:170-184) masks the last word of a range with (1 << (s1 + 1)) - 1, which
//     JavaScript evaluates to 0 for s1 == 31: a range that ends on a word boundary loses its
//     last word;
//   * emitIntervals (cg2:197-324) shifts the word it scans but adds bit positions to the word's
//     base only, so the second and later runs of ones inside one word come out displaced.
// The results depend on both, so both are reproduced on the same words: each thread keeps its
// bitsets in LDS (word-major, stride = block size), and the run scanner is the same state
// machine.  Heights compare as integers: every height involved is q*zscale for an integer q,
// and the reference's eps (1e-9 nm) is far below one step.
#pragma once
#include "cg_kernels.hip.h"

namespace cgk {

// Monotone device-wide maxima / minima: one same-address atomic per wave serialises (a million
// of them cost 10 ms at 8192^2), so a wave first looks at the current value and only joins in when
// it would change it.
__device__ __forceinline__ void lazy_max(int* p, int v) { if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v); }
__device__ __forceinline__ void lazy_min(int* p, int v) { if (v < __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(p, v); }

// ------------------------------------------------------------------------------------------
// writing a column as a stream of (z0, z1, material) pieces in ascending order
// ------------------------------------------------------------------------------------------
struct ColWriter {
    uint32_t* rec; size_t N, c; int lmax;
    int w = 0; unsigned top = 0;
    bool gap = false, cav = false, bad = false;
    __device__ ColWriter(const GridView& d, size_t c_) : rec(d.rec), N(d.N), c(c_), lmax(d.lmax) {}
    __device__ void put(unsigned q0, unsigned q1, int m) {
        if (q0 > top) { if (w < lmax) rec[(size_t)w * N + c] = (q0 & 0xffffu) | REC_HOLE; w++; gap = true; }
        else if (q0 < top) bad = true;                         // cannot be represented
        if (w < lmax) rec[(size_t)w * N + c] = rec_make(q1, m);
        w++; top = q1;
        if (m == 2) cav = true;
    }
    __device__ void raw(uint32_t r) {                          // verbatim copy of an interval
        if (w < lmax) rec[(size_t)w * N + c] = r;
        w++;
        if (rec_hole(r)) gap = true; else if (rec_mat(r) == 2) cav = true;
    }
    // flags[1] cavity layer seen, [2] gap seen, [3] largest column, [0] unrepresentable column
    __device__ void finish(uint8_t* len, int* __restrict__ flags) {
        const int old = len[c];
        for (int k = min(w, lmax); k < old; k++) rec[(size_t)k * N + c] = 0;
        len[c] = (uint8_t)min(min(w, lmax), 255);
        if (gap) flags[2] = 1;
        if (cav) flags[1] = 1;
        if (bad || w > 255) flags[0] = 1;
        lazy_max(&flags[3], w);
    }
};

__device__ __forceinline__ void copy_column(const GridView& s, const GridView& d, size_t c, int* __restrict__ flags) {
    ColWriter W(d, c);
    const int l = s.len[c];
    for (int k = 0; k < l; k++) W.raw(s.rec[(size_t)k * s.N + c]);
    W.finish(d.len, flags);
}

// ------------------------------------------------------------------------------------------
// etch_wet: one increment (stepOnce, script_columngrid3.js:250-268, without identify_cavity)
// ------------------------------------------------------------------------------------------
// Air runs of a column inside [0, Hq): everything that is not covered by a layer of material
// >= 2 (cg3:12-36).  f(a, b) is called for each maximal run, bottom to top.  The column is given
// as a ColRef, so that it may be a ghost column of the neighbouring slab.
template <class F>
__device__ __forceinline__ void for_each_air_q(const ColRef& n, unsigned Hq, F&& f) {
    unsigned prev = 0; bool open = false; unsigned a = 0;
    for (int k = 0; k < n.len; k++) {
        const uint32_t r = n.rec[(size_t)k * n.stride];
        const unsigned z1 = rec_z1(r);
        if (z1 <= prev) continue;                              // zero-thickness entries are not segments (cg2:983)
        const bool occ = !rec_hole(r) && rec_mat(r) >= 2;
        if (!occ) { if (!open) { open = true; a = prev; } }
        else if (open) { f(a, prev); open = false; }
        prev = z1;
    }
    if (!open) { open = true; a = prev; }
    if (Hq > a) f(a, Hq);
}

// Removal intervals thrown at column (i, j) by the air of its neighbours (cg3:91-155); beyond the
// domain there is no neighbour (a slab edge inside the domain has ghost columns).
template <class F>
__device__ __forceinline__ void for_each_lateral_cut(const GridView& g, const Ghost& gh, int i, int j, unsigned Hq, int d, bool diag, F&& f) {
    const int di4[4] = {1, -1, 0, 0}, dj4[4] = {0, 0, 1, -1};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const ColRef n = column_ref(g, gh, i + di4[q], j + dj4[q]);
        if (!n.ok) continue;
        for_each_air_q(n, Hq, [&](unsigned a, unsigned b) {
            const int q0 = max(0, (int)a - d), q1 = min((int)Hq, (int)b + d);
            if (q1 > q0) f(q0, q1);
        });
    }
    if (diag) {
        const int di8[4] = {1, 1, -1, -1}, dj8[4] = {1, -1, 1, -1};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const ColRef n = column_ref(g, gh, i + di8[q], j + dj8[q]);
            if (!n.ok) continue;
            for_each_air_q(n, Hq, [&](unsigned a, unsigned b) {
                int q0 = (int)a + d, q1 = (int)b - d;
                if (q1 <= q0) return;
                q0 = max(q0, 0); q1 = min(q1, (int)Hq);
                if (q1 > q0) f(q0, q1);
            });
        }
    }
}

__device__ __forceinline__ bool wet_target(int matId, int m) { return matId == 255 ? m >= 3 : m == matId; }   // cg3:46-49

struct WetOut {                                                // merge of touching pieces of one material (cg3:183-193)
    ColWriter W; bool have = false; unsigned p0 = 0, p1 = 0; int pm = 0;
    __device__ WetOut(const GridView& d, size_t c) : W(d, c) {}
    __device__ void flush() { if (have) W.put(p0, p1, pm); have = false; }
    __device__ void put(unsigned s0, unsigned s1, int m) {
        if (have && s0 <= p1 && m == pm) { p1 = max(p1, s1); return; }
        flush(); have = true; p0 = s0; p1 = s1; pm = m;
    }
};

// The removal intervals the neighbours throw at a column are collected once (up to WET_CUTS of them per lane, in LDS,
// as q0 | q1 << 16: both are at most Hq < 65536) and walked from there for every piece of every target layer; a column
// that receives more goes back to its neighbours each time.
constexpr int WET_CUTS = 24;
static __global__ __launch_bounds__(256) void k_wet_step(GridView s, Ghost gh, GridView d, int matId, int dtQ, int dtQDiag, int diag, unsigned Hq,
                                                  int* __restrict__ flags) {
    __shared__ uint32_t s_cut[WET_CUTS][256];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int tid = threadIdx.x;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < s.N; c += stride) {
        const int l = s.len[c];
        const int i = (int)(c / s.ny), j = (int)(c - (size_t)i * s.ny);
        // is there any removal interval for this column?  (cg3:166-167: none -> untouched)
        bool any = false;
        {                                                      // a target segment whose top or bottom faces air (cg3:52-87)
            unsigned prev = 0, below_end = 0; bool seg_below = false; int below_m = 0;
            for (int k = 0; k < l; k++) {
                const uint32_t r = s.rec[(size_t)k * s.N + c];
                const unsigned q0 = prev, q1 = rec_z1(r);
                if (q1 <= q0) continue;
                prev = q1;
                if (rec_hole(r)) continue;
                const int m = rec_mat(r);
                if (seg_below && below_end < q0 && (wet_target(matId, below_m) || wet_target(matId, m))) any = true;
                seg_below = true; below_end = q1; below_m = m;
            }
            if (seg_below && wet_target(matId, below_m)) any = true;
        }
        int ncut = 0;
        for_each_lateral_cut(s, gh, i, j, Hq, dtQDiag, diag != 0, [&](int q0, int q1) {
            if (ncut < WET_CUTS) s_cut[ncut][tid] = (uint32_t)q0 | ((uint32_t)q1 << 16);
            ncut++;
        });
        if (!any && ncut == 0) { copy_column(s, d, c, flags); continue; }

        WetOut out(d, c);
        unsigned prev = 0;
        bool seg_below = false; unsigned below_end = 0;        // the previous segment, if any
        for (int k = 0; k < l; k++) {
            const uint32_t r = s.rec[(size_t)k * s.N + c];
            const unsigned q0 = prev, q1 = rec_z1(r);
            if (q1 <= q0) continue;
            prev = q1;
            if (rec_hole(r)) continue;
            const int m = rec_mat(r);
            const bool gap_below = seg_below && below_end < q0;
            seg_below = true; below_end = q1;
            if (!wet_target(matId, m)) { out.put(q0, q1, m); continue; }
            // the next segment above, if any: a gap in between (or nothing above) exposes the top
            bool top_exposed = true;
            {
                unsigned p = q1;
                for (int kk = k + 1; kk < l; kk++) {
                    const uint32_t rr = s.rec[(size_t)kk * s.N + c];
                    const unsigned z = rec_z1(rr);
                    if (z <= p) continue;
                    if (!rec_hole(rr)) { top_exposed = p > q1; break; }
                    p = z;
                }
            }
            const int vt0 = top_exposed ? max((int)q0, (int)q1 - dtQ) : 0, vt1 = top_exposed ? (int)q1 : 0;      // cg3:58-78
            const int vb0 = gap_below ? (int)q0 : 0, vb1 = gap_below ? min((int)q1, (int)q0 + dtQ) : 0;          // cg3:79-84
            int cur = (int)q0;
            while (cur < (int)q1) {                            // [q0, q1) minus the union of the removal intervals
                int cover = cur, nxt = (int)q1;
                auto see = [&](int a, int b) {
                    if (b <= a) return;
                    if (a <= cur) { if (b > cover) cover = b; }
                    else if (a < nxt) nxt = a;
                };
                see(vt0, vt1); see(vb0, vb1);
                if (ncut <= WET_CUTS) { for (int q = 0; q < ncut; q++) { const uint32_t v = s_cut[q][tid]; see((int)(v & 0xffffu), (int)(v >> 16)); } }
                else for_each_lateral_cut(s, gh, i, j, Hq, dtQDiag, diag != 0, see);
                if (cover > cur) { cur = cover; continue; }
                out.put((unsigned)cur, (unsigned)nxt, m);
                cur = nxt;
            }
        }
        out.flush();
        out.W.finish(d.len, flags);
    }
}

// ------------------------------------------------------------------------------------------
// deposit_ALD
// ------------------------------------------------------------------------------------------
// One column's bitset in LDS: word w of the thread's column at base[w * stride].
struct BitCol {
    uint32_t* base; int stride; int wpc;                       // LDS: word offsets fit 32 bits
    __device__ uint32_t get(int w) const { return base[w * stride]; }
    __device__ void set(int w, uint32_t v) const { base[w * stride] = v; }
    __device__ void orw(int w, uint32_t v) const { if ((unsigned)w < (unsigned)wpc) base[w * stride] |= v; }
    __device__ void andn(int w, uint32_t v) const { if ((unsigned)w < (unsigned)wpc) base[w * stride] &= ~v; }
    __device__ void clear() const { for (int w = 0; w < wpc; w++) base[w * stride] = 0; }
};

// The same column written by OTHER lanes of the wave (they OR ranges in for its owner): LDS atomics.
struct BitColShared {
    uint32_t* base; int stride; int wpc;
    __device__ void orw(int w, uint32_t v) const { if ((unsigned)w < (unsigned)wpc && v) atomicOr(&base[w * stride], v); }
    __device__ void set(int w, uint32_t v) const { atomicOr(&base[w * stride], v); }          // (only ever all ones)
};

__device__ __forceinline__ uint32_t js_shl1(int n) { return 1u << (n & 31); }          // 1 << n in JavaScript

// markRangeQ (cg2:170-184).  Words outside the column are dropped.  CLEAR: the bits markRangeQ would
// set are cleared instead, i.e. b &= ~(what marking an empty set would give).
template <bool CLEAR = false, class B = BitCol>
__device__ __forceinline__ void mark_range_q(const B& b, int q0, int q1) {
    if (q1 <= q0) return;
    const int w0 = (int)((unsigned)q0 >> 5), w1 = (int)((unsigned)(q1 - 1) >> 5);
    const int s0 = q0 & 31, s1 = (q1 - 1) & 31;
    auto put = [&](int w, uint32_t m) { if constexpr (CLEAR) b.andn(w, m); else b.orw(w, m); };
    if (w0 == w1) put(w0, (~(js_shl1(s0) - 1u)) & (js_shl1(s1 + 1) - 1u));
    else {
        put(w0, ~(js_shl1(s0) - 1u));
        for (int w = w0 + 1; w < w1; w++) if (w < b.wpc) b.set(w, CLEAR ? 0u : 0xFFFFFFFFu);
        put(w1, js_shl1(s1 + 1) - 1u);
    }
}

// Exactly the bits of [q0, q1), q1 > q0 (what markRangeQ does for a range that does not end on a word boundary).
template <class B>
__device__ __forceinline__ void mark_range_exact(const B& b, int q0, int q1) {
    const int w0 = (int)((unsigned)q0 >> 5), w1 = (int)((unsigned)(q1 - 1) >> 5);
    const int e1 = ((q1 - 1) & 31) + 1;                        // 1 .. 32 bits of the last word
    const uint32_t lo = ~0u << (q0 & 31), hi = e1 == 32 ? ~0u : ((1u << e1) - 1u);
    if (w0 == w1) { b.orw(w0, lo & hi); return; }
    b.orw(w0, lo);
    for (int w = w0 + 1; w < w1; w++) if (w < b.wpc) b.set(w, 0xFFFFFFFFu);
    b.orw(w1, hi);
}

// qZ of deposit_ALD (cg2:144-145) applied to a de-quantised integer height
__device__ __forceinline__ int ald_requant(const GridView& g, unsigned q) { return (int)js_round(((double)q * g.zs) * g.inv_zs); }

// markRange (cg2:185-190) for the integer interval [q0, q1) of a stored layer
template <bool CLEAR = false>
__device__ __forceinline__ void mark_layer(const GridView& g, const BitCol& b, unsigned z0q, unsigned z1q) {
    int q0 = ald_requant(g, z0q); if (q0 < 0) q0 = 0;
    int q1 = ald_requant(g, z1q); if (q1 <= q0) q1 = q0 + 1;
    mark_range_q<CLEAR>(b, q0, q1);
}

__device__ __forceinline__ int ctz32(uint32_t x) { return x ? __builtin_ctz(x) : 32; }

// emitIntervals (cg2:197-324): f(s, e) for each run the reference reports, in its order.
template <class F>
__device__ __forceinline__ void emit_intervals(const BitCol& b, int Zmax, F&& f) {
    bool inRun = false; int runStartQ = 0;
    auto EMIT = [&](int sq, int eq) { const int s_ = min(sq, Zmax), e_ = min(eq, Zmax); if (e_ > s_) f(s_, e_); };
    int qBase = 0;
    for (int w = 0; w < b.wpc; w++, qBase += 32) {
        uint32_t word = b.get(w);
        if (!inRun) {
            if (word == 0) continue;
            while (word != 0) {
                const int tz = ctz32(word);
                const uint32_t v = word >> tz;
                const int ones = ctz32(~v);
                if (ones < 32 - tz) { EMIT(qBase + tz, qBase + tz + ones); word = v >> ones; }
                else { runStartQ = qBase + tz; inRun = true; break; }
            }
        } else {
            if (word == 0xFFFFFFFFu) continue;
            if (word == 0) { EMIT(runStartQ, qBase); inRun = false; continue; }
            const int prefixOnes = ctz32(~word);
            EMIT(runStartQ, qBase + prefixOnes);               // prefixOnes == 0: the run ended on the word boundary
            inRun = false;
            uint32_t v = word >> prefixOnes;
            while (v != 0) {
                const int tz = ctz32(v);
                v >>= tz;
                const int sQ = qBase + prefixOnes + tz;
                const int ones = ctz32(~v);
                if (ones < 32) { EMIT(sQ, sQ + ones); v >>= ones; }
                else { runStartQ = sQ; inRun = true; break; }
            }
        }
    }
    if (inRun) EMIT(runStartQ, Zmax);
}

// occupied bins of a column: layers of material >= 2 with positive thickness (cg2:338-364).
// CLEAR: b &= ~occupied, without a bitset of its own for the occupied bins.
template <bool CLEAR = false>
__device__ __forceinline__ void mark_occupied(const GridView& g, size_t c, const BitCol& b) {
    const int l = g.len[c];
    unsigned prev = 0;
    for (int k = 0; k < l; k++) {
        const uint32_t r = g.rec[(size_t)k * g.N + c];
        const unsigned z1 = rec_z1(r);
        if (!rec_hole(r) && rec_mat(r) >= 2 && z1 > prev) mark_layer<CLEAR>(g, b, prev, z1);
        prev = z1;
    }
}

struct AldLayout { int Zmax, wpc; uint32_t lastMask; };        // makeBitLayout (cg2:161-167), evaluated on the host

// --- air intervals of every column (buildAirIntervalsAll, cg2:367-385) ---------------------
// FILL = false: counts only (air_n); FILL = true: writes the intervals (s, e) too.  Column c goes to
// entry off + c of arrays whose planes are `stride` entries apart: a slab keeps the air of its
// neighbours' boundary rows (ghost columns, viewed as a one-row grid) in the same arrays.
template <bool FILL>
__global__ void k_ald_air(GridView g, AldLayout L, uint16_t* __restrict__ air_n, uint2* __restrict__ air, int planes,
                          size_t stride_out, size_t off, int* __restrict__ flags) {
    extern __shared__ uint32_t lds[];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const BitCol b{lds + threadIdx.x, (int)blockDim.x, L.wpc};
    int nmax = 0;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        b.clear();
        mark_occupied(g, c, b);
        for (int w = 0; w < L.wpc - 1; w++) b.set(w, ~b.get(w));
        b.set(L.wpc - 1, (~b.get(L.wpc - 1)) & L.lastMask);
        int n = 0;
        emit_intervals(b, L.Zmax, [&](int s, int e) {
            if (FILL && n < planes) air[(size_t)n * stride_out + off + c] = make_uint2((unsigned)s, (unsigned)e);
            n++;
        });
        if (!FILL) { air_n[off + c] = (uint16_t)min(n, 65535); nmax = max(nmax, n); }
    }
    if (!FILL) {
        for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_down(nmax, o, 64));
        if ((threadIdx.x & 63) == 0 && nmax > 0) lazy_max(&flags[3], nmax);
    }
}

// --- seeds (collectALDSeeds, cg2:405-463) ---------------------------------------------------
__device__ __forceinline__ bool air_has_z(const uint16_t* air_n, const uint2* air, size_t N, size_t c, unsigned z) {
    const int n = air_n[c];
    for (int a = 0; a < n; a++) { const uint2 iv = air[(size_t)a * N + c]; if (z >= iv.x && z <= iv.y) return true; }
    return false;
}

// FILL = false: seed_n only.  bbox = {imin, imax, jmin, jmax} of the columns that own a seed.
// ArrLayout: where column (i, j) sits in the air / seed arrays, and how many rows below row 0 and
// above row nx - 1 they hold (rows of the neighbouring slabs; none at the ends of the domain).
struct ArrLayout {
    size_t stride, off; int rows_lo, rows_hi;
    __device__ size_t at(const GridView& g, int i, int j) const { return (size_t)((long long)off + (long long)i * g.ny + j); }
};

template <bool FILL>
__global__ __launch_bounds__(256) void k_ald_seeds(GridView g, unsigned H0q, const uint16_t* __restrict__ air_n, const uint2* __restrict__ air, ArrLayout A,
                                                   uint16_t* __restrict__ seed_n, uint2* __restrict__ seed, ArrLayout S, int planes, int* __restrict__ bbox,
                                                   int* __restrict__ flags, unsigned long long* __restrict__ total) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int imin = 0x7fffffff, imax = -1, jmin = 0x7fffffff, jmax = -1, nmax = 0;      // reduced per wave before the atomics
    unsigned long long nsum = 0;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < g.N; c += stride) {
        const int l = g.len[c];
        const int i = (int)(c / g.ny), j = (int)(c - (size_t)i * g.ny);
        int n = 0;
        auto push = [&](unsigned a, unsigned b) {
            if (FILL && n < planes) seed[(size_t)n * S.stride + S.off + c] = make_uint2(a, b);
            n++;
        };
        const size_t ac = A.off + c;
        unsigned prev = 0;
        for (int k = 0; k < l; k++) {
            const uint32_t r = g.rec[(size_t)k * g.N + c];
            const unsigned z0 = prev, z1 = rec_z1(r);
            prev = z1;
            if (rec_hole(r) || rec_mat(r) < 3) continue;
            if (air_has_z(air_n, air, A.stride, ac, z1) || z1 == H0q) push(z1, z1);                    // top face
            if (z0 > 0 && air_has_z(air_n, air, A.stride, ac, z0)) push(z0, z0);                       // bottom face
            const int di[4] = {1, -1, 0, 0}, dj[4] = {0, 0, 1, -1};
#pragma unroll
            for (int d = 0; d < 4; d++) {                                                       // side faces
                const int ni = i + di[d], nj = j + dj[d];
                if (ni < -A.rows_lo || nj < 0 || ni >= g.nx + A.rows_hi || nj >= g.ny) continue;
                if (z1 <= z0) continue;                                                         // intersectWith: v <= u + eps
                const size_t nc = A.at(g, ni, nj);
                const int na = air_n[nc];
                for (int p = 0; p < na; p++) {
                    const uint2 iv = air[(size_t)p * A.stride + nc];
                    if (iv.y <= z0) continue;
                    if (iv.x >= z1) break;
                    push(max(iv.x, z0), min(iv.y, z1));
                }
            }
        }
        if (!FILL) { seed_n[S.off + c] = (uint16_t)min(n, 65535); nmax = max(nmax, n); nsum += (unsigned long long)n; }
        else if (n > 0) { imin = min(imin, i); imax = max(imax, i); jmin = min(jmin, j); jmax = max(jmax, j); }
    }
    for (int o = 32; o > 0; o >>= 1) {
        imin = min(imin, __shfl_down(imin, o, 64)); imax = max(imax, __shfl_down(imax, o, 64));
        jmin = min(jmin, __shfl_down(jmin, o, 64)); jmax = max(jmax, __shfl_down(jmax, o, 64));
        nmax = max(nmax, __shfl_down(nmax, o, 64)); nsum += __shfl_down(nsum, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (!FILL) { if (nmax > 0) lazy_max(&flags[3], nmax); if (nsum) atomicAdd(total, nsum); }
        else if (imax >= 0) { lazy_min(&bbox[0], imin); lazy_max(&bbox[1], imax); lazy_min(&bbox[2], jmin); lazy_max(&bbox[3], jmax); }
    }
}



// --- inserting a bitset's runs into a column (applyCoatFromBits, cg2:585-682) ---------------
struct AldOut {                                                // the adjacency pass, cg2:634-655
    const GridView& g; ColWriter W; bool have = false; unsigned p0 = 0, p1 = 0; int pm = 0;
    __device__ AldOut(const GridView& g_, const GridView& d, size_t c) : g(g_), W(d, c) {}
    __device__ void flush() {
        if (!have) return;
        const int q0 = ald_requant(g, p0); int q1 = ald_requant(g, p1);
        if (q1 <= q0) q1 = q0 + 1;
        W.put((unsigned)q0, (unsigned)q1, pm);
        have = false;
    }
    __device__ void put(unsigned z0, unsigned z1, int m) {
        if (!have) { have = true; p0 = z0; p1 = z1; pm = m; return; }
        if (z0 <= p1) {
            if (pm == m) { if (z1 > p1) p1 = z1; }
            else { const unsigned nz0 = max(z0, p1); if (z1 > nz0) { flush(); have = true; p0 = nz0; p1 = z1; pm = m; } }
        } else { flush(); have = true; p0 = z0; p1 = z1; pm = m; }
    }
};

// Column c of s, with the runs of `bits` inserted as material mat, written to d.
__device__ __forceinline__ void insert_runs(const GridView& s, const GridView& d, size_t c, const BitCol& bits, int Zmax, int mat,
                                            int* __restrict__ flags) {
    bool any = false;
    emit_intervals(bits, Zmax, [&](int, int) { any = true; });
    if (!any) { copy_column(s, d, c, flags); return; }        // cg2:612
    AldOut out(s, d, c);
    const int l = s.len[c];
    int k = 0; unsigned prev = 0;
    // next layer of positive thickness at or after k: (a0, a1, am); k is left on it
    unsigned a0 = 0, a1 = 0; int am = 0; bool haveA = false;
    auto nextA = [&]() {
        haveA = false;
        while (k < l) {
            const uint32_t r = s.rec[(size_t)k * s.N + c];
            const unsigned z1 = rec_z1(r);
            const unsigned z0 = prev;
            prev = z1; k++;
            if (!rec_hole(r) && z1 > z0) { a0 = z0; a1 = z1; am = rec_mat(r); haveA = true; return; }
        }
    };
    nextA();
    emit_intervals(bits, Zmax, [&](int bs, int be) {           // two-pointer merge, cg2:622-631
        while (haveA && (a0 < (unsigned)bs || (a0 == (unsigned)bs && a1 <= (unsigned)be))) { out.put(a0, a1, am); nextA(); }
        out.put((unsigned)bs, (unsigned)be, mat);
    });
    while (haveA) { out.put(a0, a1, am); nextA(); }
    out.flush();
    out.W.finish(d.len, flags);
}

// One tap of the coat kernel: a seed [x, y] (height steps) at offset (du, dv) grows to [x*zs - add, y*zs + add], is clipped
// to [0, Hcap] and quantised (cg2:542-557).  exact: the host has checked that the quantised ends are x - lo_q and y + hi_q
// for EVERY integer x, y (add / zs is further than 1e-6 from a rounding boundary, where the double arithmetic of the
// reference, good to 1e-10 steps here, cannot end on the other side), so the kernel works in integers for this tap.
struct AldTap { int du, dv; double add; int lo_q, hi_q, exact, cls; };      // cls: taps with the same `add` (same distance) share a class, see k_ald_grow

// The quantised range [q0, q1) a seed grows to under a tap, or false (cg2:542-557).  In integers when the tap allows
// it and the lower end stays a step clear of the cap (neither of the reference's two "empty after clipping" exits can
// then be taken: the range is 2*add > 1e-6 long before clipping, and at least most of a step long after it).
__device__ __forceinline__ bool ald_seed_range(const uint2 sd, const AldTap& tp, double zs, double inv_zs, double Hcap, int HcapQ, int& q0, int& q1) {
    if (tp.exact && (int)sd.x - tp.lo_q < HcapQ - 1) {
        q0 = max(0, (int)sd.x - tp.lo_q);
        q1 = min(HcapQ, (int)sd.y + tp.hi_q); if (q1 <= q0) q1 = q0 + 1;
        return true;
    }
    const double add = tp.add;
    const double lo = (double)sd.x * zs - add, hi = (double)sd.y * zs + add;
    if (hi <= lo + CG_EPS) return false;
    const double z0 = lo > 0 ? lo : 0, z1 = hi < Hcap ? hi : Hcap;
    if (z1 <= z0 + CG_EPS) return false;
    double d0 = js_round(z0 * inv_zs); if (d0 < 0) d0 = 0;
    double d1 = js_round(z1 * inv_zs); if (d1 <= d0) d1 = d0 + 1;
    q0 = (int)d0; q1 = (int)d1;
    return true;
}

// --- the seeds of every column grown once per tap class --------------------------------------------------------------
// A coat evaluates, for every column, every tap and every seed of the column the tap points at, the range that seed
// grows to, and ORs it into the column's bitset (computeCoatBitsSingleT, cg2:499-583).  The range depends on the seed and
// on the tap's `add` only, and taps at the same distance share their add: a 29-tap kernel has 7 distinct ones.  So the
// ranges of a column's seeds are worked out once per class here -- with markRangeQ's word-boundary behaviour applied to
// each (a range that ends on a multiple of 32 sets [q0, q1 - 32) or nothing, see the top of this file), then united
// where they touch or overlap, which is exact for the bits they set -- and the coat kernel ORs the united runs: one or
// two per (column, class) where the seeds were four or ten.  Up to ALD_RUNS runs are kept per (column, class); a column
// with more says so (ALD_RUNS_MANY) and the coat kernel goes back to its seeds.
constexpr int ALD_RUNS = 3;
constexpr int ALD_RUNS_MANY = 255;
static __global__ __launch_bounds__(256) void k_ald_grow(const uint16_t* __restrict__ seed_n, const uint2* __restrict__ seed, int seed_planes, size_t stride,
                                                  const AldTap* __restrict__ cls_taps, int ncls, double zs, double inv_zs, double Hcap, int HcapQ,
                                                  uint8_t* __restrict__ runs_n, uint32_t* __restrict__ runs) {
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < stride; e += step) {
        const int n = min((int)seed_n[e], seed_planes);
        for (int c = 0; c < ncls; c++) {
            const AldTap tp = cls_taps[c];
            int ua[ALD_RUNS], ub[ALD_RUNS];
#pragma unroll
            for (int u = 0; u < ALD_RUNS; u++) { ua[u] = 0x7fffffff; ub[u] = -0x7fffffff - 1; }
            bool many = false;
            for (int q = 0; q < n; q++) {
                int q0, q1;
                if (!ald_seed_range(seed[(size_t)q * stride + e], tp, zs, inv_zs, Hcap, HcapQ, q0, q1)) continue;
                if ((q1 & 31) == 0) { q1 -= 32; if (q1 <= q0) continue; }
                bool placed = false;
#pragma unroll
                for (int u = 0; u < ALD_RUNS; u++) {
                    const bool hit = !placed && q0 <= ub[u] && q1 >= ua[u];
                    ua[u] = hit ? min(ua[u], q0) : ua[u]; ub[u] = hit ? max(ub[u], q1) : ub[u];
                    placed |= hit;
                }
#pragma unroll
                for (int u = 0; u < ALD_RUNS; u++) {
                    const bool take = !placed && ub[u] < ua[u];
                    ua[u] = take ? q0 : ua[u]; ub[u] = take ? q1 : ub[u];
                    placed |= take;
                }
                if (!placed) { many = true; break; }
            }
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < ALD_RUNS; u++) {
                if (ub[u] > ua[u]) { runs[((size_t)c * ALD_RUNS + cnt) * stride + e] = (uint32_t)ua[u] | ((uint32_t)ub[u] << 16); cnt++; }
            }
            runs_n[(size_t)c * stride + e] = many ? (uint8_t)ALD_RUNS_MANY : (uint8_t)cnt;
        }
    }
}


// --- coat at thickness t (computeCoatBitsSingleT + applyCoatFromBits, cg2:699-707, 585-682) --
// bbox = the seeds' bounding box grown by the kernel radius and clipped to the domain (cg2:510-524).
static __global__ void k_ald_coat(GridView s, GridView d, AldLayout L, double Hcap, const AldTap* __restrict__ taps, int ntaps,
                           const uint16_t* __restrict__ seed_n, const uint2* __restrict__ seed, int seed_planes, ArrLayout S,
                           int imin, int imax, int jmin, int jmax, int mat, int HcapQ, int* __restrict__ flags) {
    extern __shared__ uint32_t lds[];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const BitCol fill{lds + threadIdx.x, (int)blockDim.x, L.wpc};
    const int lane = threadIdx.x & 63, nl = min(64, (int)blockDim.x);      // lanes of this wave (a block is 2^k threads)
    auto seed_range = [&](const uint2 sd, const AldTap& tp, int& q0, int& q1) -> bool {
        return ald_seed_range(sd, tp, s.zs, s.inv_zs, Hcap, HcapQ, q0, q1);
    };
    // (the loop runs the same number of times in every lane: the lanes of a wave work together further down)
    for (size_t base = (size_t)blockIdx.x * blockDim.x; base < s.N; base += stride) {
        const size_t c = base + threadIdx.x;
        const bool inside = c < s.N;
        const int i = inside ? (int)(c / s.ny) : 0, j = inside ? (int)(c - (size_t)i * s.ny) : 0;
        fill.clear();
        // Neighbouring taps give ranges that overlap, so ranges are united in registers and reach the bitset
        // only when one starts away from the running union.  markRangeQ sets exactly the bits of [q0, q1) unless
        // q1 is a multiple of 32 (see the top of this file): then its mask for the last word is empty, which leaves
        // [q0, q1 - 32) if the range spans words and nothing if it does not.  United here are the bits really set.
        // (One range in 32 ends like that: in a wave of 64 columns nearly every tap has one, so sending them to
        // the bitset one by one made every lane wait through a marking pass per tap.)
        // Three unions are kept at a time (an empty one has ub < ua): next to a step in the surface the taps bring
        // ranges from the upper level and from the lower one in turn, and with one union every such turn sent a range
        // to the bitset -- with 64 columns in a wave some lane did at nearly every tap, and all lanes waited.
        constexpr int NU = 3;
        int ua[NU], ub[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) { ua[u] = 0x7fffffff; ub[u] = -0x7fffffff - 1; }
        auto add_range = [&](int q0, int q1) {
            if ((q1 & 31) == 0) { q1 -= 32; if (q1 <= q0) return; }
            bool placed = false;
#pragma unroll
            for (int u = 0; u < NU; u++) {                     // the first union it touches takes it in
                const bool hit = !placed && q0 <= ub[u] && q1 >= ua[u];
                ua[u] = hit ? min(ua[u], q0) : ua[u]; ub[u] = hit ? max(ub[u], q1) : ub[u];
                placed |= hit;
            }
#pragma unroll
            for (int u = 0; u < NU; u++) {                     // else the first empty place
                const bool take = !placed && ub[u] < ua[u];
                ua[u] = take ? q0 : ua[u]; ub[u] = take ? q1 : ub[u];
                placed |= take;
            }
            if (placed) return;
            mark_range_exact(fill, ua[0], ub[0]);              // else the oldest union goes to the bitset
#pragma unroll
            for (int u = 0; u + 1 < NU; u++) { ua[u] = ua[u + 1]; ub[u] = ub[u + 1]; }
            ua[NU - 1] = q0; ub[NU - 1] = q1;
        };
        const bool in_box = inside && i >= imin && i <= imax && j >= jmin && j <= jmax;
        for (int tc = 0; tc < ntaps; tc += 64) {
            // Most columns hold one seed per neighbour, a few (pattern edges) ten or twenty, and every wave meets some
            // of those.  The first seed of every tap is done in step by the whole wave, four taps' loads in flight
            // together, its ranges united in the lane's registers.  The (column, tap) pairs that hold more seeds are
            // noted in `more` and then dealt out over ALL lanes of the wave, whoever owns the column: a lane ORs the
            // ranges of its pairs into the owner's bitset (LDS atomics).  Left to their owners, these seeds kept a
            // whole wave waiting for its busiest lane: two thirds of the kernel's instructions at 29 taps.
            const int tend = min(ntaps, tc + 64);
            unsigned long long more = 0;
            if (in_box)
                for (int t0 = tc; t0 < tend; t0 += 4) {
                    size_t sc[4]; int ns[4]; uint2 first[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        ns[u] = 0; sc[u] = 0;
                        if (t0 + u >= tend) continue;
                        const int si = i - taps[t0 + u].du, sj = j - taps[t0 + u].dv;
                        if (si < -S.rows_lo || sj < 0 || si >= s.nx + S.rows_hi || sj >= s.ny) continue;
                        sc[u] = S.at(s, si, sj);
                        ns[u] = min((int)seed_n[sc[u]], seed_planes);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) first[u] = ns[u] > 0 ? seed[sc[u]] : make_uint2(0u, 0u);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        int q0, q1;
                        if (ns[u] > 0 && seed_range(first[u], taps[min(t0 + u, tend - 1)], q0, q1)) add_range(q0, q1);
                        if (ns[u] > 1) more |= 1ull << (t0 + u - tc);
                    }
                }
            // ---- the pairs with more seeds, dealt out: pair number `it` belongs to the lane whose running count passes it
            const int cnt = __popcll(more);
            int incl = cnt;
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
            const int total = __shfl(incl, nl - 1, 64), excl = incl - cnt;
            const unsigned more_lo = (unsigned)more, more_hi = (unsigned)(more >> 32);
            for (int it0 = 0; it0 < total; it0 += nl) {
                const int it = it0 + lane;
                int owner = 0;                                  // lanes whose running count is <= it (the counts do not fall)
                for (int step = 32; step > 0; step >>= 1) {
                    const int cand = owner + step;
                    const int v = __shfl(incl, min(cand, nl) - 1, 64);
                    if (cand <= nl && v <= it) owner = cand;
                }
                owner = min(owner, nl - 1);
                const unsigned long long om = (unsigned long long)(unsigned)__shfl((int)more_lo, owner, 64)
                                              | ((unsigned long long)(unsigned)__shfl((int)more_hi, owner, 64) << 32);
                const int oex = __shfl(excl, owner, 64), oi = __shfl(i, owner, 64), oj = __shfl(j, owner, 64);
                if (it < total) {
                    unsigned long long m2 = om;
                    for (int k = it - oex; k > 0; k--) m2 &= m2 - 1;
                    const AldTap tp = taps[tc + __builtin_ctzll(m2)];
                    const size_t scl = S.at(s, oi - tp.du, oj - tp.dv);
                    const int n = min((int)seed_n[scl], seed_planes);
                    const BitColShared ob{lds + ((int)threadIdx.x - lane + owner), (int)blockDim.x, L.wpc};
                    // (the seeds of one column mostly touch one another, and so do their ranges: united like the owner's)
                    int va = 0, vb = 0; bool vhave = false;
                    for (int q = 1; q < n; q++) {
                        int q0, q1;
                        if (!seed_range(seed[(size_t)q * S.stride + scl], tp, q0, q1)) continue;
                        if ((q1 & 31) == 0) { q1 -= 32; if (q1 <= q0) continue; }
                        if (vhave && q0 <= vb && q1 >= va) { va = min(va, q0); vb = max(vb, q1); continue; }
                        if (vhave) mark_range_exact(ob, va, vb);
                        va = q0; vb = q1; vhave = true;
                    }
                    if (vhave) mark_range_exact(ob, va, vb);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (!inside) continue;
#pragma unroll
        for (int u = 0; u < NU; u++) if (ub[u] > ua[u]) mark_range_exact(fill, ua[u], ub[u]);
        mark_occupied<true>(s, c, fill);                       // fill & ~occ, cg2:566-582
        fill.set(L.wpc - 1, fill.get(L.wpc - 1) & L.lastMask);
        insert_runs(s, d, c, fill, L.Zmax, mat, flags);
    }
}

// --- the coat from the grown runs (the usual path; k_ald_coat above when heights pass 16 bits or CG_ALD_SEEDS=1) --------
// One lane per column, all taps in step: the runs of the column a tap points at, for the tap's class, are united in the
// lane's registers with what the other taps brought and reach the bitset only when they touch none of three running
// unions.  A (column, class) that holds more runs than k_ald_grow keeps is evaluated from its seeds, in the lane.
static __global__ void k_ald_coat_runs(GridView s, GridView d, AldLayout L, double Hcap, const AldTap* __restrict__ taps, int ntaps,
                                const uint16_t* __restrict__ seed_n, const uint2* __restrict__ seed, int seed_planes, ArrLayout S,
                                const uint8_t* __restrict__ runs_n, const uint32_t* __restrict__ runs,
                                int imin, int imax, int jmin, int jmax, int mat, int HcapQ, int* __restrict__ flags) {
    extern __shared__ uint32_t lds[];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const BitCol fill{lds + threadIdx.x, (int)blockDim.x, L.wpc};
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < s.N; c += stride) {
        const int i = (int)(c / s.ny), j = (int)(c - (size_t)i * s.ny);
        fill.clear();
        constexpr int NU = 3;
        int ua[NU], ub[NU];
#pragma unroll
        for (int u = 0; u < NU; u++) { ua[u] = 0x7fffffff; ub[u] = -0x7fffffff - 1; }
        auto add_bits = [&](int q0, int q1) {                 // exactly the bits of [q0, q1); q1 < q0: nothing
            bool placed = q1 < q0;
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const bool hit = !placed && q0 <= ub[u] && q1 >= ua[u];
                ua[u] = hit ? min(ua[u], q0) : ua[u]; ub[u] = hit ? max(ub[u], q1) : ub[u];
                placed |= hit;
            }
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const bool take = !placed && ub[u] < ua[u];
                ua[u] = take ? q0 : ua[u]; ub[u] = take ? q1 : ub[u];
                placed |= take;
            }
            if (!__ballot(!placed)) return;
            if (!placed) {
                mark_range_exact(fill, ua[0], ub[0]);
#pragma unroll
                for (int u = 0; u + 1 < NU; u++) { ua[u] = ua[u + 1]; ub[u] = ub[u + 1]; }
                ua[NU - 1] = q0; ub[NU - 1] = q1;
            }
        };
        // (32-bit entry numbers: the host checks that classes x runs x stride fits; a lane outside the seeds' box does the
        // same loads at its own column and drops them, so that every branch below is one the whole wave takes or skips)
        const bool in_box = i >= imin && i <= imax && j >= jmin && j <= jmax;
        const unsigned own = (unsigned)S.at(s, i, j), ustride = (unsigned)S.stride;
        for (int t0 = 0; t0 < ntaps; t0 += 4) {
            unsigned sc[4], at[4]; int nr[4]; uint32_t r0[4], r1[4], r2[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const AldTap& tp = taps[min(t0 + u, ntaps - 1)];
                const int si = i - tp.du, sj = j - tp.dv;
                const bool ok = in_box && t0 + u < ntaps && si >= -S.rows_lo && sj >= 0 && si < s.nx + S.rows_hi && sj < s.ny;
                sc[u] = ok ? own - (unsigned)(tp.du * s.ny + tp.dv) : own;
                nr[u] = runs_n[(unsigned)tp.cls * ustride + sc[u]];
                at[u] = (unsigned)tp.cls * (unsigned)ALD_RUNS * ustride + sc[u];
                if (!ok) nr[u] = 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                r0[u] = runs[at[u]];
                r1[u] = __ballot(nr[u] >= 2 && nr[u] != ALD_RUNS_MANY) ? runs[at[u] + ustride] : 0u;
                r2[u] = __ballot(nr[u] >= 3 && nr[u] != ALD_RUNS_MANY) ? runs[at[u] + 2 * ustride] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool many = nr[u] == ALD_RUNS_MANY;
                if (__ballot(many)) {                         // more runs than are kept: from the seeds (markRangeQ's rule as in k_ald_grow)
                    if (many) {
                        const AldTap tp = taps[t0 + u];
                        const int n = min((int)seed_n[sc[u]], seed_planes);
                        for (int q = 0; q < n; q++) {
                            int q0, q1;
                            if (!ald_seed_range(seed[(size_t)q * S.stride + sc[u]], tp, s.zs, s.inv_zs, Hcap, HcapQ, q0, q1)) continue;
                            if ((q1 & 31) == 0) { q1 -= 32; if (q1 <= q0) continue; }
                            add_bits(q0, q1);
                        }
                    }
                }
                // an absent run is the empty range, which touches no union and takes no place
                const int k = many ? 0 : nr[u];
                add_bits(k >= 1 ? (int)(r0[u] & 0xffffu) : 0x7fffffff, k >= 1 ? (int)(r0[u] >> 16) : -0x7fffffff - 1);
                if (__ballot(k >= 2)) add_bits(k >= 2 ? (int)(r1[u] & 0xffffu) : 0x7fffffff, k >= 2 ? (int)(r1[u] >> 16) : -0x7fffffff - 1);
                if (__ballot(k >= 3)) add_bits(k >= 3 ? (int)(r2[u] & 0xffffu) : 0x7fffffff, k >= 3 ? (int)(r2[u] >> 16) : -0x7fffffff - 1);
            }
        }
#pragma unroll
        for (int u = 0; u < NU; u++) if (ub[u] > ua[u]) mark_range_exact(fill, ua[u], ub[u]);
        mark_occupied<true>(s, c, fill);                       // fill & ~occ, cg2:566-582
        fill.set(L.wpc - 1, fill.get(L.wpc - 1) & L.lastMask);
        insert_runs(s, d, c, fill, L.Zmax, mat, flags);
    }
}

// --- keep only the cavities that appeared under the coat (mergeNewCavityFromTmp, cg2:766-814) --
// Merged cavity intervals of a column, bottom to top (collectMatIntervals, cg2:710-742).
struct CavIter {
    const GridView& g; size_t c; int l, k = 0; unsigned prev = 0;
    bool pend = false; unsigned p0 = 0, p1 = 0;
    __device__ CavIter(const GridView& g_, size_t c_) : g(g_), c(c_), l(g_.len[c_]) {}
    __device__ bool next(unsigned& a0, unsigned& a1) {
        while (k < l) {
            const uint32_t r = g.rec[(size_t)k * g.N + c];
            const unsigned z0 = prev, z1 = rec_z1(r);
            prev = z1; k++;
            if (rec_hole(r) || rec_mat(r) != 2 || z1 <= z0) continue;
            if (!pend) { pend = true; p0 = z0; p1 = z1; continue; }
            if (z0 <= p1) { if (z1 > p1) p1 = z1; continue; }
            a0 = p0; a1 = p1; p0 = z0; p1 = z1;
            return true;
        }
        if (pend) { pend = false; a0 = p0; a1 = p1; return true; }
        return false;
    }
};

// a = the state after coat + identify_cavity, b = the state before; d = b + (cavity(a) \ cavity(b))
static __global__ void k_ald_merge_cavity(GridView a, GridView b, GridView d, AldLayout L, int* __restrict__ flags) {
    extern __shared__ uint32_t lds[];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const BitCol bits{lds + threadIdx.x, (int)blockDim.x, L.wpc};
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < a.N; c += stride) {
        bits.clear();
        auto mark = [&](unsigned z0, unsigned z1) {            // cg2:803-808
            if (z1 <= z0) return;
            int q0 = ald_requant(b, z0); if (q0 < 0) q0 = 0;
            int q1 = ald_requant(b, z1); if (q1 <= q0) q1 = q0 + 1;
            if (q1 > L.Zmax) q1 = L.Zmax;
            mark_range_q(bits, q0, q1);
        };
        CavIter after(a, c);
        unsigned s0, e0;
        while (after.next(s0, e0)) {                           // diffIntervals, cg2:744-764
            CavIter before(b, c);
            unsigned bs, be, cur = s0;
            bool done = false;
            while (!done && before.next(bs, be)) {
                if (be <= s0) continue;
                if (!(bs < e0)) break;
                if (bs > cur) mark(cur, min(bs, e0));
                if (be > cur) cur = be;
                if (cur >= e0) done = true;
            }
            if (cur < e0) mark(cur, e0);
        }
        insert_runs(b, d, c, bits, L.Zmax, 2, flags);
    }
}

}  // namespace cgk
