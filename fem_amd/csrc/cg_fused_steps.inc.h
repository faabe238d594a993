// The body of a run of vertical steps on the 4 columns of a lane, shared by k_vertical_fused and the FOLD form of
// k_stencil_lean (textually, so that each kernel is compiled exactly as if the code stood in it).  Expects in scope:
//   g (GridView), F (FusedOps), c0 (size_t: first of the 4 columns), open[FUSE_MAX] (mask bytes of step k, non-zero byte =
//   open), r[4] (the tops, in and out), pushed[FUSE_MAX][4] (what deposit k pushed into its plane, 0 = nothing; filled in
//   by the includer for the steps before KB), and the macros CG_FS_KB = first step to apply, CG_FS_KE = one past the last
//   (compile-time numbers: the steps outside them cost nothing), and below_mem(g, c0, want, r, b): the intervals below the
//   records r[] of the columns in `want` that no deposit of the run holds (BelowFromMemory, or a form that knows more).
// CG_FS_PART 1: steps CG_FS_KB .. F.n-1.   CG_FS_PART 2: the stores of what deposits CG_FS_KB .. F.n-1 pushed.
#if CG_FS_PART == 1
#define CG_ON(w, j) (((w) & (0xffu << (8 * (j)))) != 0u)
    // (The common outcomes of a step are computed for all 4 columns without branches and selected by the column's mask
    // bit; only the rare ones -- no integer licence, a height past 65535, a gap under a removed layer, an etch that goes
    // on into further layers -- branch.)
#pragma unroll
    for (int k = 0; k < FUSE_MAX; k++) {
        if (k >= F.n || k < CG_FS_KB || k >= CG_FS_KE || !open[k]) continue;
        const FusedOp& op = F.op[k];
        const int matId = op.mat;
        const unsigned opn = open[k];
        if (!op.etch) {
            // _depoAtTop_ (cg1:397-426), as depo_group4_int / depo_group4
            uint32_t qn[4]; unsigned slow = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                qn[j] = rec_z1(r[j]) + (uint32_t)op.vo.depo_k;
                if (CG_ON(opn, j) & ((op.vo.depo_k < 0) | (qn[j] > 65535u))) slow |= 1u << j;
            }
            if (slow) {
#pragma unroll
                for (int j = 0; j < 4; j++) if ((slow >> j) & 1) qn[j] = quantz((double)rec_z1(r[j]) * g.zs + op.t, g.zs, g.inv_zs);
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t rc = r[j];
                // (& and | on purpose: both sides are cheap, and a short circuit would cost a branch per column)
                const bool on = CG_ON(opn, j), some = rc != 0, same = some & (rec_mat(rc) == matId);
                const uint32_t ext = rec_with_z1(rc, qn[j]), app = rec_make_l(qn[j], matId, some ? op.P : LNK_NONE);
                pushed[k][j] = (on & some & !same) ? rc : 0u;
                r[j] = on ? (same ? ext : app) : rc;
            }
            continue;
        }
        // _applyEtchAt (cg1:553-589), as etch_group4_int followed by etch_group4_regs
        const bool isAll = (matId == 255);
        unsigned act = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) act |= (CG_ON(opn, j) & (r[j] != 0) & (isAll | (rec_mat(r[j]) == matId))) ? (1u << j) : 0u;
        if (!act) continue;
        // the intervals below the tops: from the registers of the run's deposits where the link names one of their planes
        uint32_t b[4]; unsigned mem = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int p = rec_lnk(r[j]);
            uint32_t bj = 0u; bool inreg = false;
#pragma unroll
            for (int q = 0; q < FUSE_MAX; q++) {
                if (q >= k) continue;
                const bool m = (F.op[q].etch == 0) & (F.op[q].P == p);
                bj = m ? pushed[q][j] : bj; inreg = inreg | m;
            }
            b[j] = bj;
            if ((((act >> j) & 1) != 0) & (p != LNK_NONE) & !inreg) mem |= 1u << j;
        }
        if (mem) below_mem(g, c0, mem, r, b);
        double rem[4] = {op.t, op.t, op.t, op.t};
        if (op.vo.etch_kc >= 0) {
            unsigned general = 0, holes = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t rc = r[j], below = rec_lnk(rc) != LNK_NONE ? b[j] : 0u;
                const int d = (int)rec_z1(rc) - (int)rec_z1(below);
                const bool a = ((act >> j) & 1) != 0, cut = d >= op.vo.etch_kc, on = d <= op.vo.etch_kr;       // cg1:574-578 / goes on below
                const uint32_t rnew = cut ? rec_with_z1(rc, rec_z1(rc) - (uint32_t)op.vo.etch_ke) : below;   // cg1:579-587: the layer goes, and done
                r[j] = (a & (cut | !on)) ? rnew : rc;
                general |= (a & !cut & on) ? (1u << j) : 0u;
                holes |= (a & !cut & !on & rec_hole(below)) ? (1u << j) : 0u;
            }
            if (holes) {                                       // a gap under the removed layer goes with it
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (!((holes >> j) & 1)) continue;
                    if (rec_lnk(r[j]) == LNK_NONE) r[j] = 0u;
                    else { uint32_t bb[4] = {0, 0, 0, 0}; fused_below4(g, F, k, c0, 1u << j, r, pushed, bb, below_mem); r[j] = bb[j]; }
                }
            }
            act = general;
        }
        if (!(op.t > CG_EPS)) act = 0;                     // cg1:565: nothing to take
        bool have = true;                                  // b[] holds the intervals below the columns in act
        while (act) {
            if (!have) fused_below4(g, F, k, c0, act, r, pushed, b, below_mem);
            have = false;
            unsigned next = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!((act >> j) & 1)) continue;
                const uint32_t below = rec_lnk(r[j]) != LNK_NONE ? b[j] : 0u;
                const double z1 = (double)rec_z1(r[j]) * g.zs, z0 = (double)rec_z1(below) * g.zs, thick = z1 - z0;
                if (rem[j] < thick - CG_EPS) { r[j] = rec_with_z1(r[j], quantz(z1 - rem[j], g.zs, g.inv_zs)); continue; }      // cg1:574-578
                rem[j] -= thick;                                                                                              // cg1:579-587
                r[j] = below;
                if (rec_hole(r[j])) {                              // a gap under the removed layer goes with it
                    if (rec_lnk(r[j]) == LNK_NONE) { r[j] = 0u; continue; }
                    uint32_t bb[4] = {0, 0, 0, 0}; fused_below4(g, F, k, c0, 1u << j, r, pushed, bb, below_mem); r[j] = bb[j];
                }
                if (r[j] != 0 && rem[j] > CG_EPS && (isAll || rec_mat(r[j]) == matId)) next |= 1u << j;
            }
            act = next;
        }
    }
#undef CG_ON
#else
#pragma unroll
    for (int k = 0; k < FUSE_MAX; k++) {
        if (k >= F.n || k < CG_FS_KB || k >= CG_FS_KE || F.op[k].etch) continue;
        const bool p0 = pushed[k][0] != 0, p1 = pushed[k][1] != 0, p2 = pushed[k][2] != 0, p3 = pushed[k][3] != 0;
        if (!(p0 || p1 || p2 || p3)) continue;
        const size_t at = (size_t)F.op[k].P * g.N + c0;
        // (the plane is this deposit's own and a slot of it that no chain names may hold anything: the group goes out in
        // one store, zeros where a column pushed nothing)
        { u32x4 v; v[0] = pushed[k][0]; v[1] = pushed[k][1]; v[2] = pushed[k][2]; v[3] = pushed[k][3]; st_rec4(g.rec + at, v); }
    }

#endif
