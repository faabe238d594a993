// k_stencil_sym instantiations for the radii CG_RLO..CG_RHI (one translation unit per pair of radii).
#include <cmath>
#include <cstring>
#include "cg_launch.h"

#define CG_CAT2(a, b) a##b
#define CG_CAT(a, b) CG_CAT2(a, b)

namespace cgk {

template <int RR>
static void sym_one(int tjs, bool etch, bool stats, unsigned blocks, hipStream_t s, const SymArgs& a) {
    AddTab<RR> tab; memcpy(tab.a, a.quad_host, sizeof tab.a);
    for (int du = 0; du <= RR; du++) {
        tab.lim[du] = -1;
        for (int dv = 0; dv <= RR; dv++) if (tab.a[du * (RR + 1) + dv] != -INFINITY) tab.lim[du] = dv;
    }
#define L(TT, E, S) hipLaunchKernelGGL((k_stencil_sym<RR, TT, E, S>), dim3(blocks), dim3(256), 0, s, a.gv, a.mask, a.mat, a.t_eff, a.Rimg, \
                                       a.halo_lo, a.halo_up, tab, a.d_taps64, a.ntaps64, a.flags, a.stats)
#define LT(E, S) do { if (tjs == 64) L(64, E, S); else if (tjs == 128) L(128, E, S); else L(256, E, S); } while (0)
    if (etch) { if (stats) LT(true, true); else LT(true, false); }
    else { if (stats) LT(false, true); else LT(false, false); }
#undef LT
#undef L
}

bool CG_CAT(launch_sym_r, CG_RLO)(int Re, int tjs, bool etch, bool stats, unsigned blocks, hipStream_t s, const SymArgs& a) {
    if (Re == CG_RLO) { sym_one<CG_RLO>(tjs, etch, stats, blocks, s, a); return true; }
    if (Re == CG_RHI) { sym_one<CG_RHI>(tjs, etch, stats, blocks, s, a); return true; }
    return false;
}

}  // namespace cgk
