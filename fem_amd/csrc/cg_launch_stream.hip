// k_stencil_stream instantiations for the radii CG_RLO..CG_RHI (one translation unit per pair of radii).
#include "cg_launch.h"
#include "cg_stencil_lean.hip.h"

#define CG_CAT2(a, b) a##b
#define CG_CAT(a, b) CG_CAT2(a, b)

namespace cgk {

template <int RR>
static void stream_one(bool etch, bool msg, bool masked, unsigned blocks, hipStream_t s, const StreamArgs& a) {
    if (a.lean && !msg) {
        static const FusedOps none = {};
        static const FreshTab no_run = {};
#define LL(E, K, NF, W) hipLaunchKernelGGL((k_stencil_lean<RR, E, K, NF, W>), dim3(blocks), dim3(64), 0, s, a.gv, a.mask, a.mat, a.t_eff, a.d_quad, a.so, \
                                          a.fold ? *a.fold : none, a.fresh ? *a.fresh : no_run)
        constexpr int ANY = FUSE_MAX - 1;
        const int behind = (a.fold && !etch) ? a.fold->n - 1 : 0;       // steps the launch takes along behind its deposit
        if (a.fresh && !etch && !masked) {                  // (the host asks for this form only where it exists)
            if constexpr (RR <= LEAN_FOLD_MAX_R) { if (behind > 0) { LL(false, false, ANY, true); return; } }
            LL(false, false, 0, true); return;
        }
        if constexpr (RR <= LEAN_FOLD_MAX_R) {
            if (behind == 1) { if (masked) LL(false, true, 1, false); else LL(false, false, 1, false); return; }
            if (behind == 2) { if (masked) LL(false, true, 2, false); else LL(false, false, 2, false); return; }
            if (behind > 2) { if (masked) LL(false, true, ANY, false); else LL(false, false, ANY, false); return; }
        }
        if (etch) { if (masked) LL(true, true, 0, false); else LL(true, false, 0, false); }
        else { if (masked) LL(false, true, 0, false); else LL(false, false, 0, false); }
#undef LL
        return;
    }
    const bool lic = a.so.deq_f32 && a.so.grow_f32 && !msg;
#define L(E, M, K) do { if (lic) hipLaunchKernelGGL((k_stencil_stream<RR, E, M, K, true>), dim3(blocks), dim3(64), 0, s, a.gv, a.mask, a.mat, a.t_eff, a.Rimg, \
                                      a.halo_lo, a.halo_up, a.d_quad, a.so, a.flags, a.stats); \
                        else hipLaunchKernelGGL((k_stencil_stream<RR, E, M, K, false>), dim3(blocks), dim3(64), 0, s, a.gv, a.mask, a.mat, a.t_eff, a.Rimg, \
                                      a.halo_lo, a.halo_up, a.d_quad, a.so, a.flags, a.stats); } while (0)
    if (etch) { if (msg) { if (masked) L(true, true, true); else L(true, true, false); } else { if (masked) L(true, false, true); else L(true, false, false); } }
    else { if (msg) { if (masked) L(false, true, true); else L(false, true, false); } else { if (masked) L(false, false, true); else L(false, false, false); } }
#undef L
}

bool CG_CAT(launch_stream_r, CG_RLO)(int Re, bool etch, bool msg, bool masked, unsigned blocks, hipStream_t s, const StreamArgs& a) {
    if (Re == CG_RLO) { stream_one<CG_RLO>(etch, msg, masked, blocks, s, a); return true; }
    if (Re == CG_RHI) { stream_one<CG_RHI>(etch, msg, masked, blocks, s, a); return true; }
    return false;
}

}  // namespace cgk
