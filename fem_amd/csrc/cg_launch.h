// Launch wrappers of the two lateral-stencil kernel families.  Their many instantiations (radius x etch x ...)
// dominate the build, so they are compiled in translation units of their own, a few radii each
// (cg_launch_stream.hip, cg_launch_sym.hip; see the Makefile), and colgrid.hip calls them through these functions.
#pragma once
#include "cg_kernels.hip.h"

namespace cgk {

struct StreamArgs {
    GridView gv; const uint8_t* mask; int mat; double t_eff; int Rimg;
    const uint8_t* halo_lo; const uint8_t* halo_up; const float* d_quad; StreamOpts so; int* flags; unsigned long long* stats;
    bool lean;      // every licence k_stencil_lean asks for holds (see cg_stencil_lean.hip.h): tiles that read no message take that form
    const FusedOps* fold = nullptr;      // lean deposit of radius <= LEAN_FOLD_MAX_R: the vertical steps it takes along (op[0] stands for the deposit itself)
    const FreshTab* fresh = nullptr;     // lean deposit without a mask: the run of vertical steps on the reset grid in front of it (FRESH form)
};
constexpr int LEAN_FOLD_MAX_R = 4;

// k_stencil_stream<Re, etch, msg, masked> on `blocks` one-wave workgroups.  Re = 1..8 (larger: 8).
void launch_stream(int Re, bool etch, bool msg, bool masked, unsigned blocks, hipStream_t stream, const StreamArgs& a);

struct SymArgs {
    GridView gv; const uint8_t* mask; int mat; double t_eff; int Rimg;
    const uint8_t* halo_lo; const uint8_t* halo_up; const float* quad_host;      // (Re+1)^2 increments on the HOST
    const Tap* d_taps64; int ntaps64; int* flags; unsigned long long* stats;
};
// k_stencil_sym<Re, tjs, etch, stats> on `blocks` workgroups of 256.  Re = 1..8, tjs = 64, 128 or 256.
void launch_sym(int Re, int tjs, bool etch, bool stats, unsigned blocks, hipStream_t stream, const SymArgs& a);

}  // namespace cgk
