// Dispatch over the per-radius translation units of the lateral-stencil kernels (see cg_launch.h).
#include "cg_launch.h"

namespace cgk {
#define DECL(lo) bool launch_stream_r##lo(int, bool, bool, bool, unsigned, hipStream_t, const StreamArgs&); \
                 bool launch_sym_r##lo(int, int, bool, bool, unsigned, hipStream_t, const SymArgs&);
DECL(1) DECL(3) DECL(5) DECL(7)
#undef DECL

void launch_stream(int Re, bool etch, bool msg, bool masked, unsigned blocks, hipStream_t s, const StreamArgs& a) {
    if (Re > 8) Re = 8;
    (void)(launch_stream_r1(Re, etch, msg, masked, blocks, s, a) || launch_stream_r3(Re, etch, msg, masked, blocks, s, a) ||
           launch_stream_r5(Re, etch, msg, masked, blocks, s, a) || launch_stream_r7(Re, etch, msg, masked, blocks, s, a));
}
void launch_sym(int Re, int tjs, bool etch, bool stats, unsigned blocks, hipStream_t s, const SymArgs& a) {
    if (Re > 8) Re = 8;
    (void)(launch_sym_r1(Re, tjs, etch, stats, blocks, s, a) || launch_sym_r3(Re, tjs, etch, stats, blocks, s, a) ||
           launch_sym_r5(Re, tjs, etch, stats, blocks, s, a) || launch_sym_r7(Re, tjs, etch, stats, blocks, s, a));
}
}  // namespace cgk
