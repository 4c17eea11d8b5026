// rmb_symx_coop.hip -- the workgroup-cooperative instances of the generic symmetric skeleton (symx_coop_kernels.h), in
// a translation unit of their own (they compile in parallel with rmb_sym.hip) and handed to symx_device as launch thunks.
#include "rmb_internal.h"

#include "symx_coop_kernels.h"

namespace rmbi {

namespace {

template <class OP, bool WALL, bool PER>
void launch_coop(const void* args, const rmb::PairConsts&, unsigned blocks, size_t dyn_lds, hipStream_t s) {
  hipLaunchKernelGGL((rmb::symx_coop_kernel<OP, WALL, PER>), dim3(blocks), dim3(64 * rmb::kSymWaves), dyn_lds, s,
                     *static_cast<const rmb::SymXArgs*>(args));
}

template <class OP, bool WALL, bool PER>
Kernel32 one() {
  static int occ = 0;
  return Kernel32{(const void*)rmb::symx_coop_kernel<OP, WALL, PER>, rmb::SymXCoopLds<OP>::bytes, &occ, launch_coop<OP, WALL, PER>};
}

template <class OP>
Kernel32 of(bool wall, bool periodic) {
  if (wall) return periodic ? one<OP, true, true>() : one<OP, true, false>();
  return periodic ? one<OP, false, true>() : one<OP, false, false>();
}

}  // namespace

// (the struct is shared with the fp32 thunks: fn / static LDS / occupancy cache / launch)
Kernel32 symx_coop(int sx, bool wall, bool periodic) {
  switch (sx) {
    case SX_TT: return of<rmb::OpSingle<rmb::KIND_TT>>(wall, periodic);
    case SX_TR: return of<rmb::OpSingle<rmb::KIND_TR>>(wall, periodic);
    case SX_RT: return of<rmb::OpSingle<rmb::KIND_RT>>(wall, periodic);
    case SX_RR: return of<rmb::OpSingle<rmb::KIND_RR>>(wall, periodic);
    case SX_FUSED: return of<rmb::OpFusedRow>(wall, periodic);
    case SX_GRAND: return of<rmb::OpGrand>(wall, periodic);
    case SX_COLF: return of<rmb::OpColumnF>(wall, periodic);
    case SX_FREE: return of<rmb::OpFreeSurface>(false, periodic);     // raw heights: the wall = 0 instance serves both
    case SX_RADII: return of<rmb::OpRadiiTT>(wall, periodic);
    default: break;
  }
  if (sx >= SX_K2 && sx < SX_COUNT) {
    const int k = 2 + (sx - SX_K2) / 4, kind = (sx - SX_K2) % 4;
#define RMB_KC(KIND, K) if (kind == KIND && k == K) return of<rmb::OpKindK<KIND, K>>(wall, periodic);
#define RMB_KC_ROW(K) RMB_KC(rmb::KIND_TT, K) RMB_KC(rmb::KIND_TR, K) RMB_KC(rmb::KIND_RT, K) RMB_KC(rmb::KIND_RR, K)
    RMB_KC_ROW(2) RMB_KC_ROW(3) RMB_KC_ROW(4)
#undef RMB_KC_ROW
#undef RMB_KC
  }
  return Kernel32{nullptr, 0, nullptr, nullptr};
}

}  // namespace rmbi
