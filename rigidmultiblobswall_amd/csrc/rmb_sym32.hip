// rmb_sym32.hip -- the single-precision twins of the symmetric kernels (the reference's `precision = 'single'` build,
// mobility/mobility_pycuda.py:7-19; sym32_kernels.h, symx32_kernels.h), handed to the fp64 launchers of rmb_sym.hip as
// launch thunks so that this translation unit compiles on its own.
#include "rmb_internal.h"

#include "sym32_kernels.h"
#include "symx32_kernels.h"

namespace rmbi {

namespace {

rmb::f32::PairConsts pair_consts32(const rmb::PairConsts& k) {
  rmb::f32::PairConsts f;
  f.a2 = (float)k.a2; f.four_a2 = (float)k.four_a2; f.tt_k1 = (float)k.tt_k1; f.tt_k2 = (float)k.tt_k2; f.tt_k3 = (float)k.tt_k3;
  f.tt_n0 = (float)k.tt_n0; f.tt_n1 = (float)k.tt_n1; f.tt_n2 = (float)k.tt_n2;
  f.rr_m0 = (float)k.rr_m0; f.rr_m1 = (float)k.rr_m1; f.rr_m2 = (float)k.rr_m2; f.rr_m3 = (float)k.rr_m3; f.rr_m4 = (float)k.rr_m4;
  f.c_q0 = (float)k.c_q0; f.c_q1 = (float)k.c_q1; f.m7 = (float)k.m7; f.m6 = (float)k.m6; f.c15 = (float)k.c15; f.c30 = (float)k.c30;
  return f;
}

template <bool WALL>
void launch_sym32_tt(const void* args, const rmb::PairConsts& k, unsigned blocks, size_t dyn_lds, hipStream_t s) {
  hipLaunchKernelGGL((rmb::sym32_tt_kernel<WALL>), dim3(blocks), dim3(64 * rmb::kSymWaves), dyn_lds, s,
                     *static_cast<const rmb::SymArgs*>(args), pair_consts32(k));
}

template <class OP32, bool WALL>
void launch_symx32(const void* args, const rmb::PairConsts& k, unsigned blocks, size_t dyn_lds, hipStream_t s) {
  hipLaunchKernelGGL((rmb::symx32_kernel<OP32, WALL>), dim3(blocks), dim3(64 * rmb::kSymWaves), dyn_lds, s,
                     *static_cast<const rmb::SymXArgs*>(args), pair_consts32(k));
}

template <bool RADII>
void launch_force32(const void* args, const rmb::PairConsts&, unsigned blocks, size_t dyn_lds, hipStream_t s) {
  hipLaunchKernelGGL((rmb::sym_force32_kernel<RADII>), dim3(blocks), dim3(64 * rmb::kSymWaves), dyn_lds, s,
                     *static_cast<const rmb::SymForceArgs*>(args));
}

template <class OP32>
Kernel32 symx32_of(bool wall) {
  static int occ[2] = {0, 0};
  if (wall) return Kernel32{(const void*)rmb::symx32_kernel<OP32, true>, rmb::SymX32Lds<OP32>::bytes, &occ[1], launch_symx32<OP32, true>};
  return Kernel32{(const void*)rmb::symx32_kernel<OP32, false>, rmb::SymX32Lds<OP32>::bytes, &occ[0], launch_symx32<OP32, false>};
}

}  // namespace

Kernel32 sym32_tt(bool wall) {
  static int occ[2] = {0, 0};
  const size_t lds = (sizeof(float) * 9 + sizeof(double) * 3) * rmb::kSymWaves * 64;
  if (wall) return Kernel32{(const void*)rmb::sym32_tt_kernel<true>, lds, &occ[1], launch_sym32_tt<true>};
  return Kernel32{(const void*)rmb::sym32_tt_kernel<false>, lds, &occ[0], launch_sym32_tt<false>};
}

Kernel32 sym_force32(bool radii) {
  static int occ[2] = {0, 0};
  if (radii) return Kernel32{(const void*)rmb::sym_force32_kernel<true>, 0, &occ[1], launch_force32<true>};
  return Kernel32{(const void*)rmb::sym_force32_kernel<false>, 0, &occ[0], launch_force32<false>};
}

// open boundaries only; the free-surface operation takes raw heights, so its wall = 0 instance serves both columns
Kernel32 symx32(int sx, bool wall) {
  switch (sx) {
    case SX_TT: return symx32_of<rmb::OpSingle32<rmb::KIND_TT>>(wall);
    case SX_TR: return symx32_of<rmb::OpSingle32<rmb::KIND_TR>>(wall);
    case SX_RT: return symx32_of<rmb::OpSingle32<rmb::KIND_RT>>(wall);
    case SX_RR: return symx32_of<rmb::OpSingle32<rmb::KIND_RR>>(wall);
    case SX_FUSED: return symx32_of<rmb::OpFusedRow32>(wall);
    case SX_GRAND: return symx32_of<rmb::OpGrand32>(wall);
    case SX_COLF: return symx32_of<rmb::OpColumnF32>(wall);
    case SX_FREE: return symx32_of<rmb::OpFreeSurface32>(false);
    case SX_RADII: return symx32_of<rmb::OpRadiiTT32>(wall);
    default: break;
  }
  if (sx >= SX_K2 && sx < SX_COUNT) {
    const int k = 2 + (sx - SX_K2) / 4, kind = (sx - SX_K2) % 4;
#define RMB_K32(KIND, K) if (kind == KIND && k == K) return symx32_of<rmb::OpKindK32<KIND, K>>(wall);
#define RMB_K32_ROW(K) RMB_K32(rmb::KIND_TT, K) RMB_K32(rmb::KIND_TR, K) RMB_K32(rmb::KIND_RT, K) RMB_K32(rmb::KIND_RR, K)
    RMB_K32_ROW(2) RMB_K32_ROW(3) RMB_K32_ROW(4)
#undef RMB_K32_ROW
#undef RMB_K32
  }
  return Kernel32{nullptr, 0, nullptr, nullptr};
}

}  // namespace rmbi
