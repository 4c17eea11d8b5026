// symx2t_instances.h -- launch thunks of the two-targets-per-lane instances of the generic symmetric skeleton
// (symx2t_kernels.h), shared by the two translation units that compile them in parallel: rmb_symx2t.hip (open
// boundaries) and rmb_symx2t_per.hip (pseudo-periodic).  Same thunk shape as the cooperative instances (Kernel32).
#pragma once
#include "rmb_internal.h"

#include "symx2t_kernels.h"

namespace rmbi {
namespace symx2t_detail {

template <class OP, bool WALL, bool PER, int WPE>
void launch(const void* args, const rmb::PairConsts&, unsigned blocks, size_t dyn_lds, hipStream_t s) {
  hipLaunchKernelGGL((rmb::symx2t_kernel<OP, WALL, PER, WPE>), dim3(blocks), dim3(64 * rmb::kSymWaves), dyn_lds, s,
                     *static_cast<const rmb::SymXArgs*>(args));
}

template <class OP, bool WALL, bool PER, int WPE>
Kernel32 one() {
  static int occ = 0;
  constexpr size_t lds = sizeof(double2) * rmb::kSymWaves * 64 * rmb::SymXRec<OP::NIN, rmb::SymXExtra<OP>::value>::d2 +
                         sizeof(double) * rmb::kSymWaves * 3 * OP::NOUT * 64;
  return Kernel32{(const void*)rmb::symx2t_kernel<OP, WALL, PER, WPE>, lds, &occ, launch<OP, WALL, PER, WPE>};
}

// Waves per SIMD the instances are compiled for (the register budget is 512 / WPE): one wave per SIMD already issues
// at the rate four do (profiles/r5_wave_timeline_1e4.txt), so the budget follows the operation's live state.
template <class OP, bool PER> struct Wpe { static constexpr int value = PER ? 2 : (OP::NIN + OP::NOUT >= 4 ? 2 : 3); };

template <class OP, bool PER>
Kernel32 of(bool wall, int* wpe) {
  *wpe = Wpe<OP, PER>::value;
  return wall ? one<OP, true, PER, Wpe<OP, PER>::value>() : one<OP, false, PER, Wpe<OP, PER>::value>();
}

// the operations that have a two-target instance: the four single-vector blocks (periodic only: open boundaries have
// sym2t_kernel), fused row, grand, force column, one block on two vectors
template <bool PER>
Kernel32 table(int sx, bool wall, int* wpe) {
  switch (sx) {
    case SX_TT: if constexpr (PER) return of<rmb::OpSingle<rmb::KIND_TT>, PER>(wall, wpe); else break;
    case SX_TR: if constexpr (PER) return of<rmb::OpSingle<rmb::KIND_TR>, PER>(wall, wpe); else break;
    case SX_RT: if constexpr (PER) return of<rmb::OpSingle<rmb::KIND_RT>, PER>(wall, wpe); else break;
    case SX_RR: if constexpr (PER) return of<rmb::OpSingle<rmb::KIND_RR>, PER>(wall, wpe); else break;
    case SX_FUSED: return of<rmb::OpFusedRow, PER>(wall, wpe);
    case SX_GRAND: return of<rmb::OpGrand, PER>(wall, wpe);
    case SX_COLF: return of<rmb::OpColumnF, PER>(wall, wpe);
    case SX_K2 + 0: return of<rmb::OpKindK<rmb::KIND_TT, 2>, PER>(wall, wpe);
    case SX_K2 + 1: return of<rmb::OpKindK<rmb::KIND_TR, 2>, PER>(wall, wpe);
    case SX_K2 + 2: return of<rmb::OpKindK<rmb::KIND_RT, 2>, PER>(wall, wpe);
    case SX_K2 + 3: return of<rmb::OpKindK<rmb::KIND_RR, 2>, PER>(wall, wpe);
    default: break;
  }
  *wpe = 0;
  return Kernel32{nullptr, 0, nullptr, nullptr};
}

}  // namespace symx2t_detail
}  // namespace rmbi
