// rmb_symx2t_per.hip -- two-targets-per-lane instances of the generic symmetric skeleton, pseudo-periodic boundaries
// (symx2t_kernels.h; image convention mobility/mobility_numba.py:170-197).
#include "symx2t_instances.h"

namespace rmbi {
Kernel32 symx_two_periodic(int sx, bool wall, int* waves_per_eu) { return symx2t_detail::table<true>(sx, wall, waves_per_eu); }
}  // namespace rmbi
