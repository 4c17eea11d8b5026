// rmb_symx2t.hip -- two-targets-per-lane instances of the generic symmetric skeleton, open boundaries (symx2t_kernels.h).
#include "symx2t_instances.h"

namespace rmbi {
Kernel32 symx_two_open(int sx, bool wall, int* waves_per_eu) { return symx2t_detail::table<false>(sx, wall, waves_per_eu); }
}  // namespace rmbi
