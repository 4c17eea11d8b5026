// symx_coop_kernels.h -- workgroup-cooperative variant of the generic symmetric skeleton (symx_kernels.h), the same
// restructuring as sym_coop_kernels.h: the WORKGROUP owns a contiguous range of rotation steps, its four waves split
// the steps of each tile pair, read one staged copy of tile J and add into one shared pair of LDS accumulators.
//
// Besides the shared loads and flushes (what sym_coop_kernel is for) this matters for the operations with many
// vectors: symx_kernel keeps a private tile-J slab and accumulator per WAVE, 61 KB of LDS per workgroup for four
// vectors (OpKindK<.., 4>) and 47 KB for three -- two resp. three workgroups per CU, i.e. two or three waves per SIMD
// where the registers would allow three or four.  One slab per workgroup is 15 + 2 x 6 KB: LDS stops limiting residency.
#pragma once
#include "symx_kernels.h"

namespace rmb {

template <class OP> struct SymXCoopLds {
  static constexpr int RD2 = SymXRec<OP::NIN, SymXExtra<OP>::value>::d2;
  static constexpr size_t bytes = sizeof(double2) * 64 * RD2 + 2 * sizeof(double) * 3 * OP::NOUT * 64;
};

template <class OP, bool WALL, bool PERIODIC>
__global__ __launch_bounds__(64 * kSymWaves) void symx_coop_kernel(const SymXArgs a) {
  constexpr int NI = OP::NIN, NO = OP::NOUT, NX = SymXExtra<OP>::value;
  constexpr int RD2 = SymXRec<NI, NX>::d2;
  constexpr int RECB = RD2 * 16;
  __shared__ double2 rec[64 * RD2];
  __shared__ double accj[3 * NO * 64];
  __shared__ double acci[3 * NO * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* rec_bytes = reinterpret_cast<const char*>(rec);

  // a.steps_per_wave carries the steps per WORKGROUP here (rmb_sym.hip)
  // strided chunks: wave / workgroup `id` takes the step ranges id, id + n, id + 2 n, ... of `spw` steps each (one range when
  // the launch is planned that way: n spw >= the steps of the launch).  Waves that run at the same time then work on
  // NEIGHBOURING ranges whatever the size of the problem -- with the blocked unit order and the XCD-aware numbering
  // that keeps a launch's tile loads in one L2 (profiles/r4_unit_order.txt).
  for (long chunk = (a.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (long)blockIdx.x);; chunk += (long)gridDim.x) {
  long s = a.step_begin + chunk * a.steps_per_wave;
  if (s >= a.step_end) break;
  long s_end = s + a.steps_per_wave;
  if (s_end > a.step_end) s_end = a.step_end;
  int I = 0, J = 0;
  if (s < s_end) unit_seek(a.order, s >> 6, a.n_tiles, I, J);
  if (wave == 1) {
#pragma unroll
    for (int c = 0; c < 3 * NO; ++c) acci[c * 64 + lane] = 0.0;
  }

  int I_cur = -1;
  long i = 0;
  double xi = 0, yi = 0, zi = 1.0;
  double vi[3 * NI + NX];
#pragma unroll
  for (int c = 0; c < 3 * NI + NX; ++c) vi[c] = 0.0;

  auto flush_row = [&]() {      // wave 1, after a workgroup barrier that follows the row's last adds
    if (i < a.n) {
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c)
        __hip_atomic_fetch_add(&a.acc[(long)c * a.n_pad + i], acci[c * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int c = 0; c < 3 * NO; ++c) acci[c * 64 + lane] = 0.0;
  };

  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    s += k1 - k0;

    if (I != I_cur) {
      if (I_cur >= 0 && wave == 1) flush_row();
      I_cur = I;
      i = 64L * I + lane;
      xi = 1e100; yi = 1e100; zi = 1.0;
#pragma unroll
      for (int c = 0; c < 3 * NI + NX; ++c) vi[c] = 0.0;
      if (i < a.n) {
        const double4 p = a.pos[i];
        xi = p.x; yi = p.y; zi = p.z;
        if constexpr (NX > 0) vi[3 * NI] = a.extra[i];
#pragma unroll
        for (int v = 0; v < NI; ++v) {
          vi[3 * v] = a.in[v][3 * i] * p.w; vi[3 * v + 1] = a.in[v][3 * i + 1] * p.w;
          vi[3 * v + 2] = a.in_plane ? 0.0 : a.in[v][3 * i + 2] * p.w;
        }
      }
    }
    if (wave == 0) {   // tile J -> the workgroup's slab (record l = blob 64 J + l), zero its accumulators
      const long j = 64L * J + lane;
      double rd[2 * RD2];
#pragma unroll
      for (int c = 0; c < 2 * RD2; ++c) rd[c] = 0.0;
      rd[0] = -1e100; rd[1] = -1e100; rd[2] = 1.0;
      if (j < a.n) {
        const double4 p = a.pos[j];
        rd[0] = p.x; rd[1] = p.y; rd[2] = p.z;
#pragma unroll
        for (int v = 0; v < NI; ++v) {
          rd[3 + 3 * v] = a.in[v][3 * j] * p.w; rd[4 + 3 * v] = a.in[v][3 * j + 1] * p.w;
          rd[5 + 3 * v] = a.in_plane ? 0.0 : a.in[v][3 * j + 2] * p.w;
        }
        if constexpr (NX > 0) rd[3 + 3 * NI] = a.extra[j];
      }
#pragma unroll
      for (int c = 0; c < RD2; ++c) rec[lane * RD2 + c] = make_double2(rd[2 * c], rd[2 * c + 1]);
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c) accj[c * 64 + lane] = 0.0;
    }
    __syncthreads();

    const int px = PERIODIC && a.Lx > 0, py = PERIODIC && a.Ly > 0, pz = PERIODIC && a.Lz > 0;
    const bool diag = I == J;
    // this wave's share of the piece [k0, k1); diagonal units visit every ordered pair once (forward only), step 0 is the
    // blob itself: its central-box term is the self term (finalize), its periodic images use the pair formula
    const int q = (k1 - k0 + kSymWaves - 1) / kSymWaves;
    int ka = k0 + wave * q;
    const int kb = ka + q < k1 ? ka + q : k1;
    if (diag && !PERIODIC && ka < 1) ka = 1;
    if (a.skip_pairs & 1) ka = kb;
    double ui[3 * NO];
#pragma unroll
    for (int c = 0; c < 3 * NO; ++c) ui[c] = 0.0;
    for (int k = ka; k < kb; ++k) {
      const int jj = (lane + k) & 63;
      const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * RECB);
      double rd[2 * RD2];
#pragma unroll
      for (int c = 0; c < RD2; ++c) { const double2 qq = r[c]; rd[2 * c] = qq.x; rd[2 * c + 1] = qq.y; }
      double dx = xi - rd[0], dy = yi - rd[1], dz = zi - rd[2];
      double t[3 * NO];
      if constexpr (!PERIODIC) {
        OP::template pair<WALL>(a.k, dx, dy, dz, zi, rd[2], vi, rd + 3, ui, t);
      } else {
        if (px) dx = wrap_nearest_pad_safe(dx, a.Lx, a.iLx);
        if (py) dy = wrap_nearest_pad_safe(dy, a.Ly, a.iLy);
        if (pz) dz = wrap_nearest_pad_safe(dz, a.Lz, a.iLz);
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c) t[c] = 0.0;
        for (int bx = -px; bx <= px; ++bx)
          for (int by = -py; by <= py; ++by)
            for (int bz = -pz; bz <= pz; ++bz) {
              if (diag && k == 0 && bx == 0 && by == 0 && bz == 0) continue;
              OP::template pair<WALL, true>(a.k, dx + bx * a.Lx, dy + by * a.Ly, dz + bz * a.Lz, zi, rd[2], vi, rd + 3, ui, t);      // accumulates (fused multiply-adds)
            }
      }
      if (!diag) {   // workgroup-uniform
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c)
          __hip_atomic_fetch_add(&accj[c * 64 + jj], t[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
#pragma unroll
    for (int c = 0; c < 3 * NO; ++c)
      __hip_atomic_fetch_add(&acci[c * 64 + lane], ui[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    if (wave == 0 && !diag && !(a.skip_pairs & 2)) {   // one flush of u_J per piece; wave 0 re-stages the slab next
      const long j = 64L * J + lane;
      if (j < a.n) {
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c)
          __hip_atomic_fetch_add(&a.acc[(long)c * a.n_pad + j], accj[c * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (k1 == 64) {
      unit_next(a.order, a.n_tiles, I, J);
    }
  }
  if (I_cur >= 0 && wave == 1) flush_row();
  }   // chunks
}

}  // namespace rmb
