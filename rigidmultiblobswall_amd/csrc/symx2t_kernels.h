// symx2t_kernels.h -- the generic symmetric pair sweep (symx_kernels.h: several blocks / several vectors per pair) with
// TWO target blobs per lane (gfx950, fp64), open and pseudo-periodic boundaries.
//
// sym2t_kernels.h did this for the single-vector products tt / tr / rt / rr with open boundaries (+3.5-4.5 %): a lane keeps
// blob `lane` of TWO tile rows (2p, 2p + 1) in registers, a rotation step reads the LDS record of blob jj ONCE, evaluates
// both pairs and adds the SUM of the two transposed contributions with ONE set of ds_add_f64 -- the LDS instructions, the
// rotation's address arithmetic, the tile staging and the u_J flushes are paid per step instead of per pair.  The
// operations of symx_kernels.h carry more per blob (6 input and up to 6 output doubles for the grand product), so those
// costs are a larger share of their pair -- the k-vector passes are bound by the LDS pipe outright -- and the
// pseudo-periodic products read one record for 3^d image pairs of EACH row.  Same policy classes (OP::pair / OP::self,
// OP::NIN / NOUT), same record layout, same accumulators and the SAME finalize kernel as symx_kernel; the unit grid, its
// blocked order and the step schedule are sym2t_kernel's (unit2_seek / unit2_next: unit = (row pair p, tile J >= 2p), 64
// rotation steps of two pairs each; a pair shard is a step range).  The two columns at the diagonal of a row pair run row by
// row through one-target loops.
//
// Registers: the second target costs 2 x (3 + 3 NIN + 3 NOUT) VGPRs plus whatever of the pair algebra the compiler keeps
// alive across the two evaluations.  The per-wave timeline of the single-vector kernel (profiles/r5_wave_timeline_1e4.txt)
// shows that ONE wave per SIMD already issues at the rate four do -- the pair arithmetic of two independent pairs is
// enough instruction-level parallelism -- so these kernels are compiled for WPE waves per SIMD (a template parameter: the
// launcher picks the budget per operation) instead of being squeezed into the 128 registers of four.
#pragma once
#include <type_traits>

#include "sym2t_kernels.h"
#include "symx_kernels.h"

namespace rmb {

template <class OP, bool WALL, bool PERIODIC, int WPE>
__global__ __launch_bounds__(64 * kSymWaves) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void symx2t_kernel(const SymXArgs a) {
  constexpr int NI = OP::NIN, NO = OP::NOUT, NX = SymXExtra<OP>::value;
  constexpr int RD2 = SymXRec<NI, NX>::d2;
  constexpr int RECB = RD2 * 16;
  constexpr int NV = 3 * NI + NX;
  __shared__ double2 rec_all[kSymWaves][64 * RD2];
  __shared__ double accj_all[kSymWaves][3 * NO * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double2* rec = rec_all[wave];
  double* accj = accj_all[wave];
  const char* rec_bytes = reinterpret_cast<const char*>(rec);
  const int px = PERIODIC && a.Lx > 0, py = PERIODIC && a.Ly > 0, pz = PERIODIC && a.Lz > 0;

  const long w = (a.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : (long)blockIdx.x) * kSymWaves + wave;
  for (long chunk = w;; chunk += (long)gridDim.x * kSymWaves) {
  long s = a.step_begin + chunk * a.steps_per_wave;
  if (s >= a.step_end) break;
  long s_end = s + a.steps_per_wave;
  if (s_end > a.step_end) s_end = a.step_end;
  int p = 0, J = 0;
  unit2_seek(a.order, s >> 6, a.n_tiles, p, J);

  int p_cur = -1;
  long i0 = 0;
  bool ok0 = false, ok1 = false;
  // the two target rows: blob 64 (2p) + lane and blob 64 (2p + 1) + lane (constant indices after unrolling: registers)
  double x0 = 0, y0 = 0, z0 = 1.0, x1 = 0, y1 = 0, z1 = 1.0;
  double v0[NV], v1[NV], u0[3 * NO], u1[3 * NO];
#pragma unroll
  for (int c = 0; c < NV; ++c) { v0[c] = 0.0; v1[c] = 0.0; }
#pragma unroll
  for (int c = 0; c < 3 * NO; ++c) { u0[c] = 0.0; u1[c] = 0.0; }

  auto flush_rows = [&]() {
    if (ok0) {
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c)
        __hip_atomic_fetch_add(&a.acc[(long)c * a.n_pad + i0], u0[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (ok1) {
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c)
        __hip_atomic_fetch_add(&a.acc[(long)c * a.n_pad + i0 + 64], u1[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto load_row = [&](long i, bool ok, double& x, double& y, double& z, double* v) {
    x = 1e100; y = 1e100; z = 1.0;
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c] = 0.0;
    if (ok) {
      const double4 q = a.pos[i];
      x = q.x; y = q.y; z = q.z;
      if constexpr (NX > 0) v[3 * NI] = a.extra[i];
#pragma unroll
      for (int m = 0; m < NI; ++m) {
        v[3 * m] = a.in[m][3 * i] * q.w; v[3 * m + 1] = a.in[m][3 * i + 1] * q.w;
        v[3 * m + 2] = a.in_plane ? 0.0 : a.in[m][3 * i + 2] * q.w;
      }
    }
  };
  // all image pairs of (target row, record): ui += forward rows, t (+)= transposed rows.  `skip_self`: the central-box
  // term of a blob with itself (step 0 of a diagonal unit) belongs to the finalize kernel
  // `acc` (a compile-time tag): t += instead of t = -- open boundaries only (the image loop always accumulates into a
  // zeroed t; 0.0 + x is not x for x = -0.0, so the compiler would keep those adds)
  auto pair_all = [&](double xi, double yi, double zi, const double* vi, const double* rd, double* ui, double* t, bool skip_self,
                      auto acc) {
    double dx = xi - rd[0], dy = yi - rd[1], dz = zi - rd[2];
    if constexpr (!PERIODIC) {
      // the second row ACCUMULATES its transposed rows into the first one's inside the contraction (fused multiply-adds
      // instead of finished contributions summed afterwards: 3 NOUT v_add_f64 less per step)
      OP::template pair<WALL, decltype(acc)::value>(a.k, dx, dy, dz, zi, rd[2], vi, rd + 3, ui, t);
    } else {
      if (px) dx = wrap_nearest_pad_safe(dx, a.Lx, a.iLx);
      if (py) dy = wrap_nearest_pad_safe(dy, a.Ly, a.iLy);
      if (pz) dz = wrap_nearest_pad_safe(dz, a.Lz, a.iLz);
      for (int bx = -px; bx <= px; ++bx)
        for (int by = -py; by <= py; ++by)
          for (int bz = -pz; bz <= pz; ++bz) {
            if (skip_self && bx == 0 && by == 0 && bz == 0) continue;
            OP::template pair<WALL, true>(a.k, dx + bx * a.Lx, dy + by * a.Ly, dz + bz * a.Lz, zi, rd[2], vi, rd + 3, ui, t);   // t +=
          }
    }
  };
  auto read_record = [&](int jj, double* rd) {
    const double2* r = reinterpret_cast<const double2*>(rec_bytes + jj * RECB);
#pragma unroll
    for (int c = 0; c < RD2; ++c) { const double2 q = r[c]; rd[2 * c] = q.x; rd[2 * c + 1] = q.y; }
  };

  while (s < s_end) {
    const int k0 = (int)(s & 63);
    const long left = s_end - s;
    const int k1 = (left < 64 - k0) ? (int)(k0 + left) : 64;
    s += k1 - k0;

    if (p != p_cur) {
      if (p_cur >= 0) flush_rows();
      p_cur = p;
      i0 = 64L * (2 * p) + lane;
      ok0 = i0 < a.n; ok1 = i0 + 64 < a.n;
      load_row(i0, ok0, x0, y0, z0, v0);
      load_row(i0 + 64, ok1, x1, y1, z1, v1);
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c) { u0[c] = 0.0; u1[c] = 0.0; }
    }
    {   // tile J -> this wave's LDS slab (record l = blob 64 J + l), its accumulators zeroed
      const long j = 64L * J + lane;
      double rd[2 * RD2];
#pragma unroll
      for (int c = 0; c < 2 * RD2; ++c) rd[c] = 0.0;
      rd[0] = -1e100; rd[1] = -1e100; rd[2] = 1.0;
      if (j < a.n) {
        const double4 q = a.pos[j];
        rd[0] = q.x; rd[1] = q.y; rd[2] = q.z;
#pragma unroll
        for (int m = 0; m < NI; ++m) {
          rd[3 + 3 * m] = a.in[m][3 * j] * q.w; rd[4 + 3 * m] = a.in[m][3 * j + 1] * q.w;
          rd[5 + 3 * m] = a.in_plane ? 0.0 : a.in[m][3 * j + 2] * q.w;
        }
        if constexpr (NX > 0) rd[3 + 3 * NI] = a.extra[j];
      }
#pragma unroll
      for (int c = 0; c < RD2; ++c) rec[lane * RD2 + c] = make_double2(rd[2 * c], rd[2 * c + 1]);
#pragma unroll
      for (int c = 0; c < 3 * NO; ++c) accj[c * 64 + lane] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const int kb = (a.skip_pairs & 1) ? k1 : k0;
    // first step of a diagonal unit: with open boundaries step 0 (the blob itself) is skipped altogether, with periodic
    // ones only its central-box term is (pair_all's skip_self)
    const int kd = (a.skip_pairs & 1) ? k1 : ((PERIODIC || k0 > 1) ? k0 : 1);
    if (J >= 2 * p + 2) {
      // both rows off the diagonal: one record read and one set of LDS adds for the two pairs (x 3^d images each)
      for (int k = kb; k < k1; ++k) {
        const int jj = (lane + k) & 63;
        double rd[2 * RD2], t[3 * NO];
        read_record(jj, rd);
        if constexpr (PERIODIC) {
#pragma unroll
          for (int c = 0; c < 3 * NO; ++c) t[c] = 0.0;
        }
        pair_all(x0, y0, z0, v0, rd, u0, t, false, std::false_type{});
        pair_all(x1, y1, z1, v1, rd, u1, t, false, std::true_type{});
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c)
          __hip_atomic_fetch_add(&accj[c * 64 + jj], t[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    } else if (J == 2 * p) {
      // row 2p: its diagonal unit (every ordered pair of the tile once, forward only); row 2p + 1 lies below the diagonal
      for (int k = kd; k < k1; ++k) {
        const int jj = (lane + k) & 63;
        double rd[2 * RD2], t[3 * NO];
        read_record(jj, rd);
        if constexpr (PERIODIC) {
#pragma unroll
          for (int c = 0; c < 3 * NO; ++c) t[c] = 0.0;
        }
        pair_all(x0, y0, z0, v0, rd, u0, t, k == 0, std::false_type{});
      }
    } else {
      // J == 2p + 1: row 2p sees a full unit, row 2p + 1 its diagonal unit
      for (int k = kb; k < k1; ++k) {
        const int jj = (lane + k) & 63;
        double rd[2 * RD2], t[3 * NO];
        read_record(jj, rd);
        if constexpr (PERIODIC) {
#pragma unroll
          for (int c = 0; c < 3 * NO; ++c) t[c] = 0.0;
        }
        pair_all(x0, y0, z0, v0, rd, u0, t, false, std::false_type{});
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c)
          __hip_atomic_fetch_add(&accj[c * 64 + jj], t[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      for (int k = kd; k < k1; ++k) {
        const int jj = (lane + k) & 63;
        double rd[2 * RD2], t[3 * NO];
        read_record(jj, rd);
        if constexpr (PERIODIC) {
#pragma unroll
          for (int c = 0; c < 3 * NO; ++c) t[c] = 0.0;
        }
        pair_all(x1, y1, z1, v1, rd, u1, t, k == 0, std::false_type{});
      }
    }
    if (J != 2 * p) {     // the slab holds transposed contributions (none in the pure diagonal column)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const long j = 64L * J + lane;
      if (j < a.n && !(a.skip_pairs & 2)) {
#pragma unroll
        for (int c = 0; c < 3 * NO; ++c)
          __hip_atomic_fetch_add(&a.acc[(long)c * a.n_pad + j], accj[c * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __builtin_amdgcn_wave_barrier();   // accj / rec are rewritten by the next unit
    if (k1 == 64) unit2_next(a.order, a.n_tiles, p, J);
  }
  if (p_cur >= 0) flush_rows();
  }   // chunks
}

}  // namespace rmb
