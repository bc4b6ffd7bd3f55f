// hso_wave_reduce.h — sum N per-lane values over the 64 lanes of a wavefront by recursive halving (reduce-scatter): at each
// step (lane distance 32, 16, 8, ...) a lane hands the half of its values it is not responsible for to its partner and adds
// the partner's contribution to the half it keeps, so N values cost ~N exchanges instead of 6 N for N butterflies.  Afterwards
// the lane whose `slot` is k (< N) holds the total of value k (every slot is held by 64 / N adjacent lanes when N < 64; all
// of them hold the same total).  Fixed order => deterministic floating point.  Exchanges use the cheapest gfx950 form per
// distance: 32 / 16 v_permlane32_swap / v_permlane16_swap (the instruction swaps the upper half of one register with the lower
// half of another: exactly "hand over the half you do not keep"), 8 / 4 / 2 / 1 DPP (hso_dev_math.h: lane_xor).
// The same scheme as the tracker's exchange (hso_tracker_core.h), here for the later stages.
#pragma once
#include "hso_dev_math.h"

namespace hso_dev {

template <int M> HSO_DEV double halve_swap_sum(double lo, double hi)
{
  static_assert(M == 32 || M == 16, "swap distances");
  const unsigned long long bl = (unsigned long long)__double_as_longlong(lo), bh = (unsigned long long)__double_as_longlong(hi);
  lane_u32x2 r0, r1;
  if constexpr (M == 32) {
    r0 = __builtin_amdgcn_permlane32_swap((unsigned)bl, (unsigned)bh, false, false);
    r1 = __builtin_amdgcn_permlane32_swap((unsigned)(bl >> 32), (unsigned)(bh >> 32), false, false);
  } else {
    r0 = __builtin_amdgcn_permlane16_swap((unsigned)bl, (unsigned)bh, false, false);
    r1 = __builtin_amdgcn_permlane16_swap((unsigned)(bl >> 32), (unsigned)(bh >> 32), false, false);
  }
  return __longlong_as_double((long long)(((unsigned long long)r1[0] << 32) | r0[0])) +
         __longlong_as_double((long long)(((unsigned long long)r1[1] << 32) | r0[1]));
}
template <int M> HSO_DEV float halve_swap_sum(float lo, float hi)
{
  static_assert(M == 32 || M == 16, "swap distances");
  lane_u32x2 r;
  if constexpr (M == 32) r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
  else r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <typename T, int D> HSO_DEV void wave_butterfly_rest(T& x)
{
  if constexpr (D >= 1) {
    if constexpr (D == 32 || D == 16) x = halve_swap_sum<D>(x, x);
    else x += lane_xor<D>(x);
    wave_butterfly_rest<T, D / 2>(x);
  }
}

template <typename T, int N, int M>
struct WaveHalve {
  static HSO_DEV void run(T (&v)[N], int lane, int& slot, T& out)
  {
    static_assert((N & (N - 1)) == 0 && N >= 2, "N must be a power of two");
    constexpr int HALF = N / 2;
    const bool up = (lane & M) != 0;
    T keep[HALF];
#pragma unroll
    for (int i = 0; i < HALF; i++) {
      if constexpr (M == 32 || M == 16) {
        keep[i] = halve_swap_sum<M>(v[i], v[HALF + i]);
      } else {
        const T send = up ? v[i] : v[HALF + i];
        keep[i] = (up ? v[HALF + i] : v[i]) + lane_xor<M>(send);
      }
    }
    if (up) slot += HALF;
    if constexpr (HALF == 1) {
      // a single value left: plain butterflies over the remaining distances; every lane of the group ends with the total
      T x = keep[0];
      wave_butterfly_rest<T, M / 2>(x);
      out = x;
    } else {
      WaveHalve<T, HALF, M / 2>::run(keep, lane, slot, out);
    }
  }
};

// N = 32 doubles (or floats) per lane -> lane holds the wave total of value `slot` (two adjacent lanes per slot)
template <typename T>
HSO_DEV T wave_reduce_scatter32(T (&v)[32], int lane, int& slot)
{
  T out;
  slot = 0;
  WaveHalve<T, 32, 32>::run(v, lane, slot, out);
  return out;
}

}  // namespace hso_dev
