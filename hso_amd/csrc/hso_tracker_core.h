// hso_tracker_core.h — the device code of the tracker, parametrised by the workgroup shape.  Included (twice) by
// hso_tracker.hip inside a namespace, with
//   TRK_THREADS    threads per workgroup (a multiple of 64),
//   TRK_LDS_KB     LDS the workgroup may use (image + state),
//   TRK_OLD_SHARE  sixteenths of the features given to the first half of the waves (8 = even split)
// defined by the includer.  No include guard on purpose.
#define TRK_WAVES (TRK_THREADS / 64)
// LDS-resident state of one workgroup
struct Shared {
  double red[N_RED];                 // block-reduced sums of the last evaluation
  double wave_part[TRK_WAVES][N_RED + 2];
  double H[28], b[7];                // accepted normal equations
  double step[8];                    // LM step of the current proposal
  Se3 T, Tn;                         // m_T_cur_ref, new_T_cur_ref
  double energy_old, step_norm;
  float a, a_new;
  float huber, outlier, lambda;
  int level, PA, pad, S;
  int pi;                            // pattern index of the level
  int job, stop, n_select;
  int n_cand;                        // select_kth: keys appended to the candidate list
  int use_lds;
  hso_camera cam;                    // LDS copy of the camera for the out-of-line projection
  int keys_lds_off;                  // byte offset of the level's key array in LDS, 0 = keys in memory
  unsigned sel[4096];                // selection histogram (SEL_WORDS)
  int wave_cnt[TRK_WAVES];
  int poff[32];                      // byte offset oy*stride+ox of every pattern pixel of this level
  hso_track_result res;              // the job's result record, written to global memory once at the end
#if TRK_COOP
  // cooperative shape: this workgroup is one of coop_K that share ONE job (a slice of its features each)
  alignas(8) unsigned coop_in[COOP_KMAX][COOP_NG + 2];  // the peers' partial sums of the running exchange, as granule values
  int coop_fast;                     // all workgroups of the job sit on one XCD: exchange through its L2
  CoopJobState* coop_state;          // the job's exchange area in device memory
  int coop_K, coop_rank;
  unsigned coop_epoch;               // exchanges done (granule tag of the next one = coop_epoch + 1)
  int coop_region;                   // next unused histogram / list region
  unsigned coop_base;
  int coop_fail;                     // a peer did not answer within the spin bound (not resident): results are invalid
#endif
#ifdef HSO_PHASE_TIMERS
  unsigned long long dbg[8];
#endif
};

// the staged level image is addressed either in LDS (explicit address space 3, so the
// taps compile to ds_read2_b32) or in global memory (level too large for LDS)
typedef const __attribute__((address_space(3))) uint32_t* LdsPtr;
typedef const uint32_t* GlbPtr;

struct LevelCtx {
  const TrackConsts* C;
  const TrackJobDev* job;
  Scratch sc;
  GlbPtr cur_glb;          // current level image in global memory (aligned dwords)
  GlbPtr ref_glb;          // reference level image in global memory
  int cols, rows, level;
  float scale;
  double fxl, fyl;
  const __attribute__((address_space(3))) hso_camera* cam_lds;  // Shared::cam
};

// 4 consecutive bytes starting at byte address `addr` of a dword-aligned buffer
template <typename Ptr>
HSO_DEV uint32_t fetch4(Ptr w32, int addr)
{
  const int a = addr >> 2;
  const uint32_t lo = w32[a], hi = w32[a + 1];
  return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(addr & 3));
}
HSO_DEV float b0f(uint32_t v) { return (float)(v & 0xffu); }
HSO_DEV float b1f(uint32_t v) { return (float)((v >> 8) & 0xffu); }
HSO_DEV float b2f(uint32_t v) { return (float)((v >> 16) & 0xffu); }
HSO_DEV float b3f(uint32_t v) { return (float)(v >> 24); }

HSO_DEV double shfl_d(double v, int src)
{
  const int lo = __shfl(__double2loint(v), src), hi = __shfl(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
HSO_DEV double shfl_xor_d(double v, int m)
{
  const int lo = __shfl_xor(__double2loint(v), m), hi = __shfl_xor(__double2hiint(v), m);
  return __hiloint2double(hi, lo);
}

// FOV (atan) camera: rare, transcendental-heavy — kept out of line so that it does not
// inflate the register budget of the hot loops
__device__ __noinline__ void world2cam_fov(const hso_camera* cam, double x, double y, double z, double* pu, double* pv)
{
  world2cam(*cam, x, y, z, *pu, *pv);
}

// projection of one reference feature into the current level
// (CoarseTracker.cpp:290-323 / :557-583)
struct Proj {
  bool ok;
  int base;                      // byte address of pixel (u_i - 1, v_i) in the level image
  float w_tl, w_tr, w_bl, w_br;
  double x, y, z;
};

// one feature's record as it sits in memory; loaded a round ahead of its use so that the
// L2 latency of the five loads hides behind the previous feature's pixel loop
struct FeatRaw {
  double bx, by, bz, dist;
  int vis;
};

HSO_DEV FeatRaw load_feature(const LevelCtx& L, int f)
{
  FeatRaw r;
  r.vis = 0; r.bx = r.by = r.bz = 0; r.dist = -1;
  if (f < L.job->n) {
    const TrackJobDev& J = *L.job;
    const int ns = J.n_stride;
    // explicit global address space: a generic pointer would make these FLAT loads, which count on
    // lgkmcnt as well, so every LDS wait of the pixel loop would also wait for this prefetch
    typedef const __attribute__((address_space(1))) double* GlbF64;
    typedef const __attribute__((address_space(1))) uint8_t* GlbU8;
    const GlbF64 ft = (GlbF64)J.feats;
    r.vis = ((GlbU8)L.sc.visible)[f];
    r.dist = ft[5 * ns + f];
    r.bx = ft[2 * ns + f]; r.by = ft[3 * ns + f]; r.bz = ft[4 * ns + f];
  }
  return r;
}

typedef const __attribute__((address_space(3))) hso_camera* CamPtr;
// Inlined: as a call (the choice while the kernel had 168 registers) the result record came back through scratch memory
// plus 74 register moves per feature; the 256-register budget of two waves per SIMD has room for the fp64 temporaries
// (k_track 19.1 -> 18.0 ms on 4096 EuRoC pairs).  The camera is read from the workgroup's LDS copy (broadcast reads).
__device__ __forceinline__ Proj project_feature_nl(CamPtr camp, Se3 T, double bx, double by, double bz, double dist, int vis, int border,
                                                int cols, int rows, float scale)
{
  Proj p;
  p.ok = false;
  p.base = 0; p.w_tl = p.w_tr = p.w_bl = p.w_br = 0; p.x = p.y = p.z = 0;
  if (!vis) return p;
  if (dist < 0) return p;
  se3_apply(T, bx * dist, by * dist, bz * dist, p.x, p.y, p.z);
  if (p.z < 0) return p;
  double pu, pv;
  const int model = camp->model, distortion = camp->distortion;
  if (model == HSO_CAM_FOV && distortion) {
    hso_camera cam;
    cam.model = model; cam.distortion = distortion; cam.width = camp->width; cam.height = camp->height;
    cam.fx = camp->fx; cam.fy = camp->fy; cam.cx = camp->cx; cam.cy = camp->cy;
    for (int i = 0; i < 5; i++) cam.d[i] = camp->d[i];
    world2cam_fov(&cam, p.x, p.y, p.z, &pu, &pv);
  } else {
    // AbstractCamera::world2cam, src/camera.cpp:89-125 (pinhole, optional radtan)
    const double u = p.x / p.z, v = p.y / p.z;
    if (model == HSO_CAM_PINHOLE && distortion) {
      const double r2 = u * u + v * v;
      const double r4 = r2 * r2;
      const double r6 = r4 * r2;
      const double a1 = 2 * u * v;
      const double a2 = r2 + 2 * u * u;
      const double a3 = r2 + 2 * v * v;
      const double cdist = 1 + camp->d[0] * r2 + camp->d[1] * r4 + camp->d[4] * r6;
      const double xd = u * cdist + camp->d[2] * a1 + camp->d[3] * a2;
      const double yd = v * cdist + camp->d[2] * a3 + camp->d[3] * a1;
      pu = xd * camp->fx + camp->cx;
      pv = yd * camp->fy + camp->cy;
    } else {
      pu = camp->fx * u + camp->cx;
      pv = camp->fy * v + camp->cy;
    }
  }
  const float u_cur = (float)pu * scale;
  const float v_cur = (float)pv * scale;
  const int u_i = (int)floorf(u_cur);
  const int v_i = (int)floorf(v_cur);
  if (u_i - border < 0 || v_i - border < 0 || u_i + border >= cols || v_i + border >= rows) return p;
  const float su = u_cur - (float)u_i;
  const float sv = v_cur - (float)v_i;
  p.w_tl = (float)((1.0 - su) * (1.0 - sv));
  p.w_tr = (float)(su * (1.0 - sv));
  p.w_bl = (float)((1.0 - su) * sv);
  p.w_br = su * sv;
  p.base = v_i * cols + u_i - 1;
  p.ok = true;
  return p;
}

HSO_DEV Proj project_feature(const LevelCtx& L, const Se3& T, const FeatRaw& raw, int border)
{
  return project_feature_nl(L.cam_lds, T, raw.bx, raw.by, raw.bz, raw.dist, raw.vis, border, L.cols, L.rows, L.scale);
}

// ------------------------------------------------------- workgroup reductions

struct Acc {
  float H[32];   // [0..27] H, fp32 like the reference's Accumulator7 (MatrixAccumulator.h:33); [28] n_terms, [29] n_saturated
                 // (integers far below 2^24: exact in fp32); [30..31] pad
  double d[8];   // b[0..6] (fp64 like CoarseTracker.cpp:520), E — eight values: the halving exchange needs no pad
                 // (with the two counts as doubles it carried 6 zero values = 12 of its 32 dword exchanges)
};               // sizes padded to powers of two for the halving exchange (the pads stay 0)

// Sum N (a power of two <= 64) per-lane values over the 64 lanes of a wave by recursive
// halving: at each step (lane distance 32, 16, ...) a lane hands the half of its values it
// is not responsible for to its partner and adds the partner's contribution to the half it
// keeps, so N values cost ~N exchanges instead of 6N.  Afterwards the lane whose `slot` is
// k (< N) holds the total of value k.  Fixed order => deterministic floating point.
// Lane exchanges of the halving sum, cheapest form per distance (gfx950):
//   32, 16  v_permlane32_swap / v_permlane16_swap: the instruction swaps the upper half (odd rows) of one register with the
//           lower half (even rows) of another — exactly "hand over the half you do not keep": afterwards both registers
//           hold, in every lane, the two values that lane has to add.  No select, no LDS crossbar trip;
//   8, 2, 1 DPP (row_ror:8, quad_perm) on the value handed over;
//   4       two DPP moves (quad_perm [3,2,1,0], then row_half_mirror).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int M> HSO_DEV unsigned xor_lane_u32(unsigned v)
{
  if constexpr (M == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);       // row_ror:8
  else if constexpr (M == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);  // quad_perm:[2,3,0,1]
  else if constexpr (M == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);  // quad_perm:[1,0,3,2]
  else return lane_xor_u32<M>(v);   // 4: two DPP moves (hso_dev_math.h)
}
template <int M> HSO_DEV float xor_lane(float v) { return __uint_as_float(xor_lane_u32<M>(__float_as_uint(v))); }
template <int M> HSO_DEV double xor_lane(double v)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = xor_lane_u32<M>((unsigned)b), hi = xor_lane_u32<M>((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// lanes with (lane & M) == 0 get lo + lo', the others hi + hi' (primes: the partner lane's values)
template <int M> HSO_DEV float swap_sum(float lo, float hi)
{
  static_assert(M == 32 || M == 16, "swap distances");
  u32x2 r;
  if constexpr (M == 32) r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
  else r = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int M> HSO_DEV double swap_sum(double lo, double hi)
{
  const unsigned long long bl = (unsigned long long)__double_as_longlong(lo), bh = (unsigned long long)__double_as_longlong(hi);
  u32x2 r0, r1;
  if constexpr (M == 32) {
    r0 = __builtin_amdgcn_permlane32_swap((unsigned)bl, (unsigned)bh, false, false);
    r1 = __builtin_amdgcn_permlane32_swap((unsigned)(bl >> 32), (unsigned)(bh >> 32), false, false);
  } else {
    r0 = __builtin_amdgcn_permlane16_swap((unsigned)bl, (unsigned)bh, false, false);
    r1 = __builtin_amdgcn_permlane16_swap((unsigned)(bl >> 32), (unsigned)(bh >> 32), false, false);
  }
  const double a = __longlong_as_double((long long)(((unsigned long long)r1[0] << 32) | r0[0]));
  const double b = __longlong_as_double((long long)(((unsigned long long)r1[1] << 32) | r0[1]));
  return a + b;
}

template <typename T, int N, int M>
struct Halve {
  static HSO_DEV void run(T (&v)[N], int lane, int& slot, T& out)
  {
    static_assert((N & (N - 1)) == 0 && N >= 2, "N must be a power of two");
    constexpr int HALF = N / 2;
    const bool up = (lane & M) != 0;
    T keep[HALF];
#pragma unroll
    for (int i = 0; i < HALF; i++) {
      const T lo = v[i];
      const T hi = v[HALF + i];
      if constexpr (M == 32 || M == 16) {
        keep[i] = swap_sum<M>(lo, hi);
      } else {
        const T send = up ? lo : hi;
        keep[i] = (up ? hi : lo) + xor_lane<M>(send);
      }
    }
    if (up) slot += HALF;
    if constexpr (M == 1) {
      out = keep[0];  // HALF == 1 here
    } else {
      Halve<T, HALF, M / 2>::run(keep, lane, slot, out);
    }
  }
};
template <typename T, int M> HSO_DEV void butterfly_from(T& x)
{
  if constexpr (M == 32 || M == 16) x = swap_sum<M>(x, x);
  else x += xor_lane<M>(x);
  if constexpr (M > 1) butterfly_from<T, M / 2>(x);
}
template <typename T, int M>
struct Halve<T, 1, M> {
  static HSO_DEV void run(T (&v)[1], int lane, int& slot, T& out)
  {
    // a single value left before the lane distance reached 1: finish with plain butterflies
    T x = v[0];
    butterfly_from<T, M>(x);
    // every lane of the remaining group holds the total; only the group's first lane reports it
    if ((lane & (2 * M - 1)) != 0) slot = 1 << 20;
    out = x;
  }
};

HSO_DEV int block_sum_int(Shared& s, int v)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  __syncthreads();
  if (lane == 0) s.wave_cnt[wave] = v;
  __syncthreads();
  int tot = 0;
  for (int w = 0; w < TRK_WAVES; w++) tot += s.wave_cnt[w];
  return tot;
}


#if TRK_COOP
// ------------------------------------------------- exchange between the workgroups of one job (cooperative shape)
// K workgroups own feature slices of ONE (ref, cur) pair.  Everything that depends on all features is exchanged through the
// job's CoopJobState in device memory, and every workgroup then computes the same continuation from the same numbers in the
// same order (the LM step, the accept decision, the thresholds) — no broadcast, no leader, one exchange per evaluation.
// Visibility follows cdna_hip_programming.md section 6, Guideline 16: every shared word is an agent-scope access in the
// global address space; no result depends on workgroup placement or timing.
//   sums        8-byte {tag, value} granules, one sc1 store each; a reader re-reads a granule until its tag is the exchange
//               number (form R2: the data is the flag).  Two granule sets alternate, so a fast workgroup's next exchange
//               cannot overwrite what a slow one still reads.
//   histograms  device-scope atomic adds into a region that no exchange has touched before (zeroed by the launch's memset),
//               drained (vmcnt), then one arrival counter; read back with agent-scope loads.
typedef __attribute__((address_space(1))) unsigned long long* CoopG64;
typedef __attribute__((address_space(1))) unsigned* CoopG32;
#define COOP_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Transport primitives.  `fast` (workgroup-uniform): every workgroup of the job runs on the same XCD (verified at run time by
// coop_hello), so that XCD's L2 is their common point of coherence: a plain store lands there (the vector L1 is write-through)
// and an L2-scope atomic executes there; the agent-scope forms (sc1: write through to memory, line dropped from the L2) are
// what any other placement needs.  Loads are agent-scope relaxed in both cases (they bypass the CU's L1 and are L2-served).
HSO_DEV void coop_store64(CoopG64 p, unsigned long long v, bool fast)
{
  if (fast) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
  else __hip_atomic_store(p, v, COOP_RLX);
}
HSO_DEV void coop_store32(CoopG32 p, unsigned v, bool fast)
{
  if (fast) asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
  else __hip_atomic_store(p, v, COOP_RLX);
}
HSO_DEV void coop_add32(CoopG32 p, unsigned v, bool fast)
{
  if (fast) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_fetch_add(p, v, COOP_RLX);
}
HSO_DEV unsigned coop_fetch_add32(CoopG32 p, unsigned v, bool fast)
{
  if (fast) return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return __hip_atomic_fetch_add(p, v, COOP_RLX);
}
HSO_DEV unsigned long long coop_poll64(Shared& s, CoopG64 src, unsigned tag)
{
  unsigned long long x;
  for (unsigned spins = 0;;) {
    x = __hip_atomic_load(src, COOP_RLX);
    if ((unsigned)(x >> 32) == tag) break;
    if (++spins > COOP_SPIN_LIMIT) { s.coop_fail = 1; break; }
    __builtin_amdgcn_s_sleep(1);
  }
  return x;
}

// Placement census through the safe transport: every workgroup publishes the id of the XCD it runs on
// (s_getreg_b32 HW_REG_XCC_ID); the fast transport is chosen only when all K agree.
HSO_DEV void coop_hello(Shared& s)
{
  const int tid = threadIdx.x, K = s.coop_K;
  if (K == 1) return;   // a job on one workgroup: nothing to exchange (coop_fast stays 0)
  CoopJobState* const st = s.coop_state;
  const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u;   // HW_REG_XCC_ID[3:0]
  if (tid == 0) {
    s.coop_fast = 1;
    __hip_atomic_store((CoopG64)&st->hello[s.coop_rank], (1ull << 32) | xcc, COOP_RLX);
  }
  __syncthreads();
  if (tid < K) {
    const unsigned long long x = coop_poll64(s, (CoopG64)&st->hello[tid], 1u);
    if ((unsigned)x != xcc) s.coop_fast = 0;
  }
  __syncthreads();
#ifdef HSO_COOP_FORCE_SAFE
  if (tid == 0) s.coop_fast = 0;
  __syncthreads();
#endif
}

HSO_DEV void coop_allreduce(Shared& s)
{
#ifdef HSO_PHASE_TIMERS
  unsigned long long ct = __builtin_readcyclecounter();
#define COOP_T(k) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); s.dbg[k] += n_ - ct; ct = n_; } } while (0)
#else
#define COOP_T(k) do { } while (0)
#endif
  if (s.coop_K == 1) return;   // the sums are already the job's
  __syncthreads();   // s.red is complete; nobody reads coop_in of the previous exchange any more
  const int tid = threadIdx.x, K = s.coop_K;
  const unsigned e = s.coop_epoch + 1;
  const bool fast = s.coop_fast != 0;
  CoopJobState* const st = s.coop_state;
  if (tid < COOP_NG)
    coop_store64((CoopG64)&st->gran[e & 1][s.coop_rank][tid],
                 ((unsigned long long)e << 32) | (unsigned long long)reinterpret_cast<const unsigned*>(s.red)[tid], fast);
  COOP_T(5);
  // every thread polls its (<= NPT) granules together: all loads of a pass are in flight at once, so a pass costs one memory
  // round trip whatever K is
  constexpr int NPT = (COOP_KMAX * COOP_NG + TRK_THREADS - 1) / TRK_THREADS;
  const int total = K * COOP_NG;
  unsigned done = 0;
#pragma unroll
  for (int q = 0; q < NPT; q++) if (tid + q * TRK_THREADS >= total) done |= 1u << q;
  for (unsigned spins = 0; done != (1u << NPT) - 1u;) {
    unsigned long long x[NPT];
#pragma unroll
    for (int q = 0; q < NPT; q++) {
      const int g = tid + q * TRK_THREADS;
      x[q] = 0;
      if (!((done >> q) & 1u)) { const int r = g / COOP_NG, i = g - r * COOP_NG; x[q] = __hip_atomic_load((CoopG64)&st->gran[e & 1][r][i], COOP_RLX); }
    }
#pragma unroll
    for (int q = 0; q < NPT; q++) {
      if (!((done >> q) & 1u) && (unsigned)(x[q] >> 32) == e) {
        const int g = tid + q * TRK_THREADS;
        const int r = g / COOP_NG, i = g - r * COOP_NG;
        s.coop_in[r][i] = (unsigned)x[q];
        done |= 1u << q;
      }
    }
    if (done != (1u << NPT) - 1u) {
      if (++spins > COOP_SPIN_LIMIT) { s.coop_fail = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  COOP_T(6);
  if (tid < N_RED) {   // rank order, the same in every workgroup => identical bits everywhere
    double t = 0;
    for (int r = 0; r < K; r++) t += reinterpret_cast<const double*>(s.coop_in[r])[tid];
    s.red[tid] = t;
  }
  if (tid == 0) s.coop_epoch = e;
  __syncthreads();
  COOP_T(7);
}

// all K workgroups have passed this point of region R (their drained stores / atomics to R included)
HSO_DEV void coop_arrive_wait(Shared& s, CoopRegion* R, bool fast)
{
  if (threadIdx.x == 0) {
    const CoopG32 a = (CoopG32)&R->arrive;
    coop_add32(a, 1u, fast);
    for (unsigned spins = 0; __hip_atomic_load(a, COOP_RLX) < (unsigned)s.coop_K;) {
      if (++spins > COOP_SPIN_LIMIT) { s.coop_fail = 1; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

// s.sel[0, words) := sum over the workgroups of the job; *extra (a workgroup-uniform count) likewise
HSO_DEV void coop_merge(Shared& s, int words, int* extra)
{
  if (s.coop_K == 1) return;
  __syncthreads();
  const int tid = threadIdx.x;
  const bool fast = s.coop_fast != 0;
  CoopRegion* const R = &s.coop_state->region[s.coop_region];
  const CoopG32 w = (CoopG32)R->w;
  for (int i = tid; i < words; i += TRK_THREADS) {
    const unsigned c = s.sel[i];
    if (c) coop_add32(w + i, c, fast);
  }
  if (extra && tid == 0 && *extra) coop_add32((CoopG32)&R->extra, (unsigned)*extra, fast);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  coop_arrive_wait(s, R, fast);
  for (int i = tid; i < words; i += TRK_THREADS) s.sel[i] = __hip_atomic_load(w + i, COOP_RLX);
  if (extra) *extra = (int)__hip_atomic_load((CoopG32)&R->extra, COOP_RLX);
  if (tid == 0) s.coop_region++;
  __syncthreads();
}

// cand[0, s.n_cand) holds this workgroup's keys of the winning bin; afterwards cand[0, count) holds the job's (any order)
HSO_DEV void coop_gather(Shared& s, unsigned* cand, int count)
{
  if (s.coop_K == 1) return;
  __syncthreads();
  const int tid = threadIdx.x;
  const bool fast = s.coop_fast != 0;
  CoopRegion* const R = &s.coop_state->region[s.coop_region];
  const CoopG32 w = (CoopG32)R->w;
  const int n_local = s.n_cand;
  if (tid == 0) s.coop_base = coop_fetch_add32((CoopG32)&R->count, (unsigned)n_local, fast);
  __syncthreads();
  const unsigned base = s.coop_base;
  for (int i = tid; i < n_local; i += TRK_THREADS)
    if (base + (unsigned)i < COOP_REGION_WORDS) coop_store32(w + base + i, cand[i], fast);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  coop_arrive_wait(s, R, fast);
  for (int i = tid; i < count; i += TRK_THREADS) cand[i] = __hip_atomic_load(w + i, COOP_RLX);
  if (tid == 0) s.coop_region++;
  __syncthreads();
}
#endif  // TRK_COOP

// ------------------------------------------------------------- level set-up

// copy one level image (+ the zero row below it) into LDS; false if it does not fit.
// gfx950 LDS-DMA: each wavefront moves 1 KiB per instruction straight from memory into LDS
// (global_load_lds_dwordx4: per-lane source address, destination = uniform base + lane * 16),
// no staging registers and no ds_write pass; the loads of all chunks are in flight together
// and the barrier that follows drains them.
HSO_DEV bool stage_image(const LevelCtx& L, const uint8_t* src, uint32_t* lds_img)
{
  const int bytes = L.cols * L.rows;
  const int padded = (bytes + L.cols + 32 + 15) & ~15;
  if (padded > L.C->lds_img_cap) return false;
  typedef __attribute__((address_space(3))) uint8_t* LdsBytes;
  typedef const __attribute__((address_space(1))) uint8_t* GlbBytes;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = wave * 1024; c < padded; c += TRK_WAVES * 1024) {
    const int off = c + lane * 16;
    if (off < padded) __builtin_amdgcn_global_load_lds((GlbBytes)(src + off), (LdsBytes)lds_img + c, 16, 0, 0);
  }
  return true;
}

template <typename Ptr>
HSO_DEV void precompute_reference(const Shared& s, const LevelCtx& L, Ptr ref32)
{
  const TrackJobDev& J = *L.job;
  const int n = J.n, ns = J.n_stride, nm = L.C->n_max;
  const int PA = s.PA, border = s.pad + 1;
  const bool ic = L.C->inverse != 0;
  const int stride = L.cols;
  // one thread per feature (S lanes share a feature when the table is small): position and
  // bilinear weights once, then the pattern pixels
  const int S = s.S, G = TRK_THREADS / S;
  const int sub = (int)threadIdx.x % S, grp = (int)threadIdx.x / S;
  for (int f = grp; f < n; f += G) {
    const float u_ref = (float)(J.feats[0 * ns + f] * (double)L.scale);
    const float v_ref = (float)(J.feats[1 * ns + f] * (double)L.scale);
    const int u_i = (int)floorf(u_ref), v_i = (int)floorf(v_ref);
    const double dist = J.feats[5 * ns + f];
    const bool vis = dist >= 0 && !(u_i - border < 0 || v_i - border < 0 || u_i + border >= L.cols || v_i + border >= L.rows);
    if (sub == 0) L.sc.visible[f] = vis ? 1 : 0;
    if (!vis) continue;
    const float su = u_ref - (float)u_i, sv = v_ref - (float)v_i;
    const float w_tl = (float)((1.0 - su) * (1.0 - sv));
    const float w_tr = (float)(su * (1.0 - sv));
    const float w_bl = (float)((1.0 - su) * sv);
    const float w_br = (float)(1.0 - ((w_tl + w_tr) + w_bl));
    const int base = v_i * stride + u_i - 1;
    for (int pidx = sub; pidx < PA; pidx += S) {
      const int a0 = base + s.poff[pidx];
      const uint32_t o = (uint32_t)(pidx * nm + f);
      const uint32_t r1 = fetch4(ref32, a0), r2 = fetch4(ref32, a0 + stride);
      L.sc.ref_patch[o] = ((w_tl * b1f(r1) + w_tr * b2f(r1)) + w_bl * b1f(r2)) + w_br * b2f(r2);
      if (ic) {
        const uint32_t r0 = fetch4(ref32, a0 - stride), r3 = fetch4(ref32, a0 + 2 * stride);
        const float dx = 0.5f * ((((w_tl * b2f(r1) + w_tr * b3f(r1)) + w_bl * b2f(r2)) + w_br * b3f(r2))
                               - (((w_tl * b0f(r1) + w_tr * b1f(r1)) + w_bl * b0f(r2)) + w_br * b1f(r2)));
        const float dy = 0.5f * ((((w_tl * b1f(r2) + w_tr * b2f(r2)) + w_bl * b1f(r3)) + w_br * b2f(r3))
                               - (((w_tl * b1f(r0) + w_tr * b2f(r0)) + w_bl * b1f(r1)) + w_br * b2f(r1)));
        L.sc.ref_dx[o] = dx;
        L.sc.ref_dy[o] = dy;
      }
    }
  }
}

// ------------------------------------------------------ robust thresholds

#ifndef TRK_ROW_WINDOWS
#define TRK_ROW_WINDOWS 1
#endif
#ifndef TRK_PRECOMPUTE_ROWS
#define TRK_PRECOMPUTE_ROWS 1   // pattern-specialised reference patch precompute (forward mode, reference level in LDS)
#endif
#ifndef TRK_IC_ROWS
#define TRK_IC_ROWS 1       // pattern-specialised pixel loop for the inverse-compositional mode too (round 6)
#endif
#ifndef TRK_IC_PREFETCH
#define TRK_IC_PREFETCH 4
#endif
#ifndef TRK_GLOBAL_ROWS
#define TRK_GLOBAL_ROWS 1   // pattern-specialised loops also for images left in device memory (level 0 when relocalising)
#endif
template <int PI>
struct PatRows {
  static constexpr int N = h_pattern_num[PI];
  struct T { int idx[TRK_MAX_PA]; int min_ox, max_ox, min_oy, max_oy; };
  static constexpr T make()
  {
    T t{};
    t.min_ox = t.min_oy = 127; t.max_ox = t.max_oy = -127;
    for (int k = 0; k < N; k++) {
      t.idx[k] = k;
      const int ox = h_pattern[PI][k][0], oy = h_pattern[PI][k][1];
      if (ox < t.min_ox) t.min_ox = ox;
      if (ox > t.max_ox) t.max_ox = ox;
      if (oy < t.min_oy) t.min_oy = oy;
      if (oy > t.max_oy) t.max_oy = oy;
    }
    for (int i = 1; i < N; i++) {                      // stable insertion sort by oy
      const int v = t.idx[i];
      int j = i - 1;
      while (j >= 0 && h_pattern[PI][t.idx[j]][1] > h_pattern[PI][v][1]) { t.idx[j + 1] = t.idx[j]; j--; }
      t.idx[j + 1] = v;
    }
    return t;
  }
  static constexpr T v = make();
};

// Byte j of a row window as a float.  Written as the conversion instruction itself: from `(float)byte_a - (float)byte_b` the
// optimiser makes an integer subtract + int-to-float convert per DIFFERENCE (two instructions each, nothing shared), whereas
// a pixel converted once is shared by every term and every difference that touches it (~50 conversions per 13-pixel feature
// instead of ~210 subtract/convert pairs).  Not volatile: identical conversions are merged.
#ifndef TRK_CVT_ASM
#define TRK_CVT_ASM 1
#endif
HSO_DEV float win_byte(const uint32_t (&w)[3], int j)
{
#if TRK_CVT_ASM
  float f;
  const uint32_t v = w[j >> 2];
  switch (j & 3) {
    case 0: asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(v)); break;
    case 1: asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(v)); break;
    case 2: asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(v)); break;
    default: asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(f) : "v"(v)); break;
  }
  return f;
#else
  return (float)((w[j >> 2] >> (8 * (j & 3))) & 0xffu);
#endif
}

typedef const __attribute__((address_space(1))) float* GlbF32;

// precompute_reference for the forward mode with the pattern known at compile time and the reference level in LDS: the
// taps come from per-row windows (two rows live, as in collect_terms_rows below), tap and cache offsets are compile-time
// expressions, and the next feature's record is in flight while this one's patch is interpolated (the generic loop above
// waited for three dependent fp64 loads per feature).  Same expression order, so the cache is bit-identical.
template <int PI>
HSO_DEV void precompute_reference_rows(const LevelCtx& L, LdsPtr img, int border)
{
  constexpr auto P = PatRows<PI>::v;
  constexpr int PA = h_pattern_num[PI];
  constexpr int NB = P.max_ox - P.min_ox + 4;
  constexpr int NW = (NB + 3) / 4;
  constexpr int R0 = P.min_oy, R1 = P.max_oy + 1;
  typedef const __attribute__((address_space(1))) double* GlbF64;
  typedef __attribute__((address_space(1))) float* GlbF32W;
  typedef __attribute__((address_space(1))) uint8_t* GlbU8W;
  const int n = L.job->n, ns = L.job->n_stride;
  const uint32_t nm = (uint32_t)L.C->n_max;
  const GlbF64 ft = (GlbF64)L.job->feats;
  const GlbF32W rp = (GlbF32W)L.sc.ref_patch;
  const GlbU8W visible = (GlbU8W)L.sc.visible;
  const int stride = L.cols, cols = L.cols, rows = L.rows;
  const double scale = (double)L.scale;
  int f = (int)threadIdx.x;
  double nu = 0, nv = 0, nd = -1;
  if (f < n) { nu = ft[f]; nv = ft[ns + f]; nd = ft[5 * ns + f]; }
  for (; f < n; f += TRK_THREADS) {
    const double fu = nu, fv = nv, dist = nd;
    const int fn = f + TRK_THREADS;
    if (fn < n) { nu = ft[fn]; nv = ft[ns + fn]; nd = ft[5 * ns + fn]; }
    const float u_ref = (float)(fu * scale);
    const float v_ref = (float)(fv * scale);
    const int u_i = (int)floorf(u_ref), v_i = (int)floorf(v_ref);
    const bool vis = dist >= 0 && !(u_i - border < 0 || v_i - border < 0 || u_i + border >= cols || v_i + border >= rows);
    visible[f] = vis ? 1 : 0;
    if (!vis) continue;
    const float su = u_ref - (float)u_i, sv = v_ref - (float)v_i;
    const float w_tl = (float)((1.0 - su) * (1.0 - sv));
    const float w_tr = (float)(su * (1.0 - sv));
    const float w_bl = (float)((1.0 - su) * sv);
    const float w_br = (float)(1.0 - ((w_tl + w_tr) + w_bl));
    const int c0 = v_i * stride + u_i - 1 + P.min_ox;
    uint32_t win[2][3];
#pragma unroll
    for (int R = R0; R <= R1; R++) {
      {
        const int addr = c0 + R * stride;
        const int A = addr >> 2;
        const uint32_t sh = (uint32_t)(addr & 3);
        uint32_t (&w)[3] = win[(R - R0) & 1];
        const uint32_t d0 = img[A], d1 = img[A + 1], d2 = img[A + 2];
        w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
        w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
        if (NW == 3) { const uint32_t d3 = img[A + 3]; w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh); } else w[2] = 0;
      }
      const int oy = R - 1;
#pragma unroll
      for (int q = 0; q < PA; q++) {
        if (h_pattern[PI][P.idx[q]][1] != oy) continue;
        const int kk = P.idx[q];
        const int j = h_pattern[PI][kk][0] - P.min_ox;
        const uint32_t (&w1)[3] = win[(oy - R0) & 1];
        const uint32_t (&w2)[3] = win[(oy + 1 - R0) & 1];
        const float p11 = win_byte(w1, j + 1), p12 = win_byte(w1, j + 2), p21 = win_byte(w2, j + 1), p22 = win_byte(w2, j + 2);
        rp[(uint32_t)kk * nm + (uint32_t)f] = ((w_tl * p11 + w_tr * p12) + w_bl * p21) + w_br * p22;
      }
    }
  }
}

// pass 1 of selectRobustFunctionLevel (CoarseTracker.cpp:547-606): |residual| of every
// in-bounds term, stored as float bit patterns (KEY_INVALID elsewhere).  Returns errors.size().
HSO_DEV void sel_count_a(Shared& s, uint32_t kk);

// The keys of one feature with the pattern known at compile time: the taps of the bilinear intensity come from per-row
// windows (see feature_terms_rows below: two rows live here, no gradient), the reference intensities are requested up front.
typedef const __attribute__((address_space(1))) uint32_t* GlbW32;   // a level image in device memory, dword view
template <int PI, typename KP, typename IP>
__device__ __forceinline__ void collect_terms_rows(Shared& s, IP img, GlbF32 ref_patch, KP kdst, int n, int f, int base,
                                                   float w_tl, float w_tr, float w_bl, float w_br, uint32_t fb, uint32_t nb,
                                                   int stride, float a)
{
  constexpr auto P = PatRows<PI>::v;
  constexpr int PA = h_pattern_num[PI];
  constexpr int NB = P.max_ox - P.min_ox + 4;
  constexpr int NW = (NB + 3) / 4;
  constexpr int R0 = P.min_oy, R1 = P.max_oy + 1;
  stride = __builtin_amdgcn_readfirstlane(stride);
  nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
  typedef const __attribute__((address_space(1))) char* GlbBytes;
  const GlbBytes rpb = (GlbBytes)ref_patch;
  float iref[PA];
#pragma unroll
  for (int k = 0; k < PA; k++) iref[k] = *(GlbF32)(rpb + (fb + (uint32_t)k * nb));
  uint32_t win[2][3];
  const int c0 = base + P.min_ox;
#pragma unroll
  for (int R = R0; R <= R1; R++) {
    {
      const int addr = c0 + R * stride;
      const int A = addr >> 2;
      const uint32_t sh = (uint32_t)(addr & 3);
      uint32_t (&w)[3] = win[(R - R0) & 1];
      const uint32_t d0 = img[A], d1 = img[A + 1], d2 = img[A + 2];
      w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
      w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
      if (NW == 3) { const uint32_t d3 = img[A + 3]; w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh); } else w[2] = 0;
    }
    const int oy = R - 1;                              // the terms whose two rows are now complete
#pragma unroll
    for (int q = 0; q < PA; q++) {
      if (h_pattern[PI][P.idx[q]][1] != oy) continue;
      const int kk = P.idx[q];
      const int j = h_pattern[PI][kk][0] - P.min_ox;
      const uint32_t (&w1)[3] = win[(oy - R0) & 1];
      const uint32_t (&w2)[3] = win[(oy + 1 - R0) & 1];
      const float p11 = win_byte(w1, j + 1), p12 = win_byte(w1, j + 2), p21 = win_byte(w2, j + 1), p22 = win_byte(w2, j + 2);
      const float cur = ((w_tl * p11 + w_tr * p12) + w_bl * p21) + w_br * p22;   // CoarseTracker.cpp:339-348 order
      const float res = cur - a * iref[kk];
      const uint32_t key = __float_as_uint(fabsf(res));
      sel_count_a(s, key);
      kdst[kk * n + f] = key;
    }
  }
}

template <bool S1, typename Ptr, typename KP, int PI = -1>
HSO_DEV int select_collect(Shared& s, const LevelCtx& L, Ptr img, const Se3& T, float a, KP kdst)
{
  const int n = L.job->n, nm = L.C->n_max;
  const int PA = s.PA, border = s.pad + 1, S = S1 ? 1 : s.S;
  const int G = TRK_THREADS / S;
  const int sub = S1 ? 0 : (int)(threadIdx.x % S);
  const int stride = L.cols;
  int cnt = 0;
  const int grp = S1 ? (int)threadIdx.x : (int)(threadIdx.x / S);
  FeatRaw nxt = load_feature(L, grp);
  for (int base = 0; base < n; base += G) {
    const int f = base + grp;
    const FeatRaw raw = nxt;
    nxt = load_feature(L, f + G);
    if (f >= n) continue;
    const Proj p = project_feature(L, T, raw, border);
    if constexpr (PI >= 0) {
      if (p.ok) {
        collect_terms_rows<PI, KP>(s, img, (GlbF32)L.sc.ref_patch, kdst, n, f, p.base, p.w_tl, p.w_tr, p.w_bl, p.w_br, (uint32_t)f * 4u,
                               (uint32_t)nm * 4u, stride, a);
        cnt += PA;
      } else {
        for (int pidx = 0; pidx < PA; pidx++) kdst[pidx * n + f] = KEY_INVALID;
      }
      continue;
    }
    for (int pidx = sub; pidx < PA; pidx += S) {
      uint32_t key = KEY_INVALID;
      if (p.ok) {
        const int a0 = p.base + s.poff[pidx];
        const uint32_t r1 = fetch4(img, a0), r2 = fetch4(img, a0 + stride);
        const float cur = ((p.w_tl * b1f(r1) + p.w_tr * b2f(r1)) + p.w_bl * b1f(r2)) + p.w_br * b2f(r2);
        const float res = cur - a * L.sc.ref_patch[(size_t)pidx * nm + f];
        key = __float_as_uint(fabsf(res));
        sel_count_a(s, key);  // round A of the median select, fused (select_robust zeroed the bins)
        cnt++;
      }
      kdst[pidx * n + f] = key;
    }
  }
  return block_sum_int(s, cnt);
}

// ---- exact order statistics -------------------------------------------------------------
// Keys are bit patterns of non-negative floats (bit 31 clear), so unsigned order = float order;
// KEY_INVALID (bit 31 set) marks slots without a term.  The k-th smallest is found MSB-first:
//   round A  histogram of the digits [30:19] (exponent + 4 mantissa bits, 4096 bins) over all keys — for the median it is
//            fused into the pass that produces the keys.  Residual magnitudes crowd into a handful of octaves; the four
//            mantissa bits spread them over ~80 bins, so the LDS atomics rarely collide (a histogram of the exponent alone
//            needed 16 replicas per bin) and the bin that holds the rank is left with a few percent of the keys;
//   compact  one more pass over all keys copies the keys of that bin into LDS (wave-aggregated append);
//   rounds B, C  digits [18:8] and [7:0] over the compacted keys only.
// So the median costs one pass over the keys beyond the one that wrote them, the MAD two — instead of two and three passes
// with three full-size scans each.  If the bin holds more keys than the LDS list (SEL_CAND_CAP; degenerate inputs such as
// constant residuals), rounds B and C run over all keys instead.  After each round every wave locates the bin that holds the
// rank by itself (scan_find), so (prefix, rank) live in registers and are identical in all threads by construction.  The
// value returned is the element nth_element would leave at position k, whatever the input order.
#define SEL_WORDS 4096
#define SEL_A_SHIFT 19
#define SEL_CAND_CAP 2048   // candidates live in sel[0, 2048), the round B / C histograms in sel[2048, 4096)

template <int NB>
HSO_DEV void scan_find(const unsigned* hist, unsigned& rank, unsigned& bin, unsigned& count)
{
  constexpr int SEG = NB / 64;
  int lane = threadIdx.x & 63;
  // opaque to the optimiser: without this the 8 inlined copies share hoisted LDS addresses, which
  // then get spilled and are re-read from scratch inside the loops below
  asm volatile("" : "+v"(lane));
  unsigned local = 0;
  for (int j = 0; j < SEG; j++) local += hist[lane * SEG + ((j + lane) & (SEG - 1))];  // rotated start: spreads the banks
  unsigned incl = local;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(incl, d);
    if (lane >= d) incl += o;
  }
  const unsigned excl = incl - local;
  const unsigned long long m = __ballot(rank >= excl && rank < incl);
  const int seg = (m != 0ull) ? (__ffsll((long long)m) - 1) : 0;
  const unsigned r2 = rank - __shfl(excl, seg);
  const unsigned c = (lane < SEG) ? hist[seg * SEG + lane] : 0u;
  unsigned incl2 = c;
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = __shfl_up(incl2, d);
    if (lane >= d) incl2 += o;
  }
  const unsigned long long m2 = __ballot(lane < SEG && r2 >= incl2 - c && r2 < incl2);
  const int bl = (m2 != 0ull) ? (__ffsll((long long)m2) - 1) : 0;
  bin = (unsigned)(seg * SEG + bl);
  rank = r2 - __shfl(incl2 - c, bl);
  count = __shfl(c, bl);
}

HSO_DEV void sel_zero(Shared& s, int words)
{
  __syncthreads();
  for (int i = threadIdx.x; i < words; i += TRK_THREADS) s.sel[i] = 0;
  __syncthreads();
}

HSO_DEV void sel_count_a(Shared& s, uint32_t kk)
{
  if ((int)kk >= 0) atomicAdd(&s.sel[kk >> SEL_A_SHIFT], 1u);
}

// keys.each(f) calls f(key) for every key this thread owns.  round_a_done: the caller already
// histogrammed the leading digit (fused into the pass that produced the keys).
template <typename Keys>
HSO_DEV uint32_t select_kth(Shared& s, unsigned k, const Keys& keys, bool round_a_done)
{
#ifdef HSO_SEL_PROBE
  unsigned long long kt = __builtin_readcyclecounter();
#define KSEL_T(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); s.dbg[i] += n_ - kt; kt = n_; } } while (0)
#else
#define KSEL_T(i) do { } while (0)
#endif
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));  // see scan_find
  unsigned rank = k, bin = 0, count = 0;
  if (!round_a_done) {
    sel_zero(s, SEL_WORDS);
    keys.each([&](uint32_t kk) { sel_count_a(s, kk); });
#if TRK_COOP
    coop_merge(s, SEL_WORDS, nullptr);
#endif
  }
  __syncthreads();
  KSEL_T(0);
  scan_find<4096>(s.sel, rank, bin, count);
  uint32_t prefix = bin << SEL_A_SHIFT;
  __syncthreads();   // every wave has read the histogram: its words are free
  KSEL_T(1);
  if (count <= SEL_CAND_CAP) {
    unsigned* const cand = s.sel;
    unsigned* const hist = s.sel + SEL_CAND_CAP;
    if (tid == 0) s.n_cand = 0;
    for (int i = tid; i < 2048; i += TRK_THREADS) hist[i] = 0;
    __syncthreads();
    keys.compact(bin, cand, &s.n_cand);
    __syncthreads();
#if TRK_COOP
    coop_gather(s, cand, (int)count);
#endif
    KSEL_T(2);
    const int nc = (int)count;
    for (int i = tid; i < nc; i += TRK_THREADS) atomicAdd(&hist[(cand[i] >> 8) & 2047u], 1u);
    __syncthreads();
    unsigned bin_b = 0, cnt_b = 0;
    scan_find<2048>(hist, rank, bin_b, cnt_b);
    prefix |= bin_b << 8;
    __syncthreads();
    for (int i = tid; i < 256; i += TRK_THREADS) hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < nc; i += TRK_THREADS) { const uint32_t kk = cand[i]; if ((kk & 0xFFFFFF00u) == prefix) atomicAdd(&hist[kk & 255u], 1u); }
    __syncthreads();
    unsigned bin_c = 0, cnt_c = 0;
    scan_find<256>(hist, rank, bin_c, cnt_c);
    prefix |= bin_c;
    __syncthreads();
    KSEL_T(3);
    return prefix;
  }
  // the bin is too full for the LDS list: digits [18:8] and [7:0] over all keys
  sel_zero(s, 2048);
  keys.each([&](uint32_t kk) { if ((kk >> SEL_A_SHIFT) == bin) atomicAdd(&s.sel[(kk >> 8) & 2047u], 1u); });
#if TRK_COOP
  coop_merge(s, 2048, nullptr);
#endif
  __syncthreads();
  scan_find<2048>(s.sel, rank, bin, count);
  prefix |= bin << 8;
  sel_zero(s, 256);
  keys.each([&](uint32_t kk) { if ((kk & 0xFFFFFF00u) == prefix) atomicAdd(&s.sel[kk & 255u], 1u); });
#if TRK_COOP
  coop_merge(s, 256, nullptr);
#endif
  __syncthreads();
  scan_find<256>(s.sel, rank, bin, count);
  prefix |= bin;
  __syncthreads();
  KSEL_T(4);
  return prefix;
}

// keys in memory or LDS (the |residual| array of select_collect), optionally transformed on the
// fly; read as 16-byte vectors, four per thread in flight, so a pass is a handful of wide loads
// instead of dozens of dependent dword loads
typedef __attribute__((address_space(3))) uint32_t* LdsKeys;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <typename KP> struct Vec4Ptr;
template <> struct Vec4Ptr<uint32_t*> { typedef const u32x4* type; };
template <> struct Vec4Ptr<LdsKeys> { typedef const __attribute__((address_space(3))) u32x4* type; };

#define SEL_VPT 4   // 16-byte key vectors a thread has in flight per step (8 costs more in spills around the passes than it hides in load latency)
template <typename KP, typename Xf>
struct MemKeys {
  KP keys;
  int n_slots;
  Xf xf;
  template <typename F>
  HSO_DEV void each(F f) const
  {
    typedef typename Vec4Ptr<KP>::type V4;
    const V4 kv = (V4)keys;
    const int nvec = n_slots >> 2;
    for (int v0 = threadIdx.x; v0 < nvec; v0 += TRK_THREADS * SEL_VPT) {
      u32x4 q[SEL_VPT];
#pragma unroll
      for (int u = 0; u < SEL_VPT; u++) {
        const int v = v0 + u * TRK_THREADS;
        if (v < nvec) q[u] = kv[v];
        else q[u] = (u32x4)(KEY_INVALID);
      }
#pragma unroll
      for (int u = 0; u < SEL_VPT; u++) { f(xf(q[u].x)); f(xf(q[u].y)); f(xf(q[u].z)); f(xf(q[u].w)); }
    }
    if ((int)threadIdx.x < (n_slots & 3)) f(xf(keys[(nvec << 2) + (int)threadIdx.x]));
  }
  // Append every key whose leading digit is `bin` to cand[] (order irrelevant).  4 * SEL_VPT (+1) keys per thread and step:
  // a thread counts its matches, one wave-wide prefix sum and ONE LDS atomic per wave and step reserve the slots (a ballot +
  // atomic per key made this pass twice as slow as the histogram pass it replaces).
  HSO_DEV void compact(unsigned bin, unsigned* cand, int* n_cand) const
  {
    typedef typename Vec4Ptr<KP>::type V4;
    const V4 kv = (V4)keys;
    const int nvec = n_slots >> 2;
    const int lane = threadIdx.x & 63;
    const int tail = n_slots & 3;
    constexpr int NK = 4 * SEL_VPT + 1;
    const int n_steps = max(1, (nvec + TRK_THREADS * SEL_VPT - 1) / (TRK_THREADS * SEL_VPT));
    for (int step = 0; step < n_steps; step++) {   // the same trip count in every lane: the prefix sum needs the whole wave
      const int v0 = (int)threadIdx.x + step * TRK_THREADS * SEL_VPT;
      uint32_t k[NK];
#pragma unroll
      for (int u = 0; u < SEL_VPT; u++) {
        const int v = v0 + u * TRK_THREADS;
        u32x4 q;
        if (v < nvec) q = kv[v];
        else q = (u32x4)(KEY_INVALID);
        k[4 * u + 0] = xf(q.x); k[4 * u + 1] = xf(q.y); k[4 * u + 2] = xf(q.z); k[4 * u + 3] = xf(q.w);
      }
      // the (< 4) keys behind the last full vector: one each for the first threads, in the first step
      k[NK - 1] = (step == 0 && (int)threadIdx.x < tail) ? xf(keys[(nvec << 2) + (int)threadIdx.x]) : KEY_INVALID;
      unsigned long long mbits = 0;
#pragma unroll
      for (int i = 0; i < NK; i++) mbits |= (unsigned long long)((k[i] >> SEL_A_SHIFT) == bin ? 1u : 0u) << i;
      const int c = (int)__popcll(mbits);
      int incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
      const int total = __shfl(incl, 63);
      if (total == 0) continue;
      int base = 0;
      if (lane == 63) base = atomicAdd(n_cand, total);
      base = __shfl(base, 63) + incl - c;
#pragma unroll
      for (int i = 0; i < NK; i++)
        if ((mbits >> i) & 1ull) cand[base + (int)__popcll(mbits & ((1ull << i) - 1ull))] = k[i];
    }
  }
};
template <typename KP, typename Xf>
HSO_DEV MemKeys<KP, Xf> mem_keys(KP keys, int n_slots, Xf xf) { return MemKeys<KP, Xf>{ keys, n_slots, xf }; }

HSO_DEV void set_thresholds(Shared& s, float med, uint32_t mad_bits)
{
  if (threadIdx.x == 0) {
    const float standard_deviation = (float)(1.4826 * (double)__uint_as_float(mad_bits));
    const float huber = med + standard_deviation;
    float outlier = 3 * huber;
    if (outlier < 10) outlier = 10;
    s.huber = huber;
    s.outlier = outlier;
  }
  __syncthreads();
}

// selectRobustFunctionLevel, CoarseTracker.cpp:530-644
template <typename KP>
HSO_DEV void select_robust_k(Shared& s, const LevelCtx& L, LdsPtr lds_img, const Se3& T, float a, KP keys)
{
#ifdef HSO_SEL_PROBE
  unsigned long long sel_t = __builtin_readcyclecounter();
#define SELR_T(k) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); s.dbg[k] += n_ - sel_t; sel_t = n_; } } while (0)
#else
#define SELR_T(k) do { } while (0)
#endif
  sel_zero(s, SEL_WORDS);
  int n_err;
  if (s.S == 1) {
    if (s.use_lds) {
      switch (s.pi) {  // pattern-specialised taps for the patterns levels 4..1 use
        case 2: n_err = select_collect<true, LdsPtr, KP, 2>(s, L, lds_img, T, a, keys); break;
        case 3: n_err = select_collect<true, LdsPtr, KP, 3>(s, L, lds_img, T, a, keys); break;
        case 4: n_err = select_collect<true, LdsPtr, KP, 4>(s, L, lds_img, T, a, keys); break;
        case 5: n_err = select_collect<true, LdsPtr, KP, 5>(s, L, lds_img, T, a, keys); break;
        case 6: n_err = select_collect<true, LdsPtr, KP, 6>(s, L, lds_img, T, a, keys); break;
        default: n_err = select_collect<true, LdsPtr>(s, L, lds_img, T, a, keys); break;
      }
    } else {
#if TRK_GLOBAL_ROWS
      switch (s.pi) {
        case 2: n_err = select_collect<true, GlbW32, KP, 2>(s, L, (GlbW32)L.cur_glb, T, a, keys); break;
        case 3: n_err = select_collect<true, GlbW32, KP, 3>(s, L, (GlbW32)L.cur_glb, T, a, keys); break;
        case 4: n_err = select_collect<true, GlbW32, KP, 4>(s, L, (GlbW32)L.cur_glb, T, a, keys); break;
        case 5: n_err = select_collect<true, GlbW32, KP, 5>(s, L, (GlbW32)L.cur_glb, T, a, keys); break;
        case 6: n_err = select_collect<true, GlbW32, KP, 6>(s, L, (GlbW32)L.cur_glb, T, a, keys); break;
        default: n_err = select_collect<true, GlbPtr>(s, L, L.cur_glb, T, a, keys); break;
      }
#else
      n_err = select_collect<true, GlbPtr>(s, L, L.cur_glb, T, a, keys);
#endif
    }
  } else {
    n_err = s.use_lds ? select_collect<false, LdsPtr>(s, L, lds_img, T, a, keys) : select_collect<false, GlbPtr>(s, L, L.cur_glb, T, a, keys);
  }
#if TRK_COOP
  coop_merge(s, SEL_WORDS, &n_err);   // the job's leading-digit histogram and errors.size()
#endif
  const int n_slots = L.job->n * s.PA;
  if (threadIdx.x == 0) s.n_select = n_err;
  SELR_T(5);
  if (n_err < 30) {
    if (threadIdx.x == 0) { s.huber = 5.2f; s.outlier = 100.f; }
    __syncthreads();
    return;
  }
  const uint32_t med_bits = select_kth(s, (unsigned)(n_err / 2), mem_keys(keys, n_slots, [](uint32_t k) { return k; }), true);
  const float med = __uint_as_float(med_bits);
  SELR_T(6);
  const uint32_t mad_bits = select_kth(s, (unsigned)(n_err / 2), mem_keys(keys, n_slots, [med](uint32_t k) {
    return (k == KEY_INVALID) ? KEY_INVALID : __float_as_uint(fabsf(__uint_as_float(k) - med));
  }), false);
  set_thresholds(s, med, mad_bits);
  SELR_T(7);
}

// The |residual| keys of a level live in LDS above the staged image when they fit (levels 4..2 of
// a 2000-feature VGA frame: <= 104 KB), else in the workgroup's scratch in memory.
HSO_DEV void select_robust(Shared& s, const LevelCtx& L, LdsPtr lds_img, const Se3& T, float a)
{
  if (s.keys_lds_off > 0) {
    LdsKeys kl = (LdsKeys)lds_img + (s.keys_lds_off >> 2);
    select_robust_k<LdsKeys>(s, L, lds_img, T, a, kl);
  } else {
    select_robust_k<uint32_t*>(s, L, lds_img, T, a, L.sc.keys);
  }
}

// ------------------------------------------ residuals + normal equations

// Weighted moments of one feature's pattern pixels (this lane's share of them).
struct Moments {
  float ee, ex, ey, xx, xy, yy, re, rx, ry;  // sum w*{e e, e dx, e dy, dx dx, dx dy, dy dy, r e, r dx, r dy}
  float E;
  int nt, nsat;
};

// One term's raw inputs, fetched ahead of the arithmetic that consumes them.
struct TermIn {
  uint32_t r0, r1, r2, r3;  // rows y-1 .. y+2, bytes x-1 .. x+2
  float iref, dxr, dyr;
};

template <bool IC, typename Ptr>
HSO_DEV TermIn load_term(const LevelCtx& L, const Shared& s, Ptr img, const float* rp, int base, int pidx, int nm, int f)
{
  TermIn t;
  const int a0 = base + s.poff[pidx];
  t.r1 = fetch4(img, a0);
  t.r2 = fetch4(img, a0 + L.cols);
  if (!IC) {
    t.r0 = fetch4(img, a0 - L.cols);
    t.r3 = fetch4(img, a0 + 2 * L.cols);
    t.dxr = t.dyr = 0;
  } else {
    t.r0 = t.r3 = 0;
    t.dxr = L.sc.ref_dx[(uint32_t)(pidx * nm + f)];
    t.dyr = L.sc.ref_dy[(uint32_t)(pidx * nm + f)];
  }
  t.iref = rp[(uint32_t)(pidx * nm + f)];  // 32-bit offset from a uniform base: saddr + voffset addressing
  return t;
}

// The per-term arithmetic of computeResiduals (CoarseTracker.cpp:328-410) for the pattern
// pixels sub, sub+S, ... of feature f.
template <bool IC, typename Ptr>
HSO_DEV Moments feature_terms(const Shared& s, const LevelCtx& L, Ptr img, const Proj& p, int f, float a,
                              int sub, int S, int PA, bool top, float huber, float outlier, float max_energy)
{
  Moments m;
  m.ee = m.ex = m.ey = m.xx = m.xy = m.yy = m.re = m.rx = m.ry = 0; m.E = 0; m.nt = 0; m.nsat = 0;
  if (!p.ok) return m;
  const int nm = L.C->n_max;
  const float* rp = L.sc.ref_patch;  // uniform base
  int pidx = sub;
  if (pidx >= PA) return m;
  TermIn nx = load_term<IC>(L, s, img, rp, p.base, pidx, nm, f);
  for (; pidx < PA; pidx += S) {
    const TermIn t = nx;
    if (pidx + S < PA) nx = load_term<IC>(L, s, img, rp, p.base, pidx + S, nm, f);  // next term in flight
    const float p11 = b1f(t.r1), p12 = b2f(t.r1), p21 = b1f(t.r2), p22 = b2f(t.r2);
    // decision arithmetic: exactly the reference's expression order (:339-348)
    const float cur = ((p.w_tl * p11 + p.w_tr * p12) + p.w_bl * p21) + p.w_br * p22;
    const float res = cur - a * t.iref;
    const float ares = fabsf(res);
    // The Huber weight only enters tolerance-compared sums (E, H, b), never a decision: one
    // v_rcp_f32 (1 ulp) and a multiply instead of the ten-instruction IEEE division
    const float hw = ares < huber ? 1.0f : huber * __builtin_amdgcn_rcpf(ares);
    // cutoff_error = m_outlier_thresh is a float value, so the reference's double compare (:350)
    // equals this float compare
    const bool sat = (ares > outlier) && !top;
    const float e_term = top ? (hw * res) * res : ((hw * res) * res) * (2 - hw);
    m.E += sat ? max_energy : e_term;
    m.nt++;
    m.nsat += sat ? 1 : 0;
    // image gradient; forward mode: twice the central difference (the 1/2 is folded into A, B)
    float dx, dy;
    if (!IC) {
      dx = fmaf(p.w_tl, p12 - b0f(t.r1), fmaf(p.w_tr, b3f(t.r1) - p11, fmaf(p.w_bl, p22 - b0f(t.r2), p.w_br * (b3f(t.r2) - p21))));
      dy = fmaf(p.w_tl, p21 - b1f(t.r0), fmaf(p.w_tr, p22 - b2f(t.r0), fmaf(p.w_bl, b1f(t.r3) - p11, p.w_br * (b2f(t.r3) - p12))));
    } else {
      dx = t.dxr; dy = t.dyr;
    }
    // saturated terms contribute no Jacobian row (:350-355): weight 0
    const float w = sat ? 0.0f : hw;
    const float e = -t.iref;
    const float we = w * e, wx = w * dx, wy = w * dy, wr = w * res;
    m.ee = fmaf(we, e, m.ee); m.ex = fmaf(we, dx, m.ex); m.ey = fmaf(we, dy, m.ey);
    m.xx = fmaf(wx, dx, m.xx); m.xy = fmaf(wx, dy, m.xy); m.yy = fmaf(wy, dy, m.yy);
    m.re = fmaf(wr, e, m.re); m.rx = fmaf(wr, dx, m.rx); m.ry = fmaf(wr, dy, m.ry);
  }
  return m;
}

// The same arithmetic (forward mode, image in LDS) with the pattern known at compile time
// (PI = index into the static pattern tables), fully unrolled: tap offsets oy*stride+ox and
// patch-cache offsets k*n_max are scalar expressions, there is no per-term table read and no
// loop-carried prefetch record — about a quarter fewer instructions per term in a loop that is
// VALU-issue-bound.  Deliberately NOT inlined: inside the megakernel the unrolled body competes
// with the state of every other phase for the 168 VGPRs and spills; as a separate function it gets
// its own register allocation, at the price of one call per feature.
template <int PI>
__device__ __forceinline__ Moments feature_terms_static(LdsPtr img, GlbF32 ref_patch, int base, float w_tl, float w_tr, float w_bl,
                                                      float w_br, uint32_t fb, uint32_t nb, int stride, float a, float huber,
                                                      float outlier, float max_energy, int top)
{
  constexpr int PA = h_pattern_num[PI];
  Moments m;
  m.ee = m.ex = m.ey = m.xx = m.xy = m.yy = m.re = m.rx = m.ry = 0; m.E = 0; m.nt = PA; m.nsat = 0;
  stride = __builtin_amdgcn_readfirstlane(stride);
  nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
  typedef const __attribute__((address_space(1))) char* GlbBytes;
  const GlbBytes rpb = (GlbBytes)ref_patch;
  constexpr int PF = 4;  // reference intensities requested PF terms ahead of their use
  float ipf[PF];
#pragma unroll
  for (int k = 0; k < PF && k < PA; k++) ipf[k] = *(GlbF32)(rpb + (fb + (uint32_t)k * nb));
#pragma unroll
  for (int k = 0; k < PA; k++) {
    const int ox = h_pattern[PI][k][0], oy = h_pattern[PI][k][1];
    const int a0 = base + (oy * stride + ox);
    const uint32_t r1 = fetch4(img, a0), r2 = fetch4(img, a0 + stride);
    const uint32_t r0 = fetch4(img, a0 - stride), r3 = fetch4(img, a0 + 2 * stride);
    const float iref = ipf[k % PF];
    if (k + PF < PA) ipf[k % PF] = *(GlbF32)(rpb + (fb + (uint32_t)(k + PF) * nb));
    const float p11 = b1f(r1), p12 = b2f(r1), p21 = b1f(r2), p22 = b2f(r2);
    const float cur = ((w_tl * p11 + w_tr * p12) + w_bl * p21) + w_br * p22;
    const float res = cur - a * iref;
    const float ares = fabsf(res);
    const float hw = ares < huber ? 1.0f : huber * __builtin_amdgcn_rcpf(ares);
    const bool sat = (ares > outlier) && !top;
    const float e_term = top ? (hw * res) * res : ((hw * res) * res) * (2 - hw);
    m.E += sat ? max_energy : e_term;
    m.nsat += sat ? 1 : 0;
    const float dx = fmaf(w_tl, p12 - b0f(r1), fmaf(w_tr, b3f(r1) - p11, fmaf(w_bl, p22 - b0f(r2), w_br * (b3f(r2) - p21))));
    const float dy = fmaf(w_tl, p21 - b1f(r0), fmaf(w_tr, p22 - b2f(r0), fmaf(w_bl, b1f(r3) - p11, w_br * (b2f(r3) - p12))));
    const float w = sat ? 0.0f : hw;
    const float e = -iref;
    const float we = w * e, wx = w * dx, wy = w * dy, wr = w * res;
    m.ee = fmaf(we, e, m.ee); m.ex = fmaf(we, dx, m.ex); m.ey = fmaf(we, dy, m.ey);
    m.xx = fmaf(wx, dx, m.xx); m.xy = fmaf(wx, dy, m.xy); m.yy = fmaf(wy, dy, m.yy);
    m.re = fmaf(wr, e, m.re); m.rx = fmaf(wr, dx, m.rx); m.ry = fmaf(wr, dy, m.ry);
  }
  return m;
}

// The same per-term arithmetic with the image taps organised by ROW instead of by term.  A feature's pattern touches the
// rows min_oy-1 .. max_oy+2 and, in each, the bytes u_i-1+min_ox .. u_i+2+max_ox (<= 10 for the 21-pixel pattern): one
// row = two ds_read2_b32 + three v_alignbyte that bring those bytes to FIXED positions of three registers, whatever the
// lane's alignment.  Every tap of every term is then a byte select at a compile-time position (v_cvt_f32_ubyteN, SDWA
// operands) — no per-tap address arithmetic, no per-tap LDS read: 20 LDS reads per feature instead of 74 at level 1.
// Terms are visited row by row (ascending oy), so four rows of windows are live at a time.  The order of the moment sums
// changes with it (they are tolerance-compared); the per-term decision arithmetic is untouched.
template <int PI, typename IP>
__device__ __forceinline__ Moments feature_terms_rows(IP img, GlbF32 ref_patch, int base, float w_tl, float w_tr, float w_bl,
                                                    float w_br, uint32_t fb, uint32_t nb, int stride, float a, float huber,
                                                    float outlier, float max_energy, int top)
{
  constexpr auto P = PatRows<PI>::v;
  constexpr int PA = h_pattern_num[PI];
  constexpr int NB = P.max_ox - P.min_ox + 4;          // bytes of a row the pattern touches
  constexpr int NW = (NB + 3) / 4;                     // registers per row window (2 or 3)
  constexpr int R0 = P.min_oy - 1, R1 = P.max_oy + 2;  // first / last row
  Moments m;
  m.ee = m.ex = m.ey = m.xx = m.xy = m.yy = m.re = m.rx = m.ry = 0; m.E = 0; m.nt = PA; m.nsat = 0;
  stride = __builtin_amdgcn_readfirstlane(stride);
  nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
  typedef const __attribute__((address_space(1))) char* GlbBytes;
  const GlbBytes rpb = (GlbBytes)ref_patch;
  constexpr int PF = 4;  // reference intensities requested PF terms ahead of their use
  float ipf[PF];
#pragma unroll
  for (int t = 0; t < PF && t < PA; t++) ipf[t] = *(GlbF32)(rpb + (fb + (uint32_t)P.idx[t] * nb));
  uint32_t win[4][3];
  const int c0 = base + P.min_ox;                      // byte address of the window's first byte in row 0 (v_i)
  int t = 0;                                           // next term (in row order)
#pragma unroll
  for (int R = R0; R <= R1; R++) {
    {
      const int addr = c0 + R * stride;
      const int A = addr >> 2;
      const uint32_t sh = (uint32_t)(addr & 3);
      uint32_t (&w)[3] = win[(R - R0) & 3];
      const uint32_t d0 = img[A], d1 = img[A + 1], d2 = img[A + 2];
      w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
      w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
      if (NW == 3) { const uint32_t d3 = img[A + 3]; w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh); } else w[2] = 0;
    }
    const int oy = R - 2;                              // the terms whose four rows are now complete
#pragma unroll
    for (int q = 0; q < PA; q++) {
      if (h_pattern[PI][P.idx[q]][1] != oy) continue;
      const int kk = P.idx[q];
      const int j = h_pattern[PI][kk][0] - P.min_ox;   // window byte of pixel (u_i - 1 + ox)
      const uint32_t (&w0)[3] = win[(oy - 1 - R0) & 3];
      const uint32_t (&w1)[3] = win[(oy - R0) & 3];
      const uint32_t (&w2)[3] = win[(oy + 1 - R0) & 3];
      const uint32_t (&w3)[3] = win[(oy + 2 - R0) & 3];
      const float iref = ipf[t % PF];
      if (t + PF < PA) ipf[t % PF] = *(GlbF32)(rpb + (fb + (uint32_t)P.idx[(t + PF) < PA ? (t + PF) : 0] * nb));
      t++;
      const float p10 = win_byte(w1, j), p11 = win_byte(w1, j + 1), p12 = win_byte(w1, j + 2), p13 = win_byte(w1, j + 3);
      const float p20 = win_byte(w2, j), p21 = win_byte(w2, j + 1), p22 = win_byte(w2, j + 2), p23 = win_byte(w2, j + 3);
      const float p01 = win_byte(w0, j + 1), p02 = win_byte(w0, j + 2), p31 = win_byte(w3, j + 1), p32 = win_byte(w3, j + 2);
      // decision arithmetic: exactly the reference's expression order (:339-348)
      const float cur = ((w_tl * p11 + w_tr * p12) + w_bl * p21) + w_br * p22;
      const float res = cur - a * iref;
      const float ares = fabsf(res);
      const float hw = ares < huber ? 1.0f : huber * __builtin_amdgcn_rcpf(ares);
      const bool sat = (ares > outlier) && !top;
      const float e_term = top ? (hw * res) * res : ((hw * res) * res) * (2 - hw);
      m.E += sat ? max_energy : e_term;
      m.nsat += sat ? 1 : 0;
      const float dx = fmaf(w_tl, p12 - p10, fmaf(w_tr, p13 - p11, fmaf(w_bl, p22 - p20, w_br * (p23 - p21))));
      const float dy = fmaf(w_tl, p21 - p01, fmaf(w_tr, p22 - p02, fmaf(w_bl, p31 - p11, w_br * (p32 - p12))));
      const float wgt = sat ? 0.0f : hw;
      const float e = -iref;
      const float we = wgt * e, wx = wgt * dx, wy = wgt * dy, wr = wgt * res;
      m.ee = fmaf(we, e, m.ee); m.ex = fmaf(we, dx, m.ex); m.ey = fmaf(we, dy, m.ey);
      m.xx = fmaf(wx, dx, m.xx); m.xy = fmaf(wx, dy, m.xy); m.yy = fmaf(wy, dy, m.yy);
      m.re = fmaf(wr, e, m.re); m.rx = fmaf(wr, dx, m.rx); m.ry = fmaf(wr, dy, m.ry);
    }
  }
  return m;
}

// The inverse-compositional twin of feature_terms_rows: the residual needs only the bilinear intensity of the CURRENT frame (two
// rows of windows live, as in collect_terms_rows), the gradient is the cached reference gradient (precompute_reference: ref_dx /
// ref_dy beside the patch cache, [pattern pixel][feature]).  The reference takes this mode whenever the new frame's gradient mean
// does not exceed the last frame's by 0.5 (src/frame_handler_mono.cpp:184) — nearly every frame of a sequence — so this is the
// loop the sequence engine spends its tracker time in; until round 6 it ran the generic per-tap loop (run-time pattern table, two
// LDS reads and three dependent cache loads per term, one term of prefetch).  Here: tap and cache offsets are compile-time
// expressions, 8-14 LDS reads per feature instead of 2 per term, the three cached values of a term requested PF terms ahead.
// The per-term decision arithmetic (intensity, residual, saturation test) is the generic loop's expression by expression; the
// moments are summed in row order instead of pattern order (tolerance-compared quantities, like the forward mode's).
template <int PI, typename IP>
__device__ __forceinline__ Moments feature_terms_rows_ic(IP img, GlbF32 ref_patch, GlbF32 ref_dx, GlbF32 ref_dy, int base, float w_tl, float w_tr,
                                                       float w_bl, float w_br, uint32_t fb, uint32_t nb, int stride, float a, float huber,
                                                       float outlier, float max_energy, int top)
{
  constexpr auto P = PatRows<PI>::v;
  constexpr int PA = h_pattern_num[PI];
  constexpr int NB = P.max_ox - P.min_ox + 4;
  constexpr int NW = (NB + 3) / 4;
  constexpr int R0 = P.min_oy, R1 = P.max_oy + 1;
  Moments m;
  m.ee = m.ex = m.ey = m.xx = m.xy = m.yy = m.re = m.rx = m.ry = 0; m.E = 0; m.nt = PA; m.nsat = 0;
  stride = __builtin_amdgcn_readfirstlane(stride);
  nb = (uint32_t)__builtin_amdgcn_readfirstlane((int)nb);
  typedef const __attribute__((address_space(1))) char* GlbBytes;
  const GlbBytes rpb = (GlbBytes)ref_patch, rxb = (GlbBytes)ref_dx, ryb = (GlbBytes)ref_dy;
  constexpr int PF = TRK_IC_PREFETCH;   // cached values requested PF terms ahead of their use (row order)
  float ipf[PF], xpf[PF], ypf[PF];
#pragma unroll
  for (int t = 0; t < PF && t < PA; t++) {
    const uint32_t off = fb + (uint32_t)P.idx[t] * nb;
    ipf[t] = *(GlbF32)(rpb + off); xpf[t] = *(GlbF32)(rxb + off); ypf[t] = *(GlbF32)(ryb + off);
  }
  uint32_t win[2][3];
  const int c0 = base + P.min_ox;
  int t = 0;
#pragma unroll
  for (int R = R0; R <= R1; R++) {
    {
      const int addr = c0 + R * stride;
      const int A = addr >> 2;
      const uint32_t sh = (uint32_t)(addr & 3);
      uint32_t (&w)[3] = win[(R - R0) & 1];
      const uint32_t d0 = img[A], d1 = img[A + 1], d2 = img[A + 2];
      w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
      w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
      if (NW == 3) { const uint32_t d3 = img[A + 3]; w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh); } else w[2] = 0;
    }
    const int oy = R - 1;                              // the terms whose two rows are now complete
#pragma unroll
    for (int q = 0; q < PA; q++) {
      if (h_pattern[PI][P.idx[q]][1] != oy) continue;
      const int kk = P.idx[q];
      const int j = h_pattern[PI][kk][0] - P.min_ox;
      const uint32_t (&w1)[3] = win[(oy - R0) & 1];
      const uint32_t (&w2)[3] = win[(oy + 1 - R0) & 1];
      const float iref = ipf[t % PF], dx = xpf[t % PF], dy = ypf[t % PF];
      if (t + PF < PA) {
        const uint32_t off = fb + (uint32_t)P.idx[(t + PF) < PA ? (t + PF) : 0] * nb;
        ipf[t % PF] = *(GlbF32)(rpb + off); xpf[t % PF] = *(GlbF32)(rxb + off); ypf[t % PF] = *(GlbF32)(ryb + off);
      }
      t++;
      const float p11 = win_byte(w1, j + 1), p12 = win_byte(w1, j + 2), p21 = win_byte(w2, j + 1), p22 = win_byte(w2, j + 2);
      // decision arithmetic: exactly the reference's expression order (:339-348)
      const float cur = ((w_tl * p11 + w_tr * p12) + w_bl * p21) + w_br * p22;
      const float res = cur - a * iref;
      const float ares = fabsf(res);
      const float hw = ares < huber ? 1.0f : huber * __builtin_amdgcn_rcpf(ares);
      const bool sat = (ares > outlier) && !top;
      const float e_term = top ? (hw * res) * res : ((hw * res) * res) * (2 - hw);
      m.E += sat ? max_energy : e_term;
      m.nsat += sat ? 1 : 0;
      const float wgt = sat ? 0.0f : hw;
      const float e = -iref;
      const float we = wgt * e, wx = wgt * dx, wy = wgt * dy, wr = wgt * res;
      m.ee = fmaf(we, e, m.ee); m.ex = fmaf(we, dx, m.ex); m.ey = fmaf(we, dy, m.ey);
      m.xx = fmaf(wx, dx, m.xx); m.xy = fmaf(wx, dy, m.xy); m.yy = fmaf(wy, dy, m.yy);
      m.re = fmaf(wr, e, m.re); m.rx = fmaf(wr, dx, m.rx); m.ry = fmaf(wr, dy, m.ry);
    }
  }
  return m;
}

// Expand one feature's moments into the 28 + 7 normal-equation entries (computeGS, :499-525).
// A = fx_l * J.row(0), B = fy_l * J.row(1) (frame.h:192-212); in inverse-compositional mode the
// Jacobian is taken at the reference point and scaled by the exposure ratio
// (m_jacobian_cache_true = exposure_rat * m_jacobian_cache_raw, CoarseTracker.cpp:245).
#ifndef TRK_EXPAND_F32
#define TRK_EXPAND_F32 1
#endif
template <bool IC>
HSO_DEV void expand_feature(Acc& acc, const LevelCtx& L, const Proj& p, const Moments& m, int f, float a)
{
  double J0[6], J1[6];
  double sA, sB;
  if (!IC) {
    jacobian_xyz2uv(p.x, p.y, p.z, J0, J1);
    sA = 0.5 * L.fxl; sB = 0.5 * L.fyl;  // dx, dy are twice the central differences
  } else {
    const int ns = L.job->n_stride;
    const double dist = L.job->feats[5 * ns + f];
    jacobian_xyz2uv(L.job->feats[2 * ns + f] * dist, L.job->feats[3 * ns + f] * dist,
                    L.job->feats[4 * ns + f] * dist, J0, J1);
    sA = L.fxl * (double)a; sB = L.fyl * (double)a;
  }
  double A[6], B[6];
#pragma unroll
  for (int k = 0; k < 6; k++) { A[k] = J0[k] * sA; B[k] = J1[k] * sB; }
  const double d_rx = m.rx, d_ry = m.ry;
  acc.H[0] += m.ee;
#if TRK_EXPAND_F32
  // H is an fp32 quantity in the reference (Accumulator7, MatrixAccumulator.h:33: products and sums in float): expanding the
  // 27 entries in fp32 as well drops 34 fp64 -> fp32 conversions (quarter rate) and the fp64 products per feature
  float Af[6], Bf[6];
#pragma unroll
  for (int k = 0; k < 6; k++) { Af[k] = (float)A[k]; Bf[k] = (float)B[k]; }
#pragma unroll
  for (int k = 0; k < 6; k++) acc.H[1 + k] += fmaf(m.ex, Af[k], m.ey * Bf[k]);
  int idx = 7;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const float xa = fmaf(m.xx, Af[k], m.xy * Bf[k]);  // coefficient of A[l]
    const float xb = fmaf(m.xy, Af[k], m.yy * Bf[k]);  // coefficient of B[l]
#pragma unroll
    for (int l = k; l < 6; l++) { acc.H[idx] += fmaf(xa, Af[l], xb * Bf[l]); idx++; }
  }
#else
  const double d_ex = m.ex, d_ey = m.ey, d_xx = m.xx, d_xy = m.xy, d_yy = m.yy;
#pragma unroll
  for (int k = 0; k < 6; k++) acc.H[1 + k] += (float)fma(d_ex, A[k], d_ey * B[k]);
  int idx = 7;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double xa = fma(d_xx, A[k], d_xy * B[k]);  // coefficient of A[l]
    const double xb = fma(d_xy, A[k], d_yy * B[k]);  // coefficient of B[l]
#pragma unroll
    for (int l = k; l < 6; l++) { acc.H[idx] += (float)fma(xa, A[l], xb * B[l]); idx++; }
  }
#endif
  acc.d[0] -= (double)m.re;
#pragma unroll
  for (int k = 0; k < 6; k++) acc.d[1 + k] -= fma(d_rx, A[k], d_ry * B[k]);
  acc.d[7] += (double)m.E;
  acc.H[28] += (float)m.nt;
  acc.H[29] += (float)m.nsat;
}

// computeResiduals (CoarseTracker.cpp:242-414) fused with computeGS (:499-525).
// Leaves the block-reduced sums in s.red: [0..27] H upper triangle (row-major),
// [28..34] b, [35] E, [36] m_total_terms, [37] m_saturated_terms.
//
// Each thread owns FPT features per round (lane groups of S threads share one feature when
// the table is small).  The 38 per-feature contributions live in registers only between
// the expansion and the wave-wide halving exchange that follows it, so the pixel loop —
// where the time goes — runs without the accumulators' register footprint; what persists
// across rounds is the single float and the single double each lane is responsible for.
#ifndef TRK_FPT
#define TRK_FPT 2  // features per thread and round: halves the number of wave exchanges per evaluation
#endif
#ifndef TRK_FPT_P21
#define TRK_FPT_P21 TRK_FPT  // the 21-pixel pattern (finest level) can be set apart: with the old pixel loop it spilled at 2
#endif
#define HSO_PHASE __device__ __forceinline__
template <bool IC, bool S1, typename Ptr, int PI = -1>
HSO_PHASE void eval_terms(Shared& s, const LevelCtx& L, Ptr img, const Se3& T, float a)
{
  constexpr int FPT = (PI == 5) ? TRK_FPT_P21 : TRK_FPT;   // features per thread and round
  const int n = L.job->n;
  const int PA = s.PA, border = s.pad + 1, S = S1 ? 1 : s.S;
  const int G = TRK_THREADS / S;
  const int sub = S1 ? 0 : (int)(threadIdx.x % S);
  const int grp = S1 ? (int)threadIdx.x : (int)(threadIdx.x / S);
  const bool top = (L.level == L.C->max_level);
  const float huber = s.huber;
  const float outlier = s.outlier;
  const float max_energy = (float)((double)(2 * huber) * (double)outlier - (double)(huber * huber));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

#if defined(HSO_PHASE_TIMERS) && !defined(HSO_SEL_PROBE)
#define DBG_T(k) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); s.dbg[k] += n_ - dbg_t; dbg_t = n_; } } while (0)
  unsigned long long dbg_t = __builtin_readcyclecounter();
#else
#define DBG_T(k) do { } while (0)
#endif
  float totH = 0;
  double totD = 0;
  int slotH = 0, slotD = 0;
  // Which features this thread owns.  With one thread per feature the split between the two
  // wavefronts of a SIMD is deliberately uneven: the older wavefront (waves 0..3) wins issue
  // arbitration and would finish ~25 % early, leaving the younger one to run alone without latency
  // hiding; giving the older half TRK_OLD_SHARE/16 of the features lets both finish together.
  // The mapping is static, so results stay bit-reproducible.
  int gbase = 0, gn = n, gthreads = G, gt = grp;
  if (S1) {
    const int n_old = (int)(((long long)n * TRK_OLD_SHARE + 15) / 16);
    const bool old = threadIdx.x < TRK_THREADS / 2;
    gthreads = TRK_THREADS / 2;
    gbase = old ? 0 : n_old;
    gn = old ? n_old : n - n_old;
    gt = old ? (int)threadIdx.x : (int)threadIdx.x - TRK_THREADS / 2;
  }
  auto fidx = [&](int k) { const int i = gt + k * gthreads; return i < gn ? gbase + i : n; };  // n = "none"
  const int n_rounds = max(1, (((gn + gthreads - 1) / gthreads) + FPT - 1) / FPT);  // >= 1: the exchange assigns the slots
  FeatRaw nxt = load_feature(L, fidx(0));
  for (int r = 0; r < n_rounds; r++) {
    Proj p[FPT];
    Moments m[FPT];
    int ff[FPT];
#pragma unroll
    for (int q = 0; q < FPT; q++) {
      const int f = fidx(r * FPT + q);
      ff[q] = f;
      const FeatRaw raw = nxt;
      nxt = load_feature(L, fidx(r * FPT + q + 1));  // next feature's record in flight during this pixel loop
      p[q] = project_feature(L, T, raw, border);
      DBG_T(0);
      if constexpr (PI >= 0 && IC) {
        if (p[q].ok)
          m[q] = feature_terms_rows_ic<PI>(img, (GlbF32)L.sc.ref_patch, (GlbF32)L.sc.ref_dx, (GlbF32)L.sc.ref_dy, p[q].base, p[q].w_tl, p[q].w_tr, p[q].w_bl,
                                           p[q].w_br, (uint32_t)f * 4u, (uint32_t)L.C->n_max * 4u, L.cols, a, huber, outlier, max_energy, top ? 1 : 0);
        else
          m[q] = Moments{};
      } else if constexpr (PI >= 0) {
        if (p[q].ok)
#if TRK_ROW_WINDOWS
          m[q] = feature_terms_rows<PI>(img, (GlbF32)L.sc.ref_patch, p[q].base, p[q].w_tl, p[q].w_tr, p[q].w_bl, p[q].w_br,
                                        (uint32_t)f * 4u, (uint32_t)L.C->n_max * 4u, L.cols, a, huber, outlier, max_energy, top ? 1 : 0);
#else
          m[q] = feature_terms_static<PI>(img, (GlbF32)L.sc.ref_patch, p[q].base, p[q].w_tl, p[q].w_tr, p[q].w_bl, p[q].w_br,
                                          (uint32_t)f * 4u, (uint32_t)L.C->n_max * 4u, L.cols, a, huber, outlier, max_energy, top ? 1 : 0);
#endif
        else
          m[q] = Moments{};
      } else {
        m[q] = feature_terms<IC>(s, L, img, p[q], f, a, sub, S, PA, top, huber, outlier, max_energy);
      }
      DBG_T(1);
      if (!S1) {
        // combine the S lanes of the feature group (power of two <= 64, never straddles a wave)
        for (int k = S >> 1; k > 0; k >>= 1) {
          m[q].ee += __shfl_xor(m[q].ee, k); m[q].ex += __shfl_xor(m[q].ex, k); m[q].ey += __shfl_xor(m[q].ey, k);
          m[q].xx += __shfl_xor(m[q].xx, k); m[q].xy += __shfl_xor(m[q].xy, k); m[q].yy += __shfl_xor(m[q].yy, k);
          m[q].re += __shfl_xor(m[q].re, k); m[q].rx += __shfl_xor(m[q].rx, k); m[q].ry += __shfl_xor(m[q].ry, k);
          m[q].E += __shfl_xor(m[q].E, k); m[q].nt += __shfl_xor(m[q].nt, k); m[q].nsat += __shfl_xor(m[q].nsat, k);
        }
      }
    }
    Acc acc;
#pragma unroll
    for (int i = 0; i < 32; i++) acc.H[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc.d[i] = 0;
#pragma unroll
    for (int q = 0; q < FPT; q++)
      if (p[q].ok && sub == 0) expand_feature<IC>(acc, L, p[q], m[q], ff[q], a);
    DBG_T(2);
    float th; double td;
    slotH = 0; slotD = 0;
    Halve<float, 32, 32>::run(acc.H, lane, slotH, th);
    Halve<double, 8, 32>::run(acc.d, lane, slotD, td);
    totH += th;
    totD += td;
    DBG_T(3);
  }
  if (slotH < 28) s.wave_part[wave][slotH] = (double)totH;
  else if (slotH < 30) s.wave_part[wave][8 + slotH] = (double)totH;   // n_terms, n_saturated -> [36], [37]
  if (slotD < 8) s.wave_part[wave][28 + slotD] = totD;
  __syncthreads();
  if (threadIdx.x < N_RED) {
    double t = 0;
    for (int w = 0; w < TRK_WAVES; w++) t += s.wave_part[w][threadIdx.x];
    s.red[threadIdx.x] = t;
  }
  __syncthreads();
#if TRK_COOP
  coop_allreduce(s);
#endif
  DBG_T(4);
}

template <bool IC>
HSO_DEV void eval_dispatch(Shared& s, const LevelCtx& L, LdsPtr lds_img, const Se3& T, float a)
{
  if (s.S == 1) {
    if (s.use_lds) {
      if constexpr (!IC || TRK_IC_ROWS) {
        switch (s.pi) {  // pattern-specialised pixel loops for the patterns levels 4..1 use
          case 2: eval_terms<IC, true, LdsPtr, 2>(s, L, lds_img, T, a); return;
          case 3: eval_terms<IC, true, LdsPtr, 3>(s, L, lds_img, T, a); return;
          case 4: eval_terms<IC, true, LdsPtr, 4>(s, L, lds_img, T, a); return;
          case 5: eval_terms<IC, true, LdsPtr, 5>(s, L, lds_img, T, a); return;
          case 6: eval_terms<IC, true, LdsPtr, 6>(s, L, lds_img, T, a); return;   // level 0 (relocalisation) of a small image
          default: break;
        }
      }
      eval_terms<IC, true, LdsPtr>(s, L, lds_img, T, a);
    } else {
#if TRK_GLOBAL_ROWS
      if constexpr (!IC || TRK_IC_ROWS) {
        switch (s.pi) {  // the same row-window loops on the image in device memory (a level that does not fit this shape's LDS)
          case 2: eval_terms<IC, true, GlbW32, 2>(s, L, (GlbW32)L.cur_glb, T, a); return;
          case 3: eval_terms<IC, true, GlbW32, 3>(s, L, (GlbW32)L.cur_glb, T, a); return;
          case 4: eval_terms<IC, true, GlbW32, 4>(s, L, (GlbW32)L.cur_glb, T, a); return;
          case 5: eval_terms<IC, true, GlbW32, 5>(s, L, (GlbW32)L.cur_glb, T, a); return;
          case 6: eval_terms<IC, true, GlbW32, 6>(s, L, (GlbW32)L.cur_glb, T, a); return;
          default: break;
        }
      }
#endif
      eval_terms<IC, true, GlbPtr>(s, L, L.cur_glb, T, a);
    }
  } else {
    if (s.use_lds) eval_terms<IC, false, LdsPtr>(s, L, lds_img, T, a);
    else eval_terms<IC, false, GlbPtr>(s, L, L.cur_glb, T, a);
  }
}

// ----------------------------------------------------------- level + LM loop

HSO_DEV void begin_level(Shared& s, LevelCtx& L, const TrackConsts& C, const TrackJobDev& job,
                         const Scratch& sc, int level, uint32_t* lds_img)
{
  __syncthreads();
  L.C = &C; L.job = &job; L.sc = sc; L.level = level;
  L.cam_lds = (const __attribute__((address_space(3))) hso_camera*)&s.cam;
  if (threadIdx.x == 0) s.cam = C.cam;  // visible after the barriers below, before any projection
  const TrackLevel& V = C.lv[level];
  L.cols = V.w; L.rows = V.h;
  L.scale = 1.0f / (float)(1 << level);
  L.fxl = C.cam.fx * (double)L.scale;
  L.fyl = C.cam.fy * (double)L.scale;
  L.ref_glb = reinterpret_cast<GlbPtr>(job.ref_base + V.off);
  L.cur_glb = reinterpret_cast<GlbPtr>(job.cur_base + V.off);
  if (threadIdx.x == 0) {
    s.level = level; s.PA = V.pa; s.pad = V.pad; s.pi = V.pi;
    int S = 1;
    // small feature tables: several lanes per feature only in inverse-compositional mode; the forward mode's pattern-specialised
    // loops (one thread per feature) beat the lane-shared generic loop even with most threads idle (200 features: 0.50 -> 0.45 ms)
    while (C.inverse && S < 16 && job.n * S * 2 <= TRK_THREADS) S *= 2;
    s.S = S;
  }
  if (threadIdx.x < TRK_MAX_PA) s.poff[threadIdx.x] = V.poff[threadIdx.x];
  // reference image through LDS for the patch precompute, then the current image stays resident
#ifdef HSO_STAGE_PROBE
  unsigned long long sp_t = __builtin_readcyclecounter();
#define SP_T(k) do { if (threadIdx.x == 0) { const unsigned long long n_ = __builtin_readcyclecounter(); s.dbg[k] += n_ - sp_t; sp_t = n_; } } while (0)
#else
#define SP_T(k) do { } while (0)
#endif
  const bool ref_in_lds = stage_image(L, job.ref_base + V.off, lds_img);
  __syncthreads();
  SP_T(5);
  bool pre_done = false;
#if TRK_PRECOMPUTE_ROWS
  if (ref_in_lds && !C.inverse) {
    pre_done = true;
    switch (V.pi) {
      case 2: precompute_reference_rows<2>(L, (LdsPtr)lds_img, V.pad + 1); break;
      case 3: precompute_reference_rows<3>(L, (LdsPtr)lds_img, V.pad + 1); break;
      case 4: precompute_reference_rows<4>(L, (LdsPtr)lds_img, V.pad + 1); break;
      case 5: precompute_reference_rows<5>(L, (LdsPtr)lds_img, V.pad + 1); break;
      case 6: precompute_reference_rows<6>(L, (LdsPtr)lds_img, V.pad + 1); break;
      default: pre_done = false; break;
    }
  }
#endif
  if (pre_done) { }
  else if (ref_in_lds) precompute_reference<LdsPtr>(s, L, (LdsPtr)lds_img);
  else precompute_reference<GlbPtr>(s, L, L.ref_glb);
  __syncthreads();
  SP_T(6);
  const bool cur_in_lds = stage_image(L, job.cur_base + V.off, lds_img);
  if (threadIdx.x == 0) {
    s.use_lds = cur_in_lds ? 1 : 0;
    const size_t padded = (size_t)((L.cols * L.rows + L.cols + 32 + 15) & ~15);
    const size_t need = (size_t)job.n * (size_t)V.pa * 4;
    s.keys_lds_off = (cur_in_lds && !C.keys_in_memory && padded + need <= (size_t)C.lds_img_cap) ? (int)padded : 0;
  }
  __syncthreads();
  SP_T(7);
}

// Hl.ldlt().solve(b) of CoarseTracker.cpp:112-114 by one lane, entirely in registers: every
// loop below has compile-time bounds, so after unrolling all array indices are static and the
// 28 + 7 doubles never touch scratch or LDS (the earlier eight-lane v_readlane variant spent
// 16k cycles per solve on readlane hazards and 28 fp64 divisions; this one ~4k).  Right-looking
// LDL^T on the lower triangle with diagonal pivoting (largest |diagonal|, first on ties, like
// Eigen::LDLT); one reciprocal per pivot; z = D^-1 y (zero where the pivot vanished, Eigen's
// pseudo-inverse); back substitution; un-permute into s.step[0..6].
HSO_DEV void swap_d(double& x, double& y) { const double t = x; x = y; y = t; }

HSO_DEV void lane_ldlt7_solve(Shared& s, float lambda)
{
  double A[7][7], y[7], invd[7];
  int perm[7];
#pragma unroll
  for (int r = 0; r < 7; r++) {
#pragma unroll
    for (int c = r; c < 7; c++) {
      double v = s.H[7 * r - (r * (r - 1)) / 2 + (c - r)];
      if (r == c) v *= (double)(1 + lambda);  // Hl(i,i) *= (1+lambda), CoarseTracker.cpp:113
      A[c][r] = v;
    }
    y[r] = s.b[r];
    perm[r] = r;
  }
#pragma unroll
  for (int k = 0; k < 7; k++) {
    double best = -1;
    int idx = k;
#pragma unroll
    for (int q = k; q < 7; q++) {
      const double d = fabs(A[q][q]);
      if (d > best) { best = d; idx = q; }
    }
#pragma unroll
    for (int q = k + 1; q < 7; q++) {
      if (idx == q) {  // symmetric swap of indices k and q on the lower triangle
#pragma unroll
        for (int i = 0; i < k; i++) swap_d(A[k][i], A[q][i]);
        swap_d(A[k][k], A[q][q]);
#pragma unroll
        for (int i = k + 1; i < q; i++) swap_d(A[i][k], A[q][i]);
#pragma unroll
        for (int i = q + 1; i < 7; i++) swap_d(A[i][k], A[i][q]);
        swap_d(y[k], y[q]);
        const int tp = perm[k]; perm[k] = perm[q]; perm[q] = tp;
      }
    }
    const double akk = A[k][k];
    const bool valid = fabs(akk) > 0;
    const double inv = valid ? 1.0 / akk : 1.0;
    invd[k] = inv;
    double l[7];
#pragma unroll
    for (int i = k + 1; i < 7; i++) l[i] = A[i][k] * inv;
#pragma unroll
    for (int i = k + 1; i < 7; i++) {
#pragma unroll
      for (int j = k + 1; j <= i; j++) A[i][j] -= l[i] * A[j][k];
      y[i] -= l[i] * y[k];
    }
#pragma unroll
    for (int i = k + 1; i < 7; i++) A[i][k] = l[i];
  }
  const double tolerance = 1.0 / 1.7976931348623157e308;
  double x[7];
#pragma unroll
  for (int i = 0; i < 7; i++) x[i] = (fabs(A[i][i]) > tolerance) ? y[i] * invd[i] : 0.0;
#pragma unroll
  for (int k = 6; k >= 1; k--) {
#pragma unroll
    for (int i = 0; i < k; i++) x[i] -= A[k][i] * x[k];
  }
#pragma unroll
  for (int i = 0; i < 7; i++) s.step[perm[i]] = x[i];
}

// lane 0 after lane_ldlt7_solve: extrapolation, NaN guard, exposure and pose proposal
// (CoarseTracker.cpp:120-133)
HSO_DEV void lm_finish(Shared& s, bool inverse)
{
  double step[7];
  const float lambda = s.lambda;
  float extrap_fac = 1;
  if ((double)lambda < 0.001) extrap_fac = (float)sqrt(sqrt(0.001 / (double)lambda));
  double ssum = 0;
#pragma unroll
  for (int i = 0; i < 7; i++) { step[i] = s.step[i] * (double)extrap_fac; ssum += step[i]; }
  if (!isfinite(ssum) || isnan(step[0])) {
#pragma unroll
    for (int i = 0; i < 7; i++) step[i] = 0;
  }
  s.a_new = (float)((double)s.a + step[0]);
  double neg[6];
#pragma unroll
  for (int i = 0; i < 6; i++) neg[i] = -step[1 + i];
  const Se3 dT = se3_exp(neg);
  const Se3 T = s.T;
  s.Tn = inverse ? se3_mul(T, dT) : se3_mul(dT, T);
  double nrm = 0;
#pragma unroll
  for (int i = 0; i < 7; i++) nrm += step[i] * step[i];
  s.step_norm = nrm;  // squared; compared with (1e-4)^2 below — no square root on the serial lane
}

#ifdef HSO_PHASE_TIMERS
#ifndef HSO_PHASE_TIMERS_BASE
#define HSO_PHASE_TIMERS_BASE 0  /* 3: report dbg[3..7] = exchange, combine, select collect / median / MAD */
#endif
#define PH_START() unsigned long long ph_t = __builtin_readcyclecounter(); const unsigned long long ph_job = ph_t
#define PH_ADD(k) do { if (threadIdx.x == 0) { const unsigned long long ph_n = __builtin_readcyclecounter(); \
                         out->phase_cycles[k] += ph_n - ph_t; ph_t = ph_n; } } while (0)
#define PH_JOB() do { if (threadIdx.x == 0) { out->phase_cycles[4] = __builtin_readcyclecounter() - ph_job; for (int k_ = 0; k_ < 5; k_++) out->phase_cycles[5 + k_] = s.dbg[k_ + HSO_PHASE_TIMERS_BASE]; } } while (0)
#else
#define PH_START() do { } while (0)
#define PH_ADD(k) do { } while (0)
#define PH_JOB() do { } while (0)
#endif

HSO_DEV void publish_result(Shared& s, hso_track_result* gout)
{
  __syncthreads();
#if TRK_COOP
  if (s.coop_rank != 0) return;   // every workgroup of the job holds the same record; the first one writes it
#endif
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&s.res);
  uint32_t* dst = reinterpret_cast<uint32_t*>(gout);
  for (int i = threadIdx.x; i < (int)(sizeof(hso_track_result) / 4); i += TRK_THREADS) dst[i] = src[i];
}

template <bool IC>
HSO_DEV void track_one(Shared& s, const TrackConsts& C, const TrackJobDev& job, const Scratch& sc,
                       uint32_t* lds_img, hso_track_result* gout)
{
  const int tid = threadIdx.x;
  hso_track_result* const out = &s.res;  // bookkeeping stays in LDS; one coalesced copy at the end
  if (C.resume) {
    // an earlier launch worked through the levels above: its record (pose, exposure, per-level bookkeeping) is the start
    const uint32_t* src = reinterpret_cast<const uint32_t*>(gout);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out);
    for (int i = tid; i < (int)(sizeof(hso_track_result) / 4); i += TRK_THREADS) dst[i] = src[i];
    __syncthreads();
    if (tid == 0) { s.T = se3_from(out->T_cur_ref); s.a = out->exposure_rat; }
  } else if (tid == 0) {
    memset(out, 0, sizeof(*out));
    s.T = se3_from(job.T);
    s.a = job.a;
  }
  __syncthreads();
#if TRK_COOP
  if (job.n_total == 0) {  // CoarseTracker.cpp:53-54 (the whole job's table, not this workgroup's slice)
#else
  if (job.n == 0) {  // CoarseTracker.cpp:53-54
#endif
    if (tid == 0) { se3_to(s.T, out->T_cur_ref); out->exposure_rat = s.a; }
    publish_result(s, gout);
    return;
  }
  LevelCtx L;
#ifdef HSO_PHASE_TIMERS
  if (tid == 0) for (int k_ = 0; k_ < 8; k_++) s.dbg[k_] = 0;
#endif
  PH_START();
  for (int level = C.level_first; level >= C.level_last; --level) {
    begin_level(s, L, C, job, sc, level, lds_img);
    PH_ADD(0);
    {
      const Se3 T0 = s.T; const float a0 = s.a;
      select_robust(s, L, (LdsPtr)lds_img, T0, a0);
      PH_ADD(1);
      eval_dispatch<IC>(s, L, (LdsPtr)lds_img, T0, a0);
      PH_ADD(2);
    }
    if (tid < 35) { if (tid < 28) s.H[tid] = s.red[tid]; else s.b[tid - 28] = s.red[tid]; }
    if (tid == 0) {
      s.energy_old = (double)((float)s.red[35] / (float)(int)s.red[36]);
      s.lambda = 0.1f;
      s.stop = 0;
      out->huber[level] = s.huber; out->outlier[level] = s.outlier;
      out->n_select[level] = s.n_select;
      out->n_eval[level] = 1;
    }
    __syncthreads();
    for (int iter = 0; iter < C.n_iter; iter++) {
      if (tid == 0) {
        lane_ldlt7_solve(s, s.lambda);
        lm_finish(s, IC);
      }
      __syncthreads();
      PH_ADD(3);
      {
        const Se3 Tn = s.Tn; const float an = s.a_new;
        eval_dispatch<IC>(s, L, (LdsPtr)lds_img, Tn, an);
      }
      PH_ADD(2);
      if (tid < 64) {
        // the first wavefront: every lane forms the same decision from the same LDS words; the accepted normal equations are
        // copied by 35 lanes instead of one (one lane: ~2 k cycles per iteration of a job that has nothing else to run)
        const double energy_new = (double)((float)s.red[35] / (float)(int)s.red[36]);
        const bool accept = energy_new < s.energy_old;
        if (accept && tid < 35) { if (tid < 28) s.H[tid] = s.red[tid]; else s.b[tid - 28] = s.red[tid]; }
        if (tid == 0) {
          out->n_eval[level]++;
          out->iters[level] = iter + 1;
          if (accept) {
            s.energy_old = energy_new;
            s.a = s.a_new;
            s.T = s.Tn;
            s.lambda = (float)((double)s.lambda * 0.5);
            if (iter < 64) out->accept_mask[level] |= (1ull << iter);
          } else {
            s.lambda = s.lambda * 4;
            if ((double)s.lambda < 0.001) s.lambda = (float)0.001;
          }
          if (!(s.step_norm > 1e-8)) s.stop = 1;  // step.norm() > 1e-4 (CoarseTracker.cpp:169), on the squared norm
          // the last evaluation defines m_total_terms / m_saturated_terms (CoarseTracker.cpp:207)
          out->n_terms_last = (int)s.red[36];
          out->n_saturated_last = (int)s.red[37];
        }
      }
      __syncthreads();
      if (s.stop) break;
    }
    if (tid == 0) {
      out->energy[level] = s.energy_old;
      if (C.n_iter == 0) { out->n_terms_last = (int)s.red[36]; out->n_saturated_last = (int)s.red[37]; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    se3_to(s.T, out->T_cur_ref);
    out->exposure_rat = s.a;
    out->n_tracked = (int)((float)out->n_terms_last / (float)s.PA);
    out->status = 0;
#if TRK_COOP
    if (s.coop_fail) out->status = HSO_E_HIP;   // a peer workgroup never answered: the sums above are not the job's
    out->coop_workgroups = (int16_t)s.coop_K; out->coop_same_xcd = (int16_t)s.coop_fast;
#endif
  }
  PH_JOB();
  publish_result(s, gout);
}

// dynamic LDS: [0, kImgCap) staged level image (address 0 => tap addresses need no base add),
// then the Shared block
constexpr int kLdsTotal = TRK_LDS_KB * 1024;
constexpr int kImgCap = (int)(((kLdsTotal - 256 - sizeof(Shared)) / 256) * 256);
extern __shared__ __attribute__((aligned(16))) char g_smem[];

template <bool IC>
__global__ __launch_bounds__(TRK_THREADS, TRK_WAVES_PER_EU) void k_track(TrackConsts C, const TrackJobDev* jobs, int n_jobs,
                                                       int* job_counter, char* scratch, size_t scratch_stride,
                                                       hso_track_result* results)
{
  Shared& s = *reinterpret_cast<Shared*>(g_smem + kImgCap);
  uint32_t* lds_img = reinterpret_cast<uint32_t*>(g_smem);
  const Scratch sc = scratch_at(scratch + (size_t)blockIdx.x * scratch_stride, C.n_max);
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s.job = atomicAdd(job_counter, 1);
    __syncthreads();
    const int j = s.job;
    if (j >= n_jobs) break;
    track_one<IC>(s, C, jobs[j], sc, lds_img, &results[j]);
  }
}

#if TRK_COOP
// Cooperative shape: coop_K workgroups per job, each with its own slice of the job's feature table (a TrackJobDev of its own:
// the same frames, `feats` advanced to the slice, `n` its length, `n_total` the job's).  Block b runs on XCD b % 8 (observed,
// relied on for speed only), so the workgroups of job j are the blocks b = (j % 8) + 8 * i: they share one L2.
template <bool IC>
__global__ __launch_bounds__(TRK_THREADS, TRK_WAVES_PER_EU) void k_track_coop(TrackConsts C, const TrackJobDev* subjobs, int n_jobs, int k_stride,
                                                                                int scatter, CoopJobState* state, unsigned* fail_flag,
                                                                                char* scratch, size_t scratch_stride, hso_track_result* results)
{
  const int b = blockIdx.x;
  const int j = scatter ? b / k_stride : (b & 7), rank = scatter ? b % k_stride : (b >> 3);
  if (j >= n_jobs) return;
  const int w = j * k_stride + rank;
  const int K = subjobs[j * k_stride].coop_K;
  if (rank >= K) return;
  Shared& s = *reinterpret_cast<Shared*>(g_smem + kImgCap);
  uint32_t* lds_img = reinterpret_cast<uint32_t*>(g_smem);
  if (threadIdx.x == 0) {
    s.coop_state = state + j; s.coop_K = K; s.coop_rank = rank;
    s.coop_epoch = 0; s.coop_region = 0; s.coop_base = 0; s.coop_fail = 0; s.coop_fast = 0;
  }
  __syncthreads();
  coop_hello(s);
  const Scratch sc = scratch_at(scratch + (size_t)w * scratch_stride, C.n_max);
  track_one<IC>(s, C, subjobs[w], sc, lds_img, &results[j]);
  if (threadIdx.x == 0 && s.coop_fail) atomicOr(fail_flag, 1u);
}
#endif

// parity hook: one level, optional threshold selection, one evaluation
struct EvalArgs {
  int level;
  hso_se3 T;
  float a, huber, outlier;
};

template <bool IC>
__global__ __launch_bounds__(TRK_THREADS) void k_eval(TrackConsts C, const TrackJobDev* jobs, EvalArgs ea,
                                                      char* scratch, hso_eval_out* out)
{
  Shared& s = *reinterpret_cast<Shared*>(g_smem + kImgCap);
  uint32_t* lds_img = reinterpret_cast<uint32_t*>(g_smem);
  const Scratch sc = scratch_at(scratch, C.n_max);
  const TrackJobDev& job = jobs[0];
  LevelCtx L;
  if (threadIdx.x == 0) { s.T = se3_from(ea.T); s.a = ea.a; s.n_select = 0; }
  begin_level(s, L, C, job, sc, ea.level, lds_img);
  const Se3 T0 = s.T; const float a0 = s.a;
  if (ea.huber <= 0) {
    select_robust(s, L, (LdsPtr)lds_img, T0, a0);
  } else {
    if (threadIdx.x == 0) { s.huber = ea.huber; s.outlier = ea.outlier; }
    __syncthreads();
  }
  eval_dispatch<IC>(s, L, (LdsPtr)lds_img, T0, a0);
  int nv = 0;
  for (int i = threadIdx.x; i < job.n; i += TRK_THREADS) nv += sc.visible[i];
  nv = block_sum_int(s, nv);
  if (threadIdx.x == 0) {
    int idx = 0;
    for (int r = 0; r < 7; r++)
      for (int c = r; c < 7; c++) { out->H[r * 7 + c] = out->H[c * 7 + r] = s.red[idx]; idx++; }
    for (int i = 0; i < 7; i++) out->b[i] = s.red[28 + i];
    out->energy_sum = s.red[35];
    out->n_terms = (int)s.red[36];
    out->n_saturated = (int)s.red[37];
    out->energy = (double)((float)s.red[35] / (float)out->n_terms);
    out->n_select = s.n_select;
    out->n_visible = nv;
    out->huber = s.huber; out->outlier = s.outlier;
  }
}

#undef TRK_WAVES
#ifdef SEL_WORDS
#undef SEL_WORDS
#endif
#ifdef SEL_A_SHIFT
#undef SEL_A_SHIFT
#endif
#ifdef SEL_VPT
#undef SEL_VPT
#endif
#ifdef SEL_CAND_CAP
#undef SEL_CAND_CAP
#endif
#ifdef KSEL_T
#undef KSEL_T
#endif
#ifdef SELR_T
#undef SELR_T
#endif
#ifdef DBG_T
#undef DBG_T
#endif
#ifdef HSO_PHASE
#undef HSO_PHASE
#endif
#ifdef PH_START
#undef PH_START
#endif
#ifdef PH_ADD
#undef PH_ADD
#endif
#ifdef PH_JOB
#undef PH_JOB
#endif
#ifdef HSO_PHASE_TIMERS_BASE
#undef HSO_PHASE_TIMERS_BASE
#endif
#ifdef TRK_FPT
#undef TRK_FPT
#endif
#ifdef TRK_FPT_P21
#undef TRK_FPT_P21
#endif
