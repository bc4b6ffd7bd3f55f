// hso_select.hip — the grid selection of Reprojector::reprojectMap on the device (remainder):
// which of a frame's projected candidates are examined, in which order, and which of them become features.
//
// Reference: src/reprojector.cpp — reprojectMap :253-306 (reprojectCellAll when fewer than max_fts + 50 candidates were
// projected, else three passes over the cells in `cell_order`), reprojectCell :352-429 (first visit sorts the cell with
// pointQualityComparator :333-345 — point type, then feature type, both descending, stable; every examined candidate is
// erased; a deleted point costs a trial and nothing else; pass 1 and 2 stop at the cell's first success, pass 3 takes up to
// three), reprojectCellAll :556-612.  The matching itself has already happened for every candidate (hso_gpu_reproject_match
// matches all projected points in one launch), so the policy is a pure function of (cell, quality, deleted, matched) per
// candidate, the cell order and the budget — it only decides which results the caller applies, and in which order
// (n_failed_reproj_ / n_succeeded_reproj_ bookkeeping and the order of frame->fts_ follow from it).
//
// One 256-thread workgroup per frame.  Per-cell work (sorting the few candidates of a cell, locating its successes) is one
// thread per cell; the budget — "stop after the cell in which n_matches reaches max_fts" — is a prefix sum over the cells in
// visiting order with a search for the cut, done per pass by the whole workgroup.  No atomics decide anything: the result is
// deterministic, and equal to the sequential walk.
#include "hso_ctx.h"
#include "hso_pose_dev.h"
#include "hso_match_dev.h"
#include <string.h>
#include <algorithm>
#include <functional>
#include <mutex>
#include <vector>

using namespace hso_dev;

// 1024 threads per frame: these kernels are rounds of barrier-separated passes over ~5600 grid cells / ~3000 candidates with
// dependent loads from global scratch at one workgroup per frame; four times fewer rounds is what shortens them (256 threads:
// k_select 102 us, k_sel_emit_feats 48 us for ONE 2000-feature frame).  All results are integers or per-thread fp64: order-free.
#define SEL_THREADS 1024
#define SEL_WAVES (SEL_THREADS / 64)

struct SelFrame {
  int first, n;           // candidate range (projection order)
  const int* n_dev;       // when set: the candidate count is only known on the device (the chained call)
  int* cnt;               // [n_cells + 1] cell start offsets into list
  int* fill;              // [n_cells]
  int* list;              // [n] candidates by cell, each cell ordered by (quality desc, projection order)
  int* e1; int* e2;       // [n_cells] candidates examined in pass 1 / 2 when the cell is visited
  int* p3;                // [3 * n_cells] pass 3: examined count when stopping at the 1st / 2nd / 3rd remaining success
  int* a3;                // [n_cells] pass 3: successes among the remaining candidates
  int* scan;              // [n_cells] work array of the budget scans
  int* out;               // [n] examined candidates in order: index | taken << 31
  int* counts;            // [4] n_examined (= n_trials), n_matches, passes run, branch (0 = all, 1 = cells)
};

struct SelArgs {
  const int32_t* cell; const uint8_t* quality; const uint8_t* flags;   // per candidate; flags bit 0 matched, bit 1 deleted
  const int32_t* cell_order;
  int n_cells, max_fts;
};

// inclusive scan of v over a workgroup of W wavefronts; *total = sum (all threads must call)
template <int W> __device__ int sel_block_scan_w(int v, int* s_wave, int& total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  __syncthreads();
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0; total = 0;
#pragma unroll
  for (int w = 0; w < W; w++) { if (w < wave) base += s_wave[w]; total += s_wave[w]; }
  return incl + base;
}
// over the selection kernels' workgroup (SEL_THREADS values)
__device__ int sel_block_scan(int v, int* s_wave, int& total) { return sel_block_scan_w<SEL_WAVES>(v, s_wave, total); }

// For the visiting sequence k = 0 .. m-1 (cell = visit(k)) with per-cell gains gain(cell): the first k at which the running
// sum reaches `budget` (m if never), the total gained (capped at budget) and, in `scan`, the EXCLUSIVE prefix per k.
template <typename Visit, typename Gain>
__device__ void sel_budget_scan(int m, int budget, Visit visit, Gain gain, int* scan, int* s_wave, int* s_cut, int& cut, int& gained)
{
  if (threadIdx.x == 0) *s_cut = m;
  __syncthreads();
  int carry = 0;
  for (int k0 = 0; k0 < m; k0 += SEL_THREADS) {
    const int k = k0 + (int)threadIdx.x;
    const int g = k < m ? gain(visit(k)) : 0;
    int tot;
    const int incl = sel_block_scan(g, s_wave, tot) + carry;
    if (k < m) {
      scan[k] = incl - g;
      if (g > 0 && incl >= budget && incl - g < budget) atomicMin(s_cut, k);
    }
    carry += tot;
    __syncthreads();
  }
  __syncthreads();
  cut = *s_cut;
  gained = carry < budget ? carry : budget;
}

__global__ __launch_bounds__(SEL_THREADS) void k_select(SelArgs A, const SelFrame* frames)
{
  __shared__ int s_wave[SEL_WAVES];
  __shared__ int s_cut, s_n;
  const SelFrame F = frames[blockIdx.x];
  const int tid = threadIdx.x, n = F.n_dev ? *F.n_dev : F.n, nc = A.n_cells, budget = A.max_fts;
  const int32_t* cell = A.cell + F.first; const uint8_t* qual = A.quality + F.first; const uint8_t* flg = A.flags + F.first;
  auto matched = [&](int i) { return (flg[i] & 3) == 1; };   // matched and not deleted
  auto deleted = [&](int i) { return (flg[i] & 2) != 0; };
  if (n == 0 || budget <= 0) { if (tid == 0) { F.counts[0] = F.counts[1] = F.counts[2] = 0; F.counts[3] = 0; } return; }

  if (n < budget + 50) {
    // reprojectCellAll: projection order; every candidate costs a trial; stop when the budget is met
    int cut, gained;
    sel_budget_scan(n, budget, [](int k) { return k; }, [&](int i) { return matched(i) ? 1 : 0; }, F.scan /* >= n_cells? see host */, s_wave, &s_cut, cut, gained);
    const int n_ex = cut < n ? cut + 1 : n;
    for (int i = tid; i < n_ex; i += SEL_THREADS) F.out[i] = i | (matched(i) ? (int)0x80000000 : 0);
    if (tid == 0) { F.counts[0] = n_ex; F.counts[1] = gained; F.counts[2] = 0; F.counts[3] = 0; }
    return;
  }

  // ---- candidates by cell
  for (int c = tid; c <= nc; c += SEL_THREADS) F.cnt[c] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += SEL_THREADS) atomicAdd(&F.cnt[cell[i] + 1], 1);   // integer counts: order-free
  __syncthreads();
  {
    int carry = 0;
    for (int c0 = 0; c0 <= nc; c0 += SEL_THREADS) {
      const int c = c0 + tid;
      const int v = c <= nc ? F.cnt[c] : 0;
      int tot;
      const int incl = sel_block_scan(v, s_wave, tot) + carry;
      if (c <= nc) F.cnt[c] = incl;     // cnt[c] = start of cell c (cnt[0] = 0 since the count of "cell -1" is 0)
      carry += tot;
      __syncthreads();
    }
  }
  for (int c = tid; c < nc; c += SEL_THREADS) F.fill[c] = F.cnt[c];
  __syncthreads();
  for (int i = tid; i < n; i += SEL_THREADS) F.list[atomicAdd(&F.fill[cell[i]], 1)] = i;   // any order: sorted next
  __syncthreads();
  // ---- per cell: order by (quality desc, projection order) = the stable sort of reprojectCell; locate the successes
  for (int c = tid; c < nc; c += SEL_THREADS) {
    const int b = F.cnt[c], e = F.cnt[c + 1];
    for (int x = b + 1; x < e; x++) {                 // insertion sort, cells hold a handful of candidates
      const int v = F.list[x];
      int y = x - 1;
      while (y >= b) {
        const int w = F.list[y];
        if (qual[w] > qual[v] || (qual[w] == qual[v] && w < v)) break;
        F.list[y + 1] = w; y--;
      }
      F.list[y + 1] = v;
    }
    // pass 1: up to and including the first success (or everything)
    int x = b, e1 = 0, e2 = 0, h1 = 0, h2 = 0;
    for (; x < e; x++) { e1++; if (matched(F.list[x])) { h1 = 1; x++; break; } }
    for (; x < e; x++) { e2++; if (matched(F.list[x])) { h2 = 1; x++; break; } }
    // pass 3 starts behind what passes 1 and 2 erased; whether pass 2 ran for this cell is decided later: keep both variants
    F.e1[c] = e1 | (h1 << 30);
    F.e2[c] = e2 | (h2 << 30);
  }
  __syncthreads();

  int n_out = 0, n_match = 0, passes = 1;
  // ---- pass 1 (:268-278): cells in order, one match each, stop when the budget is met
  int cut1, g1;
  sel_budget_scan(nc, budget, [&](int k) { return A.cell_order[k]; }, [&](int c) { return (F.e1[c] >> 30) & 1; }, F.scan, s_wave, &s_cut, cut1, g1);
  const int m1 = cut1 < nc ? cut1 + 1 : nc;       // cells visited
  {
    // emit: exclusive prefix of the examined counts over the visited cells
    int carry = 0;
    for (int k0 = 0; k0 < m1; k0 += SEL_THREADS) {
      const int k = k0 + tid;
      const int c = k < m1 ? A.cell_order[k] : 0;
      const int ex = k < m1 ? (F.e1[c] & 0x3fffffff) : 0;
      int tot;
      const int off = sel_block_scan(ex, s_wave, tot) + carry - ex;
      if (k < m1) {
        const int b = F.cnt[c];
        for (int x = 0; x < ex; x++) { const int i = F.list[b + x]; F.out[off + x] = i | ((x == ex - 1 && ((F.e1[c] >> 30) & 1)) ? (int)0x80000000 : 0); }
      }
      carry += tot;
      __syncthreads();
    }
    n_out = carry; n_match = g1;
  }
  // ---- pass 2 (:281-293): cells in reverse order without index 0, the next match of each
  if (n_match < budget) {
    passes = 2;
    const int m = nc - 1;   // k = nc-1 .. 1
    auto visit2 = [&](int j) { return A.cell_order[nc - 1 - j]; };
    int cut2, g2;
    sel_budget_scan(m, budget - n_match, visit2, [&](int c) { return (F.e2[c] >> 30) & 1; }, F.scan, s_wave, &s_cut, cut2, g2);
    const int m2 = cut2 < m ? cut2 + 1 : m;
    int carry = 0;
    for (int k0 = 0; k0 < m2; k0 += SEL_THREADS) {
      const int k = k0 + tid;
      const int c = k < m2 ? visit2(k) : 0;
      const int ex = k < m2 ? (F.e2[c] & 0x3fffffff) : 0;
      int tot;
      const int off = sel_block_scan(ex, s_wave, tot) + carry - ex;
      if (k < m2) {
        const int b = F.cnt[c] + (F.e1[c] & 0x3fffffff);
        for (int x = 0; x < ex; x++) { const int i = F.list[b + x]; F.out[n_out + off + x] = i | ((x == ex - 1 && ((F.e2[c] >> 30) & 1)) ? (int)0x80000000 : 0); }
        F.e2[c] |= 1 << 29;   // visited in pass 2: its candidates are erased
      }
      carry += tot;
      __syncthreads();
    }
    n_out += carry; n_match += g2;
    // ---- pass 3 (:296-305): cells in order, up to three more matches each, stop when the budget is met
    if (n_match < budget) {
      passes = 3;
      __syncthreads();
      for (int c = tid; c < nc; c += SEL_THREADS) {
        const int b = F.cnt[c] + (F.e1[c] & 0x3fffffff) + (((F.e2[c] >> 29) & 1) ? (F.e2[c] & 0x1fffffff) : 0), e = F.cnt[c + 1];
        int a = 0, p[3] = { 0, 0, 0 };
        for (int x = b; x < e; x++) if (matched(F.list[x])) { if (a < 3) p[a] = x - b + 1; a++; }
        F.a3[c] = (a < 255 ? a : 255) | ((e - b) << 8);   // successes (at most 3 matter: clamped to the byte) and the remaining length
        F.p3[3 * c] = p[0]; F.p3[3 * c + 1] = p[1]; F.p3[3 * c + 2] = p[2];
      }
      __syncthreads();
      const int R = budget - n_match;
      int cut3, g3;
      sel_budget_scan(nc, R, [&](int k) { return A.cell_order[k]; }, [&](int c) { const int a = F.a3[c] & 0xff; return a < 3 ? a : 3; }, F.scan, s_wave, &s_cut, cut3, g3);
      const int m3 = cut3 < nc ? cut3 + 1 : nc;
      int carry3 = 0;
      for (int k0 = 0; k0 < m3; k0 += SEL_THREADS) {
        const int k = k0 + tid;
        int ex = 0, take = 0, c = 0;
        if (k < m3) {
          c = A.cell_order[k];
          const int a = F.a3[c] & 0xff, len = F.a3[c] >> 8;
          take = a < 3 ? a : 3;
          if (k == cut3) take = R - F.scan[k];                 // the budget runs out inside this cell
          ex = (take == 3 || k == cut3) ? F.p3[3 * c + take - 1] : len;   // stopped at a success, or walked to the end
        }
        int tot;
        const int off = sel_block_scan(ex, s_wave, tot) + carry3 - ex;
        if (k < m3) {
          const int b = F.cnt[c] + (F.e1[c] & 0x3fffffff) + (((F.e2[c] >> 29) & 1) ? (F.e2[c] & 0x1fffffff) : 0);
          for (int x = 0; x < ex; x++) { const int i = F.list[b + x]; F.out[n_out + off + x] = i | (matched(i) ? (int)0x80000000 : 0); }
        }
        carry3 += tot;
        __syncthreads();
      }
      n_out += carry3; n_match += g3;
    }
  }
  if (tid == 0) { F.counts[0] = n_out; F.counts[1] = n_match; F.counts[2] = passes; F.counts[3] = 1; }
  (void)s_n; (void)deleted;
}

extern "C" int hso_gpu_reproject_select(hso_gpu_ctx* ctx, const int32_t* frame_begin, int n_frames, const int32_t* cell,
                                        const uint8_t* quality, const uint8_t* flags, const int32_t* cell_order, int n_cells,
                                        int max_fts, int32_t* examined_out, int32_t* counts_out)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_frames < 0 || (n_frames > 0 && (!frame_begin || !counts_out)) || n_cells <= 0 || !cell_order || max_fts < 0)
    return hso_fail(ctx, HSO_E_INVALID, "reproject_select: bad argument");
  if (n_frames == 0) return HSO_OK;
  const int n_total = frame_begin[n_frames];
  if (frame_begin[0] != 0 || n_total < 0 || (n_total > 0 && (!cell || !quality || !flags || !examined_out)))
    return hso_fail(ctx, HSO_E_INVALID, "reproject_select: bad candidate tables");
  for (int f = 0; f < n_frames; f++)
    if (frame_begin[f + 1] < frame_begin[f]) return hso_fail(ctx, HSO_E_INVALID, "reproject_select: frame ranges must ascend");
  for (int i = 0; i < n_total; i++)
    if (cell[i] < 0 || cell[i] >= n_cells) return hso_fail(ctx, HSO_E_INVALID, "reproject_select: cell out of range");
  {
    std::vector<uint8_t> seen(n_cells, 0);
    for (int k = 0; k < n_cells; k++) {
      if (cell_order[k] < 0 || cell_order[k] >= n_cells || seen[cell_order[k]]) return hso_fail(ctx, HSO_E_INVALID, "reproject_select: cell_order is not a permutation");
      seen[cell_order[k]] = 1;
    }
  }
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  // inputs | frame records | per-frame scratch | outputs
  size_t o = 0;
  const size_t o_cell = o; o += al(sizeof(int32_t) * (size_t)n_total);
  const size_t o_q = o; o += al((size_t)n_total);
  const size_t o_f = o; o += al((size_t)n_total);
  const size_t o_ord = o; o += al(sizeof(int32_t) * (size_t)n_cells);
  const size_t o_fr = o; o += al(sizeof(SelFrame) * (size_t)n_frames);
  const size_t in_bytes = o;
  const size_t o_out = o; o += al(sizeof(int32_t) * (size_t)std::max(n_total, 1));
  const size_t o_cnt = o; o += al(sizeof(int32_t) * 4 * (size_t)n_frames);
  const size_t o_list = o; o += al(sizeof(int32_t) * (size_t)std::max(n_total, 1));
  const size_t per_frame = al(sizeof(int32_t) * (size_t)(n_cells + 1)) + 4 * al(sizeof(int32_t) * (size_t)n_cells) + al(sizeof(int32_t) * 3 * (size_t)n_cells);
  // the scan array also serves reprojectCellAll (indexed by candidate): n < max_fts + 50 entries
  const size_t scan_bytes = al(sizeof(int32_t) * (size_t)std::max(n_cells, max_fts + 50));
  const size_t o_scr = o; o += (per_frame + scan_bytes) * (size_t)n_frames;
  if (ctx->batch_cap < o) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(o)));
    ctx->batch_cap = hso_grown(o);
  }
  char* d = ctx->d_batch;
  char* h = hso_pinned(ctx, 0, in_bytes);
  if (!h) return HSO_E_NOMEM;
  memcpy(h + o_cell, cell, sizeof(int32_t) * (size_t)n_total);
  memcpy(h + o_q, quality, (size_t)n_total);
  memcpy(h + o_f, flags, (size_t)n_total);
  memcpy(h + o_ord, cell_order, sizeof(int32_t) * (size_t)n_cells);
  SelFrame* hf = reinterpret_cast<SelFrame*>(h + o_fr);
  for (int f = 0; f < n_frames; f++) {
    char* s = d + o_scr + (per_frame + scan_bytes) * (size_t)f;
    SelFrame& F = hf[f];
    F.first = frame_begin[f]; F.n = frame_begin[f + 1] - frame_begin[f]; F.n_dev = nullptr;
    F.cnt = reinterpret_cast<int*>(s); s += al(sizeof(int32_t) * (size_t)(n_cells + 1));
    F.fill = reinterpret_cast<int*>(s); s += al(sizeof(int32_t) * (size_t)n_cells);
    F.e1 = reinterpret_cast<int*>(s); s += al(sizeof(int32_t) * (size_t)n_cells);
    F.e2 = reinterpret_cast<int*>(s); s += al(sizeof(int32_t) * (size_t)n_cells);
    F.a3 = reinterpret_cast<int*>(s); s += al(sizeof(int32_t) * (size_t)n_cells);
    F.p3 = reinterpret_cast<int*>(s); s += al(sizeof(int32_t) * 3 * (size_t)n_cells);
    F.scan = reinterpret_cast<int*>(s);
    F.list = reinterpret_cast<int*>(d + o_list) + F.first;
    F.out = reinterpret_cast<int*>(d + o_out) + F.first;
    F.counts = reinterpret_cast<int*>(d + o_cnt) + 4 * f;
  }
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  SelArgs A;
  A.cell = reinterpret_cast<const int32_t*>(d + o_cell); A.quality = reinterpret_cast<const uint8_t*>(d + o_q);
  A.flags = reinterpret_cast<const uint8_t*>(d + o_f); A.cell_order = reinterpret_cast<const int32_t*>(d + o_ord);
  A.n_cells = n_cells; A.max_fts = max_fts;
  hipLaunchKernelGGL(k_select, dim3(n_frames), dim3(SEL_THREADS), 0, ctx->stream, A, reinterpret_cast<const SelFrame*>(d + o_fr));
  HSO_HIP_CHECK(ctx, hipGetLastError());
  if (n_total > 0) HSO_HIP_CHECK(ctx, hipMemcpyAsync(examined_out, d + o_out, sizeof(int32_t) * (size_t)n_total, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(counts_out, d + o_cnt, sizeof(int32_t) * 4 * (size_t)n_frames, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return HSO_OK;
}


// ------------------------------------------------------------------------------------------------ the resident per-frame chain
// hso_gpu_seq_chain (include/hso_gpu.h): tracker table -> CoarseTracker -> visiting order -> point list -> projection + matching
// (hso_align.hip) -> grid selection -> the frame's features + pose optimisation -> the examined candidates' bookkeeping and the
// inputs of the keyframe decision.  Everything between the job records going up and the result records coming back stays on the
// device; one workgroup per sequence in every stage that is not per point.

// the projected points of one job, in list order, as the candidate tables of k_select; cand_pt = candidate -> slice entry
__global__ __launch_bounds__(SEL_THREADS) void k_sel_gather(const hso_reproj_point* proj, const hso_match_brief* brief, const ChainJobDev* jobs, const ChainCur* cur,
                                                           int32_t* cell, uint8_t* quality, uint8_t* flags, int32_t* cand_pt, int* n_cand,
                                                           uint8_t* projected_out)
{
  __shared__ int s_wave[SEL_WAVES];
  const int c = blockIdx.x, b = jobs[c].slice_begin, e = b + cur[c].n_listed;
  int carry = 0;
  for (int i0 = b; i0 < e; i0 += SEL_THREADS) {
    const int i = i0 + (int)threadIdx.x;
    const int is = (i < e && proj[i].projected) ? 1 : 0;
    if (i < e) projected_out[i] = (uint8_t)is;   // reprojectPoint's return value per listed point
    int tot;
    const int pos = sel_block_scan(is, s_wave, tot) + carry - is;
    if (is) {
      const int q = proj[i].pad_ & 0xff;
      cell[b + pos] = proj[i].cell;
      quality[b + pos] = (uint8_t)q;
      flags[b + pos] = (uint8_t)(((proj[i].ref_obs >= 0 && brief[i].success) ? 1 : 0) | (((q >> 4) == 0) ? 2 : 0));
      cand_pt[b + pos] = i;
    }
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) n_cand[c] = carry;
}

#define SEL_KF_WORDS 64         // 4096 keyframes per sequence map take part in the compaction bitmask

// One workgroup per job finishes what the selection left: (1) where the job's examined records start in the packed output — the sum
// of the earlier jobs' counts, a few hundred integers at most, so every workgroup forms its own; (2) the examined candidates'
// records, packed; (3) the frame's features — the candidates that became features, in examination order = the order
// Reprojector::reprojectCell pushes `new Feature` into fts_ (src/reprojector.cpp:395-425): f = cam->cam2world(px_cur), level = the
// matcher's search level, type = the reference feature's, grad = the rotated reference gradient — once as the pose optimiser's
// table (host bearing / inverse depth / host keyframe of the point, temporary = Point::TYPE_TEMPORARY; the keyframe table
// compacted to the hosts of the selected features), once as the frame's own feature table in the sequence map.
struct EmitArgs {
  const hso_match_brief* brief; const int32_t* examined; const int32_t* cand_pt; const int* counts;
  int* offs; hso_match_brief* out; hso_frame_match* records; int n_calls;
};

__global__ __launch_bounds__(SEL_THREADS) void k_sel_emit_feats(EmitArgs E, hso_camera cam, const ChainJobDev* cjobs, const int32_t* ids, const uint8_t* quality,
                                                               int feat_cap, hso_pose_feat* feats, PoseJobDev* jobs, hso_se3* poses_out, int* n_poses_out, int* n_feats)
{
  __shared__ int s_wave[SEL_WAVES];
  __shared__ unsigned long long s_used[SEL_KF_WORDS];
  __shared__ int s_base[SEL_KF_WORDS];
  const int c = blockIdx.x;
  const ChainJobDev& J = cjobs[c];
  const int lb = J.slice_begin;
  int b, e;
  {
    int part = 0;
    for (int q = threadIdx.x; q < c; q += SEL_THREADS) part += E.counts[4 * q];
    int before;
    (void)sel_block_scan(part, s_wave, before);
    const int n_ex = E.counts[4 * c];
    b = before; e = before + n_ex;
    if (threadIdx.x == 0) { E.offs[c] = b; if (c == E.n_calls - 1) E.offs[E.n_calls] = e; }
    for (int k = threadIdx.x; k < n_ex; k += SEL_THREADS) {
      const int v = E.examined[lb + k];
      const int g = E.cand_pt[lb + (v & 0x7fffffff)];
      hso_match_brief r = E.brief[g];
      r.success = (v < 0) ? 1 : 0;          // became a feature (a matched candidate the budget never reached stays 0)
      r.pad_ = g - lb;                      // the point's position in its job's list
      E.out[b + k] = r;
      hso_frame_match m;
      m.px_cur[0] = r.px_cur[0]; m.px_cur[1] = r.px_cur[1]; m.grad[0] = r.grad[0]; m.grad[1] = r.grad[1];
      m.point = r.pad_; m.success = r.success; m.search_level = r.search_level; m.ref_type = r.ref_type; m.pad_ = 0;
      E.records[b + k] = m;
    }
    __threadfence_block();
    __syncthreads();                        // the records this workgroup wrote are what it reads below
  }
  const hso_match_brief* out = E.out;
  hso_pose_feat* F = feats + (size_t)c * feat_cap;
  for (int w = threadIdx.x; w < SEL_KF_WORDS; w += SEL_THREADS) s_used[w] = 0;
  __syncthreads();
  int carry = 0;
  for (int i0 = b; i0 < e; i0 += SEL_THREADS) {
    const int i = i0 + (int)threadIdx.x;
    const int is = (i < e && out[i].success) ? 1 : 0;
    int tot;
    const int pos = sel_block_scan(is, s_wave, tot) + carry - is;
    if (is && pos < feat_cap) {
      const hso_match_brief& r = out[i];
      const int at = lb + r.pad_;
      const int pid = ids[at];
      const hso_map_point& p = J.M.pts[pid];
      hso_pose_feat f;
      f.has_point = 1; f.type = r.ref_type; f.level = r.search_level; f.temporary = ((quality[at] >> 4) == 1) ? 1 : 0;
      f.host_pose = p.host_kf; f._pad = 0;
      hso_dev::cam2world_dev(cam, r.px_cur[0], r.px_cur[1], f.f);
      f.grad[0] = (double)r.grad[0]; f.grad[1] = (double)r.grad[1];
      f.host_f[0] = p.host_f[0]; f.host_f[1] = p.host_f[1]; f.host_f[2] = p.host_f[2];
      f.idist = p.idist;
      F[pos] = f;
      if (pos < J.M.ff_cap) {
        hso_seq_feature q;
        q.px[0] = r.px_cur[0]; q.px[1] = r.px_cur[1]; q.f[0] = f.f[0]; q.f[1] = f.f[1]; q.f[2] = f.f[2];
        q.grad[0] = r.grad[0]; q.grad[1] = r.grad[1]; q.point = pid; q.level = r.search_level; q.type = r.ref_type; q.pad_ = 0;
        J.M.ff_cur[pos] = q;
      }
      if (p.host_kf < SEL_KF_WORDS * 64) atomicOr(&s_used[p.host_kf >> 6], 1ull << (p.host_kf & 63));
    }
    carry += tot;
    __syncthreads();
  }
  const int n = carry < feat_cap ? carry : feat_cap;
  // compact keyframe index = rank of the keyframe among the used ones, in map order
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < SEL_KF_WORDS; w++) { s_base[w] = t; t += __popcll(s_used[w]); }
    n_poses_out[c] = t;
    jobs[c].n_feats = n; jobs[c].n_poses = t < HSO_POSE_MAX_POSES ? t : HSO_POSE_MAX_POSES; n_feats[c] = n;
  }
  __syncthreads();
  hso_se3* PO = poses_out + (size_t)c * HSO_POSE_MAX_POSES;
  for (int k = threadIdx.x; k < J.M.n_kfs && k < SEL_KF_WORDS * 64; k += SEL_THREADS) {
    const unsigned long long word = s_used[k >> 6], bit = 1ull << (k & 63);
    if (!(word & bit)) continue;
    const int idx = s_base[k >> 6] + __popcll(word & (bit - 1));
    if (idx < HSO_POSE_MAX_POSES) PO[idx] = J.M.kfs[k].T_f_w;
  }
  for (int i = threadIdx.x; i < n; i += SEL_THREADS) {
    const int k = F[i].host_pose;
    int idx = HSO_POSE_MAX_POSES;
    if (k < SEL_KF_WORDS * 64) { const unsigned long long word = s_used[k >> 6], bit = 1ull << (k & 63); idx = s_base[k >> 6] + __popcll(word & (bit - 1)); }
    if (idx < HSO_POSE_MAX_POSES) F[i].host_pose = idx;
    else { F[i].has_point = 0; F[i].host_pose = 0; }     // more host keyframes than the optimiser's table holds: the feature sits out
  }
}

// ---- after the pose optimiser: the bookkeeping of the examined candidates, the culling mask on the frame's feature table, the
// inputs of the keyframe decision, and the job's result record.  One workgroup per job.
struct FinishArgs {
  const ChainJobDev* jobs; const ChainCur* cur; const hso_track_result* track; const hso_pose_result* pose; const int* counts; const int* offs;
  const hso_frame_match* records; const int32_t* ids; const uint8_t* projected; const uint8_t* mask; const int* n_feats;
  const int32_t* kf_nfts; int32_t* events; hso_seq_result* results;
  SeedFrameDev* seed_frames;   // the seed table's per-group frame records (null: no observation chained)
  int feat_cap, quality_min_fts;
};

#define FIN_SORT_N 4096
// 1024 threads per sequence: the kernel is chains of dependent loads (record -> point row, feature -> point -> observation list)
// at one workgroup per sequence; more of them in flight is what shortens it (256 threads: 0.48 ms per 128 sequences)
#define FIN_THREADS 1024
#define FIN_WAVES (FIN_THREADS / 64)
// ascending bitonic sort of FIN_SORT_N doubles in LDS by FIN_THREADS threads
__device__ void fin_sort(double* a)
{
  for (int k = 2; k <= FIN_SORT_N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < FIN_SORT_N; i += FIN_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const double x = a[i], y = a[l];
          const bool up = (i & k) == 0;
          if ((x > y) == up) { a[i] = y; a[l] = x; }
        }
      }
      __syncthreads();
    }
}

__global__ __launch_bounds__(FIN_THREADS) void k_chain_finish(FinishArgs A, hso_camera cam)
{
  __shared__ double s_buf[FIN_SORT_N];           // the medians' keys; the flow terms (2 x 2048)
  __shared__ int s_votes[2048];
  __shared__ int s_wave[FIN_WAVES];
  __shared__ double s_min[FIN_WAVES];
  __shared__ float s_flow_full, s_flow_shift;
  __shared__ int s_flow_count;
  const int c = blockIdx.x, tid = threadIdx.x;
  const ChainJobDev& J = A.jobs[c];
  const ChainCur& C = A.cur[c];
  hso_seq_result& R = A.results[c];
  hso_map_point* pts = J.M.pts;
  const int32_t* ids = A.ids + J.slice_begin;
  int32_t* ev = A.events + J.slice_begin;
  int n_ev = 0;
  auto emit = [&](int code, int point, int flag) {   // ordered: every thread calls, flagged ones append in thread order
    int tot;
    const int pos = sel_block_scan_w<FIN_WAVES>(flag, s_wave, tot) + n_ev - flag;
    if (flag && pos < J.slice_cap) ev[pos] = (code << 28) | point;
    n_ev += tot;
    __syncthreads();
  };
  // (1) listed candidates and temporary points the projection rejected pay three failures (src/reprojector.cpp:214-222, 247-251)
  for (int i0 = C.n_kf_points; i0 < C.n_listed; i0 += FIN_THREADS) {
    const int i = i0 + tid;
    int code = 0, p = 0;
    if (i < C.n_listed && !A.projected[J.slice_begin + i]) {
      p = ids[i];
      uint32_t w = (uint32_t)pts[p].pad_;
      uint32_t nf = HSO_PT_NFAIL(w) + 3; if (nf > 1023) nf = 1023;
      w = (w & ~(0x3ffu << 8)) | (nf << 8);
      if (nf > 30) {
        if (i < C.n_kf_points + C.n_cands_listed) { w &= ~0xf0u; code = HSO_EV_ERASE_CANDIDATE; }   // kind -> TYPE_DELETED
        else { w |= HSO_PT_BAD; code = HSO_EV_TEMP_BAD; }
      }
      pts[p].pad_ = (int32_t)w;
    }
    emit(code, p, code != 0);
  }
  // (2) the examined candidates in examination order (:366-425)
  const int n_ex = A.counts[4 * c], rb = A.offs[c];
  for (int k0 = 0; k0 < n_ex; k0 += FIN_THREADS) {
    const int k = k0 + tid;
    int code = 0, p = 0;
    if (k < n_ex) {
      const hso_frame_match r = A.records[rb + k];
      p = ids[r.point];
      uint32_t w = (uint32_t)pts[p].pad_;
      const uint32_t kind = (w & 0xffu) >> 4;
      if (kind != 0) {
        if (!r.success) {
          uint32_t nf = HSO_PT_NFAIL(w) + 1; if (nf > 1023) nf = 1023;
          w = (w & ~(0x3ffu << 8)) | (nf << 8);
          if (kind == 3 && nf > 15) { w &= ~0xf0u; code = HSO_EV_ERASE_POINT; }
          else if (kind == 2 && nf > 30) { w &= ~0xf0u; code = HSO_EV_ERASE_CANDIDATE; }
          else if (kind == 1 && nf > 30) { w |= HSO_PT_BAD; code = HSO_EV_TEMP_BAD; }
        } else {
          uint32_t nk = HSO_PT_NOK(w) + 1; if (nk > 2047) nk = 2047;
          w = (w & ~(0x7ffu << 20)) | (nk << 20);
          if (kind == 3 && nk > 10) { w = (w & ~0xf0u) | (4u << 4); code = HSO_EV_GOOD; }
        }
        pts[p].pad_ = (int32_t)w;
      }
    }
    emit(code, p, code != 0);
  }
  // (3) the pose optimiser's culling (feature->point = NULL, src/pose_optimizer.cpp:722-748) — when processFrame gets that far
  const hso_pose_result& PR = A.pose[c];
  const int n_matches = A.counts[4 * c + 1], nf = A.n_feats[c];
  const bool used_pose = n_matches >= A.quality_min_fts && PR.status == 0 && !((J.flags & HSO_SEQ_SEED_BRANCH) && n_matches < 100);
  hso_seq_feature* ff = J.M.ff_cur;
  const uint8_t* mask = A.mask + (size_t)c * A.feat_cap;
  if (used_pose) for (int i = tid; i < nf; i += FIN_THREADS) if (mask[i]) ff[i].point = -1;
  __threadfence_block();
  __syncthreads();
  const Se3 Tc = used_pose ? se3_from(PR.T_f_w) : se3_from(C.T_cur_w);
  // (4a) the number of the frame's features that kept their point, and (below, once the flow sums say the frame may become a
  // keyframe) getSceneDepth / getSceneDistance (src/frame.cpp:323-366): upper medians of depth and distance over those points
  int n_pt = 0;
  {
    int mine = 0;
    for (int i = tid; i < nf && i < FIN_SORT_N; i += FIN_THREADS) mine += ff[i].point >= 0 ? 1 : 0;
    (void)sel_block_scan_w<FIN_WAVES>(mine, s_wave, n_pt);
    __syncthreads();
  }
  // (4b) createCovisibilityGraph (src/frame_handler_mono.cpp:559-647): a vote per observation of each of the frame's points
  for (int k = tid; k < HSO_SEQ_MAX_KFS; k += FIN_THREADS) s_votes[k] = 0;
  __syncthreads();
  for (int i = tid; i < nf; i += FIN_THREADS) {
    const int p = ff[i].point;
    if (p < 0) continue;
    const hso_map_point& P = pts[p];
    for (int q = 0, o = P.obs_begin; q < P.obs_count && o >= 0; q++) { const hso_obs& ob = J.M.obs[o]; if (ob.kf >= 0 && ob.kf < HSO_SEQ_MAX_KFS) atomicAdd(&s_votes[ob.kf], 1); o = ob.pad_; }
  }
  __syncthreads();
  if (tid == 0) {
    const int nk = J.M.n_kfs < HSO_SEQ_MAX_KFS ? J.M.n_kfs : HSO_SEQ_MAX_KFS;
    const int need = n_pt > 30 ? 5 : 3;
    int seen = 0, best = -1;
    for (int k = 0; k < nk; k++) if (s_votes[k] > 0) { seen++; if (best < 0 || s_votes[k] > s_votes[best]) best = k; }
    int last_v = 0x7fffffff, last_k = -1, n = 0;
    while (n < HSO_SEQ_MAX_COVIS) {          // votes descending, equal votes in table (= frame) order
      int pick = -1;
      for (int k = 0; k < nk; k++) {
        const int v = s_votes[k];
        if (v < need) continue;
        if (v > last_v || (v == last_v && k <= last_k)) continue;
        if (pick < 0 || v > s_votes[pick]) pick = k;
      }
      if (pick < 0) break;
      R.covis[n] = pick; R.covis_votes[n] = s_votes[pick]; n++;
      last_v = s_votes[pick]; last_k = pick;
    }
    if (n == 0 && best >= 0) { R.covis[0] = best; R.covis_votes[0] = s_votes[best]; n = 1; }
    for (int q = n; q < HSO_SEQ_MAX_COVIS; q++) { R.covis[q] = -1; R.covis_votes[q] = 0; }
    R.n_covis = seen; R.covis_best = best; R.n_with_point = n_pt;
  }
  __syncthreads();
  // (4c) needNewKf (:428-507): the optical flow the motion since the last keyframe induces on that keyframe's features, with the
  // full motion and with its translation alone: two float sums, added serially in list order like the reference's
  float flow_full = 0.f, flow_shift = 0.f; int flow_count = 0;
  if (J.last_kf_row >= 0) {
    const SeqKfDev& K = J.M.kfs[J.last_kf_row];
    const Se3 Tk_inv = se3_inverse(se3_from(K.T_f_w)), T_cur_kf = se3_mul(Tc, Tk_inv);
    const int32_t* lst = J.M.kf_fts + (size_t)J.last_kf_row * J.M.fts_cap;
    const int n_list = A.kf_nfts[J.kf_begin + J.last_kf_row];
    double* t_full = s_buf; double* t_shift = s_buf + FIN_SORT_N / 2;
    int mine_valid = 0;
    for (int i0 = 0; i0 < n_list; i0 += FIN_SORT_N / 2) {
      const int m = n_list - i0 < FIN_SORT_N / 2 ? n_list - i0 : FIN_SORT_N / 2;
      for (int i = tid; i < m; i += FIN_THREADS) {
        const int f = lst[i0 + i];
        const int p = J.M.obs_pt[f];
        double a = -1.0, bb = -1.0;                                // < 0: the feature has no point (squared distances are >= 0)
        if (p >= 0 && ((uint32_t)pts[p].pad_ & 0xf0u) != 0) {
          const hso_obs& o = J.M.obs[f];
          const hso_map_point& P = pts[p];
          const double ox = P.pos[0] - Tk_inv.tx, oy = P.pos[1] - Tk_inv.ty, oz = P.pos[2] - Tk_inv.tz;
          const double len = sqrt(ox * ox + oy * oy + oz * oz);
          const double kx = o.f[0] * len, ky = o.f[1] * len, kz = o.f[2] * len;   // the point in the keyframe, on the feature's bearing
          double x, y, z, u, v;
          se3_apply(T_cur_kf, kx, ky, kz, x, y, z);
          world2cam(cam, x, y, z, u, v);
          a = (u - o.px[0]) * (u - o.px[0]) + (v - o.px[1]) * (v - o.px[1]);
          world2cam(cam, kx + T_cur_kf.tx, ky + T_cur_kf.ty, kz + T_cur_kf.tz, u, v);
          bb = (u - o.px[0]) * (u - o.px[0]) + (v - o.px[1]) * (v - o.px[1]);
        }
        // a feature without a point contributes an exact zero: (float)((double)s + 0.0) == s bit for bit (s >= +0), so the serial
        // lanes below add every slot without a branch (the loads no longer wait for the comparison: 83 -> 20 us per 2000 features)
        // and the features that count are counted in parallel
        const bool valid = a >= 0.0;
        t_full[i] = valid ? a : 0.0; t_shift[i] = valid ? bb : 0.0;
        mine_valid += valid ? 1 : 0;
      }
      __syncthreads();
      if (tid == 0) { float s = flow_full; for (int i = 0; i < m; i++) s = (float)((double)s + t_full[i]); flow_full = s; }
      if (tid == 64) { float s = flow_shift; for (int i = 0; i < m; i++) s = (float)((double)s + t_shift[i]); flow_shift = s; }
      __syncthreads();
    }
    { int tot; (void)sel_block_scan_w<FIN_WAVES>(mine_valid, s_wave, tot); flow_count = tot; __syncthreads(); }
    if (tid == 64) { R.flow_shift = flow_shift; s_flow_shift = flow_shift; }
    if (tid == 0) { s_flow_full = flow_full; s_flow_count = flow_count; }
  } else { if (tid == 64) { R.flow_shift = 0.f; s_flow_shift = 0.f; } if (tid == 0) { s_flow_full = 0.f; s_flow_count = 0; } }
  __syncthreads();
  flow_full = s_flow_full; flow_count = s_flow_count;
  // (4d) getSceneDepth / getSceneDistance — only a keyframe's are ever read (depth_filter_->addKeyframe, src/frame_handler_mono.cpp:
  // 335-338; needNewKf ignores its depth argument): formed when the job says the frame is one for sure (the frame after the
  // initialisation) or the flow criterion says it will be; two 4096-key sorts otherwise saved
  bool want_depth = (J.flags & HSO_SEQ_DEPTH_STATS) != 0;
  bool make_kf = want_depth;
  if (!want_depth && J.last_kf_row >= 0 && flow_count > 0) {
    // needNewKf's test, as the caller evaluates it (:486-506)
    float ff_full = flow_full / (float)flow_count;
    if (!(ff_full < 133.f)) {
      ff_full = sqrtf(ff_full);
      const float ff_shift = sqrtf(s_flow_shift / (float)flow_count);
      const int nominal = 752 + 480;
      const float w_shift = 0.04 * nominal, w_full = 0.02 * nominal, w_global = 0.75;
      const int extent = cam.width + cam.height;
      const float score = w_global * w_shift * ff_shift / extent + w_global * w_full * ff_full / extent;
      want_depth = score > 0.9f;                                   // a margin below the threshold of 1
      make_kf = score > 1;                                          // needNewKf's answer (:506)
    }
  }
  if (want_depth) {
    double zmin = 1.7976931348623157e308;
    for (int pass = 0; pass < 2; pass++) {
      for (int i = tid; i < FIN_SORT_N; i += FIN_THREADS) {
        double key = 1.0 / 0.0;
        if (i < nf && ff[i].point >= 0) {
          const hso_map_point& P = pts[ff[i].point];
          double x, y, z;
          se3_apply(Tc, P.pos[0], P.pos[1], P.pos[2], x, y, z);
          key = pass == 0 ? z : sqrt(x * x + y * y + z * z);
          if (pass == 0) zmin = fmin(z, zmin);
        }
        s_buf[i] = key;
      }
      __syncthreads();
      if (pass == 0) {
        double m = zmin;
        for (int d = 32; d > 0; d >>= 1) m = fmin(m, __shfl_xor(m, d));
        if ((tid & 63) == 0) s_min[tid >> 6] = m;
        __syncthreads();
        zmin = s_min[0];
        for (int w = 1; w < FIN_WAVES; w++) zmin = fmin(zmin, s_min[w]);
      }
      fin_sort(s_buf);
      if (tid == 0) {
        const double med = n_pt > 0 ? s_buf[n_pt / 2] : 0.0;
        if (pass == 0) { R.depth_median = med; R.depth_min = zmin; } else R.dist_median = med;
      }
      __syncthreads();
    }
  } else if (tid == 0) { R.depth_median = 0.0; R.dist_median = 0.0; R.depth_min = -1.0; }   // depth_min < 0: not computed
  // (4e) DepthFilter::addFrame for a regular frame: the sequence's seeds are observed in it (the chain's next kernels), with the
  // pose and exposure the frame ends with — unless processFrame stops before (too few matches / inliers, :224-254), the seed branch
  // takes over, or the frame becomes a keyframe (its local BA moves the pose first: the caller observes after that)
  const bool observe = A.seed_frames && J.seed_group >= 0 && n_matches >= A.quality_min_fts && PR.status == 0 && PR.num_obs >= A.quality_min_fts &&
                       !((J.flags & HSO_SEQ_SEED_BRANCH) && n_matches < 100) && !make_kf;
  if (tid == 0 && observe) {
    SeedFrameDev f;
    f.T_f_w = PR.T_f_w; f.exposure = C.exposure; f.cur_base = J.cur_base;
    A.seed_frames[J.seed_group] = f;
  }
  // (5) the result record
  if (tid == 0) {
    R.make_kf = make_kf ? 1 : 0; R.seeds_observed = observe ? 1 : 0;
    if (J.flags & HSO_SEQ_NO_TRACK) memset(&R.track, 0, sizeof(R.track)); else R.track = A.track[c];
    R.pose = PR;
    R.T_tracked = C.T_cur_w; R.exposure = C.exposure;
    for (int q = 0; q < 4; q++) R.counts[q] = A.counts[4 * c + q];
    R.n_feats = nf; R.n_listed = C.n_listed; R.n_kf_points = C.n_kf_points; R.n_candidates = C.n_cands_listed;
    R.n_visit = C.n_visit;
    for (int q = 0; q < HSO_SEQ_MAX_VISIT; q++) R.visit[q] = q < C.n_visit ? C.visit[q] : -1;
    R.flow_full = flow_full; R.flow_count = flow_count;
    R.n_events = n_ev;
  }
  for (int q = tid; q < HSO_SEQ_EVENTS; q += FIN_THREADS) R.events[q] = q < n_ev && q < J.slice_cap ? ev[q] : 0;
}

namespace {
// where the last chain call left what hso_gpu_seq_events / _debug_list / _debug_ref_table read (valid until the work area is reused)
struct ChainLast {
  hso_gpu_ctx* ctx = nullptr;
  std::vector<int> slice_begin, slice_cap, n_listed, n_events, n_ref, n_ref_stride;
  std::vector<const double*> table;
  const int32_t* d_events = nullptr; const int32_t* d_ids = nullptr; const uint8_t* d_quality = nullptr;
};
std::mutex g_chain_last_mutex;
std::vector<ChainLast> g_chain_last;
ChainLast& chain_last_of(hso_gpu_ctx* ctx)
{
  for (ChainLast& L : g_chain_last) if (L.ctx == ctx) return L;
  g_chain_last.emplace_back();
  g_chain_last.back().ctx = ctx;
  return g_chain_last.back();
}
}  // namespace

void hso_chain_forget(hso_gpu_ctx* ctx)
{
  std::lock_guard<std::mutex> lk(g_chain_last_mutex);
  for (size_t i = 0; i < g_chain_last.size(); i++) if (g_chain_last[i].ctx == ctx) { g_chain_last.erase(g_chain_last.begin() + (std::ptrdiff_t)i); break; }
}

extern "C" int hso_gpu_seq_chain(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_seq_chain_cfg* cfg, const hso_seq_job* jobs, int n_jobs,
                                 const int32_t* temps, int n_temps_total, hso_seq_result* results)
{
  if (!ctx) return HSO_E_INVALID;
  if (!cam || !cfg || n_jobs < 0 || (n_jobs > 0 && (!jobs || !results)) || n_temps_total < 0 || (n_temps_total > 0 && !temps))
    return hso_fail(ctx, HSO_E_INVALID, "seq_chain: bad argument");
  if (cfg->n_cells <= 0 || !cfg->cell_order || cfg->max_fts < 1 || cfg->cell_size < 1 || cfg->grid_n_cols < 1 || cfg->max_kfs < 0 || cfg->pose_n_iter < 0)
    return hso_fail(ctx, HSO_E_INVALID, "seq_chain: bad configuration");
  if (cfg->max_fts > HSO_POSE_MAX_FEATS) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: max_fts above the pose optimiser's table size (4096)");
  {
    std::vector<uint8_t> seen(cfg->n_cells, 0);
    for (int k = 0; k < cfg->n_cells; k++) {
      if (cfg->cell_order[k] < 0 || cfg->cell_order[k] >= cfg->n_cells || seen[cfg->cell_order[k]]) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: cell_order is not a permutation");
      seen[cfg->cell_order[k]] = 1;
    }
  }
  if (n_jobs == 0) return HSO_OK;
  // everything about the chained seed observation is checked before the first kernel is queued: the observation updates the seeds
  // in place, so a refusal after it would report failure with the seeds' state already advanced
  if (cfg->seed_table >= 0) {
    if (cfg->n_seed_groups <= 0 || !cfg->seed_brief_out) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: seed observation without groups or a brief table");
    int n_slots = 0, n_live = 0;
    if (int rc = hso_gpu_seed_table_size(ctx, cfg->seed_table, &n_slots, &n_live)) return rc;
    if (n_slots > cfg->seed_brief_cap) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: seed_brief_out is smaller than the seed table");
  }
  HSO_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int n_cells = cfg->n_cells, max_fts = cfg->max_fts, feat_cap = max_fts;
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  // ---- per job: frames, the map's view, the list slice (as long as the list can get: every point row once, or every feature of
  // max_kfs + 5 keyframes plus candidates and temporary points, whichever is smaller)
  if (int rc = hso_seqmap_flush_kfs(ctx)) return rc;
  for (int c = 0; c < n_jobs; c++) {
    for (int q = 0; q < c; q++) if (jobs[q].map == jobs[c].map) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: a map appears twice in one call");
    if (int rc = hso_seqmap_chain_reserve(ctx, jobs[c].map, feat_cap)) return rc;
  }
  std::vector<ChainJobDev> hj((size_t)n_jobs);
  std::vector<int> kf_begin((size_t)n_jobs + 1, 0);
  std::vector<const int32_t*> nfts_host((size_t)n_jobs);
  size_t total = 0, table_doubles = 0;
  int n_max_stride = 0;
  for (int c = 0; c < n_jobs; c++) {
    const hso_seq_job& jb = jobs[c];
    ChainJobDev& J = hj[(size_t)c];
    memset(&J, 0, sizeof(J));
    int nk = 0;
    if (int rc = hso_seqmap_chain_view(ctx, jb, &J.M, &nk, &nfts_host[(size_t)c])) return rc;
    auto itc = ctx->frames.find(jb.cur_frame_id);
    auto itr = ctx->frames.find(jb.ref_frame_id);
    if (itc == ctx->frames.end() || itr == ctx->frames.end()) return hso_fail(ctx, HSO_E_NOFRAME, "seq_chain: frame not resident");
    if (jb.n_temps < 0 || jb.temps_begin < 0 || jb.temps_begin + jb.n_temps > n_temps_total) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: bad slice of the temporary points");
    for (int q = 0; q < jb.n_temps; q++) if (temps[jb.temps_begin + q] < 0 || temps[jb.temps_begin + q] >= J.M.n_pts) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: temporary point row out of range");
    J.cur_base = itc->second.base;
    J.T_ref_w = jb.T_ref_w; J.T_cur_w0 = jb.T_cur_w; J.ref_exposure = jb.ref_exposure;
    J.ref_kf_row = jb.ref_kf_row; J.n_ref = (jb.flags & HSO_SEQ_NO_TRACK) ? 0 : jb.n_ref_feats; J.n_ref_stride = (J.n_ref + 31) & ~31; J.flags = jb.flags;
    J.cur_keyframe_id = jb.cur_keyframe_id; J.last_kf_row = jb.last_kf_row;
    for (int q = 0; q < 5; q++) J.covis[q] = jb.covis[q];
    size_t by_kfs = (size_t)J.M.n_cands + (size_t)jb.n_temps;
    {
      std::vector<int32_t> len(nfts_host[(size_t)c], nfts_host[(size_t)c] + nk);
      std::sort(len.begin(), len.end(), std::greater<int32_t>());
      for (int q = 0; q < nk && q < cfg->max_kfs + 5; q++) by_kfs += (size_t)len[(size_t)q];
    }
    const size_t cap = std::min((size_t)J.M.n_pts, by_kfs);
    J.slice_begin = (int)total; J.slice_cap = (int)cap;
    total += cap;
    J.temps_begin = jb.temps_begin; J.n_temps = jb.n_temps;
    J.kf_begin = kf_begin[(size_t)c]; kf_begin[(size_t)c + 1] = kf_begin[(size_t)c] + nk;
    J.seed_group = cfg->seed_table >= 0 ? jb.seed_group : -1;
    if (J.seed_group >= cfg->n_seed_groups) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: seed_group beyond n_seed_groups");
    table_doubles += 6 * (size_t)J.n_ref_stride;
    n_max_stride = std::max(n_max_stride, J.n_ref_stride);
    if (J.n_ref == 0) J.flags |= HSO_SEQ_NO_TRACK;
  }
  const size_t total_kfs = (size_t)kf_begin[(size_t)n_jobs];
  const size_t nt = std::max(total, (size_t)1);
  // ---- the work area
  const size_t per_frame = al(sizeof(int32_t) * (size_t)(n_cells + 1)) + 4 * al(sizeof(int32_t) * (size_t)n_cells) + al(sizeof(int32_t) * 3 * (size_t)n_cells) +
                           al(sizeof(int32_t) * (size_t)std::max(n_cells, max_fts + 50));
  size_t o = 0;
  // uploaded image: [jobs | list lengths | temps | cell order | selection frames | pose jobs]
  const size_t o_jobs = o; o += al(sizeof(ChainJobDev) * (size_t)n_jobs);
  const size_t o_nfts = o; o += al(sizeof(int32_t) * std::max(total_kfs, (size_t)1));
  const size_t o_temps = o; o += al(sizeof(int32_t) * (size_t)std::max(n_temps_total, 1));
  const size_t o_order = o; o += al(sizeof(int32_t) * (size_t)n_cells);
  const size_t o_frames = o; o += al(sizeof(SelFrame) * (size_t)n_jobs);
  const size_t o_pj = o; o += al(sizeof(PoseJobDev) * (size_t)n_jobs);
  const size_t o_slices = o; o += al(sizeof(int32_t) * ((size_t)n_jobs + 1));
  const size_t in_bytes = o;
  const size_t o_cur = o; o += al(sizeof(ChainCur) * (size_t)n_jobs);
  const size_t o_table = o; o += al(sizeof(double) * std::max(table_doubles, (size_t)1));
  const size_t o_kfs = o; o += al(hso_chain_sizeof_reproj_kf() * std::max(total_kfs, (size_t)1));
  const size_t o_ids = o; o += al(sizeof(int32_t) * nt);
  const size_t o_lq = o; o += al(nt);
  const size_t o_align = o; o += al(hso_chain_sizeof_align_job() * nt);
  const size_t o_match = o; o += al(sizeof(hso_align_out) * nt);
  const size_t o_proj = o; o += al(sizeof(hso_reproj_point) * nt);
  const size_t o_brief = o; o += al(sizeof(hso_match_brief) * nt);
  const size_t o_cell = o; o += al(sizeof(int32_t) * nt);
  const size_t o_q = o; o += al(nt);
  const size_t o_f = o; o += al(nt);
  const size_t o_pt = o; o += al(sizeof(int32_t) * nt);
  const size_t o_ncand = o; o += al(sizeof(int) * (size_t)n_jobs);
  const size_t o_list = o; o += al(sizeof(int32_t) * nt);
  const size_t o_exam = o; o += al(sizeof(int32_t) * nt);
  const size_t o_counts = o; o += al(sizeof(int32_t) * 4 * (size_t)n_jobs);
  const size_t o_offs = o; o += al(sizeof(int) * (size_t)(n_jobs + 1));
  const size_t o_out = o; o += al(sizeof(hso_match_brief) * nt);
  const size_t o_flag = o; o += al(nt);
  const size_t o_rec = o; o += al(sizeof(hso_frame_match) * nt);
  const size_t o_ev = o; o += al(sizeof(int32_t) * nt);
  const size_t o_scr = o; o += per_frame * (size_t)n_jobs;
  const size_t o_pf = o; o += al(sizeof(hso_pose_feat) * (size_t)n_jobs * feat_cap);
  const size_t o_pp = o; o += al(sizeof(hso_se3) * (size_t)n_jobs * HSO_POSE_MAX_POSES);
  const size_t o_pnp = o; o += al(sizeof(int) * (size_t)n_jobs);
  const size_t o_pr = o; o += al(sizeof(hso_pose_result) * (size_t)n_jobs);
  const size_t o_pk = o; o += al((size_t)n_jobs * feat_cap);
  const size_t o_pn = o; o += al(sizeof(int) * (size_t)n_jobs);
  const size_t o_res = o; o += al(sizeof(hso_seq_result) * (size_t)n_jobs);
  const size_t need = o;
  if (ctx->batch_cap < need) {
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_batch) (void)hipFree(ctx->d_batch);
    ctx->d_batch = nullptr; ctx->batch_cap = 0;
    HSO_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_batch), hso_grown(need)));
    ctx->batch_cap = hso_grown(need);
  }
  char* d = ctx->d_batch;
  // ---- one staging image for everything the kernels read from the caller
  char* h = hso_pinned(ctx, 1, in_bytes);
  if (!h) return HSO_E_NOMEM;
  {
    size_t t_at = 0;
    for (int c = 0; c < n_jobs; c++) { hj[(size_t)c].table = reinterpret_cast<double*>(d + o_table) + t_at; t_at += 6 * (size_t)hj[(size_t)c].n_ref_stride; }
    memcpy(h + o_jobs, hj.data(), sizeof(ChainJobDev) * (size_t)n_jobs);
    int32_t* hn = reinterpret_cast<int32_t*>(h + o_nfts);
    for (int c = 0; c < n_jobs; c++) memcpy(hn + kf_begin[(size_t)c], nfts_host[(size_t)c], sizeof(int32_t) * (size_t)(kf_begin[(size_t)c + 1] - kf_begin[(size_t)c]));
    if (n_temps_total) memcpy(h + o_temps, temps, sizeof(int32_t) * (size_t)n_temps_total);
    memcpy(h + o_order, cfg->cell_order, sizeof(int32_t) * (size_t)n_cells);
    int32_t* hs = reinterpret_cast<int32_t*>(h + o_slices);
    for (int c = 0; c < n_jobs; c++) hs[c] = hj[(size_t)c].slice_begin;
    hs[n_jobs] = (int32_t)total;
    SelFrame* hf = reinterpret_cast<SelFrame*>(h + o_frames);
    PoseJobDev* pj = reinterpret_cast<PoseJobDev*>(h + o_pj);
    for (int c = 0; c < n_jobs; c++) {
      char* sc = d + o_scr + per_frame * (size_t)c;
      SelFrame& F = hf[c];
      F.first = hj[(size_t)c].slice_begin; F.n = 0; F.n_dev = reinterpret_cast<const int*>(d + o_ncand) + c;
      F.cnt = reinterpret_cast<int*>(sc); sc += al(sizeof(int32_t) * (size_t)(n_cells + 1));
      F.fill = reinterpret_cast<int*>(sc); sc += al(sizeof(int32_t) * (size_t)n_cells);
      F.e1 = reinterpret_cast<int*>(sc); sc += al(sizeof(int32_t) * (size_t)n_cells);
      F.e2 = reinterpret_cast<int*>(sc); sc += al(sizeof(int32_t) * (size_t)n_cells);
      F.a3 = reinterpret_cast<int*>(sc); sc += al(sizeof(int32_t) * (size_t)n_cells);
      F.p3 = reinterpret_cast<int*>(sc); sc += al(sizeof(int32_t) * 3 * (size_t)n_cells);
      F.scan = reinterpret_cast<int*>(sc);
      F.list = reinterpret_cast<int*>(d + o_list) + F.first;
      F.out = reinterpret_cast<int*>(d + o_exam) + F.first;
      F.counts = reinterpret_cast<int*>(d + o_counts) + 4 * c;
      pj[c].feats = reinterpret_cast<const hso_pose_feat*>(d + o_pf) + (size_t)c * feat_cap;
      pj[c].poses = reinterpret_cast<const hso_se3*>(d + o_pp) + (size_t)c * HSO_POSE_MAX_POSES;
      pj[c].mask = reinterpret_cast<uint8_t*>(d + o_pk) + (size_t)c * feat_cap;
      pj[c].n_feats = 0; pj[c].n_poses = 0; pj[c].T = jobs[c].T_cur_w; pj[c].reproj_thresh = cfg->pose_reproj_thresh;
      pj[c].n_iter = cfg->pose_n_iter; pj[c]._pad = 0;
    }
  }
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  const ChainJobDev* d_jobs = reinterpret_cast<const ChainJobDev*>(d + o_jobs);
  ChainCur* d_cur = reinterpret_cast<ChainCur*>(d + o_cur);
  // ---- (1) the reference frames' feature tables, CoarseTracker::run
  if (int rc = hso_chain_table_launch(ctx, d_jobs, n_jobs, n_max_stride)) return rc;
  const hso_track_result* d_track = nullptr;
  {
    std::vector<hso_track_job> tj; std::vector<int> who;
    for (int c = 0; c < n_jobs; c++) {
      const ChainJobDev& J = hj[(size_t)c];
      if (J.flags & HSO_SEQ_NO_TRACK) continue;
      hso_track_job t{};
      t.ref_frame_id = jobs[c].ref_frame_id; t.cur_frame_id = jobs[c].cur_frame_id;
      t.feats = reinterpret_cast<const hso_ref_feat*>(J.table); t.n_feats = J.n_ref; t.feats_soa = 2;
      Se3 Tcr = se3_mul(se3_from(jobs[c].T_cur_w), se3_inverse(se3_from(jobs[c].T_ref_w)));   // src/CoarseTracker.cpp:63
      se3_to(Tcr, t.T_cur_ref);
      t.exposure_rat = jobs[c].exposure_rat;
      tj.push_back(t); who.push_back(c);
    }
    // the tracker's result records are indexed by ITS job order: jobs it does not run (no reference features) must come last
    for (size_t i = 0; i < who.size(); i++) if (who[i] != (int)i) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: jobs without reference features (HSO_SEQ_NO_TRACK) must follow the others");
    if (!tj.empty()) {
      bool coop = false;
      if (int rc = hso_track_chain_launch(ctx, cam, &cfg->track, tj.data(), (int)tj.size(), &d_track, &coop)) return rc;
    }
  }
  // ---- (2) + (3) visiting order, list, projection, matching
  ChainFront A;
  A.d_jobs = d_jobs; A.d_cur = d_cur; A.d_track = d_track; A.d_kf_nfts = reinterpret_cast<const int32_t*>(d + o_nfts); A.d_temps = reinterpret_cast<const int32_t*>(d + o_temps);
  A.d_kfs = reinterpret_cast<ReprojKf*>(d + o_kfs); A.d_ids = reinterpret_cast<int32_t*>(d + o_ids); A.d_quality = reinterpret_cast<uint8_t*>(d + o_lq);
  A.d_align = reinterpret_cast<AlignJobDev*>(d + o_align); A.d_match = reinterpret_cast<hso_align_out*>(d + o_match); A.d_proj = reinterpret_cast<hso_reproj_point*>(d + o_proj);
  A.d_brief = reinterpret_cast<hso_match_brief*>(d + o_brief); A.d_pose_jobs = reinterpret_cast<PoseJobDev*>(d + o_pj);
  A.n_jobs = n_jobs; A.n_total = (int)total; A.max_kfs = cfg->max_kfs; A.cell_size = cfg->cell_size; A.grid_n_cols = cfg->grid_n_cols;
  if (int rc = hso_chain_front_launch(ctx, cam, A)) return rc;
  // ---- (3) selection, the frame's features, pose optimisation
  hipLaunchKernelGGL(k_sel_gather, dim3(n_jobs), dim3(SEL_THREADS), 0, ctx->stream, A.d_proj, A.d_brief, d_jobs, d_cur,
                     reinterpret_cast<int32_t*>(d + o_cell), reinterpret_cast<uint8_t*>(d + o_q), reinterpret_cast<uint8_t*>(d + o_f),
                     reinterpret_cast<int32_t*>(d + o_pt), reinterpret_cast<int*>(d + o_ncand), reinterpret_cast<uint8_t*>(d + o_flag));
  SelArgs SA;
  SA.cell = reinterpret_cast<const int32_t*>(d + o_cell); SA.quality = reinterpret_cast<const uint8_t*>(d + o_q);
  SA.flags = reinterpret_cast<const uint8_t*>(d + o_f); SA.cell_order = reinterpret_cast<const int32_t*>(d + o_order);
  SA.n_cells = n_cells; SA.max_fts = max_fts;
  hipLaunchKernelGGL(k_select, dim3(n_jobs), dim3(SEL_THREADS), 0, ctx->stream, SA, reinterpret_cast<const SelFrame*>(d + o_frames));
  HSO_HIP_CHECK(ctx, hipMemsetAsync(d + o_pk, 0, (size_t)n_jobs * feat_cap, ctx->stream));
  {
    EmitArgs E;
    E.brief = A.d_brief; E.examined = reinterpret_cast<const int32_t*>(d + o_exam); E.cand_pt = reinterpret_cast<const int32_t*>(d + o_pt);
    E.counts = reinterpret_cast<const int*>(d + o_counts); E.offs = reinterpret_cast<int*>(d + o_offs); E.out = reinterpret_cast<hso_match_brief*>(d + o_out);
    E.records = reinterpret_cast<hso_frame_match*>(d + o_rec); E.n_calls = n_jobs;
    hipLaunchKernelGGL(k_sel_emit_feats, dim3(n_jobs), dim3(SEL_THREADS), 0, ctx->stream, E, *cam, d_jobs, A.d_ids, A.d_quality,
                       feat_cap, reinterpret_cast<hso_pose_feat*>(d + o_pf), reinterpret_cast<PoseJobDev*>(d + o_pj), reinterpret_cast<hso_se3*>(d + o_pp),
                       reinterpret_cast<int*>(d + o_pnp), reinterpret_cast<int*>(d + o_pn));
    HSO_HIP_CHECK(ctx, hipGetLastError());
    if (int rc = hso_pose_launch_device(ctx, cam, reinterpret_cast<const PoseJobDev*>(d + o_pj), n_jobs, feat_cap,
                                        reinterpret_cast<hso_pose_result*>(d + o_pr))) return rc;
  }
  // ---- (4) + (5) bookkeeping, decision inputs, result records; the regular frames' seed observation behind them
  SeedFrameDev* d_seed_frames = nullptr;
  if (cfg->seed_table >= 0) {
    if (cfg->n_seed_groups <= 0 || !cfg->seed_brief_out) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: seed observation without groups or a brief table");
    if (int rc = hso_seed_table_chain_frames(ctx, cfg->seed_table, cfg->n_seed_groups, &d_seed_frames)) return rc;
  }
  {
    FinishArgs Fa;
    Fa.seed_frames = d_seed_frames;
    Fa.jobs = d_jobs; Fa.cur = d_cur; Fa.track = d_track; Fa.pose = reinterpret_cast<const hso_pose_result*>(d + o_pr);
    Fa.counts = reinterpret_cast<const int*>(d + o_counts); Fa.offs = reinterpret_cast<const int*>(d + o_offs);
    Fa.records = reinterpret_cast<const hso_frame_match*>(d + o_rec); Fa.ids = A.d_ids; Fa.projected = reinterpret_cast<const uint8_t*>(d + o_flag);
    Fa.mask = reinterpret_cast<const uint8_t*>(d + o_pk); Fa.n_feats = reinterpret_cast<const int*>(d + o_pn); Fa.kf_nfts = A.d_kf_nfts;
    Fa.events = reinterpret_cast<int32_t*>(d + o_ev); Fa.results = reinterpret_cast<hso_seq_result*>(d + o_res);
    Fa.feat_cap = feat_cap; Fa.quality_min_fts = cfg->quality_min_fts;
    hipLaunchKernelGGL(k_chain_finish, dim3(n_jobs), dim3(FIN_THREADS), 0, ctx->stream, Fa, *cam);
    HSO_HIP_CHECK(ctx, hipGetLastError());
  }
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(results, d + o_res, sizeof(hso_seq_result) * (size_t)n_jobs, hipMemcpyDeviceToHost, ctx->stream));
  if (cfg->seed_table >= 0) {
    const hso_seed_brief* d_brief = nullptr; int n_slots = 0;
    if (int rc = hso_seed_table_chain_launch(ctx, cam, cfg->seed_table, cfg->n_seed_groups, cfg->px_error_angle, &d_brief, &n_slots)) return rc;
    if (n_slots > cfg->seed_brief_cap) return hso_fail(ctx, HSO_E_INVALID, "seq_chain: seed_brief_out is smaller than the seed table");
    if (n_slots > 0 && d_brief) HSO_HIP_CHECK(ctx, hipMemcpyAsync(cfg->seed_brief_out, d_brief, sizeof(hso_seed_brief) * (size_t)n_slots, hipMemcpyDeviceToHost, ctx->stream));
  }
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int c = 0; c < n_jobs; c++) hso_seqmap_chain_commit(ctx, jobs[c], results[c].n_feats);
  {
    std::lock_guard<std::mutex> lk(g_chain_last_mutex);
    ChainLast& L = chain_last_of(ctx);
    L.slice_begin.resize((size_t)n_jobs); L.slice_cap.resize((size_t)n_jobs); L.n_listed.resize((size_t)n_jobs); L.n_events.resize((size_t)n_jobs);
    L.n_ref.resize((size_t)n_jobs); L.n_ref_stride.resize((size_t)n_jobs); L.table.resize((size_t)n_jobs);
    for (int c = 0; c < n_jobs; c++) {
      const ChainJobDev& J = hj[(size_t)c];
      L.slice_begin[(size_t)c] = J.slice_begin; L.slice_cap[(size_t)c] = J.slice_cap; L.n_listed[(size_t)c] = results[c].n_listed; L.n_events[(size_t)c] = results[c].n_events;
      L.n_ref[(size_t)c] = J.n_ref; L.n_ref_stride[(size_t)c] = J.n_ref_stride; L.table[(size_t)c] = J.table;
    }
    L.d_events = reinterpret_cast<const int32_t*>(d + o_ev); L.d_ids = A.d_ids; L.d_quality = A.d_quality;
  }
  if (cfg->want_debug) {
    hso_seqmaps_debug_set(ctx, HSO_DBG_SLICES, d + o_slices, sizeof(int32_t) * ((size_t)n_jobs + 1));
    hso_seqmaps_debug_set(ctx, HSO_DBG_PROJ, A.d_proj, sizeof(hso_reproj_point) * total);
    hso_seqmaps_debug_set(ctx, HSO_DBG_MATCH, A.d_match, sizeof(hso_align_out) * total);
    hso_seqmaps_debug_set(ctx, HSO_DBG_PROJECTED, d + o_flag, total);
    hso_seqmaps_debug_set(ctx, HSO_DBG_EXAMINED_BEGIN, d + o_offs, sizeof(int32_t) * ((size_t)n_jobs + 1));
    size_t n_exam = 0;
    for (int c = 0; c < n_jobs; c++) n_exam += (size_t)results[c].counts[0];
    hso_seqmaps_debug_set(ctx, HSO_DBG_BRIEF, d + o_out, sizeof(hso_match_brief) * n_exam);
    hso_seqmaps_debug_set(ctx, HSO_DBG_POSE_FEATS, d + o_pf, sizeof(hso_pose_feat) * (size_t)n_jobs * feat_cap);
    hso_seqmaps_debug_set(ctx, HSO_DBG_POSE_POSES, d + o_pp, sizeof(hso_se3) * (size_t)n_jobs * HSO_POSE_MAX_POSES);
    hso_seqmaps_debug_set(ctx, HSO_DBG_POSE_NPOSES, d + o_pnp, sizeof(int) * (size_t)n_jobs);
    hso_seqmaps_debug_set(ctx, HSO_DBG_POSE_MASK, d + o_pk, (size_t)n_jobs * feat_cap);
  }
  return HSO_OK;
}

extern "C" int hso_gpu_seq_events(hso_gpu_ctx* ctx, int job, int32_t* events_out, int cap)
{
  if (!ctx) return HSO_E_INVALID;
  int n = 0; const int32_t* src = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_chain_last_mutex);
    ChainLast& L = chain_last_of(ctx);
    if (job < 0 || (size_t)job >= L.n_events.size() || !events_out) return hso_fail(ctx, HSO_E_INVALID, "seq_events: no such job in the last chain call");
    n = std::min(L.n_events[(size_t)job], L.slice_cap[(size_t)job]);
    src = L.d_events + L.slice_begin[(size_t)job];
  }
  if (n > cap) return hso_fail(ctx, HSO_E_INVALID, "seq_events: cap is smaller than the number of events");
  if (n == 0) return 0;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(events_out, src, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return n;
}

extern "C" int hso_gpu_seq_debug_list(hso_gpu_ctx* ctx, int job, int32_t* ids_out, uint8_t* quality_out, int cap)
{
  if (!ctx) return HSO_E_INVALID;
  int n = 0; const int32_t* si = nullptr; const uint8_t* sq = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_chain_last_mutex);
    ChainLast& L = chain_last_of(ctx);
    if (job < 0 || (size_t)job >= L.n_listed.size() || !ids_out || !quality_out) return hso_fail(ctx, HSO_E_INVALID, "seq_debug_list: no such job in the last chain call");
    n = L.n_listed[(size_t)job]; si = L.d_ids + L.slice_begin[(size_t)job]; sq = L.d_quality + L.slice_begin[(size_t)job];
  }
  if (n > cap) return hso_fail(ctx, HSO_E_INVALID, "seq_debug_list: cap is smaller than the list");
  if (n == 0) return 0;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(ids_out, si, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(quality_out, sq, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return n;
}

extern "C" int hso_gpu_seq_debug_ref_table(hso_gpu_ctx* ctx, int job, hso_ref_feat* out, int cap)
{
  if (!ctx) return HSO_E_INVALID;
  int n = 0, ns = 0; const double* src = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_chain_last_mutex);
    ChainLast& L = chain_last_of(ctx);
    if (job < 0 || (size_t)job >= L.n_ref.size() || !out) return hso_fail(ctx, HSO_E_INVALID, "seq_debug_ref_table: no such job in the last chain call");
    n = L.n_ref[(size_t)job]; ns = L.n_ref_stride[(size_t)job]; src = L.table[(size_t)job];
  }
  if (n > cap) return hso_fail(ctx, HSO_E_INVALID, "seq_debug_ref_table: cap is smaller than the table");
  if (n == 0) return 0;
  std::vector<double> t(6 * (size_t)ns);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(t.data(), src, sizeof(double) * t.size(), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < n; i++) { out[i].px[0] = t[(size_t)i]; out[i].px[1] = t[(size_t)ns + i]; out[i].f[0] = t[2 * (size_t)ns + i]; out[i].f[1] = t[3 * (size_t)ns + i]; out[i].f[2] = t[4 * (size_t)ns + i]; out[i].dist = t[5 * (size_t)ns + i]; }
  return n;
}
