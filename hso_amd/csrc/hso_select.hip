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
#include <vector>

using namespace hso_dev;

#define SEL_THREADS 256
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

// inclusive scan of v over the workgroup (SEL_THREADS values); *total = sum
__device__ int sel_block_scan(int v, int* s_wave, int& total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
  __syncthreads();
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0; total = 0;
  for (int w = 0; w < SEL_WAVES; w++) { if (w < wave) base += s_wave[w]; total += s_wave[w]; }
  return incl + base;
}

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


// ------------------------------------------------------------------------------------------------ chained behind the stored maps

// the projected points of one call, in point order, as the candidate tables of k_select; cand_pt = candidate -> record
__global__ __launch_bounds__(SEL_THREADS) void k_sel_gather(const hso_reproj_point* proj, const hso_match_brief* brief, const int* begin,
                                                           int32_t* cell, uint8_t* quality, uint8_t* flags, int32_t* cand_pt, int* n_cand,
                                                           uint8_t* projected_out)
{
  __shared__ int s_wave[SEL_WAVES];
  const int c = blockIdx.x, b = begin[c], e = begin[c + 1];
  int carry = 0;
  for (int i0 = b; i0 < e; i0 += SEL_THREADS) {
    const int i = i0 + (int)threadIdx.x;
    const int is = (i < e && proj[i].projected) ? 1 : 0;
    if (projected_out && i < e) projected_out[i] = (uint8_t)is;   // reprojectPoint's return value per listed point, for the caller
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

// where each call's examined records start in the packed output, and their total
__global__ void k_sel_offsets(int n_calls, const int* counts, int* offs)
{
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int t = 0;
  for (int c = 0; c < n_calls; c++) { offs[c] = t; t += counts[4 * c]; }
  offs[n_calls] = t;
}

__global__ __launch_bounds__(SEL_THREADS) void k_sel_emit(const hso_match_brief* brief, const int* begin, const int32_t* examined,
                                                         const int32_t* cand_pt, const int* counts, const int* offs, hso_match_brief* out,
                                                         hso_frame_match* records = nullptr)
{
  const int c = blockIdx.x, b = begin[c], n_ex = counts[4 * c], o = offs[c];
  for (int k = threadIdx.x; k < n_ex; k += SEL_THREADS) {
    const int v = examined[b + k];
    const int g = cand_pt[b + (v & 0x7fffffff)];
    hso_match_brief r = brief[g];
    r.success = (v < 0) ? 1 : 0;          // became a feature (a matched candidate the budget never reached stays 0)
    r.pad_ = g - b;                       // the point's index in its map
    out[o + k] = r;
    if (records) {
      hso_frame_match m;
      m.px_cur[0] = r.px_cur[0]; m.px_cur[1] = r.px_cur[1]; m.grad[0] = r.grad[0]; m.grad[1] = r.grad[1];
      m.point = r.pad_; m.success = r.success; m.search_level = r.search_level; m.ref_type = r.ref_type; m.pad_ = 0;
      records[o + k] = m;
    }
  }
}


// The frame's features as pose_optimizer sees them, built where the selection left its result: every examined candidate that
// became a feature (success), in examination order = the order Reprojector::reprojectCell pushes `new Feature` into fts_
// (src/reprojector.cpp:395-425): f = cam->cam2world(px_cur), level = the matcher's search level, type = the reference
// feature's, grad = the rotated reference gradient, the point's host bearing / inverse depth / host keyframe from the stored
// map, temporary = Point::TYPE_TEMPORARY (quality key >> 4 == 1).  One workgroup per call; ranks by a block scan.
__global__ __launch_bounds__(SEL_THREADS) void k_pose_feats_from_sel(hso_camera cam, const hso_match_brief* out, const int* offs, const int* maps,
                                                                    const hso_map_point* pts, int max_points, int feat_cap,
                                                                    hso_pose_feat* feats, PoseJobDev* jobs, uint8_t* masks, int* n_feats)
{
  __shared__ int s_wave[SEL_WAVES];
  const int c = blockIdx.x, b = offs[c], e = offs[c + 1];
  const hso_map_point* P = pts + (size_t)maps[c] * max_points;
  hso_pose_feat* F = feats + (size_t)c * feat_cap;
  int carry = 0;
  for (int i0 = b; i0 < e; i0 += SEL_THREADS) {
    const int i = i0 + (int)threadIdx.x;
    const int is = (i < e && out[i].success) ? 1 : 0;
    int tot;
    const int pos = sel_block_scan(is, s_wave, tot) + carry - is;
    if (is && pos < feat_cap) {
      const hso_match_brief& r = out[i];
      const hso_map_point& p = P[r.pad_];
      hso_pose_feat f;
      f.has_point = 1; f.type = r.ref_type; f.level = r.search_level; f.temporary = ((p.pad_ >> 4) == 1) ? 1 : 0;
      f.host_pose = p.host_kf; f._pad = 0;
      hso_dev::cam2world_dev(cam, r.px_cur[0], r.px_cur[1], f.f);
      f.grad[0] = (double)r.grad[0]; f.grad[1] = (double)r.grad[1];
      f.host_f[0] = p.host_f[0]; f.host_f[1] = p.host_f[1]; f.host_f[2] = p.host_f[2];
      f.idist = p.idist;
      F[pos] = f;
    }
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) { const int n = carry < feat_cap ? carry : feat_cap; jobs[c].n_feats = n; n_feats[c] = n; }
}

static int reproject_select_maps_impl(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_map_call* calls, int n_calls, int cell_size,
                                             int grid_n_cols, const int32_t* cell_order, int n_cells, int max_fts, hso_match_brief* out,
                                             int out_capacity, int32_t* begin_out, int32_t* counts_out, const hso_pose_chain* pose)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_calls < 0 || n_cells <= 0 || !cell_order || max_fts < 0 || (n_calls > 0 && (!begin_out || !counts_out)))
    return hso_fail(ctx, HSO_E_INVALID, "reproject_select_maps: bad argument");
  if (pose && (n_calls > 0 && (!pose->results || pose->n_iter < 0))) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_maps: bad pose argument");
  if (pose && max_fts > HSO_POSE_MAX_FEATS) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_maps: max_fts above the pose optimiser's table size (4096)");
  {
    std::vector<uint8_t> seen(n_cells, 0);
    for (int k = 0; k < n_cells; k++) {
      if (cell_order[k] < 0 || cell_order[k] >= n_cells || seen[cell_order[k]]) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_maps: cell_order is not a permutation");
      seen[cell_order[k]] = 1;
    }
  }
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  // the points of all calls are known on the host (stored maps): size the selection scratch before the launch chain
  MapArenaSizes Z;
  if (int rc = hso_map_call_sizes(ctx, calls, n_calls, &Z)) return rc;
  const size_t n_total = (size_t)Z.total;
  if (n_total == 0) { for (int c = 0; c <= n_calls; c++) if (begin_out) begin_out[c] = 0; for (int c = 0; c < 4 * n_calls; c++) counts_out[c] = 0; return 0; }
  const size_t per_frame = al(sizeof(int32_t) * (size_t)(n_cells + 1)) + 4 * al(sizeof(int32_t) * (size_t)n_cells) + al(sizeof(int32_t) * 3 * (size_t)n_cells) +
                           al(sizeof(int32_t) * (size_t)std::max(n_cells, max_fts + 50));
  size_t o = 0;
  const size_t o_begin = o; o += al(sizeof(int) * (size_t)(n_calls + 1));
  const size_t o_order = o; o += al(sizeof(int32_t) * (size_t)n_cells);
  const size_t o_frames = o; o += al(sizeof(SelFrame) * (size_t)n_calls);
  const size_t in_bytes = o;
  const size_t o_cell = o; o += al(sizeof(int32_t) * n_total);
  const size_t o_q = o; o += al(n_total);
  const size_t o_f = o; o += al(n_total);
  const size_t o_pt = o; o += al(sizeof(int32_t) * n_total);
  const size_t o_ncand = o; o += al(sizeof(int) * (size_t)n_calls);
  const size_t o_list = o; o += al(sizeof(int32_t) * n_total);
  const size_t o_exam = o; o += al(sizeof(int32_t) * n_total);
  const size_t o_counts = o; o += al(sizeof(int32_t) * 4 * (size_t)n_calls);
  const size_t o_offs = o; o += al(sizeof(int) * (size_t)(n_calls + 1));
  const size_t o_out = o; o += al(sizeof(hso_match_brief) * n_total);
  const size_t o_scr = o; o += per_frame * (size_t)n_calls;
  // the chained pose optimisation: feature tables, job records, keyframe poses, results, cull masks
  const int feat_cap = std::max(max_fts, 1);
  MapArena* const A_ = ctx->maps;
  const int max_kfs = pose ? hso_map_max_kfs(ctx) : 0;
  const size_t o_pf = o; if (pose) o += al(sizeof(hso_pose_feat) * (size_t)n_calls * feat_cap);
  const size_t o_pj = o; if (pose) o += al(sizeof(PoseJobDev) * (size_t)n_calls);
  const size_t o_pp = o; if (pose) o += al(sizeof(hso_se3) * (size_t)n_calls * max_kfs);
  const size_t o_pm = o; if (pose) o += al(sizeof(int) * (size_t)n_calls);
  const size_t o_pr = o; if (pose) o += al(sizeof(hso_pose_result) * (size_t)n_calls);
  const size_t o_pk = o; if (pose) o += al((size_t)n_calls * feat_cap);
  const size_t o_pn = o; if (pose) o += al(sizeof(int) * (size_t)n_calls);
  (void)A_;
  HsoMapsRun R;
  const int total = hso_reproject_maps_run(ctx, cam, calls, n_calls, cell_size, grid_n_cols, o, &R);
  if (total < 0) return total;
  char* d = R.d_extra;
  // staging: hso_reproject_maps_run used pinned slot 0 for its own tables and its copy may still be in flight: use slot 1
  char* h = hso_pinned(ctx, 1, std::max(in_bytes, sizeof(int32_t) * (5 * (size_t)n_calls + 2)));
  if (!h) return HSO_E_NOMEM;
  int* hb = reinterpret_cast<int*>(h + o_begin);
  for (int c = 0; c <= n_calls; c++) hb[c] = R.begin[c];
  memcpy(h + o_order, cell_order, sizeof(int32_t) * (size_t)n_cells);
  SelFrame* hf = reinterpret_cast<SelFrame*>(h + o_frames);
  for (int c = 0; c < n_calls; c++) {
    char* sc = d + o_scr + per_frame * (size_t)c;
    SelFrame& F = hf[c];
    F.first = R.begin[c]; F.n = 0; F.n_dev = reinterpret_cast<const int*>(d + o_ncand) + c;
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
  }
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  const int* d_begin = reinterpret_cast<const int*>(d + o_begin);
  hipLaunchKernelGGL(k_sel_gather, dim3(n_calls), dim3(SEL_THREADS), 0, ctx->stream, R.d_proj, R.d_brief, d_begin,
                     reinterpret_cast<int32_t*>(d + o_cell), reinterpret_cast<uint8_t*>(d + o_q), reinterpret_cast<uint8_t*>(d + o_f),
                     reinterpret_cast<int32_t*>(d + o_pt), reinterpret_cast<int*>(d + o_ncand), static_cast<uint8_t*>(nullptr));
  SelArgs A;
  A.cell = reinterpret_cast<const int32_t*>(d + o_cell); A.quality = reinterpret_cast<const uint8_t*>(d + o_q);
  A.flags = reinterpret_cast<const uint8_t*>(d + o_f); A.cell_order = reinterpret_cast<const int32_t*>(d + o_order);
  A.n_cells = n_cells; A.max_fts = max_fts;
  hipLaunchKernelGGL(k_select, dim3(n_calls), dim3(SEL_THREADS), 0, ctx->stream, A, reinterpret_cast<const SelFrame*>(d + o_frames));
  hipLaunchKernelGGL(k_sel_offsets, dim3(1), dim3(64), 0, ctx->stream, n_calls, reinterpret_cast<const int*>(d + o_counts), reinterpret_cast<int*>(d + o_offs));
  hipLaunchKernelGGL(k_sel_emit, dim3(n_calls), dim3(SEL_THREADS), 0, ctx->stream, R.d_brief, d_begin, reinterpret_cast<const int32_t*>(d + o_exam),
                     reinterpret_cast<const int32_t*>(d + o_pt), reinterpret_cast<const int*>(d + o_counts), reinterpret_cast<const int*>(d + o_offs),
                     reinterpret_cast<hso_match_brief*>(d + o_out));
  HSO_HIP_CHECK(ctx, hipGetLastError());
  if (pose) {
    // job records (poses = the stored maps' keyframe poses, start = the call's pose) and the map index per call: one small
    // upload; the feature tables are built on the device from the records k_sel_emit just wrote
    const size_t b_pj = al(sizeof(PoseJobDev) * (size_t)n_calls), b_pp = al(sizeof(hso_se3) * (size_t)n_calls * max_kfs), b_pm = al(sizeof(int) * (size_t)n_calls);
    char* hp = hso_pinned(ctx, 0, b_pj + b_pp + b_pm);      // slot 0: hso_reproject_maps_run's upload has been enqueued before its kernels above
    if (!hp) return HSO_E_NOMEM;
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // ... and must have left the staging buffer before it is rewritten
    PoseJobDev* pj = reinterpret_cast<PoseJobDev*>(hp);
    hso_se3* pp = reinterpret_cast<hso_se3*>(hp + b_pj);
    int* pm = reinterpret_cast<int*>(hp + b_pj + b_pp);
    for (int c = 0; c < n_calls; c++) {
      const int nk = hso_map_kf_poses(ctx, calls[c].map, pp + (size_t)c * max_kfs);
      pj[c].feats = reinterpret_cast<const hso_pose_feat*>(d + o_pf) + (size_t)c * feat_cap;
      pj[c].poses = reinterpret_cast<const hso_se3*>(d + o_pp) + (size_t)c * max_kfs;
      pj[c].mask = reinterpret_cast<uint8_t*>(d + o_pk) + (size_t)c * feat_cap;
      pj[c].n_feats = 0; pj[c].n_poses = nk; pj[c].T = calls[c].T_cur_w; pj[c].reproj_thresh = pose->reproj_thresh;
      pj[c].n_iter = pose->n_iter; pj[c]._pad = 0;
      pm[c] = calls[c].map;
    }
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_pj, hp, b_pj, hipMemcpyHostToDevice, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_pp, hp + b_pj, b_pp, hipMemcpyHostToDevice, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_pm, hp + b_pj + b_pp, b_pm, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_pose_feats_from_sel, dim3(n_calls), dim3(SEL_THREADS), 0, ctx->stream, *cam, reinterpret_cast<const hso_match_brief*>(d + o_out),
                       reinterpret_cast<const int*>(d + o_offs), reinterpret_cast<const int*>(d + o_pm), hso_map_points_dev(ctx), hso_map_max_points(ctx),
                       feat_cap, reinterpret_cast<hso_pose_feat*>(d + o_pf), reinterpret_cast<PoseJobDev*>(d + o_pj),
                       reinterpret_cast<uint8_t*>(d + o_pk), reinterpret_cast<int*>(d + o_pn));
    if (int rc = hso_pose_launch_device(ctx, cam, reinterpret_cast<const PoseJobDev*>(d + o_pj), n_calls, feat_cap,
                                        reinterpret_cast<hso_pose_result*>(d + o_pr))) return rc;
    char* hr = hso_pinned(ctx, 0, al(sizeof(hso_pose_result) * (size_t)n_calls) + al(sizeof(int) * (size_t)n_calls) + (size_t)n_calls * feat_cap);
    if (!hr) return HSO_E_NOMEM;
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // the uploads above have left slot 0
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(hr, d + o_pr, sizeof(hso_pose_result) * (size_t)n_calls, hipMemcpyDeviceToHost, ctx->stream));
    char* hn = hr + al(sizeof(hso_pose_result) * (size_t)n_calls);
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(hn, d + o_pn, sizeof(int) * (size_t)n_calls, hipMemcpyDeviceToHost, ctx->stream));
    char* hk = hn + al(sizeof(int) * (size_t)n_calls);
    if (pose->outlier_mask) HSO_HIP_CHECK(ctx, hipMemcpyAsync(hk, d + o_pk, (size_t)n_calls * feat_cap, hipMemcpyDeviceToHost, ctx->stream));
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(pose->results, hr, sizeof(hso_pose_result) * (size_t)n_calls);
    if (pose->n_feats) memcpy(pose->n_feats, hn, sizeof(int) * (size_t)n_calls);
    if (pose->outlier_mask) memcpy(pose->outlier_mask, hk, (size_t)n_calls * feat_cap);
  }
  // counts and offsets first (small), then exactly the examined records
  int32_t* hs = reinterpret_cast<int32_t*>(h);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(hs, d + o_counts, sizeof(int32_t) * 4 * (size_t)n_calls, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(hs + 4 * n_calls, d + o_offs, sizeof(int) * (size_t)(n_calls + 1), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(counts_out, hs, sizeof(int32_t) * 4 * (size_t)n_calls);
  memcpy(begin_out, hs + 4 * n_calls, sizeof(int32_t) * (size_t)(n_calls + 1));
  const int n_out = begin_out[n_calls];
  if (n_out > 0) {
    if (!out || out_capacity < n_out) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_maps: output smaller than the examined candidates");
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, d + o_out, sizeof(hso_match_brief) * (size_t)n_out, hipMemcpyDeviceToHost, ctx->stream));
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return n_out;
}

extern "C" int hso_gpu_reproject_select_maps(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_map_call* calls, int n_calls, int cell_size,
                                             int grid_n_cols, const int32_t* cell_order, int n_cells, int max_fts, hso_match_brief* out,
                                             int out_capacity, int32_t* begin_out, int32_t* counts_out)
{
  return reproject_select_maps_impl(ctx, cam, calls, n_calls, cell_size, grid_n_cols, cell_order, n_cells, max_fts, out, out_capacity, begin_out,
                                    counts_out, nullptr);
}

extern "C" int hso_gpu_reproject_select_pose_maps(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_map_call* calls, int n_calls, int cell_size,
                                                  int grid_n_cols, const int32_t* cell_order, int n_cells, int max_fts, hso_match_brief* out,
                                                  int out_capacity, int32_t* begin_out, int32_t* counts_out, const hso_pose_chain* pose)
{
  if (!pose) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_maps: null pose argument");
  return reproject_select_maps_impl(ctx, cam, calls, n_calls, cell_size, grid_n_cols, cell_order, n_cells, max_fts, out, out_capacity, begin_out,
                                    counts_out, pose);
}


// ------------------------------------------------------------------------------------------------ chained behind the sequence maps
// hso_gpu_reproject_select_pose_frames: the same chain over the points each frame LISTS (hso_align.hip: hso_reproject_frames_run).
// What differs behind the selection: the feature table takes the point row through the frame's id list, the quality key from
// the list's key array, and the pose job's keyframe table is compacted here to the keyframes that host a selected feature (a
// sequence map holds every keyframe of its sequence; k_pose keeps HSO_POSE_MAX_POSES transforms in LDS).
struct PoseSrcDev {
  const hso_map_point* pts;     // the frame's map
  const hso_se3* kf_poses;      // its keyframe poses (map order)
  int list_begin, n_kfs;
};

#define SEL_KF_WORDS 64         // 4096 keyframes per sequence map take part in the compaction bitmask

// One workgroup per frame finishes what the selection left: (1) where the frame's examined records start in the packed output —
// the sum of the earlier frames' counts, a few hundred integers at most, so every workgroup forms its own instead of waiting for a
// one-thread prefix kernel; (2) the examined candidates' records, packed (k_sel_emit's work); (3) the frame's pose-optimisation
// feature table from the records it has just written.  Three launches (offsets, emit, features) were ~50 us of a 1.5 ms call.
struct EmitArgs {
  const hso_match_brief* brief; const int* begin; const int32_t* examined; const int32_t* cand_pt; const int* counts;
  int* offs; hso_match_brief* out; hso_frame_match* records; int n_calls;
};

__global__ __launch_bounds__(SEL_THREADS) void k_sel_emit_feats(EmitArgs E, hso_camera cam, const PoseSrcDev* src,
                                                               const int32_t* ids, const uint8_t* quality, int feat_cap, hso_pose_feat* feats,
                                                               PoseJobDev* jobs, hso_se3* poses_out, int* n_poses_out, double* feat_f, int* n_feats)
{
  __shared__ int s_wave[SEL_WAVES];
  __shared__ unsigned long long s_used[SEL_KF_WORDS];
  __shared__ int s_base[SEL_KF_WORDS];
  const int c = blockIdx.x;
  int b, e;
  {
    int part = 0;
    for (int q = threadIdx.x; q < c; q += SEL_THREADS) part += E.counts[4 * q];
    int before;
    (void)sel_block_scan(part, s_wave, before);
    const int n_ex = E.counts[4 * c], lb = E.begin[c];
    b = before; e = before + n_ex;
    if (threadIdx.x == 0) { E.offs[c] = b; if (c == E.n_calls - 1) E.offs[E.n_calls] = e; }
    for (int k = threadIdx.x; k < n_ex; k += SEL_THREADS) {
      const int v = E.examined[lb + k];
      const int g = E.cand_pt[lb + (v & 0x7fffffff)];
      hso_match_brief r = E.brief[g];
      r.success = (v < 0) ? 1 : 0;          // became a feature (a matched candidate the budget never reached stays 0)
      r.pad_ = g - lb;                      // the point's index in its frame's list
      E.out[b + k] = r;
      if (E.records) {
        hso_frame_match m;
        m.px_cur[0] = r.px_cur[0]; m.px_cur[1] = r.px_cur[1]; m.grad[0] = r.grad[0]; m.grad[1] = r.grad[1];
        m.point = r.pad_; m.success = r.success; m.search_level = r.search_level; m.ref_type = r.ref_type; m.pad_ = 0;
        E.records[b + k] = m;
      }
    }
    __threadfence_block();
    __syncthreads();                        // the records this workgroup wrote are what it reads below
  }
  const hso_match_brief* out = E.out;
  const PoseSrcDev S = src[c];
  hso_pose_feat* F = feats + (size_t)c * feat_cap;
  double* FF = feat_f ? feat_f + (size_t)c * feat_cap * 3 : nullptr;
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
      const int at = S.list_begin + r.pad_;
      const hso_map_point& p = S.pts[ids[at]];
      hso_pose_feat f;
      f.has_point = 1; f.type = r.ref_type; f.level = r.search_level; f.temporary = ((quality[at] >> 4) == 1) ? 1 : 0;
      f.host_pose = p.host_kf; f._pad = 0;
      hso_dev::cam2world_dev(cam, r.px_cur[0], r.px_cur[1], f.f);
      f.grad[0] = (double)r.grad[0]; f.grad[1] = (double)r.grad[1];
      f.host_f[0] = p.host_f[0]; f.host_f[1] = p.host_f[1]; f.host_f[2] = p.host_f[2];
      f.idist = p.idist;
      F[pos] = f;
      if (FF) { FF[3 * pos] = f.f[0]; FF[3 * pos + 1] = f.f[1]; FF[3 * pos + 2] = f.f[2]; }
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
  for (int k = threadIdx.x; k < S.n_kfs && k < SEL_KF_WORDS * 64; k += SEL_THREADS) {
    const unsigned long long word = s_used[k >> 6], bit = 1ull << (k & 63);
    if (!(word & bit)) continue;
    const int idx = s_base[k >> 6] + __popcll(word & (bit - 1));
    if (idx < HSO_POSE_MAX_POSES) PO[idx] = S.kf_poses[k];
  }
  for (int i = threadIdx.x; i < n; i += SEL_THREADS) {
    const int k = F[i].host_pose;
    int idx = HSO_POSE_MAX_POSES;
    if (k < SEL_KF_WORDS * 64) { const unsigned long long word = s_used[k >> 6], bit = 1ull << (k & 63); idx = s_base[k >> 6] + __popcll(word & (bit - 1)); }
    if (idx < HSO_POSE_MAX_POSES) F[i].host_pose = idx;
    else { F[i].has_point = 0; F[i].host_pose = 0; }     // more host keyframes than the optimiser's table holds: the feature sits out
  }
}

extern "C" int hso_gpu_reproject_select_pose_frames(hso_gpu_ctx* ctx, const hso_camera* cam, const hso_map_frame* frames, int n_calls, int cell_size,
                                                    int grid_n_cols, const int32_t* cell_order, int n_cells, int max_fts, hso_match_brief* out,
                                                    int out_capacity, int32_t* begin_out, int32_t* counts_out, uint8_t* projected_out,
                                                    const hso_pose_chain* pose)
{
  if (!ctx) return HSO_E_INVALID;
  if (n_calls < 0 || n_cells <= 0 || !cell_order || max_fts < 0 || (n_calls > 0 && (!frames || !begin_out || !counts_out)) || !pose)
    return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: bad argument");
  if (n_calls > 0 && (!pose->results || pose->n_iter < 0)) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: bad pose argument");
  if (max_fts > HSO_POSE_MAX_FEATS) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: max_fts above the pose optimiser's table size (4096)");
  {
    std::vector<uint8_t> seen(n_cells, 0);
    for (int k = 0; k < n_cells; k++) {
      if (cell_order[k] < 0 || cell_order[k] >= n_cells || seen[cell_order[k]]) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: cell_order is not a permutation");
      seen[cell_order[k]] = 1;
    }
  }
  auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
  size_t n_total = 0;
  for (int c = 0; c < n_calls; c++) { if (frames[c].n_points < 0) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: negative list length"); n_total += (size_t)frames[c].n_points; }
  for (int c = 0; c <= n_calls; c++) begin_out[c] = 0;
  for (int c = 0; c < 4 * n_calls; c++) counts_out[c] = 0;
  for (int c = 0; c < n_calls; c++) {
    memset(&pose->results[c], 0, sizeof(hso_pose_result));
    pose->results[c].status = 1; pose->results[c].T_f_w = frames[c].T_cur_w;
    if (pose->n_feats) pose->n_feats[c] = 0;
  }
  if (n_total == 0) return 0;
  const int feat_cap = std::max(max_fts, 1);
  const size_t per_frame = al(sizeof(int32_t) * (size_t)(n_cells + 1)) + 4 * al(sizeof(int32_t) * (size_t)n_cells) + al(sizeof(int32_t) * 3 * (size_t)n_cells) +
                           al(sizeof(int32_t) * (size_t)std::max(n_cells, max_fts + 50));
  size_t o = 0;
  const size_t o_begin = o; o += al(sizeof(int) * (size_t)(n_calls + 1));
  const size_t o_order = o; o += al(sizeof(int32_t) * (size_t)n_cells);
  const size_t o_frames = o; o += al(sizeof(SelFrame) * (size_t)n_calls);
  const size_t in_bytes = o;
  const size_t o_cell = o; o += al(sizeof(int32_t) * n_total);
  const size_t o_q = o; o += al(n_total);
  const size_t o_f = o; o += al(n_total);
  const size_t o_pt = o; o += al(sizeof(int32_t) * n_total);
  const size_t o_ncand = o; o += al(sizeof(int) * (size_t)n_calls);
  const size_t o_list = o; o += al(sizeof(int32_t) * n_total);
  const size_t o_exam = o; o += al(sizeof(int32_t) * n_total);
  const size_t o_counts = o; o += al(sizeof(int32_t) * 4 * (size_t)n_calls);
  const size_t o_offs = o; o += al(sizeof(int) * (size_t)(n_calls + 1));
  const size_t o_out = o; o += al(sizeof(hso_match_brief) * n_total);
  const size_t o_flag = o; o += al(n_total);
  const size_t o_rec = o; o += al(sizeof(hso_frame_match) * n_total);
  const size_t o_scr = o; o += per_frame * (size_t)n_calls;
  const size_t o_pf = o; o += al(sizeof(hso_pose_feat) * (size_t)n_calls * feat_cap);
  const size_t o_pj = o; o += al(sizeof(PoseJobDev) * (size_t)n_calls);
  const size_t o_ps = o; o += al(sizeof(PoseSrcDev) * (size_t)n_calls);
  const size_t o_pp = o; o += al(sizeof(hso_se3) * (size_t)n_calls * HSO_POSE_MAX_POSES);
  const size_t o_pnp = o; o += al(sizeof(int) * (size_t)n_calls);
  const size_t o_pr = o; o += al(sizeof(hso_pose_result) * (size_t)n_calls);
  const size_t o_pk = o; o += al((size_t)n_calls * feat_cap);
  const size_t o_pn = o; o += al(sizeof(int) * (size_t)n_calls);
  const size_t o_ff = o; o += al(sizeof(double) * 3 * (size_t)n_calls * feat_cap);
  const size_t o_kp_fixed = o;   // every frame's keyframe poses follow; their number is known after the run below
  HsoMapsRun R; HsoFramesAux X;
  // the keyframe pose tables: sized by the maps' keyframe counts (a few dozen rows per frame); reserve generously, checked below
  size_t kf_rows = 0;
  {
    int nk = 0, np = 0, no = 0;
    for (int c = 0; c < n_calls; c++) { if (hso_gpu_seqmap_size(ctx, frames[c].map, &nk, &np, &no) < 0) return HSO_E_INVALID; kf_rows += (size_t)nk; }
  }
  o += al(sizeof(hso_se3) * std::max(kf_rows, (size_t)1));
  const int total = hso_reproject_frames_run(ctx, cam, frames, n_calls, cell_size, grid_n_cols, o, &R, &X);
  if (total < 0) return total;
  char* d = R.d_extra;
  // ONE staging image (slot 1: slot 0 holds hso_reproject_frames_run's upload, still in flight) for everything the chain behind the
  // projection needs — selection tables, pose job records, per-frame sources, keyframe poses — uploaded before the first kernel, so
  // the kernels of the chain run back to back with no synchronisation in between
  const size_t b_pj = al(sizeof(PoseJobDev) * (size_t)n_calls), b_ps = al(sizeof(PoseSrcDev) * (size_t)n_calls);
  const size_t b_kp = al(sizeof(hso_se3) * std::max(X.kf_poses.size(), (size_t)1));
  if (X.kf_poses.size() > kf_rows) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: keyframe tables changed during the call");
  char* h = hso_pinned(ctx, 1, in_bytes + b_pj + b_ps + b_kp);
  if (!h) return HSO_E_NOMEM;
  int* hb = reinterpret_cast<int*>(h + o_begin);
  for (int c = 0; c <= n_calls; c++) hb[c] = R.begin[c];
  memcpy(h + o_order, cell_order, sizeof(int32_t) * (size_t)n_cells);
  SelFrame* hf = reinterpret_cast<SelFrame*>(h + o_frames);
  for (int c = 0; c < n_calls; c++) {
    char* sc = d + o_scr + per_frame * (size_t)c;
    SelFrame& F = hf[c];
    F.first = R.begin[c]; F.n = 0; F.n_dev = reinterpret_cast<const int*>(d + o_ncand) + c;
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
  }
  {
    char* hp = h + in_bytes;
    PoseJobDev* pj = reinterpret_cast<PoseJobDev*>(hp);
    PoseSrcDev* ps = reinterpret_cast<PoseSrcDev*>(hp + b_pj);
    if (!X.kf_poses.empty()) memcpy(hp + b_pj + b_ps, X.kf_poses.data(), sizeof(hso_se3) * X.kf_poses.size());
    const hso_se3* d_kp = reinterpret_cast<const hso_se3*>(d + o_kp_fixed);
    for (int c = 0; c < n_calls; c++) {
      pj[c].feats = reinterpret_cast<const hso_pose_feat*>(d + o_pf) + (size_t)c * feat_cap;
      pj[c].poses = reinterpret_cast<const hso_se3*>(d + o_pp) + (size_t)c * HSO_POSE_MAX_POSES;
      pj[c].mask = reinterpret_cast<uint8_t*>(d + o_pk) + (size_t)c * feat_cap;
      pj[c].n_feats = 0; pj[c].n_poses = 0; pj[c].T = frames[c].T_cur_w; pj[c].reproj_thresh = pose->reproj_thresh;
      pj[c].n_iter = pose->n_iter; pj[c]._pad = 0;
      ps[c].pts = X.pts[c]; ps[c].kf_poses = d_kp + X.kf_begin[c]; ps[c].list_begin = R.begin[c]; ps[c].n_kfs = X.kf_begin[c + 1] - X.kf_begin[c];
    }
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_pj, hp, b_pj, hipMemcpyHostToDevice, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_ps, hp + b_pj, b_ps, hipMemcpyHostToDevice, ctx->stream));
    HSO_HIP_CHECK(ctx, hipMemcpyAsync(d + o_kp_fixed, hp + b_pj + b_ps, b_kp, hipMemcpyHostToDevice, ctx->stream));
  }
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  const int* d_begin = reinterpret_cast<const int*>(d + o_begin);
  hipLaunchKernelGGL(k_sel_gather, dim3(n_calls), dim3(SEL_THREADS), 0, ctx->stream, R.d_proj, R.d_brief, d_begin,
                     reinterpret_cast<int32_t*>(d + o_cell), reinterpret_cast<uint8_t*>(d + o_q), reinterpret_cast<uint8_t*>(d + o_f),
                     reinterpret_cast<int32_t*>(d + o_pt), reinterpret_cast<int*>(d + o_ncand), projected_out ? reinterpret_cast<uint8_t*>(d + o_flag) : nullptr);
  SelArgs A;
  A.cell = reinterpret_cast<const int32_t*>(d + o_cell); A.quality = reinterpret_cast<const uint8_t*>(d + o_q);
  A.flags = reinterpret_cast<const uint8_t*>(d + o_f); A.cell_order = reinterpret_cast<const int32_t*>(d + o_order);
  A.n_cells = n_cells; A.max_fts = max_fts;
  hipLaunchKernelGGL(k_select, dim3(n_calls), dim3(SEL_THREADS), 0, ctx->stream, A, reinterpret_cast<const SelFrame*>(d + o_frames));
  HSO_HIP_CHECK(ctx, hipMemsetAsync(d + o_pk, 0, (size_t)n_calls * feat_cap, ctx->stream));
  {
    EmitArgs E;
    E.brief = R.d_brief; E.begin = d_begin; E.examined = reinterpret_cast<const int32_t*>(d + o_exam); E.cand_pt = reinterpret_cast<const int32_t*>(d + o_pt);
    E.counts = reinterpret_cast<const int*>(d + o_counts); E.offs = reinterpret_cast<int*>(d + o_offs); E.out = reinterpret_cast<hso_match_brief*>(d + o_out);
    E.records = pose->records ? reinterpret_cast<hso_frame_match*>(d + o_rec) : nullptr; E.n_calls = n_calls;
    hipLaunchKernelGGL(k_sel_emit_feats, dim3(n_calls), dim3(SEL_THREADS), 0, ctx->stream, E, *cam, reinterpret_cast<const PoseSrcDev*>(d + o_ps), X.d_ids, X.d_quality,
                       feat_cap, reinterpret_cast<hso_pose_feat*>(d + o_pf), reinterpret_cast<PoseJobDev*>(d + o_pj), reinterpret_cast<hso_se3*>(d + o_pp),
                       reinterpret_cast<int*>(d + o_pnp), pose->feat_f ? reinterpret_cast<double*>(d + o_ff) : nullptr, reinterpret_cast<int*>(d + o_pn));
    HSO_HIP_CHECK(ctx, hipGetLastError());
    if (int rc = hso_pose_launch_device(ctx, cam, reinterpret_cast<const PoseJobDev*>(d + o_pj), n_calls, feat_cap,
                                        reinterpret_cast<hso_pose_result*>(d + o_pr))) return rc;
  }
  // read-back: counts + offsets + pose results + feature counts first; then exactly the examined records and the used rows
  hso_pose_result* const h_res = pose->results;
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(counts_out, d + o_counts, sizeof(int32_t) * 4 * (size_t)n_calls, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(begin_out, d + o_offs, sizeof(int) * (size_t)(n_calls + 1), hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(h_res, d + o_pr, sizeof(hso_pose_result) * (size_t)n_calls, hipMemcpyDeviceToHost, ctx->stream));
  std::vector<int> nf((size_t)n_calls, 0);
  HSO_HIP_CHECK(ctx, hipMemcpyAsync(pose->n_feats ? pose->n_feats : nf.data(), d + o_pn, sizeof(int) * (size_t)n_calls, hipMemcpyDeviceToHost, ctx->stream));
  if (pose->outlier_mask) HSO_HIP_CHECK(ctx, hipMemcpyAsync(pose->outlier_mask, d + o_pk, (size_t)n_calls * feat_cap, hipMemcpyDeviceToHost, ctx->stream));
  if (pose->feat_f) HSO_HIP_CHECK(ctx, hipMemcpyAsync(pose->feat_f, d + o_ff, sizeof(double) * 3 * (size_t)n_calls * feat_cap, hipMemcpyDeviceToHost, ctx->stream));
  if (projected_out) HSO_HIP_CHECK(ctx, hipMemcpyAsync(projected_out, d + o_flag, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
  HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const int n_out = begin_out[n_calls];
  if (n_out > 0) {
    if ((!out && !pose->records) || out_capacity < n_out) return hso_fail(ctx, HSO_E_INVALID, "reproject_select_pose_frames: output smaller than the examined candidates");
    if (out) HSO_HIP_CHECK(ctx, hipMemcpyAsync(out, d + o_out, sizeof(hso_match_brief) * (size_t)n_out, hipMemcpyDeviceToHost, ctx->stream));
    if (pose->records) HSO_HIP_CHECK(ctx, hipMemcpyAsync(pose->records, d + o_rec, sizeof(hso_frame_match) * (size_t)n_out, hipMemcpyDeviceToHost, ctx->stream));
    HSO_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  hso_seqmaps_debug_set(ctx, HSO_DBG_PROJ, R.d_proj, sizeof(hso_reproj_point) * (size_t)total);
  hso_seqmaps_debug_set(ctx, HSO_DBG_MATCH, X.d_match, sizeof(hso_align_out) * (size_t)total);
  hso_seqmaps_debug_set(ctx, HSO_DBG_POSE_FEATS, d + o_pf, sizeof(hso_pose_feat) * (size_t)n_calls * feat_cap);
  hso_seqmaps_debug_set(ctx, HSO_DBG_POSE_POSES, d + o_pp, sizeof(hso_se3) * (size_t)n_calls * HSO_POSE_MAX_POSES);
  hso_seqmaps_debug_set(ctx, HSO_DBG_POSE_NPOSES, d + o_pnp, sizeof(int) * (size_t)n_calls);
  return n_out;
}
