// The path's only exchange, native: one ncclAllGather (RCCL over xGMI) of the per-frame result records of all ranks —
// BASELINE.json north_star "RCCL ... used only to gather results", SURVEY §8(d) config 5 (`[n_frames][8]` doubles per
// sequence).  C interface in include/hso_vo.h (hso_gather_*).  Built as its own small library (libhso_gather.so) so that
// libhso_host.so carries no RCCL dependency for single-GPU users; one process per GPU, the caller distributes the 128-byte
// communicator id (rank 0's hso_gather_unique_id) by whatever channel launched the ranks (environment, file, MPI, torch).
#include "../../include/hso_vo.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdio.h>
#include <string.h>
#include <string>

static thread_local std::string g_err;

struct hso_gather {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int rank = 0, world = 1, device = 0;
  double* d_send = nullptr;     // grow-only device staging: 288 GB of HBM, a trajectory block is KBs
  double* d_recv = nullptr;
  size_t cap_rows = 0;
};

static int fail(const char* what, const char* detail)
{
  g_err = std::string(what) + ": " + detail;
  return HSO_E_HIP;
}

#define G_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(#x, hipGetErrorString(e_)); } while (0)
#define G_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(#x, ncclGetErrorString(r_)); } while (0)

extern "C" {

const char* hso_gather_last_error(void) { return g_err.c_str(); }

int hso_gather_unique_id(uint8_t id[HSO_GATHER_ID_BYTES])
{
  static_assert(HSO_GATHER_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  if (!id) { g_err = "hso_gather_unique_id: id is NULL"; return HSO_E_INVALID; }
  ncclUniqueId u;
  G_NCCL(ncclGetUniqueId(&u));
  memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
  return HSO_OK;
}

int hso_gather_create(hso_gather** out, const uint8_t id[HSO_GATHER_ID_BYTES], int rank, int world, int device)
{
  if (!out || !id || world < 1 || rank < 0 || rank >= world) { g_err = "hso_gather_create: bad arguments"; return HSO_E_INVALID; }
  *out = nullptr;
  G_HIP(hipSetDevice(device));
  hso_gather* g = new hso_gather();
  g->rank = rank; g->world = world; g->device = device;
  ncclUniqueId u;
  memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
  hipError_t e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete g; return fail("hipStreamCreateWithFlags", hipGetErrorString(e)); }
  ncclResult_t r = ncclCommInitRank(&g->comm, world, u, rank);
  if (r != ncclSuccess) { (void)hipStreamDestroy(g->stream); delete g; return fail("ncclCommInitRank", ncclGetErrorString(r)); }
  *out = g;
  return HSO_OK;
}

void hso_gather_destroy(hso_gather* g)
{
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->comm) ncclCommDestroy(g->comm);
  if (g->d_send) (void)hipFree(g->d_send);
  if (g->d_recv) (void)hipFree(g->d_recv);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
}

int hso_gather_size(const hso_gather* g) { return g ? g->world : 0; }
int hso_gather_rank(const hso_gather* g) { return g ? g->rank : -1; }

// every rank passes n_rows records of HSO_GATHER_RECORD doubles (same n_rows on all ranks: pad short sequences with NaN rows
// as hso_amd/dist.py:pack_trajectories does); all = [world][n_rows][HSO_GATHER_RECORD] in rank order on every rank
int hso_gather_records(hso_gather* g, const double* mine, int n_rows, double* all)
{
  if (!g || n_rows < 0 || (n_rows > 0 && (!mine || !all))) { g_err = "hso_gather_records: bad arguments"; return HSO_E_INVALID; }
  if (n_rows == 0) return HSO_OK;
  G_HIP(hipSetDevice(g->device));
  const size_t cnt = (size_t)n_rows * HSO_GATHER_RECORD;
  if ((size_t)n_rows > g->cap_rows) {
    if (g->d_send) (void)hipFree(g->d_send);
    if (g->d_recv) (void)hipFree(g->d_recv);
    g->d_send = g->d_recv = nullptr; g->cap_rows = 0;
    G_HIP(hipMalloc((void**)&g->d_send, cnt * sizeof(double)));
    G_HIP(hipMalloc((void**)&g->d_recv, cnt * sizeof(double) * (size_t)g->world));
    g->cap_rows = (size_t)n_rows;
  }
  G_HIP(hipMemcpyAsync(g->d_send, mine, cnt * sizeof(double), hipMemcpyHostToDevice, g->stream));
  G_NCCL(ncclAllGather(g->d_send, g->d_recv, cnt, ncclDouble, g->comm, g->stream));
  G_HIP(hipMemcpyAsync(all, g->d_recv, cnt * sizeof(double) * (size_t)g->world, hipMemcpyDeviceToHost, g->stream));
  G_HIP(hipStreamSynchronize(g->stream));
  return HSO_OK;
}

}  // extern "C"
