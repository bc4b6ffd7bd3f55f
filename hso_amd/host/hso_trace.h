// hso_trace.h — C-ABI call recorder of the host driver.
//
// Every device call the driver makes goes through one of the api:: wrappers below; with a trace
// file open they append the call's inputs and outputs as raw tables (the C-ABI's own POD layouts).
// tests/test_chain_gpu.py replays each record through the CPU restatement (oracle/) and compares —
// stage-by-stage parity from the same evolving state, without the product ever touching the oracle.
// The reference's counterpart is the -DTRACE performance log (include/hso/global.h:108-123): it
// records timings only; this records the data.
//
// Record: u32 magic 'HSTR', u32 name_len, name, u32 n_fields; field: u32 key_len, key, u64 n_bytes, bytes.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/hso_gpu.h"

namespace hso {
namespace api {

struct Trace {
  FILE* f = nullptr;
  uint32_t n_fields_pos = 0, n_fields = 0;
  bool open(const char* path) { close(); f = std::fopen(path, "wb"); return f != nullptr; }
  void close() { if (f) std::fclose(f); f = nullptr; }
  bool on() const { return f != nullptr; }
  void begin(const char* name, uint32_t nf)
  {
    const uint32_t magic = 0x52545348u, nl = (uint32_t)std::strlen(name);
    std::fwrite(&magic, 4, 1, f); std::fwrite(&nl, 4, 1, f); std::fwrite(name, 1, nl, f); std::fwrite(&nf, 4, 1, f);
  }
  void field(const char* key, const void* data, size_t bytes)
  {
    const uint32_t kl = (uint32_t)std::strlen(key); const uint64_t nb = bytes;
    std::fwrite(&kl, 4, 1, f); std::fwrite(key, 1, kl, f); std::fwrite(&nb, 8, 1, f);
    if (bytes) std::fwrite(data, 1, bytes, f);
  }
  void scalar(const char* key, double v) { field(key, &v, 8); }
  void flush() { if (f) std::fflush(f); }
};

Trace& trace();   // one per thread (hso_host.cpp)

// A failed device call in the middle of processFrame: unlike the reference's own exceptions (wrong image size, thrown before
// anything is touched) it can leave the pointer graph half updated (a keyframe already registered, features already
// referenced by points), so the driver treats it as fatal for the handle (hso_vo.cpp: vo_guard).
struct DeviceError : std::runtime_error { using std::runtime_error::runtime_error; };

inline void check(hso_gpu_ctx* ctx, int rc, const char* what)
{
  if (rc < 0) throw DeviceError(std::string(what) + ": " + hso_gpu_last_error(ctx));
}

}  // namespace api
}  // namespace hso
