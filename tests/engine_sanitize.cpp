// TEST INFRASTRUCTURE: the sequence engine (hso_amd/host) over tests/fakegpu (the C-ABI on the CPU restatement), driven from C++ so
// that it can run under ThreadSanitizer / AddressSanitizer + UBSan without an interpreter in the process.  tools/run_sanitizers.sh
// writes the input (a rendered sequence: camera, images, first-frame depth) with hso_amd.synth, builds this file with the engine's
// and the fake backend's sources under each sanitizer and keeps the logs (profiles/r5_sanitizers.md).
//   engine_sanitize <sequence file> <n_sequences> <max_fts> [blank frames in the middle]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/hso_vo.h"

int main(int argc, char** argv)
{
  if (argc < 4) { fprintf(stderr, "usage: engine_sanitize file n_sequences max_fts [n_blank]\n"); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  const int n_seq = atoi(argv[2]), max_fts = atoi(argv[3]), n_blank = argc > 4 ? atoi(argv[4]) : 0;
  int32_t hdr[3];
  hso_camera cam;
  if (fread(hdr, 4, 3, f) != 3 || fread(&cam, sizeof(cam), 1, f) != 1) return 2;
  const int w = hdr[0], h = hdr[1], n_frames = hdr[2];
  std::vector<std::vector<uint8_t>> img((size_t)n_frames, std::vector<uint8_t>((size_t)w * h));
  std::vector<float> depth((size_t)w * h);
  for (auto& im : img) if (fread(im.data(), 1, im.size(), f) != im.size()) return 2;
  if (fread(depth.data(), 4, depth.size(), f) != depth.size()) return 2;
  fclose(f);
  hso_vo_multi* m = nullptr;
  if (hso_vo_host_share(2) < 0) return 1;
  if (hso_vo_multi_create(&m, &cam, max_fts, n_seq, 0) < 0) { fprintf(stderr, "create failed\n"); return 1; }
  std::vector<const uint8_t*> imgs((size_t)n_seq); std::vector<const float*> dz((size_t)n_seq, depth.data()); std::vector<double> ts((size_t)n_seq, 0.0);
  for (int q = 0; q < n_seq; q++) imgs[(size_t)q] = img[0].data();
  if (hso_vo_multi_set_first_frames(m, imgs.data(), w, h, ts.data(), dz.data(), nullptr) < 0) { fprintf(stderr, "first frame: %s\n", hso_vo_multi_last_error(m)); return 1; }
  std::vector<uint8_t> blank((size_t)w * h, 117);
  int n_fail = 0, n_kf = 0;
  for (int k = 1; k < n_frames; k++) {
    // sequence q runs q frames behind the others' rhythm (sits the first q steps of the second half out): ragged banks
    for (int q = 0; q < n_seq; q++) {
      const bool lost = n_blank > 0 && q == 1 && k >= n_frames / 2 && k < n_frames / 2 + n_blank;   // one sequence loses track and relocalises
      imgs[(size_t)q] = (q == 2 && (k % 7) == 3) ? nullptr : (lost ? blank.data() : img[(size_t)k].data());
      ts[(size_t)q] = k;
    }
    if (hso_vo_multi_add_images(m, imgs.data(), w, h, ts.data()) < 0) { fprintf(stderr, "step %d: %s\n", k, hso_vo_multi_last_error(m)); return 1; }
    for (int q = 0; q < n_seq; q++) {
      hso_vo_status st;
      if (hso_vo_multi_get_status(m, q, &st) < 0) return 1;
      n_fail += st.result == 2; n_kf += st.is_keyframe;
    }
  }
  hso_vo_status st;
  hso_vo_multi_get_status(m, 0, &st);
  printf("engine_sanitize: %d sequences x %d frames, %d keyframe reports, %d failure reports, sequence 0 ends at (%.4f %.4f %.4f) with %d matches: ok\n",
         n_seq, n_frames - 1, n_kf, n_fail, st.T_f_w.t[0], st.T_f_w.t[1], st.T_f_w.t[2], st.n_matches);
  hso_vo_multi_destroy(m);
  return 0;
}
