// ref_shim_fast.cpp — C-linkage entry points over the reference's vendored FAST corner library
// (thirdparty/fast: fast_corner_detect_9_sse2, fast_corner_score_9, fast_nonmax_3x3), the calls
// FeatureExtractor::fastDetect makes (src/feature_detection.cpp:518-587 (fastDetectST per level, fastDetect)).  Built by
// oracle/Makefile into oracle/_ref/libfast_ref.so from the reference sources where they lie;
// this file contains no reference code, it only calls the library's public interface.
#include <cstddef>
#include <fast/fast.h>
#include <vector>

extern "C" {

// returns the number of corners; writes at most cap (x, y) pairs in the library's output order
int ref_fast9_detect(const unsigned char* img, int w, int h, int stride, int barrier, short* xy, int cap)
{
  std::vector<fast::fast_xy> c;
  fast::fast_corner_detect_9_sse2(img, w, h, stride, (short)barrier, c);
  for (int i = 0; i < (int)c.size() && i < cap; i++) { xy[2 * i] = c[i].x; xy[2 * i + 1] = c[i].y; }
  return (int)c.size();
}

void ref_fast9_score(const unsigned char* img, int stride, const short* xy, int n, int threshold, int* scores)
{
  std::vector<fast::fast_xy> c;
  c.reserve(n);
  for (int i = 0; i < n; i++) c.emplace_back(xy[2 * i], xy[2 * i + 1]);
  std::vector<int> s;
  fast::fast_corner_score_9(img, stride, c, threshold, s);
  for (int i = 0; i < n; i++) scores[i] = s[i];
}

// returns the number of maxima; idx[k] = index into the corner list
int ref_fast_nonmax(const short* xy, const int* scores, int n, int* idx)
{
  std::vector<fast::fast_xy> c;
  c.reserve(n);
  for (int i = 0; i < n; i++) c.emplace_back(xy[2 * i], xy[2 * i + 1]);
  std::vector<int> s(scores, scores + n), keep;
  fast::fast_nonmax_3x3(c, s, keep);
  for (std::size_t k = 0; k < keep.size(); k++) idx[k] = keep[k];
  return (int)keep.size();
}
}

// FAST-12 (FeatureExtractor::fillingHole, src/feature_detection.cpp:1125-1154)
extern "C" {
int ref_fast12_detect(const unsigned char* img, int w, int h, int stride, int barrier, short* xy, int cap)
{
  std::vector<fast::fast_xy> c;
  fast::fast_corner_detect_plain_12(img, w, h, stride, (short)barrier, c);
  for (int i = 0; i < (int)c.size() && i < cap; i++) { xy[2 * i] = c[i].x; xy[2 * i + 1] = c[i].y; }
  return (int)c.size();
}

void ref_fast12_score(const unsigned char* img, int stride, const short* xy, int n, int threshold, int* scores)
{
  std::vector<fast::fast_xy> c;
  c.reserve(n);
  for (int i = 0; i < n; i++) c.emplace_back(xy[2 * i], xy[2 * i + 1]);
  std::vector<int> s;
  fast::fast_corner_score_12(img, stride, c, threshold, s);
  for (int i = 0; i < n; i++) scores[i] = s[i];
}
}
