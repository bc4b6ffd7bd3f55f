// ref_shim_patch_score.cpp — C-linkage entry points over the reference's own patch-score classes
// (include/hso/vikit/patch_score.h, a header that compiles standalone): ZMNCC_F<4>, the score
// Matcher::doLineStereo ranks epipolar candidates with (src/matcher.cpp:918), and ZMSSD_F<4>.
// Built by oracle/Makefile into oracle/_ref/libpatch_score_ref.so from the header where it lies;
// this file contains no reference code, it only instantiates the reference's templates.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include "hso/vikit/patch_score.h"

extern "C" {
float ref_zmncc_f8(const float* host, const float* target)
{
  hso::patch_score::ZMNCC_F<4> s(const_cast<float*>(host));
  return s.computeScore(const_cast<float*>(target));
}
float ref_zmssd_f8(const float* host, const float* target)
{
  hso::patch_score::ZMSSD_F<4> s(const_cast<float*>(host));
  return s.computeScore(const_cast<float*>(target));
}
}
