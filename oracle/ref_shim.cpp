// ref_shim.cpp — C-linkage entry points over the reference's own
// hso::robust_cost classes (include/hso/vikit/robust_cost.h), so that Python
// can call the compiled reference code (src/vikit/robust_cost.cpp) to generate
// and check golden vectors.  This file contains no reference code: it only
// instantiates the reference's classes through their public interface.
#include "hso/vikit/robust_cost.h"
#include <vector>
using namespace hso::robust_cost;
extern "C" {
float ref_huber_weight(float k, float x) { HuberWeightFunction f(k); return f.value(x); }
float ref_tukey_weight(float b, float x) { TukeyWeightFunction f(b); return f.value(x); }
float ref_tdist_weight(float dof, float x) { TDistributionWeightFunction f(dof); return f.value(x); }
float ref_mad_scale(const float* errors, int n) {
  std::vector<float> v(errors, errors + n);
  MADScaleEstimator e; return e.compute(v);
}
float ref_tdist_scale(float dof, const float* errors, int n) {
  std::vector<float> v(errors, errors + n);
  TDistributionScaleEstimator e(dof); return e.compute(v);
}
float ref_huber_default_k() { return HuberWeightFunction::DEFAULT_K; }
float ref_tukey_default_b() { return TukeyWeightFunction::DEFAULT_B; }
}
