/*
 * hso_oracle_robust.c — robust weight functions and scale estimators.
 * TEST INFRASTRUCTURE (see hso_oracle.h).  Follows src/vikit/robust_cost.cpp of
 * the reference; this is the one part of the oracle that IS pinned against the
 * compiled reference (oracle/_ref/librobust_cost_ref.so, built from the
 * reference's own robust_cost.cpp) and against tests/golden/robust_cost.json
 * generated from it.
 */
#include "hso_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* HuberWeightFunction::value, robust_cost.cpp:141-148 (k = 1.345f default, :129) */
float hso_or_huber_weight(float k, float t)
{
  const float t_abs = fabsf(t);
  if (t_abs < k) return 1.0f;
  else return k / t_abs;
}

/* TukeyWeightFunction::value, robust_cost.cpp:93-103 (b = 4.6851f default, :86) */
float hso_or_tukey_weight(float b, float x)
{
  const float b_square = b * b;
  const float x_square = x * x;
  if (x_square <= b_square) {
    const float tmp = 1.0f - x_square / b_square;
    return tmp * tmp;
  } else
    return 0;
}

/* TDistributionWeightFunction::value, robust_cost.cpp:117-121 */
float hso_or_tdist_weight(float dof, float x) { return ((dof + 1.0f) / (dof + (x * x))); }

/* MADScaleEstimator::compute, robust_cost.cpp:67-74: 1.4826f * nth_element at floor(n/2) */
float hso_or_mad_scale(const float* errors, int n)
{
  float* tmp = (float*)malloc(sizeof(float) * (size_t)n);
  memcpy(tmp, errors, sizeof(float) * (size_t)n);
  const float m = hso_or_median_f(tmp, n);
  free(tmp);
  return 1.4826f * m;
}

/* TDistributionScaleEstimator::compute, robust_cost.cpp:38-63 */
float hso_or_tdist_scale(float dof, const float* errors, int n)
{
  const float initial_sigma = 5.0f;
  float initial_lamda = 1.0f / (initial_sigma * initial_sigma);
  int num = 0;
  float lambda = initial_lamda;
  do {
    initial_lamda = lambda;
    num = 0;
    lambda = 0.0f;
    for (int i = 0; i < n; i++) {
      if (isfinite(errors[i])) {
        ++num;
        const float error2 = errors[i] * errors[i];
        lambda += error2 * ((dof + 1.0f) / (dof + initial_lamda * error2));
      }
    }
    lambda = (float)num / lambda;
  } while (fabsf(lambda - initial_lamda) > 1e-3);
  return sqrtf(1.0f / lambda);
}
