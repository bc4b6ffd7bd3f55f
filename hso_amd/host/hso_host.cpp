// hso_host.cpp — bodies of the host mirror: thin adapters onto the C-ABI.
#include "hso_host.h"
#include <cmath>
#include <string>

namespace hso {

static void quat_rotate(const double q[4], const double v[3], double o[3])
{
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  o[0] = (v[0] + q[3] * uv[0]) + (q[1] * uv[2] - q[2] * uv[1]);
  o[1] = (v[1] + q[3] * uv[1]) + (q[2] * uv[0] - q[0] * uv[2]);
  o[2] = (v[2] + q[3] * uv[2]) + (q[0] * uv[1] - q[1] * uv[0]);
}
static void quat_normalize(double q[4])
{
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}

SE3 SE3::operator*(const SE3& o) const
{
  SE3 r;
  double rt[3];
  quat_rotate(v.q, o.v.t, rt);
  for (int i = 0; i < 3; i++) r.v.t[i] = v.t[i] + rt[i];
  const double *a = v.q, *b = o.v.q;
  r.v.q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r.v.q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r.v.q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  r.v.q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  quat_normalize(r.v.q);
  return r;
}

Vector3d SE3::operator*(const Vector3d& p) const
{
  double o[3];
  quat_rotate(v.q, p.data(), o);
  return {o[0] + v.t[0], o[1] + v.t[1], o[2] + v.t[2]};
}

SE3 SE3::inverse() const
{
  SE3 r;
  r.v.q[0] = -v.q[0]; r.v.q[1] = -v.q[1]; r.v.q[2] = -v.q[2]; r.v.q[3] = v.q[3];
  quat_normalize(r.v.q);
  const double nt[3] = {v.t[0] * -1., v.t[1] * -1., v.t[2] * -1.};
  quat_rotate(r.v.q, nt, r.v.t);
  return r;
}

double AbstractCamera::errorMultiplier2() const
{
  return (c_.fx * c_.fy < 0) ? std::fabs(c_.fx) : std::fabs((c_.fx + c_.fy) * 0.5);
}

Vector2d AbstractCamera::world2cam(const Vector3d& xyz) const
{
  const double u = xyz[0] / xyz[2], v = xyz[1] / xyz[2];
  if (c_.model == HSO_CAM_PINHOLE && c_.distortion) {
    const double r2 = u * u + v * v, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * u * v, a2 = r2 + 2 * u * u, a3 = r2 + 2 * v * v;
    const double cdist = 1 + c_.d[0] * r2 + c_.d[1] * r4 + c_.d[4] * r6;
    const double xd = u * cdist + c_.d[2] * a1 + c_.d[3] * a2;
    const double yd = v * cdist + c_.d[2] * a3 + c_.d[3] * a1;
    return {xd * c_.fx + c_.cx, yd * c_.fy + c_.cy};
  }
  if (c_.model == HSO_CAM_FOV && c_.distortion) {
    const double omega = c_.d[0];
    const double dist = std::sqrt(u * u + v * v);
    const double ratio = (omega == 0 || dist == 0) ? 1 : std::atan(2 * dist * std::tan(omega / 2)) / (dist * omega);
    return {ratio * c_.fx * u + c_.cx, ratio * c_.fy * v + c_.cy};
  }
  return {c_.fx * u + c_.cx, c_.fy * v + c_.cy};
}

int Frame::frame_counter_ = 0;

Frame::Frame(hso_gpu_ctx* ctx, AbstractCamera* cam, const uint8_t* img, int width, int height, double timestamp)
    : id_(frame_counter_++), timestamp_(timestamp), cam_(cam), ctx_(ctx)
{
  // src/frame.cpp:85-86
  if (!img || width != cam->width() || height != cam->height())
    throw std::runtime_error("Frame: provided image has not the same size as the camera model or image is not grayscale");
  hso_frame_stats st{};
  const int rc = hso_gpu_frame_upload(ctx_, id_, img, width, height, 0, &st);
  if (rc < 0) throw std::runtime_error(std::string("Frame: ") + hso_gpu_last_error(ctx_));
  integralImage_ = st.integral_image;
  gradMean_ = st.grad_mean;
}

Frame::~Frame()
{
  for (Feature* f : fts_) delete f;
  hso_gpu_frame_release(ctx_, id_);
}

CoarseTracker::CoarseTracker(bool inverse_composition, int max_level, int min_level, int n_iter, bool verbose)
    : m_inverse_composition(inverse_composition), m_max_level(max_level), m_min_level(min_level),
      m_n_iter(n_iter), m_verbose(verbose)
{
}

size_t CoarseTracker::run(FramePtr ref, FramePtr cur)
{
  if (ref->fts_.empty()) return 0;  // src/CoarseTracker.cpp:53-54
  // makeDepthRef (:210-240), flattened in fts_ list order; a feature without point keeps its slot
  std::vector<hso_ref_feat> feats;
  feats.reserve(ref->fts_.size());
  for (Feature* ft : ref->fts_) {
    hso_ref_feat r{};
    r.px[0] = ft->px[0]; r.px[1] = ft->px[1];
    r.f[0] = ft->f[0]; r.f[1] = ft->f[1]; r.f[2] = ft->f[2];
    r.dist = -1;
    if (ft->point != nullptr) {
      const Feature* host = ft->point->hostFeature_;
      const double inv = 1.0 / ft->point->idist_;
      const Vector3d p_host{host->f[0] * inv, host->f[1] * inv, host->f[2] * inv};
      const SE3 T_r_h = ref->T_f_w_ * host->frame->T_f_w_.inverse();
      const Vector3d p_ref = T_r_h * p_host;
      if (!(p_ref[2] < 0.00001)) r.dist = std::sqrt(p_ref[0] * p_ref[0] + p_ref[1] * p_ref[1] + p_ref[2] * p_ref[2]);
    }
    feats.push_back(r);
  }
  hso_track_job job{};
  job.ref_frame_id = ref->id_;
  job.cur_frame_id = cur->id_;
  job.feats = feats.data();
  job.n_feats = (int)feats.size();
  job.T_cur_ref = (cur->T_f_w_ * ref->T_f_w_.inverse()).v;           // :63
  job.exposure_rat = cur->integralImage_ / ref->integralImage_;       // :60
  hso_track_params p{m_inverse_composition ? 1 : 0, m_max_level, m_min_level, m_n_iter};
  const int rc = hso_gpu_coarse_track_batch(cur->ctx_, &cur->cam_->pod(), &p, &job, 1, &m_last);
  if (rc < 0) throw std::runtime_error(std::string("CoarseTracker: ") + hso_gpu_last_error(cur->ctx_));
  m_T_cur_ref.v = m_last.T_cur_ref;
  // write-back, :198-202
  cur->T_f_w_ = m_T_cur_ref * ref->T_f_w_;
  cur->m_exposure_time = (double)m_last.exposure_rat * ref->m_exposure_time;
  if (m_last.exposure_rat > 0.99 && m_last.exposure_rat < 1.01) cur->m_exposure_time = ref->m_exposure_time;
  return (size_t)m_last.n_tracked;  // :207
}

}  // namespace hso
