// Device-side quaternion (wxyz) helpers, fp32.  Formulas follow add_gym/util/torch_util.py
// (cited per function) so that results agree with the reference to a few ulp.
#pragma once
#include <hip/hip_runtime.h>

namespace addhip {

struct Quat { float w, x, y, z; };
struct Vec3 { float x, y, z; };

// torch_util.py:48-61
__device__ __forceinline__ Quat quat_mul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  return r;
}

__device__ __forceinline__ Vec3 cross(const Vec3& a, const Vec3& b) {
  return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// torch_util.py:65-70: v + w*t + qv x t with t = 2*(qv x v)
__device__ __forceinline__ Vec3 quat_rotate(const Quat& q, const Vec3& v) {
  Vec3 qv{q.x, q.y, q.z};
  Vec3 t = cross(qv, v);
  t.x *= 2.0f; t.y *= 2.0f; t.z *= 2.0f;
  Vec3 c = cross(qv, t);
  return Vec3{v.x + q.w * t.x + c.x, v.y + q.w * t.y + c.y, v.z + q.w * t.z + c.z};
}

// torch_util.py:231-242: element c (0..5) of [rot(q,x), rot(q,z)]
__device__ __forceinline__ float tan_norm_elem(const Quat& q, int c) {
  Vec3 r = (c < 3) ? quat_rotate(q, Vec3{1.0f, 0.0f, 0.0f}) : quat_rotate(q, Vec3{0.0f, 0.0f, 1.0f});
  int k = (c < 3) ? c : c - 3;
  return k == 0 ? r.x : (k == 1 ? r.y : r.z);
}

// torch_util.py:326-356: inverse heading rotation about z
__device__ __forceinline__ Quat heading_quat_inv(const Quat& q) {
  Vec3 d = quat_rotate(q, Vec3{1.0f, 0.0f, 0.0f});
  float heading = atan2f(d.y, d.x);
  float half = (-heading) / 2.0f;
  // axis_angle_to_quat (torch_util.py:186-195): normalize(axis)=z exactly; then unit-normalise
  float w = cosf(half), z = sinf(half);
  float n = sqrtf(w * w + z * z);
  n = fmaxf(n, 1e-9f);
  return Quat{w / n, 0.0f / n, 0.0f / n, z / n};
}

// torch_util.py:74-94 angle only (quat_pos, then 2*atan2(|xyz|, w); 0 when |xyz| <= 1e-5)
__device__ __forceinline__ float quat_angle(Quat q) {
  if (q.w < 0.0f) { q.w = -q.w; q.x = -q.x; q.y = -q.y; q.z = -q.z; }
  float len = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z);
  float ang = 2.0f * atan2f(len, q.w);
  return len > 1e-5f ? ang : 0.0f;
}

// torch_util.py:275-285: angle of q1 * conj(q0)
__device__ __forceinline__ float quat_diff_angle(const Quat& q0, const Quat& q1) {
  Quat c{q0.w, -q0.x, -q0.y, -q0.z};
  return quat_angle(quat_mul(q1, c));
}

}  // namespace addhip
