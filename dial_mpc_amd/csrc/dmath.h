// dmath.h -- small fp32 vector / quaternion / spatial-algebra helpers used by the kernel body.
// Conventions follow MuJoCo-MJX: quaternions (w,x,y,z); 6-vectors are [angular; linear];
// cinert = [Ixx,Iyy,Izz,Ixy,Ixz,Iyz, m*dx,m*dy,m*dz, m] about the subtree-root COM.
#pragma once
#include "wave.h"

#ifdef DIAL_EMU
#define DM_SQRT(x) std::sqrt(x)
#define DM_SIN(x) std::sin(x)
#define DM_COS(x) std::cos(x)
#define DM_ATAN2(y, x) std::atan2(y, x)
#define DM_POW(x, y) std::pow(x, y)
#define DM_FLOOR(x) std::floor(x)
#define DM_EXP(x) std::exp(x)
#define DM_RINT(x) std::nearbyint(x)
#define DM_FMA(a, b, c) std::fma((float)(a), (float)(b), (float)(c))
#define DM_OPAQUE(x) ((void)0)
#define DM_NOFOLD() ((void)0)
#define DM_UNIFORM_I(x) (x)
#else
#define DM_SQRT(x) sqrtf(x)
#define DM_SIN(x) sinf(x)
#define DM_COS(x) cosf(x)
#define DM_ATAN2(y, x) atan2f(y, x)
#define DM_POW(x, y) powf(x, y)
#define DM_FLOOR(x) floorf(x)
#define DM_EXP(x) expf(x)
#define DM_RINT(x) rintf(x)
#define DM_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
// hides a VGPR value's provenance from the optimiser (stops select chains from becoming scratch-array lookups)
#define DM_OPAQUE(x) asm("" : "+v"(x))
// between two `if (uniform condition) break;`: keeps them two s_cmp + s_cbranch_scc pairs (SimplifyCFG folds consecutive exits into one
// condition, and every term of that becomes a 64-bit lane mask: s_cselect_b64 per compare, s_or_b64 per term)
#define DM_NOFOLD() asm volatile("")
// a wave-uniform int that the compiler holds in a VGPR (loaded from LDS / computed by the VALU): move it to an SGPR
#define DM_UNIFORM_I(x) __builtin_amdgcn_readfirstlane(x)
#endif

#define DM_FLT_MIN 1.17549435e-38f   // smallest normal fp32

namespace dm {
DIAL_DEV float fminf_(float a, float b) { return a < b ? a : b; }
DIAL_DEV float fmaxf_(float a, float b) { return a > b ? a : b; }
DIAL_DEV float clip(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
DIAL_DEV float absf(float x) { return x < 0.f ? -x : x; }

DIAL_DEV float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
DIAL_DEV void cross3(float* o, const float* a, const float* b) {
  float r0 = a[1] * b[2] - a[2] * b[1], r1 = a[2] * b[0] - a[0] * b[2], r2 = a[0] * b[1] - a[1] * b[0];
  o[0] = r0; o[1] = r1; o[2] = r2;
}
DIAL_DEV void quat_mul(float* o, const float* u, const float* v) {
  float r0 = u[0] * v[0] - u[1] * v[1] - u[2] * v[2] - u[3] * v[3];
  float r1 = u[0] * v[1] + u[1] * v[0] + u[2] * v[3] - u[3] * v[2];
  float r2 = u[0] * v[2] - u[1] * v[3] + u[2] * v[0] + u[3] * v[1];
  float r3 = u[0] * v[3] + u[1] * v[2] - u[2] * v[1] + u[3] * v[0];
  o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}
// mjx math.rotate: 2(u.v)u + (s^2 - u.u)v + 2s(u x v)
DIAL_DEV void rotate(float* o, const float* vec, const float* q) {
  float s = q[0];
  float u[3] = {q[1], q[2], q[3]};
  float ud = dot3(u, vec), uu = dot3(u, u), c[3];
  cross3(c, u, vec);
  float r0 = 2.f * (ud * u[0]) + (s * s - uu) * vec[0] + 2.f * s * c[0];
  float r1 = 2.f * (ud * u[1]) + (s * s - uu) * vec[1] + 2.f * s * c[1];
  float r2 = 2.f * (ud * u[2]) + (s * s - uu) * vec[2] + 2.f * s * c[2];
  o[0] = r0; o[1] = r1; o[2] = r2;
}
DIAL_DEV void inv_rotate(float* o, const float* vec, const float* q) {
  float qc[4] = {q[0], -q[1], -q[2], -q[3]};
  rotate(o, vec, qc);
}
DIAL_DEV void quat_to_mat(float* m, const float* q) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2.f * (x * y - w * z); m[2] = 2.f * (x * z + w * y);
  m[3] = 2.f * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2.f * (y * z - w * x);
  m[6] = 2.f * (x * z - w * y); m[7] = 2.f * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
DIAL_DEV void normalize4(float* q) {
  float n = DM_SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n > 0.f) { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
// sin / cos of a joint half-angle.  On the GPU: the native v_sin_f32 / v_cos_f32 (argument in revolutions,
// abs error ~1e-6 on |x| <= pi); joint angles are bounded by the joint ranges so no range reduction is needed.
DIAL_DEV void fast_sincos(float x, float& s, float& c) {
#ifdef DIAL_EMU
  s = std::sin(x); c = std::cos(x);
#else
  const float r = x * 0.15915494309189535f;
  s = __builtin_amdgcn_sinf(r); c = __builtin_amdgcn_cosf(r);
#endif
}
DIAL_DEV void axis_angle_to_quat(float* q, const float* axis, float angle) {
  float s, c;
  fast_sincos(angle * 0.5f, s, c);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
DIAL_DEV void inert_mul(float* o, const float* I, const float* v) {
  float a0 = I[0] * v[0] + I[3] * v[1] + I[4] * v[2];
  float a1 = I[3] * v[0] + I[1] * v[1] + I[5] * v[2];
  float a2 = I[4] * v[0] + I[5] * v[1] + I[2] * v[2];
  float c1[3], c2[3];
  cross3(c1, I + 6, v + 3);
  cross3(c2, I + 6, v);
  float m = I[9];
  o[0] = a0 + c1[0]; o[1] = a1 + c1[1]; o[2] = a2 + c1[2];
  o[3] = m * v[3] - c2[0]; o[4] = m * v[4] - c2[1]; o[5] = m * v[5] - c2[2];
}
DIAL_DEV void motion_cross(float* o, const float* u, const float* v) {
  float a[3], b[3], c[3];
  cross3(a, u, v);
  cross3(b, u + 3, v);
  cross3(c, u, v + 3);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2];
  o[3] = b[0] + c[0]; o[4] = b[1] + c[1]; o[5] = b[2] + c[2];
}
DIAL_DEV void motion_cross_force(float* o, const float* v, const float* f) {
  float a[3], b[3], c[3];
  cross3(a, v, f);
  cross3(b, v + 3, f + 3);
  cross3(c, v, f + 3);
  o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2];
  o[3] = c[0]; o[4] = c[1]; o[5] = c[2];
}
}  // namespace dm
