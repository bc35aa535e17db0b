// rt_math.hpp -- 3-vector, colour and primitive arithmetic of the hot path, written once for
// host (scene preparation inside the library) and device (gfx950 kernels).
//
// Everything here that produces a value the renderer reports (hit parameter t, normals,
// reflection/refraction directions, Fresnel weights, Phong terms) keeps the reference's
// operation order so that, built with -ffp-contract=off, it rounds exactly like a CPU build
// of the reference.  Paths cited are relative to /root/reference/RayTracing/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RT_HD __host__ __device__ __forceinline__

namespace rt {

constexpr double kEps = 1e-8;    // Geometry.h:12
constexpr double kInfty = 1e+8;  // Geometry.h:48
constexpr double kMiss = 1e300;  // "no hit" for the parametric hit record

struct V3 { double x, y, z; };

RT_HD V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }       // Geometry.h:24
RT_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }       // :25
RT_HD V3 operator*(V3 a, double l) { return mk(a.x * l, a.y * l, a.z * l); }         // :26
RT_HD V3 operator/(V3 a, double l) { return mk(a.x / l, a.y / l, a.z / l); }         // :27
RT_HD V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }                                  // :90-95 (by value)
RT_HD V3 cross(V3 a, V3 b)                                                            // :51-56
{ return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           // :121-124
RT_HD double dist2(V3 a, V3 b)                                                        // :133-138
{ return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z); }
RT_HD double len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }                // :37 (x-0 == x)
RT_HD double len(V3 a) { return sqrt(len2(a)); }                                      // :36
RT_HD double dist(V3 a, V3 b) { return sqrt(dist2(a, b)); }                           // :126-131
RT_HD bool veq(V3 a, V3 b)                                                            // :97-103
{ return fabs(a.x - b.x) < kEps && fabs(a.y - b.y) < kEps && fabs(a.z - b.z) < kEps; }
RT_HD V3 infty_pt() { return mk(kInfty, kInfty, kInfty); }
RT_HD bool is_infty_pt(V3 a) { return veq(a, infty_pt()); }
RT_HD V3 unitize(V3 a)                                                                // :113-119
{
  if (veq(a, mk(0, 0, 0)) || is_infty_pt(a)) return infty_pt();
  double l = len(a);
  return mk(a.x / l, a.y / l, a.z / l);
}
RT_HD double dmin(double a, double b) { return (b < a) ? b : a; }   // std::min
RT_HD double dmax(double a, double b) { return (a < b) ? b : a; }   // std::max

// ---- sine, cosine and arc cosine by the same operations on every host and device -------------------------------------------------------
// The angle of a surface of revolution is unbounded (Object.h:864, 870 are commented out): two Newton seeds can reach the same point 2 pi k
// apart, and which root is "nearer" is decided in the last bits of t'.  With the C library's sin / cos / acos (glibc on the host, the device
// library on the GPU; both accurate, neither the other's bit pattern) the checker and the kernels could take different roots.  These use
// + - x / floor sqrt only (IEEE operations, -ffp-contract=off), so every build rounds alike: Cody-Waite reduction by a three-part pi/2
// (exact products for |x| < ~1e6; beyond that the result is still the same everywhere, only less accurate) and the polynomial kernels
// of Sun's fdlibm (k_sin.c, k_cos.c, e_acos.c without the split of sqrt's result); errors <= 2 ulp against glibc on [-200, 200] / [-1, 1].
// oracle/rt_oracle.c restates them operation for operation.
RT_HD void rt_sincos(double x, double *s_out, double *c_out)
{
  const double n = floor(x * 6.36619772367581382433e-01 + 0.5);                  // nearest multiple of pi/2
  double r = ((x - n * 1.57079632673412561417e+00) - n * 6.07710050630396597660e-11) - n * 2.02226624871116645580e-21;
  r = r - n * 8.47842766036889956997e-32;
  const double z = r * r;
  const double s = r + r * z * (-1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                   z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
  const double c = 1.0 - 0.5 * z + z * z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 +
                   z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double q = n - 4.0 * floor(n * 0.25);                                    // n mod 4 in {0, 1, 2, 3}
  if (q == 0.0) { *s_out = s; *c_out = c; }
  else if (q == 1.0) { *s_out = c; *c_out = -s; }
  else if (q == 2.0) { *s_out = -s; *c_out = -c; }
  else { *s_out = -c; *c_out = s; }
}
RT_HD double rt_sin(double x) { double s, c; rt_sincos(x, &s, &c); return s; }
RT_HD double rt_cos(double x) { double s, c; rt_sincos(x, &s, &c); return c; }
RT_HD double rt_acos_ratio(double z)
{
  const double p = z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 + z * (2.01212532134862925881e-01 + z * (-4.00555345006794114027e-02 +
                   z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
  const double q = 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 + z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
  return p / q;
}
RT_HD double rt_acos(double x)
{
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
  const double ax = fabs(x);
  if (!(ax < 1.0)) {
    if (x == 1.0) return 0.0;
    if (x == -1.0) return pi + 2.0 * pio2_lo;
    return (x - x) / (x - x);                                                     // NaN, like acos
  }
  if (ax < 0.5) {
    if (ax < 6.9388939039072284e-18) return pio2_hi + pio2_lo;
    return pio2_hi - (x - (pio2_lo - x * rt_acos_ratio(x * x)));
  }
  if (x < 0) {
    const double z = (1.0 + x) * 0.5, s = sqrt(z), w = rt_acos_ratio(z) * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  const double z = (1.0 - x) * 0.5, s = sqrt(z), w = rt_acos_ratio(z) * s;
  return 2.0 * (s + w);
}

// ---- CColour (Colour.h:26-63): every operation clamps to <= 255 -----------------------------
struct Col { double r, g, b; };
RT_HD Col mkc(double r, double g, double b) { Col c; c.r = r; c.g = g; c.b = b; return c; }
RT_HD Col cadd(Col a, Col b) { return mkc(dmin(255.0, a.r + b.r), dmin(255.0, a.g + b.g), dmin(255.0, a.b + b.b)); }
RT_HD Col cmul(Col a, double w) { return mkc(dmin(255.0, a.r * w), dmin(255.0, a.g * w), dmin(255.0, a.b * w)); }
RT_HD Col cdiv(Col a, double w) { return mkc(dmin(255.0, a.r / w), dmin(255.0, a.g / w), dmin(255.0, a.b / w)); }

// ---- Eigen fixed 3x3 determinant (the only Eigen call of the reference, Object.h:436-438) -----
RT_HD double det3(double m00, double m01, double m02, double m10, double m11, double m12,
                  double m20, double m21, double m22)
{
  return m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
}

// ---- primitive tests: return the hit parameter t (hit point = o + d*t) or kMiss ----------------
// CTriangle::FirstInterPt, Object.h:416-446
RT_HD double tri_exact_t(V3 v0, V3 v1, V3 v2, V3 m_norm, V3 o, V3 d)
{
  if (fabs(dot(d, m_norm)) < kEps) return kMiss;
  V3 E1 = v0 - v1, E2 = v0 - v2, S = v0 - o;
  double dA1 = det3(S.x, E1.x, E2.x, S.y, E1.y, E2.y, S.z, E1.z, E2.z);
  double dA2 = det3(d.x, S.x, E2.x, d.y, S.y, E2.y, d.z, S.z, E2.z);
  double dA3 = det3(d.x, E1.x, S.x, d.y, E1.y, S.y, d.z, E1.z, S.z);
  double dB = det3(d.x, E1.x, E2.x, d.y, E1.y, E2.y, d.z, E1.z, E2.z);
  double t = dA1 / dB, belta = dA2 / dB, gamma = dA3 / dB;
  if (t > kEps && belta > -kEps && belta < 1 + kEps && gamma > -kEps && gamma < 1 + kEps &&
      belta + gamma < 1 + kEps)
    return t;
  return kMiss;
}
// CSphere::FirstInterPt, Object.h:250-278
RT_HD double sphere_t(V3 c, double radius, V3 o, V3 d)
{
  V3 l = c - o;
  double tp = dot(l, d);
  double d2 = len2(l) - tp * tp;
  if (d2 > radius * radius + kEps) return kMiss;
  double t_prime = sqrt(radius * radius - d2);
  if (len2(l) < radius * radius - kEps) return tp + t_prime;
  else if (len2(l) > radius * radius + kEps) {
    double t = tp - t_prime;
    return (t > kEps) ? t : kMiss;
  } else {
    if (tp < kEps) return kMiss;
    return tp * 2;
  }
}
// CInftyPlane::FirstInterPt, Object.h:708-723 (a grazing in-plane ray returns its origin: t = 0)
RT_HD double plane_t(V3 p0, V3 n, V3 o, V3 d)
{
  if (fabs(dot(n, d)) < kEps) {
    if (fabs(dot(p0 - o, n)) < kEps) return 0.0;
    return kMiss;
  }
  double D = -dot(p0, n);
  double t = -(D + dot(n, o)) / dot(n, d);
  if (t < kEps) return kMiss;
  return t;
}
// CCircle::FirstInterPt, Object.h:667-678
RT_HD double circle_t(V3 p0, V3 n, double radius, V3 o, V3 d)
{
  if (fabs(dot(n, d)) < kEps) return kMiss;
  double D = -dot(p0, n);
  double t = -(D + dot(n, o)) / dot(n, d);
  V3 pt = o + d * t;
  if (t < kEps || dist(pt, p0) > radius - kEps) return kMiss;
  return t;
}

// ---- counter-based generator standing in for rand() (Light.h:11); same function as the oracle's ---
RT_HD uint64_t mix64(uint64_t z)
{
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
RT_HD uint32_t rng_u32(uint64_t seed, uint32_t key0, uint32_t key1, uint32_t counter)
{
  uint64_t x = mix64(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)key0 + 1));
  x = mix64(x ^ (0xD1B54A32D192ED03ULL * ((uint64_t)key1 + 1)));
  x = mix64(x ^ (0x8CB92BA72F3D8DD7ULL * ((uint64_t)counter + 1)));
  return (uint32_t)(x >> 32);
}
// counter = path(12) | light(4) | light-sample(14) | dim(2).  path < 4096 is guaranteed by the callers: the render entry points refuse a
// camera depth above 12 (a ray of level L has path < 2^(L+1)), so no two rays of a pixel sample share a stream
RT_HD uint32_t rng_counter(uint32_t path, uint32_t light, uint32_t lsample, uint32_t dim)
{ return (path << 20) | ((light & 15u) << 16) | ((lsample & 0x3FFFu) << 2) | (dim & 3u); }
// random(r, x) = 2 r (rand() % x) / x - r, Light.h:11, rand() := 31-bit draw
RT_HD double ref_random(double r, int x, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t path,
                        uint32_t light, uint32_t lsample, uint32_t dim)
{
  uint32_t k = rng_u32(seed, pixel, sample, rng_counter(path, light, lsample, dim)) >> 1;
  return 2 * r * (double)(k % (uint32_t)x) / (double)x - r;
}

}  // namespace rt
