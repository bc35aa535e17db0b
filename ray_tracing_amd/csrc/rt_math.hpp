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
