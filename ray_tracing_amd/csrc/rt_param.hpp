// rt_param.hpp -- the two parametric primitives of the reference and their Newton intersections:
// bicubic Bezier patch (CBezierSurface / CBezierObject, Bezier.h:91-138, Object.h:1085-1227) and Bezier
// surface of revolution (CBezierCurve / CRotationalSurface / CRotationalObject, Bezier.h:8-87,
// Object.h:791-1037).  Host + device.  The reference evaluates the surfaces by recursive de Casteljau; here
// the same lerps run as loops in the same order (the recursion only repeats identical sub-results), so with
// -ffp-contract=off the values round like the reference's.  Paths are relative to /root/reference/RayTracing/.
#pragma once
#include "rt_math.hpp"

namespace rt {

// pmin / pmax (both objects): a box that holds every point the Newton iterations can ACCEPT, used only to skip solves that cannot matter
// (see param_prune_box): the surface over its parameter domain lies in the convex hull of its control points; an accepted iterate has a
// step shorter than 1e-4 (|d(t,u,v)|^2 < minStepLen = 1e-8, Object.h:865, 1117), i.e. the ray point is within |J| 1e-4 of the surface point
// and the parameters within 1e-4 of the domain; the box is the hull's bounds grown by 10x that.  The reference has no such test (SURVEY
// R12: "no AABB pre-test"): it solves and finds nothing.
struct DBezier {
  double c[4][4][3]; double center[3];     // controlPts[i][j], center
  double pmin[3], pmax[3];
  double seed[25][3];                      // Point3D(i/4, j/4) + center of the 5x5 Newton seeds (Object.h:1094-1098): the same for every ray
  double sub[16][6];                       // param_prune_box: hull boxes (min xyz, max xyz) of the 4 x 4 sub-patches, grown like pmin / pmax
};
struct DBezierCtrl { double c[4][4][3]; double center[3]; };    // what the Newton steps read: a lane's private copy
struct DRotational {
  double prof[8][2];            // profile control points (u = radius, v = height), CBezierCurve::controlPts
  double center[3];
  double u_range;               // uRange
  int32_t n_prof, u_sample_num; // degree + 1, uSampleNum
  double bmin[3], bmax[3];      // m_AABB (Object.h:954-970)
  double pmin[3], pmax[3];      // see DBezier
  double shell[8][4];           // param_prune_shells: z0, z1, outer radius^2, inner radius^2 (0 = none) of 8 stacked cylinder shells around the surface
};

constexpr int kMaxStepNum = 10;          // Object.h:831
// Diagnostic build only (-DRT_SOLVE_STATS): Newton steps as the waves run them ([0] patch, [2] revolution: steps x 64) and as the lanes
// need them ([1], [3]: lanes active in each step); the ratio is the lock-step efficiency of the solve loops.
#if defined(RT_SOLVE_STATS) && defined(__HIPCC__)
__device__ unsigned long long g_solve_steps[16];  // deferred solver (rt_defer.inc): [4] patch pairs, [5] of them with a root, [6] / [7] the same for surfaces of revolution;
                                                  // [8] / [9] pairs / roots of closest-hit queries, [10] / [11] of occlusion queries (both object kinds)
#endif
#if defined(RT_SOLVE_STATS) && defined(__HIP_DEVICE_COMPILE__)
#define RT_SOLVE_LANE_STEPS(h) (h).steps++;
#define RT_SOLVE_COUNT(k) { const unsigned long long m_ = __builtin_amdgcn_ballot_w64(true); if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(m_)) { atomicAdd(&g_solve_steps[k], 64ull); atomicAdd(&g_solve_steps[k + 1], (unsigned long long)__builtin_popcountll(m_)); } }
#else
#define RT_SOLVE_COUNT(k)
#define RT_SOLVE_LANE_STEPS(h)
#endif
constexpr double kMinStepLen = 1e-8;     // :832 (= eps)
constexpr double kMaxDeviation = 0.3;    // :833
constexpr double kPiRef = 3.1415926535;  // Geometry.h:13

struct ParamHit {
  double t, u, v;
#ifdef RT_SOLVE_STATS
  int steps = 0;
#endif
};     // t = Dist(hit point, ray origin): what the scene compares (Scene.h:74); kMiss = none

RT_HD V3 lerp3(V3 a, V3 b, double s) { return a * (1 - s) + b * s; }          // P*(1-s) + Q*s, Bezier.h:33,121-127
RT_HD V3 cubic3(V3 p0, V3 p1, V3 p2, V3 p3, double s)
{
  V3 q0 = lerp3(p0, p1, s), q1 = lerp3(p1, p2, s), q2 = lerp3(p2, p3, s);
  V3 r0 = lerp3(q0, q1, s), r1 = lerp3(q1, q2, s);
  return lerp3(r0, r1, s);
}
template <class BZ> RT_HD V3 bez_ctrl(const BZ &B, int i, int j) { return mk(B.c[i][j][0], B.c[i][j][1], B.c[i][j][2]); }
// Point3D(u,v,0,j,3,0): the cubic in u down column j (Bezier.h:120-121)
template <class BZ> RT_HD V3 bez_col(const BZ &B, int j, double u) { return cubic3(bez_ctrl(B, 0, j), bez_ctrl(B, 1, j), bez_ctrl(B, 2, j), bez_ctrl(B, 3, j), u); }
// Point3D(u,v,i,0,0,3): the cubic in v along row i (Bezier.h:124-125)
template <class BZ> RT_HD V3 bez_row(const BZ &B, int i, double v) { return cubic3(bez_ctrl(B, i, 0), bez_ctrl(B, i, 1), bez_ctrl(B, i, 2), bez_ctrl(B, i, 3), v); }

// S(u,v), dS/du, dS/dv (Bezier.h:98, 130-138).  S and dS/dv share the four column points.
template <class BZ> RT_HD void bez_eval(const BZ &B, double u, double v, V3 &S, V3 &dSu, V3 &dSv)
{
  V3 c0 = bez_col(B, 0, u), c1 = bez_col(B, 1, u), c2 = bez_col(B, 2, u), c3 = bez_col(B, 3, u);
  V3 a0 = lerp3(c0, c1, v), a1 = lerp3(c1, c2, v), a2 = lerp3(c2, c3, v);
  V3 b0 = lerp3(a0, a1, v), b1 = lerp3(a1, a2, v);
  S = lerp3(b0, b1, v);                       // Point3D(u,v,0,0,3,3)
  dSv = (b1 - b0) * 3;                        // (Point3D(u,v,0,1,3,2) - Point3D(u,v,0,0,3,2)) * n
  V3 r0 = bez_row(B, 0, v), r1 = bez_row(B, 1, v), r2 = bez_row(B, 2, v), r3 = bez_row(B, 3, v);
  V3 e0 = lerp3(r0, r1, u), e1 = lerp3(r1, r2, u), e2 = lerp3(r2, r3, u);
  dSu = (lerp3(e1, e2, u) - lerp3(e0, e1, u)) * 3;   // (Point3D(u,v,1,0,2,3) - Point3D(u,v,0,0,2,3)) * m
}
template <class BZ> RT_HD V3 bez_point(const BZ &B, double u, double v)
{
  V3 c0 = bez_col(B, 0, u), c1 = bez_col(B, 1, u), c2 = bez_col(B, 2, u), c3 = bez_col(B, 3, u);
  return cubic3(c0, c1, c2, c3, v);
}
template <class BZ> RT_HD V3 bez_norm(const BZ &B, double u, double v)                        // Bezier.h:99
{
  V3 S, du, dv; bez_eval(B, u, v, S, du, dv);
  return unitize(cross(du, dv));
}

// One Newton update shared by both objects (Object.h:857-862, 1109-1114)
RT_HD void newton_step(V3 w, V3 o, V3 center, V3 S, V3 dPu, V3 dPv, double t, double u, double v, double &tp, double &up, double &vp)
{
  double D = dot(w, cross(dPu, dPv));
  V3 df = center + S - (o + w * t);
  tp = t + dot(dPu, cross(dPv, df)) / D;
  up = u + dot(w, cross(dPv, df)) / D;
  vp = v - dot(w, cross(dPu, df)) / D;
}

// CBezierObject::FirstInterPt and _Unrecorded (Object.h:1087-1202): 5x5 seeds, <= 10 Newton steps each
RT_HD ParamHit bezier_hit(const DBezier &B, V3 o, V3 d)
{
  const V3 center = mk(B.center[0], B.center[1], B.center[2]);
  double cur_u = kInfty, cur_v = kInfty, cur_t = kInfty;
  bool have = false;
  const double u_int = 1 / double(4), v_int = 1 / double(4);
  for (int i = 0; i <= 4; i++) {
    for (int j = 0; j <= 4; j++) {
      double u = i * u_int, v = j * v_int;
      double t = dist(bez_point(B, u, v) + center, o);
      for (int step = 0; step < kMaxStepNum; step++) {
        V3 S, dPu, dPv; bez_eval(B, u, v, S, dPu, dPv);
        double tp, up, vp; newton_step(d, o, center, S, dPu, dPv, t, u, v, tp, up, vp);
        if (up < -kMaxDeviation || up > 1 + kMaxDeviation || vp < -kMaxDeviation || vp > 1 + kMaxDeviation || tp < -kMaxDeviation) break;
        if (len2(mk(tp - t, up - u, vp - v)) < kMinStepLen && tp > kEps && 0 < up && up < 1 && 0 < vp && vp < 1) {
          have = true;
          if (tp < cur_t - kEps) { cur_u = up; cur_v = vp; cur_t = tp; break; }
        }
        u = up; v = vp; t = tp;
      }
    }
  }
  ParamHit h; h.t = kMiss; h.u = kInfty; h.v = kInfty;
  if (have) { h.t = dist(center + bez_point(B, cur_u, cur_v), o); h.u = cur_u; h.v = cur_v; }
  return h;
}

// The same search as the kernels run it.  Differences to the loops above, none of which changes the result:
//  * the seed points Point3D(i/4, j/4) + center come from the table made at scene creation (same arithmetic, no contraction);
//  * a seed is left as soon as its iterates repeat (x' == x, or x' == the iterate before x): every further step would reproduce an
//    iterate that has been examined already, against the same cur_t (measured on C4's camera rays: 244 -> 154 steps per hit ray,
//    bit-identical results; with 64 lanes in lock step the longest seed of the wave still sets the pace, 231);
//  * first_only (the caller only asks whether there is a hit): the first accepted iterate ends the search; the (t, u, v) returned
//    then belong to that iterate, not necessarily to the nearest root.
template <class BZ>
RT_HD ParamHit bezier_solve(const BZ &B, const double (*seed)[3], V3 o, V3 d, bool first_only)
{
  ParamHit h; h.t = kMiss; h.u = kInfty; h.v = kInfty;
  const V3 center = mk(B.center[0], B.center[1], B.center[2]);
  double cur_u = kInfty, cur_v = kInfty, cur_t = kInfty;
  bool have = false;
  for (int k = 0; k < 25; k++) {
    if (first_only && have) break;
    const int i = k / 5, j = k - i * 5;
    double u = i * (1 / double(4)), v = j * (1 / double(4));
    double t = dist(mk(seed[k][0], seed[k][1], seed[k][2]), o);
    double pt = kMiss, pu = kMiss, pv = kMiss;                     // the iterate before (t, u, v)
    for (int step = 0; step < kMaxStepNum; step++) {
      RT_SOLVE_COUNT(0)
      RT_SOLVE_LANE_STEPS(h)
      V3 S, dPu, dPv; bez_eval(B, u, v, S, dPu, dPv);
      double tp, up, vp; newton_step(d, o, center, S, dPu, dPv, t, u, v, tp, up, vp);
      if (up < -kMaxDeviation || up > 1 + kMaxDeviation || vp < -kMaxDeviation || vp > 1 + kMaxDeviation || tp < -kMaxDeviation) break;
      if (len2(mk(tp - t, up - u, vp - v)) < kMinStepLen && tp > kEps && 0 < up && up < 1 && 0 < vp && vp < 1) {
        have = true;
        if (tp < cur_t - kEps) { cur_u = up; cur_v = vp; cur_t = tp; break; }
      }
      if ((tp == t && up == u && vp == v) || (tp == pt && up == pu && vp == pv)) break;
      pt = t; pu = u; pv = v;
      u = up; v = vp; t = tp;
    }
  }
  if (have) { h.t = dist(center + bez_point(B, cur_u, cur_v), o); h.u = cur_u; h.v = cur_v; }
  return h;
}

// ---- CBezierCurve (Bezier.h:28-45) ----
// de Casteljau on control points i .. i+deg.  Written with compile-time array indices (fully unrolled, the bounds
// that depend on the degree are wave-uniform branches) so that the work array lives in registers: indexed
// dynamically it went to scratch memory, and the Newton loops spent their time there.
struct P2 { double u, v; };
RT_HD P2 curve_pt(const DRotational &R, double t, int i, int deg)              // Point2D(t, i, deg)
{
  double qu[8], qv[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { const int src = (i + k < 8) ? i + k : 7; qu[k] = R.prof[src][0]; qv[k] = R.prof[src][1]; }
#pragma unroll
  for (int lev = 1; lev < 8; lev++) {
    if (lev > deg) break;
#pragma unroll
    for (int k = 0; k + lev < 8; k++)
      if (k + lev <= deg) { qu[k] = qu[k] * (1 - t) + qu[k + 1] * t; qv[k] = qv[k] * (1 - t) + qv[k + 1] * t; }
  }
  P2 r; r.u = qu[0]; r.v = qv[0];
  return r;
}
// Point2D(t) and TanVec(t) from ONE pass: the level deg-1 pair (b0, b1) of the scheme IS Point2D(t, 0, deg-1) and
// Point2D(t, 1, deg-1) (same operations on the same control points), TanVec = (b1 - b0) * deg (Bezier.h:36-39) and
// Point2D(t) = b0 (1-t) + b1 t (Bezier.h:33)
// (RT: DRotational, or the same in the constant address space -- rt_defer.inc reads a launch's one object through the scalar cache)
template <class RT> RT_HD void curve_eval(const RT &R, double t, P2 &pt, P2 &tan)
{
  const int deg = R.n_prof - 1;
  double qu[8], qv[8];
#pragma unroll
  for (int k = 0; k < 8; k++) { qu[k] = R.prof[k][0]; qv[k] = R.prof[k][1]; }
#pragma unroll
  for (int lev = 1; lev < 7; lev++) {
    if (lev > deg - 1) break;
#pragma unroll
    for (int k = 0; k + lev < 8; k++)
      if (k + lev <= deg) { qu[k] = qu[k] * (1 - t) + qu[k + 1] * t; qv[k] = qv[k] * (1 - t) + qv[k + 1] * t; }
  }
  tan.u = (qu[1] - qu[0]) * deg; tan.v = (qv[1] - qv[0]) * deg;
  pt.u = qu[0] * (1 - t) + qu[1] * t; pt.v = qv[0] * (1 - t) + qv[1] * t;
}
RT_HD P2 curve_tan(const DRotational &R, double t)                             // TanVec
{
  P2 p, tv; curve_eval(R, t, p, tv);
  return tv;
}
RT_HD P2 curve_norm(const DRotational &R, double t)                            // Norm; CPoint2D::Unitize, Geometry.h:219-225
{
  P2 tv = curve_tan(R, t);
  P2 n; n.u = tv.v; n.v = -tv.u;
  const bool zero = fabs(n.u) < kEps && fabs(n.v) < kEps, inf = fabs(n.u - kInfty) < kEps && fabs(n.v - kInfty) < kEps;
  if (zero || inf) { n.u = kInfty; n.v = kInfty; return n; }
  double l = sqrt(n.u * n.u + n.v * n.v);
  P2 r; r.u = n.u / l; r.v = n.v / l;
  return r;
}
// ---- CRotationalSurface (Bezier.h:65-87) ----
// (sin / cos / acos of this file are rt_math.hpp's: the same bits on the host, on the device and in the oracle)
RT_HD V3 rot_point(const DRotational &R, double t, double th) { P2 p = curve_pt(R, t, 0, R.n_prof - 1); double s, c; rt_sincos(th, &s, &c); return mk(p.u * c, p.u * s, p.v); }
RT_HD V3 rot_norm(const DRotational &R, double t, double th) { P2 r = curve_norm(R, t); double s, c; rt_sincos(th, &s, &c); return unitize(mk(r.u * c, r.u * s, r.v)); }
RT_HD V3 rot_ds_dt(const DRotational &R, double t, double th) { P2 dd = curve_tan(R, t); double s, c; rt_sincos(th, &s, &c); return mk(c * dd.u, s * dd.u, dd.v); }
RT_HD V3 rot_ds_dth(const DRotational &R, double t, double th) { P2 p = curve_pt(R, t, 0, R.n_prof - 1); double s, c; rt_sincos(th, &s, &c); return mk(-(p.u) * s, p.u * c, 0); }
// S, dS/dt, dS/dtheta at once (what one Newton step needs): one curve pass, one sine, one cosine
template <class RT> RT_HD void rot_eval(const RT &R, double t, double th, V3 &S, V3 &dSt, V3 &dSth)
{
  P2 p, dd; curve_eval(R, t, p, dd);
  double s, c;
  rt_sincos(th, &s, &c);           // one argument reduction for both
  S = mk(p.u * c, p.u * s, p.v);
  dSt = mk(c * dd.u, s * dd.u, dd.v);
  dSth = mk(-(p.u) * s, p.u * c, 0);
}

inline void param_prune_box(DBezier &B)
{
  double edge = 0, lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
    for (int a = 0; a < 3; a++) { lo[a] = dmin(lo[a], B.c[i][j][a]); hi[a] = dmax(hi[a], B.c[i][j][a]); }
    if (i < 3) edge = dmax(edge, sqrt(dist2(mk(B.c[i][j][0], B.c[i][j][1], B.c[i][j][2]), mk(B.c[i + 1][j][0], B.c[i + 1][j][1], B.c[i + 1][j][2]))));
    if (j < 3) edge = dmax(edge, sqrt(dist2(mk(B.c[i][j][0], B.c[i][j][1], B.c[i][j][2]), mk(B.c[i][j + 1][0], B.c[i][j + 1][1], B.c[i][j + 1][2]))));
  }
  const double pad = 1e-3 * (1 + 2 * 3 * edge);          // |J| <= |w| + |dS/du| + |dS/dv|, a cubic's derivative <= 3 x the largest control-net edge
  for (int a = 0; a < 3; a++) { B.pmin[a] = lo[a] + B.center[a] - pad; B.pmax[a] = hi[a] + B.center[a] + pad; }
  const V3 center = mk(B.center[0], B.center[1], B.center[2]);
  for (int i = 0; i <= 4; i++) for (int j = 0; j <= 4; j++) {
    const V3 p = bez_point(B, i * (1 / double(4)), j * (1 / double(4))) + center;      // as bezier_hit computes it (no contraction: host == device)
    B.seed[i * 5 + j][0] = p.x; B.seed[i * 5 + j][1] = p.y; B.seed[i * 5 + j][2] = p.z;
  }
  // The same argument piece by piece: the surface over [a/4, (a+1)/4] x [b/4, (b+1)/4] lies in the convex hull of THAT piece's control net (the
  // blossom of the cubics at the piece's end parameters), so an accepted iterate lies in one of the 16 hull boxes grown by the pad.  A curved
  // sheet fills a small part of its one box; rays through the rest of it run 25 seeds to find nothing.
  auto blossom = [](const double p[4], double t1, double t2, double t3) {
    const double a0 = p[0] * (1 - t1) + p[1] * t1, a1 = p[1] * (1 - t1) + p[2] * t1, a2 = p[2] * (1 - t1) + p[3] * t1;
    const double b0 = a0 * (1 - t2) + a1 * t2, b1 = a1 * (1 - t2) + a2 * t2;
    return b0 * (1 - t3) + b1 * t3;
  };
  auto segment = [&](const double p[4], double a, double b, double q[4]) {
    q[0] = blossom(p, a, a, a); q[1] = blossom(p, a, a, b); q[2] = blossom(p, a, b, b); q[3] = blossom(p, b, b, b);
  };
  for (int pa = 0; pa < 4; pa++) for (int pb = 0; pb < 4; pb++) {
    const double u0 = pa / 4.0, u1 = (pa + 1) / 4.0, v0 = pb / 4.0, v1 = (pb + 1) / 4.0;
    double *S = B.sub[pa * 4 + pb];
    for (int a = 0; a < 3; a++) {
      double tmp[4][4], net[4][4];
      for (int j = 0; j < 4; j++) { const double col[4] = {B.c[0][j][a], B.c[1][j][a], B.c[2][j][a], B.c[3][j][a]}; double q[4]; segment(col, u0, u1, q); for (int i = 0; i < 4; i++) tmp[i][j] = q[i]; }
      for (int i = 0; i < 4; i++) { double q[4]; segment(tmp[i], v0, v1, q); for (int j = 0; j < 4; j++) net[i][j] = q[j]; }
      double mn = 1e300, mx = -1e300;
      for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { mn = dmin(mn, net[i][j]); mx = dmax(mx, net[i][j]); }
      S[a] = mn + B.center[a] - pad; S[3 + a] = mx + B.center[a] + pad;
    }
  }
}
inline void param_prune_box(DRotational &R)
{
  const int n = R.n_prof;
  double rmax = 0, vlo = 1e300, vhi = -1e300, edge = 0;
  for (int k = 0; k < n; k++) {
    rmax = dmax(rmax, fabs(R.prof[k][0])); vlo = dmin(vlo, R.prof[k][1]); vhi = dmax(vhi, R.prof[k][1]);
    if (k + 1 < n) edge = dmax(edge, sqrt((R.prof[k + 1][0] - R.prof[k][0]) * (R.prof[k + 1][0] - R.prof[k][0]) + (R.prof[k + 1][1] - R.prof[k][1]) * (R.prof[k + 1][1] - R.prof[k][1])));
  }
  const double pad = 1e-3 * (1 + (n - 1) * edge + rmax);   // |J| <= |w| + |dS/du| + |dS/dtheta|
  R.pmin[0] = R.center[0] - rmax - pad; R.pmax[0] = R.center[0] + rmax + pad;
  R.pmin[1] = R.center[1] - rmax - pad; R.pmax[1] = R.center[1] + rmax + pad;
  R.pmin[2] = R.center[2] + vlo - pad; R.pmax[2] = R.center[2] + vhi + pad;
}

// A tighter version of the same argument for the surface of revolution: the profile over [0, uRange] is cut into 8 pieces; each piece lies
// in the convex hull of ITS control polygon (de Casteljau subdivision), so the surface piece lies in the cylinder shell
// z in [min v, max v], radius in [min u, max |u|] around the axis through `center`; shells are grown by the same pad as the box.
// A cup is thin: most rays through its box meet no shell, and 51 seeds x ~7 Newton steps find nothing for them.
inline void param_prune_shells(DRotational &R)
{
  const int n = R.n_prof;
  double rall = 0, edge = 0;
  for (int k = 0; k < n; k++) {
    rall = dmax(rall, fabs(R.prof[k][0]));
    if (k + 1 < n) edge = dmax(edge, sqrt((R.prof[k + 1][0] - R.prof[k][0]) * (R.prof[k + 1][0] - R.prof[k][0]) + (R.prof[k + 1][1] - R.prof[k][1]) * (R.prof[k + 1][1] - R.prof[k][1])));
  }
  const double pad = 1e-3 * (1 + (n - 1) * edge + rall);
  for (int piece = 0; piece < 8; piece++) {
    const double a = R.u_range * piece / 8.0, b = R.u_range * (piece + 1) / 8.0;
    double q[8][2];
    for (int k = 0; k < n; k++) { q[k][0] = R.prof[k][0]; q[k][1] = R.prof[k][1]; }
    // left part of the split at b: the first points of every de Casteljau level
    double left[8][2];
    for (int lev = 0; lev < n; lev++) {
      left[lev][0] = q[0][0]; left[lev][1] = q[0][1];
      for (int k = 0; k + lev + 1 < n; k++) { q[k][0] = q[k][0] * (1 - b) + q[k + 1][0] * b; q[k][1] = q[k][1] * (1 - b) + q[k + 1][1] * b; }
    }
    // right part of that curve split at a / b: the last points of every level
    const double s = b > 0 ? a / b : 0;
    double lo_r = 1e300, hi_r = 0, lo_z = 1e300, hi_z = -1e300;
    for (int lev = 0; lev < n; lev++) {
      const double *pt = left[n - 1 - lev];
      lo_r = dmin(lo_r, pt[0]); hi_r = dmax(hi_r, fabs(pt[0])); lo_z = dmin(lo_z, pt[1]); hi_z = dmax(hi_z, pt[1]);
      for (int k = 0; k + lev + 1 < n; k++) { left[k][0] = left[k][0] * (1 - s) + left[k + 1][0] * s; left[k][1] = left[k][1] * (1 - s) + left[k + 1][1] * s; }
    }
    R.shell[piece][0] = lo_z - pad; R.shell[piece][1] = hi_z + pad;
    R.shell[piece][2] = (hi_r + pad) * (hi_r + pad);
    R.shell[piece][3] = lo_r - pad > 0 ? (lo_r - pad) * (lo_r - pad) : 0;
  }
}
// may the ray meet one of the shells at a parameter in (0, t_max]?
RT_HD bool rot_shells_may_matter(const DRotational &R, V3 o, V3 d, double t_max)
{
  const double ox = o.x - R.center[0], oy = o.y - R.center[1], oz = o.z - R.center[2];
  const double a = d.x * d.x + d.y * d.y, b = ox * d.x + oy * d.y, c0 = ox * ox + oy * oy;
  const bool vertical = a < 1e-18, level = fabs(d.z) < 1e-12;
  const double inv_a = vertical ? 0.0 : 1.0 / a, inv_z = level ? 0.0 : 1.0 / d.z;
  bool any = false;
  for (int k = 0; k < 8; k++) {
    const double z0 = R.shell[k][0], z1 = R.shell[k][1], r2 = R.shell[k][2], in2 = R.shell[k][3];
    double lo = 0, hi = t_max;
    bool ok = true;
    if (level) ok = oz >= z0 && oz <= z1;
    else { const double t1 = (z0 - oz) * inv_z, t2 = (z1 - oz) * inv_z; lo = dmax(lo, dmin(t1, t2)); hi = dmin(hi, dmax(t1, t2)); }
    const double c = c0 - r2;
    if (vertical) ok = ok && c <= 0;
    else {
      const double disc = b * b - a * c;
      if (disc < 0) ok = false;
      else { const double sq = sqrt(disc); lo = dmax(lo, (-b - sq) * inv_a); hi = dmin(hi, (-b + sq) * inv_a); }
    }
    ok = ok && lo <= hi;
    if (ok && in2 > 0) {                                  // the squared distance to the axis is convex in t: inside the hole at both ends = inside throughout
      const double ra = c0 + (2 * b + a * lo) * lo, rb = c0 + (2 * b + a * hi) * hi;
      if (ra < in2 && rb < in2) ok = false;
    }
    any = any || ok;
  }
  return any;
}

// CBoundingBox::Collide == inftyPt2D (Geometry.h:302-342): the reference's own slab rule (a ray parallel to a slab is
// never rejected by it; boxes behind the origin are accepted)
RT_HD double ref_plane_t(double p0a, double oa, double da)                     // IntersectWithPlane for an axis normal
{
  if (fabs(da) < kEps) return kInfty;
  double D = -p0a;
  return -(D + oa) / da;
}
RT_HD bool ref_box_miss(const double *bmin, const double *bmax, V3 o, V3 d)
{
  const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  double tmin_a[3], tmax_a[3];
  for (int a = 0; a < 3; a++) {
    double t1 = ref_plane_t(bmin[a], oo[a], dd[a]), t2 = ref_plane_t(bmax[a], oo[a], dd[a]);
    tmin_a[a] = dmin(t1, t2); tmax_a[a] = dmax(t1, t2);
  }
  double t_max = dmin(dmin(tmax_a[0], tmax_a[1]), tmax_a[2]);
  for (int a = 0; a < 3; a++) tmin_a[a] = (tmin_a[a] == kInfty) ? -kInfty : tmin_a[a];
  double t_min = dmax(dmax(tmin_a[0], tmin_a[1]), tmin_a[2]);
  if (t_min < t_max + kEps) return fabs(t_min - kInfty) < kEps && fabs(t_max - kInfty) < kEps;   // == inftyPt2D is an eps compare
  return true;
}

// CRotationalObject::t_v_Initial (Object.h:972-1037); false = inftyPt2D
RT_HD bool rot_seed(const DRotational &R, double u, V3 o, V3 d, double &t_out, double &v_out)
{
  P2 pt2 = curve_pt(R, u, 0, R.n_prof - 1);
  V3 p0 = mk(R.center[0], R.center[1], R.center[2] + pt2.v);
  V3 n = mk(0, 0, 1);
  V3 p; double t;
  if (fabs(dot(n, d)) < 0.1) {
    V3 l = p0 - o;
    double tp = dot(l, d);
    double d2 = len2(l) - tp * tp;
    if (d2 > pt2.u * pt2.u + kEps) return false;
    double t_prime = sqrt(pt2.u * pt2.u - d2);
    if (len2(l) < pt2.u * pt2.u - kEps * 1000) { t = tp + t_prime; p = o + d * t; }
    else if (len2(l) > pt2.u * pt2.u + kEps * 1000) {
      t = tp - t_prime;
      if (t > kEps) p = o + d * t; else return false;
    } else {
      if (tp < kEps * 1000) return false;
      t = tp * 2; p = o + d * (tp * 2);
    }
  } else {
    double D = -dot(p0, n);
    t = -(D + dot(n, o)) / dot(n, d);
    p = o + d * t;
    if (t < kEps) return false;
  }
  // the seed must not compare equal to inftyPt2D (Object.h:848): only possible for |t|, |v| ~ 1e8
  double cos_theta = (p - p0).x / len(p - p0);
  double theta = rt_acos(cos_theta);
  double v = (p.y > p0.y || (p.y == p0.y && p.x > p0.x)) ? theta : 2 * kPiRef - theta;
  if (fabs(t - kInfty) < kEps && fabs(v - kInfty) < kEps) return false;
  t_out = t; v_out = v;
  return true;
}

// CRotationalObject::FirstInterPt (record = true, t' > 10 eps) / _Unrecorded (t' > 5000 eps), Object.h:834-952
// the seeds and Newton iterations for a ray that passed the object's box test (Object.h:843-...)
RT_HD ParamHit rotational_solve(const DRotational &R, V3 o, V3 d, bool record)
{
  ParamHit h; h.t = kMiss; h.u = kInfty; h.v = kInfty;
  const V3 center = mk(R.center[0], R.center[1], R.center[2]);
  double cur_u = kInfty, cur_v = kInfty, cur_t = kInfty;
  bool have = false;
  const double t_thr = record ? kEps * 10 : kEps * 5000;
  const double u_int = R.u_range / double(R.u_sample_num);
  for (int i = 0; i <= R.u_sample_num; i++) {
    double u = i * u_int, v, t;
    if (!rot_seed(R, u, o, d, t, v)) continue;
    double pt = kMiss, pu = kMiss, pv = kMiss;                     // the iterate before (t, u, v)
    for (int step = 0; step < kMaxStepNum; step++) {
      RT_SOLVE_COUNT(2)
      RT_SOLVE_LANE_STEPS(h)
      V3 S, dPu, dPv; rot_eval(R, u, v, S, dPu, dPv);
      double tp, up, vp; newton_step(d, o, center, S, dPu, dPv, t, u, v, tp, up, vp);
      if (up < -kMaxDeviation || up > 1 + kMaxDeviation || tp < -kMaxDeviation) break;
      if (len2(mk(tp - t, up - u, vp - v)) < kMinStepLen && tp > t_thr && 0 < up && up < R.u_range) {
        have = true;
        if (tp < cur_t) { cur_u = up; cur_v = vp; cur_t = tp; break; }
      }
#ifndef RT_NO_REPEAT_EXIT
      // iterates that repeat (x' == x, or x' == the iterate before x) would be examined again against the same cur_t: the seed is done
      if ((tp == t && up == u && vp == v) || (tp == pt && up == pu && vp == pv)) break;
      pt = t; pu = u; pv = v;
#endif
      u = up; v = vp; t = tp;
    }
  }
  if (have) { h.t = dist(center + rot_point(R, cur_u, cur_v), o); h.u = cur_u; h.v = cur_v; }
  return h;
}
RT_HD ParamHit rotational_hit(const DRotational &R, V3 o, V3 d, bool record)
{
  if (ref_box_miss(R.bmin, R.bmax, o, d)) { ParamHit h; h.t = kMiss; h.u = kInfty; h.v = kInfty; return h; }   // Object.h:836-838
  return rotational_solve(R, o, d, record);
}

}  // namespace rt
