// rt_device.hpp -- device-side scene layout, wave-packet BSP traversal and shading for gfx950.
//
// Execution model (wave64): one ray per lane, 64 rays per wave walk the tree TOGETHER.  The node
// stack is wave-uniform (one 32-bit index per entry, kept in LDS), node boxes and the leaf's
// triangle records are addressed with wave-uniform indices, so the compiler fetches them with
// scalar loads (s_load_*) and the vector ALU only does per-ray arithmetic.  A lane takes part in a
// node only if its own ray overlaps the node's box in front of its current best hit; the wave
// skips a node when no lane wants it (__ballot).  Children are pushed far-first according to the
// majority ray direction, so the walk is front-to-back for most lanes and every lane prunes with
// its own best t.
//
// Leaf triangles are tested in two steps:
//   cull    -- a cheap plane + barycentric test on a precomputed 128-byte record, with margins
//              that are wide compared with every rounding error involved (see TriCull);
//   confirm -- the reference's own arithmetic (rt::tri_exact_t, Object.h:416-446) on the raw
//              vertices, only for lanes that survive the cull.
// Accepted hits therefore carry exactly the t the reference formula produces.
#pragma once
#include "rt_math.hpp"

namespace rt {

// ---- device scene ------------------------------------------------------------------------------
struct alignas(64) DNode {          // 64 B, box inflated by kBoxPad
  double bmin[3], bmax[3];
  int32_t axis;                     // >= 0: inner (a = left, b = right); < 0: leaf with (-1-axis) non-triangle refs
  int32_t a;                        // leaf: first cull record
  int32_t b;                        // leaf: number of cull records
  int32_t c;                        // leaf: first entry of leaf_other[]
};

// Cull record of one triangle reference, 128 B, stored in leaf order (a triangle referenced by k
// leaves is stored k times so that a leaf is one contiguous stream).
//   n, dpl    : unit normal m_norm and dpl = dot(v0, n)       -> t' = (dpl - o.n) / (d.n)
//   B, b0     : beta'  = dot(P', B) + b0                       (dual basis of E1' = v1 - v0)
//   G, g0     : gamma' = dot(P', G) + g0                       (dual basis of E2' = v2 - v0)
//   bsum      : |B|_1 + |G|_1 + 1/sin(angle at v0), scales the per-ray margin
// A record with n = 0 (degenerate or sliver triangle) always goes to the confirm step.
struct alignas(128) TriCull {
  double n[3], dpl;
  double B[3], b0;
  double G[3], g0;
  double bsum;
  int32_t prim_id, tri_idx;
  double pad[2];
};

struct alignas(32) TriExact { double v0[3], v1[3], v2[3], nrm[3]; };   // 96 B, by triangle index
struct TriShade { double vn[3][3]; int32_t has_vn; int32_t pad; };      // vertex normals, by triangle index
struct DPlane { double p0[3], n[3]; };
struct DSphere { double c[3], r; };
struct DCircle { double p0[3], n[3], r; };
struct DMaterial { double reflW, refrW, RI, diff, spec, shiness, emissive, ambient; };
struct DLight { int32_t type, sample_num; double p[3], lum, len, lum_per_sample; };

struct DevScene {
  const DNode *nodes; int32_t n_nodes;          // n_nodes == 0: no tree
  const TriCull *cull;
  const int32_t *leaf_other;                    // non-triangle prim ids referenced by leaves
  const int32_t *root_objs; int32_t n_root_objs;// bspTree.root->objects (infinite objects)
  const TriExact *tri_exact; const TriShade *tri_shade;
  const DPlane *planes; const DSphere *spheres; const DCircle *circles;
  const uint8_t *obj_type; const uint32_t *obj_index; const uint32_t *obj_mat; const double *obj_colour;
  int32_t n_objects;
  const DMaterial *materials;
  const DLight *lights; int32_t n_lights;
  double ambient;
  double coord_scale;                           // max |coordinate| of finite geometry
};

struct Stats { unsigned long long rays, shadow_rays, node_visits, prim_tests, exact_tests, wave_nodes, wave_prims, pad; };

constexpr double kBoxPad = 1e-7;   // spatial inflation of node boxes (covers slab-test rounding)
constexpr double kGMin = 1e-5;     // |d.n| below this: skip the cull, go to the confirm step

#define RT_STACK_DEPTH 64

// uniform (scalar) copy of a value that is known to be the same in all lanes
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- per-object exact test (CBaseObject::FirstInterPt dispatch, Object.h:37) --------------------
__device__ __forceinline__ double object_t(const DevScene &sc, int id, V3 o, V3 d)
{
  const int type = sc.obj_type[id];
  const uint32_t k = sc.obj_index[id];
  if (type == 1) {        // RT_HIP_TRIANGLE
    const TriExact &T = sc.tri_exact[k];
    return tri_exact_t(mk(T.v0[0], T.v0[1], T.v0[2]), mk(T.v1[0], T.v1[1], T.v1[2]), mk(T.v2[0], T.v2[1], T.v2[2]),
                       mk(T.nrm[0], T.nrm[1], T.nrm[2]), o, d);
  } else if (type == 4) { // RT_HIP_INFTY_PLANE
    const DPlane &P = sc.planes[k];
    return plane_t(mk(P.p0[0], P.p0[1], P.p0[2]), mk(P.n[0], P.n[1], P.n[2]), o, d);
  } else if (type == 0) { // RT_HIP_SPHERE
    const DSphere &S = sc.spheres[k];
    return sphere_t(mk(S.c[0], S.c[1], S.c[2]), S.r, o, d);
  } else if (type == 3) { // RT_HIP_CIRCLE
    const DCircle &C = sc.circles[k];
    return circle_t(mk(C.p0[0], C.p0[1], C.p0[2]), mk(C.n[0], C.n[1], C.n[2]), C.r, o, d);
  }
  return kMiss;
}

// The occlusion predicate of Scene.h:449 / Light.h:66,147: the hit X = o + d t lies between the
// shaded point o and the reference point.  Without a reference point (parallel light,
// Light.h:91) any hit occludes.
__device__ __forceinline__ bool occludes(V3 o, V3 d, double t, bool has_ref, V3 ref)
{
  if (!has_ref) return true;
  V3 X = o + d * t;
  return dot(o - X, ref - X) < -kEps;
}

struct RayState {
  V3 o, d;
  double inv[3];      // 1/d per axis (inf-safe: see slab())
  double best_t;      // closest accepted t so far (kMiss = none); for ANY mode: upper bound of interest
  int best_id;
  double tmargin;     // per-ray t margin of the cull
  double mscale;      // per-ray scale of the barycentric margin
  double gmin;        // per-ray |d.n| threshold
};

__device__ __forceinline__ void ray_setup(RayState &r, V3 o, V3 d, double coord_scale)
{
  r.o = o; r.d = d;
  r.inv[0] = 1.0 / d.x; r.inv[1] = 1.0 / d.y; r.inv[2] = 1.0 / d.z;
  r.best_t = kMiss; r.best_id = -1;
  double os = dmax(dmax(fabs(o.x), fabs(o.y)), fabs(o.z));
  double s = 1.0 + coord_scale + os;
  r.tmargin = 1e-6 * s;
  r.mscale = 1e-9 * s;
  r.gmin = dmax(kGMin, 1e-7 * s);
}

// slab test against an (inflated) box; returns false when the ray misses it
__device__ __forceinline__ bool slab(const RayState &r, const double *bmin, const double *bmax, double &tmin, double &tmax)
{
  const double o[3] = {r.o.x, r.o.y, r.o.z};
  const double d[3] = {r.d.x, r.d.y, r.d.z};
  tmin = -kMiss; tmax = kMiss;
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (fabs(d[a]) < 1e-30) {           // parallel to the slab: inside or never
      ok = ok && (o[a] >= bmin[a]) && (o[a] <= bmax[a]);
    } else {
      double t1 = (bmin[a] - o[a]) * r.inv[a];
      double t2 = (bmax[a] - o[a]) * r.inv[a];
      tmin = fmax(tmin, fmin(t1, t2));
      tmax = fmin(tmax, fmax(t1, t2));
    }
  }
  return ok && (tmin <= tmax);
}

// One triangle reference for all lanes of the wave.  `want` = lanes for which this leaf matters.
// Returns the confirmed t (kMiss if none) for the calling lane.
template <bool STATS>
__device__ __forceinline__ double leaf_triangle(const DevScene &sc, const TriCull &c, const RayState &r, bool want,
                                                double t_hi, unsigned long long &n_exact)
{
  // ---- cull (explicit FMAs: this part only has to be conservative) ----
  const double g = fma(r.d.z, c.n[2], fma(r.d.y, c.n[1], r.d.x * c.n[0]));
  const double on = fma(r.o.z, c.n[2], fma(r.o.y, c.n[1], r.o.x * c.n[0]));
  const double sn = c.dpl - on;
  double rg = __builtin_amdgcn_rcp(g);
  rg = fma(fma(-g, rg, 1.0), rg, rg);              // one Newton step: ~2^-50 relative
  const double tc = sn * rg;
  const double px = fma(r.d.x, tc, r.o.x), py = fma(r.d.y, tc, r.o.y), pz = fma(r.d.z, tc, r.o.z);
  const double be = fma(pz, c.B[2], fma(py, c.B[1], fma(px, c.B[0], c.b0)));
  const double ga = fma(pz, c.G[2], fma(py, c.G[1], fma(px, c.G[0], c.g0)));
  const double m = fma(r.mscale, c.bsum, 1e-6);
  const bool small_g = !(fabs(g) >= r.gmin);        // also catches n = 0 and NaN
  const bool inside = (tc > -r.tmargin) && (tc < t_hi + r.tmargin) && (be > -m) && (ga > -m) && (be + ga < 1.0 + m);
  const bool pass = want && (small_g || inside);
  double t = kMiss;
  if (pass) {
    // ---- confirm: the reference's arithmetic on the raw vertices ----
    if (STATS) n_exact++;
    const TriExact &T = sc.tri_exact[c.tri_idx];
    t = tri_exact_t(mk(T.v0[0], T.v0[1], T.v0[2]), mk(T.v1[0], T.v1[1], T.v1[2]), mk(T.v2[0], T.v2[1], T.v2[2]),
                    mk(T.nrm[0], T.nrm[1], T.nrm[2]), r.o, r.d);
  }
  return t;
}

// Packet traversal.  ANY = false: closest hit into r.best_t / r.best_id (Scene.h:85-91 semantics).
// ANY = true: returns true for a lane as soon as some hit satisfies occludes(); r.best_t must be
// preset to the largest t of interest (kMiss if unbounded).
// `stack` points to this wave's RT_STACK_DEPTH ints in LDS.  All lanes of the wave must call this
// together (inactive rays pass active = false).
template <bool ANY, bool STATS>
__device__ __forceinline__ bool traverse(const DevScene &sc, RayState &r, bool active, bool has_ref, V3 ref,
                                         int *stack, Stats &st)
{
  bool found = false;        // ANY mode result
  // update rule shared by every candidate (Scene.h:74-79): first candidate wins ties within eps
  auto candidate = [&](double t, int id, bool lane_on) {
    if (!lane_on || !(t < kMiss)) return;
    if (ANY) {
      if (occludes(r.o, r.d, t, has_ref, ref)) found = true;
    } else {
      if (r.best_id < 0 || t < r.best_t - kEps) { r.best_t = t; r.best_id = id; }
    }
  };

  if (sc.n_nodes > 0) {
    int sp = 0;
    stack[0] = 0; sp = 1;
    while (sp > 0) {
      const int ni = uni(stack[--sp]);
      const DNode &nd = sc.nodes[ni];
      double tmin, tmax;
      bool live = active && !(ANY && found);
      bool want = live && slab(r, nd.bmin, nd.bmax, tmin, tmax) && (tmax > 0.0) && (tmin <= r.best_t);
      if (STATS) { if (live) st.node_visits++; if ((threadIdx.x & 63) == 0) st.wave_nodes++; }
      const unsigned long long wmask = __ballot(want);
      if (wmask == 0ull) {
        if (__ballot(live) == 0ull) break;   // every ray of the wave is finished
        continue;
      }
      const int axis = uni(nd.axis);
      if (axis >= 0) {
        // inner node: near child first for the majority of the interested lanes
        const double da = (axis == 0) ? r.d.x : ((axis == 1) ? r.d.y : r.d.z);
        const unsigned long long pos = __ballot(want && (da >= 0.0));
        const bool left_first = (2 * __popcll(pos) >= __popcll(wmask));
        const int l = uni(nd.a), rr = uni(nd.b);
        if (sp + 2 <= RT_STACK_DEPTH) {
          stack[sp++] = left_first ? rr : l;
          stack[sp++] = left_first ? l : rr;
        }
      } else {
        const int first = uni(nd.a), cnt = uni(nd.b);
        for (int i = 0; i < cnt; i++) {
          const TriCull &c = sc.cull[first + i];
          if (STATS) { if (want) st.prim_tests++; if ((threadIdx.x & 63) == 0) st.wave_prims++; }
          const bool lane_on = want && !(ANY && found);
          double t = leaf_triangle<STATS>(sc, c, r, lane_on, r.best_t, st.exact_tests);
          candidate(t, uni(c.prim_id), lane_on);
        }
        const int n_other = -1 - axis;
        const int ofirst = uni(nd.c);
        for (int i = 0; i < n_other; i++) {
          const int id = uni(sc.leaf_other[ofirst + i]);
          const bool lane_on = want && !(ANY && found);
          double t = lane_on ? object_t(sc, id, r.o, r.d) : kMiss;
          if (STATS && lane_on) st.exact_tests++;
          candidate(t, id, lane_on);
        }
      }
    }
    // root's infinite objects (Scene.h:329-346, 369-386), once per ray
    for (int i = 0; i < sc.n_root_objs; i++) {
      const int id = uni(sc.root_objs[i]);
      const bool lane_on = active && !(ANY && found);
      double t = lane_on ? object_t(sc, id, r.o, r.d) : kMiss;
      if (STATS && lane_on) st.exact_tests++;
      candidate(t, id, lane_on);
    }
  } else {
    // no tree: CScene::FirstInterPt (Scene.h:62-83) -- every object, in order
    for (int id = 0; id < sc.n_objects; id++) {
      const bool lane_on = active && !(ANY && found);
      if (ANY && __ballot(lane_on) == 0ull) break;
      double t = lane_on ? object_t(sc, id, r.o, r.d) : kMiss;
      if (STATS && lane_on) st.exact_tests++;
      candidate(t, id, lane_on);
    }
  }
  return found;
}

// ---- shading ------------------------------------------------------------------------------------
// CBaseObject::Norm without bump map (Object.h:79-83) -> InitialNorm of the object type
__device__ __forceinline__ V3 object_norm(const DevScene &sc, int id, V3 pt)
{
  const int type = sc.obj_type[id];
  const uint32_t k = sc.obj_index[id];
  if (type == 1) {                                    // CTriangle::InitialNorm, Object.h:448-457
    const TriExact &T = sc.tri_exact[k];
    const TriShade &S = sc.tri_shade[k];
    if (!S.has_vn) return mk(T.nrm[0], T.nrm[1], T.nrm[2]);
    V3 v0 = mk(T.v0[0], T.v0[1], T.v0[2]), v1 = mk(T.v1[0], T.v1[1], T.v1[2]), v2 = mk(T.v2[0], T.v2[1], T.v2[2]);
    double S0 = len(cross(v1 - pt, v2 - pt));
    double S1 = len(cross(v0 - pt, v2 - pt));
    double S2 = len(cross(v0 - pt, v1 - pt));
    double Sa = len(cross(v1 - v0, v2 - v0));
    V3 n0 = mk(S.vn[0][0], S.vn[0][1], S.vn[0][2]), n1 = mk(S.vn[1][0], S.vn[1][1], S.vn[1][2]),
       n2 = mk(S.vn[2][0], S.vn[2][1], S.vn[2][2]);
    return n0 * (S0 / Sa) + n1 * (S1 / Sa) + n2 * (S2 / Sa);
  } else if (type == 4) {                             // Object.h:693
    const DPlane &P = sc.planes[k]; return mk(P.n[0], P.n[1], P.n[2]);
  } else if (type == 0) {                             // Object.h:239
    const DSphere &S = sc.spheres[k]; return unitize(pt - mk(S.c[0], S.c[1], S.c[2]));
  } else if (type == 3) {                             // Object.h:660
    const DCircle &C = sc.circles[k]; return mk(C.n[0], C.n[1], C.n[2]);
  }
  return infty_pt();
}

// CBaseObject::ReflectRay direction, Object.h:94-101 (Unitize twice: :100 and the CRay ctor)
__device__ __forceinline__ V3 reflect_dir(V3 N, V3 d)
{
  if (dot(d, N) > kEps) N = neg(N);
  V3 dir = d - N * (2 * dot(d, N));
  return unitize(unitize(dir));
}
// CBaseObject::RefractRay direction, Object.h:103-126; infty_pt() on total reflection
__device__ __forceinline__ V3 refract_dir(V3 N, V3 d, double RI)
{
  V3 T = infty_pt();
  if (dot(d, N) < kEps) {
    double ni_nt = 1 / RI;
    double c2 = dot(d, N) * dot(d, N);
    double ct = sqrt(1 - ni_nt * ni_nt * (1 - c2));
    T = N * (ni_nt * dot(neg(d), N) - ct) + d * ni_nt;
  } else {
    N = neg(N);
    double ni_nt = RI;
    double c2 = dot(d, N) * dot(d, N);
    if (1 - ni_nt * ni_nt * (1 - c2) > kEps) {
      double ct = sqrt(1 - ni_nt * ni_nt * (1 - c2));
      T = N * (ni_nt * dot(neg(d), N) - ct) + d * ni_nt;
    }
  }
  return unitize(unitize(T));
}
// (r_p^2 + r_v^2)/2 of Object.h:128-155 / 157-184
__device__ __forceinline__ double fresnel_R(V3 N, V3 d, double RI)
{
  double ni_nt = 0, ci = 0, ct = 0;
  if (dot(d, N) < kEps) {
    ni_nt = 1 / RI;
    double c2 = dot(d, N) * dot(d, N);
    ci = sqrt(c2);
    ct = sqrt(1 - ni_nt * ni_nt * (1 - c2));
  } else {
    N = neg(N);
    ni_nt = RI;
    double c2 = dot(d, N) * dot(d, N);
    ci = sqrt(c2);
    if (1 - ni_nt * ni_nt * (1 - c2) > kEps) ct = sqrt(1 - ni_nt * ni_nt * (1 - c2));
  }
  double r_p = (ni_nt * ci - ct) / (ni_nt * ci + ct);
  double r_v = (ci - ni_nt * ct) / (ci + ni_nt * ct);
  return (r_p * r_p + r_v * r_v) / 2;
}

struct RngKey { uint64_t seed; uint32_t pixel, sample, path; };

// luminosity of one light at pt: CBaseLight::luminosity (Light.h:34-44) with the InShadow rules of
// Light.h:62-69 / 87-94, CAreaLight::luminosity (Light.h:126-162) and luminosity_BSPTree
// (Scene.h:436-455).  Every shadow query is a wave-packet ANY traversal.
template <bool STATS>
__device__ __forceinline__ double light_luminosity(const DevScene &sc, const DLight &L, int li, V3 pt, bool active,
                                                   const RngKey &key, int *stack, Stats &st)
{
  const V3 lp = mk(L.p[0], L.p[1], L.p[2]);
  const int type = uni(L.type);
  if (type == 100 || type == 101) {            // point / parallel
    V3 dir = (type == 100) ? (lp - pt) : neg(lp);
    RayState r; ray_setup(r, pt, unitize(dir), sc.coord_scale);
    if (type == 100) r.best_t = dot(lp - pt, r.d) + 1e-6 * (1.0 + sc.coord_scale);
    if (STATS && active) { st.rays++; st.shadow_rays++; }
    bool occ = traverse<true, STATS>(sc, r, active, type == 100, lp, stack, st);
    return occ ? 0.0 : L.lum;
  }
  // area light
  double lum = 0;
  const int sn = uni(L.sample_num);
  const double rr = (L.len / double(sn)) / 2;
  const V3 O = mk(lp.x - sn * rr, lp.y - sn * rr, lp.z);
  for (int i = 0; i < sn; i++) {
    for (int j = 0; j < sn; j++) {
      double x0 = O.x + i * rr * 2, y0 = O.y + j * rr * 2, z0 = O.z;
      uint32_t ls = (uint32_t)(i * sn + j);
      double rx = ref_random(rr, 1000, key.seed, key.pixel, key.sample, key.path, (uint32_t)li, ls, 0);
      double ry = ref_random(rr, 1000, key.seed, key.pixel, key.sample, key.path, (uint32_t)li, ls, 1);
      V3 sp = mk(x0 + rx, y0 + ry, z0);
      RayState r; ray_setup(r, pt, unitize(sp - pt), sc.coord_scale);
      // hits beyond the projection of the light centre onto the ray can never satisfy the predicate
      r.best_t = fabs(dot(lp - pt, r.d)) + 1e-6 * (1.0 + sc.coord_scale);
      if (STATS && active) { st.rays++; st.shadow_rays++; }
      bool occ = traverse<true, STATS>(sc, r, active, true, lp, stack, st);
      if (!occ) lum += L.lum_per_sample;
    }
  }
  return lum;
}

// CScene::LocalIllumination, Scene.h:93-122 (untextured: objColour = m_colour)
template <bool STATS>
__device__ __forceinline__ Col local_illumination(const DevScene &sc, V3 o, V3 d, int id, V3 pt, bool active,
                                                  const RngKey &key, int *stack, Stats &st, V3 &N_out)
{
  Col colour = mkc(0, 0, 0);
  const int sid = active ? id : 0;
  V3 N = object_norm(sc, sid, pt);
  N_out = N;
  V3 norm = N;
  if (dot(d, norm) > kEps) norm = neg(norm);
  const DMaterial M = sc.materials[sc.obj_mat[sid]];
  const Col oc = mkc(sc.obj_colour[3 * sid], sc.obj_colour[3 * sid + 1], sc.obj_colour[3 * sid + 2]);
  const V3 refl = reflect_dir(N, d);
  for (int li = 0; li < sc.n_lights; li++) {
    const DLight &L = sc.lights[li];
    double lum = light_luminosity<STATS>(sc, L, li, pt, active, key, stack, st);
    V3 ldir = (L.type == 101) ? mk(L.p[0], L.p[1], L.p[2]) : unitize(pt - mk(L.p[0], L.p[1], L.p[2]));
    double diffuse = M.diff * dmax(0.0, -dot(norm, ldir) * lum);
    double specular = M.spec * lum * pow(dmax(0.0, -dot(ldir, refl)), M.shiness);
    colour = cadd(colour, cmul(oc, M.emissive + sc.ambient * M.ambient + diffuse + specular));
  }
  (void)o;
  return colour;
}

}  // namespace rt
