// rt_device.hpp -- device-side scene layout, wave-packet BSP traversal and shading for gfx950.
//
// Execution model (wave64): one ray per lane, 64 rays per wave walk the tree TOGETHER.  The node
// stack is wave-uniform (8 bytes per entry, in the lanes of two VGPRs).  An inner step fetches the boxes of BOTH
// children with one scalar load (a 64-byte DPairF of FP32 boxes, wave-uniform address) and each lane slab-tests
// its own ray (in FP32, conservatively: slab_axis_f); a child is pushed only if some lane wants it (__ballot), far child first according
// to the majority ray direction, so the walk is front-to-back for most lanes and every lane
// prunes with its own best t.  A leaf is a contiguous stream (header + 128-byte records) that the
// wave pulls into LDS with direct-to-LDS loads (16 B per lane, all loads of a chunk in flight
// together) and then reads back as broadcasts.
//
// Leaf triangles are tested in two steps:
//   cull    -- a cheap plane + barycentric test on precomputed data, with margins that are wide
//              compared with every rounding error involved (see LeafRec / leaf_record);
//   confirm -- the reference's own arithmetic (rt::tri_exact_t, Object.h:416-446) on the raw
//              vertices, only for lanes that survive the cull.
// Accepted hits therefore carry exactly the t the reference formula produces.
#pragma once
#include "rt_math.hpp"
#include "rt_param.hpp"

#ifndef RT_OBJ_INLINE
#define RT_OBJ_INLINE __forceinline__   /* noinline was tried: the call ABI costs more registers than it saves */
#endif

namespace rt {

// ---- device scene ------------------------------------------------------------------------------
// Child descriptor of the flattened tree: i0 >= 0 -> inner node, i0 = index of its DPair;
// i0 < 0 -> leaf with (-1 - i0) records, i1 = offset of its stream in 64-byte units.
struct ChildRef { int32_t i0, i1; };

// The two children of one inner node as the builder keeps them (FP64): both boxes are fetched and
// tested in ONE step of the walk.  Boxes are tight (cell /\ bounds of the content) and inflated by kBoxPad.
struct alignas(128) DPair {
  double lmin[3], lmax[3];
  double rmin[3], rmax[3];
  ChildRef l, r;
  int32_t axis, pad[3];
};
// The same pair as the device walks it: FP32 boxes rounded outward, 64 B = one s_load_dwordx16.  The split axis of a pair travels in the
// i1 of the ChildRef that points at it.
struct alignas(64) DPairF { float lmin[3], lmax[3], rmin[3], rmax[3]; ChildRef l, r; };

// Leaf stream: a 64-byte header followed by `count` 128-byte records, contiguous, so that a wave can
// pull a leaf into LDS with 16-byte-per-lane direct-to-LDS loads (1 KiB per instruction).
struct alignas(64) LeafHeader { float bmin[3], bmax[3]; int32_t count, pad[9]; };   // box in FP32, rounded outward

// One object reference of a leaf, 128 B (a triangle referenced by k leaves is stored k times).
//   n (= m_norm), dpl = dot(v0,n)  -> t' = (dpl - o.n)/(d.n)
//   B,b0 / G,g0 : beta' = dot(P',B)+b0, gamma' = dot(P',G)+g0 (dual basis of v1-v0, v2-v0)
//   bsum = |B|_1 + |G|_1 + 1/sin(angle at v0): scales the per-ray barycentric margin
//   kind : 0 triangle, 1 triangle whose cull data are unusable (degenerate / sliver): always confirmed,
//          2 any other finite object type (sphere ...): exact test only
//   tri_idx : index into tri_exact[] (raw vertices for the confirm step)
struct alignas(64) LeafRec {
  double n[3], dpl;
  double B[3], b0;
  double G[3], g0;
  double bsum;
  int32_t prim_id, kind;
  int32_t tri_idx, pad0;
  double pad1;
};
static_assert(sizeof(DPair) == 128 && sizeof(DPairF) == 64 && sizeof(LeafHeader) == 64 && sizeof(LeafRec) == 128, "layout");

#define RT_CHUNK_RECS 31                 // 64 + 31*128 = 4032 bytes, staged as 4 x 1 KiB
#define RT_CHUNK_BYTES 4096

struct TriExact { double v0[3], v1[3], v2[3], nrm[3]; };               // 96 B, by triangle index (brute force / shading)
struct TriShade { double vn[3][3]; int32_t has_vn; int32_t pad; };      // vertex normals, by triangle index
struct DPlane { double p0[3], n[3]; };
struct DSphere { double c[3], r; };
struct DCircle { double p0[3], n[3], r; };
struct DMaterial { double reflW, refrW, RI, diff, spec, shiness, emissive, ambient; };
struct DLight { int32_t type, sample_num; double p[3], lum, len, lum_per_sample; };
struct DTexture { const uint8_t *bgr; int32_t rows, cols; };           // cv::Mat CV_8UC3: bgr[(row * cols + col) * 3 + {B,G,R}]

#define RT_COVER_BOXES 16
struct DevScene {
  const DPairF *pairs; const unsigned char *leaf_stream;
  ChildRef root; int32_t has_tree;              // has_tree == 0: CScene::FirstInterPt over all objects
  const int32_t *root_objs; int32_t n_root_objs;// bspTree.root->objects (infinite objects)
  const TriExact *tri_exact; const TriShade *tri_shade;   // tri_shade == NULL: no triangle has vertex normals
  const DPlane *planes; const DSphere *spheres; const DCircle *circles;
  const DBezier *beziers; const DRotational *rotationals;
  int32_t has_param;                            // the scene holds Bezier / rotational objects: kernels built with PARAM
  // the Newton-solved objects numbered 0 .. n_rot + n_bez - 1 (pidx: a surface of revolution k is k, a patch k is n_rot + k), and the
  // global (ray, object) queue of the deferred solves (rt_defer.inc; set per launch, NULL otherwise)
  int32_t n_rot, n_bez; const int32_t *param_obj;   // param_obj[pidx] = object id
  struct PairRec *gq_pairs; unsigned int *gq_count; unsigned int gq_region;   // object pidx owns slots [pidx * gq_region, (pidx + 1) * gq_region), gq_count[pidx] of them used
  const uint8_t *obj_type; const uint32_t *obj_index; const uint32_t *obj_mat; const double *obj_colour;
  int32_t n_objects;
  // textures (Object.h:69-74); obj_tex == NULL: no object of the scene has one
  const DTexture *textures; const int32_t *obj_tex; const int32_t *obj_bump;   // per object: texture index or -1
  const double *obj_tex_scale;                                                 // per object: textureScale, bumpTextureScale (CInftyPlane)
  const double *tri_uv;                                                        // per triangle: textureCoord[3] (u, v), or NULL
  const DMaterial *materials;
  const DLight *lights; int32_t n_lights;
  double ambient;
  double coord_scale;                           // max |coordinate| of finite geometry
  double rbmin[3], rbmax[3];                    // root box (inflated); only used to order work, never for results
  float rbf_min[3], rbf_max[3];                 // the same in FP32, rounded outward: shadow rays that miss it skip the walk (trace_occluded)
  int32_t lit_skip;                             // area_light_all_clear() in front of an area light's sample loop (RT_HIP_LIT_SKIP=0 at scene creation: off)
  // a cover of all finite geometry by the boxes of a cut through the top of the hierarchy (the root box alone when the root is a leaf): FP32, rounded outward
  int32_t n_cover; float cover_min[RT_COVER_BOXES][3], cover_max[RT_COVER_BOXES][3];
};

// One deferred solve: `ray` names the ray (index of its node / photon), tag = pidx | sample << 8 | light << 20 (shadow rays), next = the
// ray's previous pair (closest-hit queries fold their pairs in rt_defer.inc: k_closest_p3), n_seeds = filled by the solver's preparation pass.
struct PairRec { uint32_t ray, tag, next, n_seeds; };

struct Stats { unsigned long long rays, shadow_rays, node_visits, prim_tests, exact_tests, wave_nodes, wave_prims, pad; };

constexpr double kBoxPad = 1e-7;   // spatial inflation of node boxes (covers slab-test rounding)
constexpr double kGMin = 1e-5;     // |d.n| below this: skip the cull, go to the confirm step

#define RT_STACK_DEPTH 64
#ifndef RT_BLOCK
#define RT_BLOCK 64
#endif
#define RT_WAVES_DEV (RT_BLOCK / 64)

// per-wave LDS: the node stack and the staged leaf chunk
struct RayLds { double o[3], d[3], bound, pad; };    // 64 B: what a lane needs to cull a record for ANOTHER lane's ray
struct WaveLds {
  alignas(16) unsigned char leaf[RT_CHUNK_BYTES];
  alignas(16) RayLds rays[64];
  unsigned int slot[64];                             // rank among the interested rays of a leaf -> lane
  unsigned long long t_start;                        // k_frame_direct: the wave's start time (its cost orders the next frame's waves)
  double cam_d[64][3];                               // k_frame_direct: the camera rays' directions across the walks (two unitize = 300 instructions to make again)
};

// uniform (scalar) copy of a value that is known to be the same in all lanes
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- per-object exact test (CBaseObject::FirstInterPt / FirstInterPt_Unrecorded dispatch, Object.h:37-38) ----
// PARAM = false: kernels for scenes without Bezier / rotational objects (their Newton code is not compiled in).
// u, v: surface parameters of the hit (cur_u / cur_v, Object.h:805-808, 1059-1062), 1e8 for the other types.
template <bool PARAM>
__device__ RT_OBJ_INLINE double object_hit(const DevScene &sc, int id, V3 o, V3 d, bool record, double &u, double &v)
{
  const int type = sc.obj_type[id];
  const uint32_t k = sc.obj_index[id];
  u = kInfty; v = kInfty;
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
  if constexpr (PARAM) {
    if (type == 6) {      // RT_HIP_BEZIER
      ParamHit h = bezier_solve(sc.beziers[k], sc.beziers[k].seed, o, d, false);
      u = h.u; v = h.v; return h.t;
    } else if (type == 5) { // RT_HIP_ROTATIONAL
      ParamHit h = rotational_hit(sc.rotationals[k], o, d, record);
      u = h.u; v = h.v; return h.t;
    }
  }
  return kMiss;
}
__device__ RT_OBJ_INLINE double object_t(const DevScene &sc, int id, V3 o, V3 d)
{ double u, v; return object_hit<false>(sc, id, o, d, true, u, v); }

// The occlusion predicate of Scene.h:449 / Light.h:66,147: the hit X = o + d t lies between the
// shaded point o and the reference point.  Without a reference point (parallel light,
// Light.h:91) any hit occludes.
__device__ __forceinline__ bool occludes(V3 o, V3 d, double t, bool has_ref, V3 ref)
{
  if (!has_ref) return true;
  V3 X = o + d * t;
  return dot(o - X, ref - X) < -kEps;
}

// Diagnostic build only (-DRT_TIMERS): shader-clock cycles per section of the walk, summed over waves.
#ifdef RT_TIMERS
__device__ unsigned long long g_timers[32];     // [closest, occluded][0-4 cycles per section, 5-15 trip counts]
#define RT_TM_DECL unsigned long long tm_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tm_t = __builtin_readcyclecounter();
#define RT_TM(k) { unsigned long long tm_n = __builtin_readcyclecounter(); tm_[k] += tm_n - tm_t; tm_t = tm_n; }
#define RT_TM_CNT(k) { tm_[k]++; }
#define RT_TM_FLUSH { if ((threadIdx.x & 63) == 0) for (int k_ = 0; k_ < 16; k_++) atomicAdd(&g_timers[k_ + (ANY ? 16 : 0)], tm_[k_]); }
#else
#define RT_TM_DECL
#define RT_TM(k)
#define RT_TM_CNT(k)
#define RT_TM_FLUSH
#endif

struct RayState {
  V3 o, d;
  V3 inv;             // 1/d per axis, +-1e30 for a (near) zero component so that the slab products stay finite
  V3 oi;              // o * inv per axis: the slab test is one FMA per plane, (b * inv - oi)
  double bound;       // prune bound: nothing beyond this t can change the result
  double best_t;      // closest CONFIRMED t so far (kMiss = none)
  int best_id;
  double tmargin;     // per-ray t margin of the cull
  double mscale;      // per-ray scale of the barycentric margin
  double gmin;        // per-ray |d.n| threshold below which the cull is not trusted
  double tlow;        // scale of the margin of the t > eps test: tlow / |d.n|
  unsigned dsign;     // bit a set: d[a] >= 0 (near-child order of the packet, never results)
  // the FP32 form of the slab test (slab_f): inv per axis, and o * inv moved by the axis' error bound towards "earlier" for the plane the
  // ray enters through (fa* pairs with bmin when inv >= 0) and "later" for the one it leaves through
  float fix, fiy, fiz, fax, fay, faz, fbx, fby, fbz;
  double best_u, best_v;            // surface parameters of the confirmed hit (PARAM kernels only)
  int mb_id; double mb_t, mb_u, mb_v; // the last non-triangle leaf object tested and its result: the reference re-solves an
                                      // object in every leaf that lists it (Scene.h:352), always with the same result
  unsigned long long pq_mask;         // deferred solves (walk<.., GQ>): the Newton-solved objects (bit pidx) listed by the leaves this ray wanted
};

// 1/x for the box tests only (they decide which nodes are visited, never a result): v_rcp_f64 + one Newton step, ~2^-50 relative -- three
// instructions instead of the ~30 of an IEEE division, far inside the margins of slab_axis_f (2^-21) and the 1e-7 by which slab()'s boxes are inflated
__device__ __forceinline__ double safe_inv(double x)
{
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return (fabs(x) < 1e-30) ? ((x < 0) ? -1e30 : 1e30) : r;
}

// Margins of the cull (DESIGN.md "cull margins"): with s = 1 + max|scene coordinate| + max|origin coordinate|,
// the cull's t' differs from the exact t by at most ~7e-10 s when |d.n| >= 1e-6 (rounding of dpl - o.n,
// about 3 ulp of s, divided by |d.n|), the point P' by the same amount, beta'/gamma' by that times |B|,|G|.
// tmargin = 1e-6 s and mscale = 1e-8 s are > 10x those bounds; they also cover the reference's own rounding
// noise in beta/gamma (~5e-11 |S| |B| / |d.n|).
// the part of the setup that depends on the origin only (the wave-wide scale of the margins): a light's shadow rays share it
__device__ __forceinline__ double ray_origin_scale(V3 o, double coord_scale)
{
  double os = dmax(dmax(fabs(o.x), fabs(o.y)), fabs(o.z));
  // one scale for the whole wave (the largest): in the redistributed cull a lane tests records for another lane's ray
  for (int off = 32; off > 0; off >>= 1) os = dmax(os, __shfl_xor(os, off, 64));
  { const unsigned long long b = __double_as_longlong(os);
    os = __longlong_as_double(((unsigned long long)(unsigned)uni((int)(b >> 32)) << 32) | (unsigned long long)(unsigned)uni((int)(unsigned)b)); }
  return 1.0 + coord_scale + os;
}
// One axis of the FP32 slab test.  slab() accepts a box on t64 = fma(b, inv, -oi) per plane; slab_f computes tf = fmaf(bf, (float)inv,
// -(float)oi -+ E) with bf = b rounded outward.  With u = 2^-24: |tf - t64| <= (4 |b inv| + 2 |oi|) u (1 + ...) from the three roundings
// to FP32 and the FMA's own, so E = 8 u (S |inv| + |oi|), S >= |b|, keeps the entry plane's tf <= t64 and the exit plane's tf >= t64 with the
// rounding of oi -+ E (<= u |oi|) paid from the same margin: every box slab() accepts, slab_f accepts.  An axis whose FP32 products could
// overflow is taken out of the test (NaN: v_min / v_max return the other operand), which only accepts more.
__device__ __forceinline__ bool slab_axis_f(double inv, double oi, float Sf, float &fi, float &fa, float &fb)
{
  fi = (float)inv; const float fo = (float)oi;
  const float E = 0x1p-21f * fmaf(Sf, fabsf(fi), fabsf(fo));
  const bool ok = (Sf * fabsf(fi) < 1e37f) && (fabsf(fo) < 1e37f);
  const float early = fo + E, late = fo - E;
  fa = ok ? ((fi >= 0.0f) ? early : late) : __builtin_nanf("");
  fb = ok ? ((fi >= 0.0f) ? late : early) : __builtin_nanf("");
  return ok;
}
__device__ __forceinline__ void ray_setup_scaled(RayState &r, V3 o, V3 d, double s)
{
  r.o = o; r.d = d;
  r.inv = mk(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  r.oi = mk(o.x * r.inv.x, o.y * r.inv.y, o.z * r.inv.z);
  r.bound = kMiss; r.best_t = kMiss; r.best_id = -1;
  r.best_u = kInfty; r.best_v = kInfty; r.mb_id = -1; r.mb_t = kMiss; r.mb_u = kInfty; r.mb_v = kInfty; r.pq_mask = 0ull;
  r.tmargin = 1e-6 * s;
  r.mscale = 1e-8 * s;
  r.gmin = 1e-6;
  // near t = 0 (rays that start on a surface) the error of t' is the rounding of dpl - o.n (a few ulp of s)
  // over |d.n|: 16 ulp(s) / |d.n| bounds it, and keeps a ray's own triangle (t' ~ 1e-15) out of the survivors
  r.tlow = 16.0 * 2.3e-16 * s;
  r.dsign = (d.x >= 0.0 ? 1u : 0u) | (d.y >= 0.0 ? 2u : 0u) | (d.z >= 0.0 ? 4u : 0u);
  // s >= 1 + max |box coordinate| (coord_scale covers the root box, which contains every node and leaf box)
  const float Sf = (float)s * 1.000001f;
  const bool kx = slab_axis_f(r.inv.x, r.oi.x, Sf, r.fix, r.fax, r.fbx);
  const bool ky = slab_axis_f(r.inv.y, r.oi.y, Sf, r.fiy, r.fay, r.fby);
  const bool kz = slab_axis_f(r.inv.z, r.oi.z, Sf, r.fiz, r.faz, r.fbz);
  if (!(kx | ky | kz)) { r.fix = 0.0f; r.fax = __builtin_inff(); r.fbx = -__builtin_inff(); }   // nothing usable in FP32: every box passes
}
__device__ __forceinline__ void ray_setup(RayState &r, V3 o, V3 d, double coord_scale)
{
  ray_setup_scaled(r, o, d, ray_origin_scale(o, coord_scale));
}

// A pair through the scalar data path: the index is wave-uniform, so the 64 bytes arrive as one
// s_load_dwordx16 in SGPRs with ONE wait, and the slab tests take the box as scalar operands.
// (Left to the compiler, the pair came in through five dependent vector-memory round trips.)
typedef const __attribute__((address_space(4))) DPairF *PairScalarPtr;
__device__ __forceinline__ DPairF load_pair(const DPairF *pairs, int i)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return *(PairScalarPtr)(unsigned long long)(pairs + i);
#else
  return pairs[i];   // host pass of the compiler only parses this
#endif
}

// branch-free slab test against an (inflated) box: does the ray overlap it for some t in (0, bound]?
// One FMA per plane: b * inv - (o * inv) differs from (b - o) * inv by at most |o * inv| 2^-52, i.e. it is the
// exact test against a plane moved by |o| 2^-52 -- far inside the 1e-7 by which every box is inflated.
__device__ __forceinline__ bool slab(const RayState &r, const double *bmin, const double *bmax)
{
  double t1 = fma(bmin[0], r.inv.x, -r.oi.x), t2 = fma(bmax[0], r.inv.x, -r.oi.x);
  double tmin = fmin(t1, t2), tmax = fmax(t1, t2);
  t1 = fma(bmin[1], r.inv.y, -r.oi.y); t2 = fma(bmax[1], r.inv.y, -r.oi.y);
  tmin = fmax(tmin, fmin(t1, t2)); tmax = fmin(tmax, fmax(t1, t2));
  t1 = fma(bmin[2], r.inv.z, -r.oi.z); t2 = fma(bmax[2], r.inv.z, -r.oi.z);
  tmin = fmax(tmin, fmin(t1, t2)); tmax = fmin(tmax, fmax(t1, t2));
  return (tmin <= tmax) && (tmax > 0.0) && (tmin <= r.bound);
}

// The walk's form of it in FP32 (half the issue cycles, a third of the registers): accepts every box slab() accepts (slab_axis_f), and a
// little more -- boxes within ~1e-6 of the ray.  Visiting a box the ray misses changes nothing but the time.  bound_f >= the ray's bound.
__device__ __forceinline__ float bound_up_f(double bound) { const float f = (float)bound; return fmaf(fabsf(f), 0x1p-22f, f); }
__device__ __forceinline__ bool slab_f(const RayState &r, float bound_f, const float *bmin, const float *bmax)
{
  float t1 = fmaf(bmin[0], r.fix, -r.fax), t2 = fmaf(bmax[0], r.fix, -r.fbx);
  const float lx = __builtin_fminf(t1, t2), hx = __builtin_fmaxf(t1, t2);
  t1 = fmaf(bmin[1], r.fiy, -r.fay); t2 = fmaf(bmax[1], r.fiy, -r.fby);
  const float ly = __builtin_fminf(t1, t2), hy = __builtin_fmaxf(t1, t2);
  t1 = fmaf(bmin[2], r.fiz, -r.faz); t2 = fmaf(bmax[2], r.fiz, -r.fbz);
  const float lz = __builtin_fminf(t1, t2), hz = __builtin_fmaxf(t1, t2);
  const float tmin = __builtin_fmaxf(__builtin_fmaxf(lx, ly), lz), tmax = __builtin_fminf(__builtin_fminf(hx, hy), hz);
  return (tmin <= tmax) && (tmax >= 0.0f) && (tmin <= bound_f);
}

// The node step's form of slab_f: the answer for all 64 rays at once, as a lane mask.  A vote over a compound condition costs two vector
// instructions beyond its compares (the condition is materialised per lane and compared again); a vote over ONE compare is that compare's own
// result register.  So the three conditions of slab_f are folded into one: with bound_f >= 0 (every bound is a t + margin or kMiss),
// tmin <= tmax && tmax >= 0 && tmin <= bound_f  <=>  max(tmin, 0) <= min(tmax, bound_f).
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool slab_hit_f(const RayState &r, float bound_f, const float *bmin, const float *bmax)
{
  float t1 = fmaf(bmin[0], r.fix, -r.fax), t2 = fmaf(bmax[0], r.fix, -r.fbx);
  const float lx = __builtin_fminf(t1, t2), hx = __builtin_fmaxf(t1, t2);
  t1 = fmaf(bmin[1], r.fiy, -r.fay); t2 = fmaf(bmax[1], r.fiy, -r.fby);
  const float ly = __builtin_fminf(t1, t2), hy = __builtin_fmaxf(t1, t2);
  t1 = fmaf(bmin[2], r.fiz, -r.faz); t2 = fmaf(bmax[2], r.fiz, -r.fbz);
  const float lz = __builtin_fminf(t1, t2), hz = __builtin_fmaxf(t1, t2);
  const float lo = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(lx, ly), lz), 0.0f), hi = __builtin_fminf(__builtin_fminf(__builtin_fminf(hx, hy), hz), bound_f);
  return lo <= hi;
}

// can a Newton-solved object still matter to this ray?  Its accepted hits lie in [pmin, pmax] (rt_param.hpp: param_prune_box), in front of
// the origin, and only those up to t_max can change the result.  Skipping the solve otherwise changes nothing but the time.
__device__ __forceinline__ bool param_may_matter(const RayState &r, const double *pmin, const double *pmax, double t_max)
{
  RayState q = r; q.bound = t_max;
  return slab(q, pmin, pmax);
}

// the same for a patch, piece by piece: its accepted hits lie in one of the 16 sub-patch hull boxes (rt_param.hpp: param_prune_box)
__device__ __forceinline__ bool bez_may_matter(const RayState &r, const DBezier &Z, double t_max)
{
  RayState q = r; q.bound = t_max;
  if (!slab(q, Z.pmin, Z.pmax)) return false;
  bool any = false;
  for (int k = 0; k < 16; k++) any = any || slab(q, Z.sub[k], Z.sub[k] + 3);
  return any;
}

// direct-to-LDS copy of `bytes` (rounded up to 1 KiB pieces, at most RT_CHUNK_BYTES) from a 16-byte
// aligned global address; every lane moves 16 bytes per instruction (wave-uniform LDS base + lane*16).
__device__ __forceinline__ void stage_chunk(const unsigned char *src, unsigned char *lds, int bytes)
{
  const int lane = threadIdx.x & 63;
  typedef __attribute__((address_space(3))) void lds_void;
  typedef const __attribute__((address_space(1))) void glb_void;
  const unsigned char *g = src + lane * 16;
  __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)lds, 16, 0, 0);
  if (bytes > 1024) __builtin_amdgcn_global_load_lds((glb_void *)(g + 1024), (lds_void *)(lds + 1024), 16, 0, 0);
  if (bytes > 2048) __builtin_amdgcn_global_load_lds((glb_void *)(g + 2048), (lds_void *)(lds + 2048), 16, 0, 0);
  if (bytes > 3072) __builtin_amdgcn_global_load_lds((glb_void *)(g + 3072), (lds_void *)(lds + 3072), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// exact test of a finite non-triangle object referenced by a leaf (sphere, Bezier patch, surface of revolution),
// through the per-ray mailbox
template <bool PARAM>
__device__ __forceinline__ double leaf_other_hit(const DevScene &sc, RayState &r, V3 o, V3 d, int id, bool record, double &u, double &v)
{
  if constexpr (!PARAM) return object_hit<false>(sc, id, o, d, record, u, v);   // spheres: cheaper to redo than to remember
  else {
    if (r.mb_id == id) { u = r.mb_u; v = r.mb_v; return r.mb_t; }
    double t = object_hit<true>(sc, id, o, d, record, u, v);
    r.mb_id = id; r.mb_t = t; r.mb_u = u; r.mb_v = v;
    return t;
  }
}

// confirm step by primitive id: the reference's arithmetic on the raw vertices (per-lane gather)
__device__ RT_OBJ_INLINE double confirm_triangle(const DevScene &sc, int prim, V3 o, V3 d)
{
  const TriExact &T = sc.tri_exact[sc.obj_index[prim]];
  return tri_exact_t(mk(T.v0[0], T.v0[1], T.v0[2]), mk(T.v1[0], T.v1[1], T.v1[2]), mk(T.v2[0], T.v2[1], T.v2[2]),
                     mk(T.nrm[0], T.nrm[1], T.nrm[2]), o, d);
}

// ---- (ray, object) pairs for the Newton-solved objects -------------------------------------------------------------
// A surface of revolution costs up to (uSampleNum + 1) x 10 Newton steps per ray that passes its box test, a patch
// 25 x 10, and nothing for a ray that does not meet the object.  Solved in place with all lanes in lock step, a wave
// pays the full price whenever ANY of its rays needs the object.  Instead the rays that need one are queued as
// (owner lane, object) pairs and solved 64 pairs at a time with every lane busy: the solving lane gathers the
// owner's ray from LDS and works on a private copy of its object, the owner folds the results.
// Without a tree the pairs of consecutive objects are folded in object order (the candidate order of
// Scene.h:65-81); in the tree walk they are folded with the walk's own order-independent rule (confirmed()).
struct PairQueue {
  unsigned short owner[128];        // lane that owns the ray
  int obj[128];                     // object id
  double t[64], u[64], v[64];       // results of the batch being folded
};
// one queue per wave, shared by every walk / no-tree loop of a kernel (they never run nested)
__device__ __forceinline__ PairQueue &wave_pair_queue()
{
  __shared__ PairQueue s_pq[RT_WAVES_DEV];
  return s_pq[threadIdx.x >> 6];
}
#ifdef RT_SOLVE_NOINLINE
#define RT_SOLVE_INLINE __noinline__
#else
#define RT_SOLVE_INLINE __forceinline__
#endif
// solve the first min(n, 64) pairs; returns how many were solved.  All lanes call this together.
// first_only: the caller only asks WHETHER the object is hit (occlusion query without a reference point, where any hit occludes)
__device__ RT_SOLVE_INLINE int pairs_solve(const DevScene &sc, WaveLds *lds, PairQueue &Q, int n, bool record, bool first_only = false)
{
  const int lane = threadIdx.x & 63;
  const int nb = n < 64 ? n : 64;
#ifdef RT_SOLVE_STATS
  int my_steps = 0, my_kind = -1;
#endif
  if (lane < nb) {
    const int src = Q.owner[lane], id = Q.obj[lane];
    const RayLds &gr = lds->rays[src];
    const V3 o = mk(gr.o[0], gr.o[1], gr.o[2]), d = mk(gr.d[0], gr.d[1], gr.d[2]);
    ParamHit h;
    if (sc.obj_type[id] == 6) {
      const DBezier &G = sc.beziers[sc.obj_index[id]];
      DBezierCtrl B;                                                  // this lane's control net: a private copy in registers
      for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int a = 0; a < 3; a++) B.c[i][j][a] = G.c[i][j][a];
      for (int a = 0; a < 3; a++) B.center[a] = G.center[a];
      h = bezier_solve(B, G.seed, o, d, first_only);
    } else {
      const DRotational R = sc.rotationals[sc.obj_index[id]];
      h = rotational_solve(R, o, d, record);                         // the box test was the condition for queueing
    }
    Q.t[lane] = h.t; Q.u[lane] = h.u; Q.v[lane] = h.v;
#ifdef RT_SOLVE_STATS
    my_steps = h.steps; my_kind = sc.obj_type[id] == 6 ? 0 : 1;
#endif
  }
#ifdef RT_SOLVE_STATS
  for (int kind = 0; kind < 2; kind++) {
    int mine = my_kind == kind ? my_steps : 0, mx = mine, sum = mine;
    for (int off = 32; off > 0; off >>= 1) { mx = max(mx, __shfl_xor(mx, off, 64)); sum += __shfl_xor(sum, off, 64); }
    if (lane == 0 && mx) { atomicAdd(&g_solve_steps[4 + 2 * kind], 64ull * mx); atomicAdd(&g_solve_steps[5 + 2 * kind], (unsigned long long)sum); }
  }
#endif
  __builtin_amdgcn_wave_barrier();
  return nb;
}
__device__ __forceinline__ void pairs_shift(PairQueue &Q, int n, int done)
{
  // entries [done, n) move to the front (n - done < 64: one entry per lane)
  const int lane = threadIdx.x & 63;
  unsigned short o = 0; int b = 0;
  if (done + lane < n) { o = Q.owner[done + lane]; b = Q.obj[done + lane]; }
  __builtin_amdgcn_wave_barrier();
  if (done + lane < n) { Q.owner[lane] = o; Q.obj[lane] = b; }
  __builtin_amdgcn_wave_barrier();
}
// append this object's interested lanes; returns the new count (< 128)
__device__ __forceinline__ int pairs_push(PairQueue &Q, int n, bool want, int id)
{
  const unsigned long long m = __ballot(want);
  const unsigned int lane = threadIdx.x & 63;
  const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
  if (want) { Q.owner[n + rank] = (unsigned short)lane; Q.obj[n + rank] = id; }
  __builtin_amdgcn_wave_barrier();
  return n + __popcll(m);
}

// How n interested rays x the records of a chunk are dealt over the 64 lanes (redistributed cull, see walk): lane L culls record L / n of the step
// for the ray of rank L % n.  Per n: the 16-bit reciprocal that divides a lane number by n exactly (floor(65536 / n) + 2; checked for
// every L < 64, n <= 64 in the tests), S = 64 / n records per step, the lanes in use (S * n of them), and the lanes that hold the records of the
// ray of rank 0 (bits 0, n, 2n, ...).  One 32-byte scalar load per chunk instead of a reciprocal in the vector unit and two scalar loops.
struct alignas(32) DealRow { unsigned int magic, S; unsigned long long stride_mask, used_mask, pad; };
struct DealTable { DealRow row[65]; };
constexpr DealTable make_deal_table()
{
  DealTable t{};
  for (unsigned int n = 1; n <= 64; n++) {
    t.row[n].magic = 65536u / n + 2u;
    const unsigned int S = 64u / n;
    t.row[n].S = S;
    unsigned long long m = 0;
    for (unsigned int k = 0; k < S; k++) m |= 1ull << (k * n);
    t.row[n].stride_mask = m;
    t.row[n].used_mask = (S * n < 64u) ? ((1ull << (S * n)) - 1ull) : ~0ull;
  }
  t.row[0] = t.row[1]; t.row[0].used_mask = 0;
  return t;
}
__constant__ const DealTable kDeal = make_deal_table();

// Deferred-confirmation state of one ray.  During the fast walk only the cull runs; the two nearest
// cull survivors (by t') are remembered and confirmed once, after the walk, by all lanes together.
// A lane whose survivors are not conclusive (see finish_closest / finish_any) is re-walked with the
// confirm step inline (walk<.., false>), so the result never depends on the cull's approximations.
struct Deferred {
  double c1_t, c2_t, c3_t;   // t' of the nearest, second nearest and nearest discarded survivor
  int c1_id, c2_id;
  bool redo;                 // met something the fast walk cannot judge (|d.n| tiny, degenerate triangle)
};

// One tree walk.  DEFER = true: fast walk (cull only, survivors into D).  DEFER = false: every cull survivor
// is confirmed inline with the reference's arithmetic.
// ANY = false: closest hit; confirmed hits go to r.best_t / r.best_id, r.bound prunes.
// ANY = true : occlusion; `found` becomes true for a lane as soon as a confirmed hit satisfies occludes()
//              (DEFER: as soon as a cull survivor MAY satisfy it; D.c1_id remembers which).
// GQ = true (with PARAM = false): deferred solves -- a Newton-solved object listed by a leaf is only noted in r.pq_mask; the caller queues the
// (ray, object) pairs after the walk (gq_push_closest / gq_push_occluded) and a kernel of its own solves them (rt_defer.inc).
template <bool ANY, bool DEFER, bool STATS, bool PARAM, bool GQ = false>
__device__ __forceinline__ void walk(const DevScene &sc, RayState &r, bool active, bool has_ref, V3 ref, double fmargin,
                                     WaveLds *lds, Stats &st, Deferred &D, bool &found, bool record)
{
  // confirmed candidate (Scene.h:74-79): first candidate wins ties within eps
  auto confirmed = [&](double t, int id, bool lane_on, double hu = kInfty, double hv = kInfty) {
    if (!lane_on || !(t < kMiss)) return;
    if (ANY) {
      if (occludes(r.o, r.d, t, has_ref, ref)) found = true;
    } else {
      // within a leaf the reference keeps the first-listed (= lower id) of two hits closer than eps (Scene.h:74-79);
      // the walk order here differs from the reference's, so the lower id is made to win explicitly
      if (r.best_id < 0 || t < r.best_t - kEps || (t <= r.best_t + kEps && id < r.best_id)) {
        r.best_t = t; r.best_id = id; r.bound = fmin(r.bound, t + r.tmargin);
        if (PARAM) { r.best_u = hu; r.best_v = hv; }
      }
    }
  };
  int sp = 0;
  ChildRef cur = sc.root;
  // Newton-solved objects met in leaves (fast walk only): queued, solved 64 pairs at a time, see PairQueue
  PairQueue *pq_ptr = nullptr;
  if constexpr (PARAM && DEFER) pq_ptr = &wave_pair_queue();
  PairQueue &PQ = *pq_ptr;
  int nq = 0;
  auto solve_pairs = [&](int keep_below) {
    if constexpr (PARAM && DEFER) {
      const int lane = threadIdx.x & 63;
      while (nq >= keep_below && nq > 0) {
        const int nb = pairs_solve(sc, lds, PQ, nq, record, ANY && __ballot(has_ref) == 0ull);
        const double bt = r.best_t;
        for (int q = 0; q < nb; q++)
          if ((int)PQ.owner[q] == lane) confirmed(PQ.t[q], PQ.obj[q], !(ANY && found), PQ.u[q], PQ.v[q]);
        if (!ANY && r.best_t != bt) lds->rays[lane].bound = r.bound;
        __builtin_amdgcn_wave_barrier();
        pairs_shift(PQ, nq, nb); nq -= nb;
      }
    }
  };
  if constexpr (DEFER) {                           // this wave's rays, for the redistributed cull (LDS is in order per wave)
    RayLds &me = lds->rays[threadIdx.x & 63];
    me.o[0] = r.o.x; me.o[1] = r.o.y; me.o[2] = r.o.z; me.d[0] = r.d.x; me.d[1] = r.d.y; me.d[2] = r.d.z; me.bound = r.bound; me.pad = 0.0;
  }
  // the node stack: entry k lives in lane k of these two registers (the entries are wave-uniform, the depth is at most RT_STACK_DEPTH = 64):
  // v_writelane / v_readlane instead of an LDS round trip per push and pop (measured +1.7 % on C2)
  int stk0 = 0, stk1 = 0;
  RT_TM_DECL
  RT_TM_CNT(8)
  // Wave-level state of the node steps (lane masks in SGPRs, so that a step's votes are scalar algebra on its compares):
  // the lanes that still walk, the lanes whose rays look along +x / +y / +z; and every lane's bound in FP32 (it only changes in leaves).
  unsigned long long live_mask = ballot64(active && !(ANY && found));
  const unsigned long long sgn_x = ballot64(r.d.x >= 0.0), sgn_y = ballot64(r.d.y >= 0.0), sgn_z = ballot64(r.d.z >= 0.0);
  float bound_f = bound_up_f(r.bound);
  if (live_mask != 0ull)                           // (0: every ray of the wave is finished)
  for (;;) {
    // ---- descent through inner nodes: fetch both child boxes at once, go on with the near child some lane wants, the far one waits on the
    //      stack.  Nothing a leaf changes (bounds, survivors) is touched in here. ----
    bool at_leaf = true;
    while (cur.i0 >= 0) {
      RT_TM_CNT(9)
      RT_TM(0)
      const DPairF P = load_pair(sc.pairs, cur.i0);
      const unsigned long long ml = ballot64(slab_hit_f(r, bound_f, P.lmin, P.lmax)) & live_mask;
      const unsigned long long mr = ballot64(slab_hit_f(r, bound_f, P.rmin, P.rmax)) & live_mask;
      if (STATS) { if ((live_mask >> (threadIdx.x & 63)) & 1ull) st.node_visits += 2; if ((threadIdx.x & 63) == 0) st.wave_nodes += 2; }
      RT_TM_CNT(5)
#ifdef RT_ASM_STEP
      // (experiment, measured -0.8 % on C2: the same decisions written out for the scalar unit -- 24 scalar instructions and 2 branches per step
      // instead of the compiler's 46 and 6, but a block the scheduler cannot move the next step's work across)
      // Which child next, which one waits: wave-uniform decisions on the two votes, written out for the scalar unit (left to the compiler,
      // uniform predicates that are and-ed / or-ed live as 64-bit lane masks and every use costs a mask-to-condition round trip: 46 scalar
      // instructions and 6 branches per step instead of 24 and 2).
      //   n_both = rays that want a child (0: dead end); near = left iff (mr == 0) or (ml != 0 and the majority of the interested rays looks
      //   along the split axis from the left side); the far child is written to stack slot sp in any case, sp moves only if both are wanted.
      const unsigned long long sg = (cur.i1 == 0) ? sgn_x : ((cur.i1 == 1) ? sgn_y : sgn_z);     // i1: the split axis of this pair
      int n_both, near0, near1, far0, far1, tl_, push_, m0_keep;
      unsigned long long t_;
      asm volatile("s_or_b64 %[t], %[ml], %[mr]\n\t"
                   "s_bcnt1_i32_b64 %[nb], %[t]\n\t"
                   "s_and_b64 %[t], %[t], %[sg]\n\t"
                   "s_bcnt1_i32_b64 %[tl], %[t]\n\t"
                   "s_lshl_b32 %[tl], %[tl], 1\n\t"
                   "s_cmp_ge_u32 %[tl], %[nb]\n\t"
                   "s_cselect_b32 %[tl], 1, 0\n\t"
                   "s_cmp_lg_u64 %[ml], 0\n\t"
                   "s_cselect_b32 %[tl], %[tl], 0\n\t"
                   "s_cselect_b32 %[pu], 1, 0\n\t"
                   "s_cmp_lg_u64 %[mr], 0\n\t"
                   "s_cselect_b32 %[tl], %[tl], 1\n\t"
                   "s_cselect_b32 %[pu], %[pu], 0\n\t"
                   "s_cmp_lg_u32 %[tl], 0\n\t"
                   "s_cselect_b32 %[n0], %[l0], %[r0]\n\t"
                   "s_cselect_b32 %[n1], %[l1], %[r1]\n\t"
                   "s_cselect_b32 %[f0], %[r0], %[l0]\n\t"
                   "s_cselect_b32 %[f1], %[r1], %[l1]\n\t"
                   "s_mov_b32 %[mk], m0\n\t"         // (M0 belongs to the compiler: put back)
                   "s_mov_b32 m0, %[sp]\n\t"
                   "s_nop 0\n\t"
                   "v_writelane_b32 %[k0], %[f0], m0\n\t"
                   "v_writelane_b32 %[k1], %[f1], m0\n\t"
                   "s_mov_b32 m0, %[mk]"
                   : [t] "=&s"(t_), [nb] "=&s"(n_both), [tl] "=&s"(tl_), [pu] "=&s"(push_), [n0] "=&s"(near0), [n1] "=&s"(near1), [f0] "=&s"(far0), [f1] "=&s"(far1),
                     [mk] "=&s"(m0_keep), [k0] "+v"(stk0), [k1] "+v"(stk1)
                   : [ml] "s"(ml), [mr] "s"(mr), [sg] "s"(sg), [l0] "s"(P.l.i0), [l1] "s"(P.l.i1), [r0] "s"(P.r.i0), [r1] "s"(P.r.i1), [sp] "s"(sp)
                   : "scc");
      // (what an asm statement returns counts as divergent: uni() says otherwise, and costs nothing on a value that is in an SGPR already)
      if (uni(n_both) == 0) { at_leaf = false; RT_TM(1) break; }    // no ray wants either child
      sp += uni(push_);
#ifdef RT_TIMERS
      tm_[14] += push_;
#endif
      cur.i0 = uni(near0); cur.i1 = uni(near1);
      RT_TM(1)
#else
      const unsigned long long both = ml | mr;
      if (both == 0ull) { at_leaf = false; RT_TM(1) break; }        // no ray wants either child
      const unsigned long long sg = (cur.i1 == 0) ? sgn_x : ((cur.i1 == 1) ? sgn_y : sgn_z);     // i1: the split axis of this pair
      const bool left_first = (2 * __popcll(both & sg) >= __popcll(both));                      // the majority's near side
      const bool take_left = (mr == 0ull) | ((ml != 0ull) & left_first);
      if ((ml != 0ull) & (mr != 0ull)) {
        const ChildRef far = take_left ? P.r : P.l;
        int m0_keep;                          // (v_writelane takes one SGPR operand besides the lane select in M0; M0 belongs to the compiler, so it is put back)
        asm volatile("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %4\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %5, m0\n\ts_mov_b32 m0, %2" : "+v"(stk0), "+v"(stk1), "=&s"(m0_keep) : "s"(far.i0), "s"(sp), "s"(far.i1)); sp++;
        RT_TM_CNT(14)
      }
      cur = take_left ? P.l : P.r;
      RT_TM(1)
#endif
    }
    if (at_leaf) {
      RT_TM_CNT(9)
      RT_TM(0)
      const int i0 = cur.i0, i1 = cur.i1;
      // ---- leaf: stage header + records in LDS chunk by chunk ----
      const int cnt = -1 - i0;
      const unsigned char *base = sc.leaf_stream + (size_t)(unsigned)i1 * 64;
      const bool live = active && !(ANY && found);
      bool want = false;
      unsigned long long Wm = 0ull;                 // the rays that want this leaf, as a mask
      for (int c0 = 0; c0 < cnt; c0 += RT_CHUNK_RECS) {
        const int nrec = (cnt - c0 < RT_CHUNK_RECS) ? (cnt - c0) : RT_CHUNK_RECS;
        stage_chunk(base + (size_t)c0 * sizeof(LeafRec), lds->leaf, 64 + nrec * (int)sizeof(LeafRec));
        RT_TM(2) RT_TM_CNT(6)
        if (c0 == 0) {
          const LeafHeader *H = (const LeafHeader *)lds->leaf;
          const bool in_box = slab_hit_f(r, bound_f, H->bmin, H->bmax);
          want = live && in_box; Wm = ballot64(in_box) & live_mask;
          if (STATS) { if (live) st.node_visits++; if ((threadIdx.x & 63) == 0) st.wave_nodes++; }
          if (Wm == 0ull) break;
        }
        RT_TM(3)
        const LeafRec *recs = (const LeafRec *)(lds->leaf + 64);
        if constexpr (DEFER) {
          // ---- redistributed cull.  Typically ~1/6 of the wave's rays want a given leaf, so instead of all 64
          // lanes stepping through every record with most of them idle, the n interested rays x nrec records are
          // dealt out over the lanes: lane L culls record (base + L / n) for the ray of rank (L % n), whose
          // origin/direction/bound it gathers from LDS.  Survivors (rare) are handed back to the owning lane,
          // in record order per ray, which then does exactly what the broadcast loop did. ----
          const unsigned int lane = threadIdx.x & 63;
          const int n = __popcll(Wm);
          RT_TM_CNT(10)
#ifdef RT_TIMERS
          tm_[13] += n;
#endif
          const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(Wm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Wm, 0u));
          if (want) lds->slot[rank] = lane;
          // the deal of this chunk (kDeal): q = L / n, j = L % n, S = 64 / n records in flight per step
#if defined(__HIP_DEVICE_COMPILE__)
          const DealRow deal = *(const __attribute__((address_space(4))) DealRow *)(unsigned long long)(&kDeal.row[n]);   // (one s_load_dwordx8, like load_pair)
#else
          const DealRow deal = kDeal.row[n];
#endif
          const unsigned long long used_mask = deal.used_mask;                           // lanes with q < S
          const unsigned int magic = deal.magic;
          const unsigned int q = (lane * magic) >> 16, j = lane - q * (unsigned)n;
          const int S = (int)deal.S;
          // deal.stride_mask: the lanes that hold the records of the ray of rank 0; shifted by k for the ray of rank k
          const unsigned long long my_events = want ? (deal.stride_mask << rank) : 0ull;  // where this lane's ray has its records culled
          const unsigned int src = lds->slot[j];
          const RayLds &gr = lds->rays[src];
          const double gox = gr.o[0], goy = gr.o[1], goz = gr.o[2], gdx = gr.d[0], gdy = gr.d[1], gdz = gr.d[2], ghi = gr.bound + r.tmargin;
          if (STATS && (threadIdx.x & 63) == 0) st.wave_prims += nrec;
          // records of kind 2 (spheres ...): exact test after the loop.  Found once per chunk (lane i looks at record i), not once per step.
          unsigned other = (unsigned)(ballot64(recs[lane < 31u ? lane : 30u].kind == 2) & ((1ull << nrec) - 1ull));
          // the interested rays that are still open (occlusion: not yet occluded), so that the step loop can end without a vote
          unsigned long long open_mask = Wm;
          if (ANY) open_mask &= ~ballot64(found);
          const unsigned int end_off = (unsigned)nrec * (unsigned)sizeof(LeafRec);
          unsigned int off = q * (unsigned)sizeof(LeafRec);   // this lane's record of the step, as a byte offset
          for (int b0i = 0; b0i < nrec; b0i += S, off += (unsigned)S * (unsigned)sizeof(LeafRec)) {
            if (ANY && open_mask == 0ull) break;                      // every interested ray of the wave is occluded
            RT_TM_CNT(11)
            const unsigned long long valid_mask = used_mask & ballot64(off < end_off);
            const unsigned int coff = off < end_off ? off : end_off - (unsigned)sizeof(LeafRec);       // (a lane without a record culls the chunk's last one; its vote is masked)
            const LeafRec &c = *(const LeafRec *)((const unsigned char *)recs + coff);
            if (STATS && ((valid_mask >> lane) & 1ull)) st.prim_tests++;
            const double n0 = c.n[0], n1 = c.n[1], n2 = c.n[2], dpl = c.dpl;
            const double B0 = c.B[0], B1 = c.B[1], B2 = c.B[2], b0 = c.b0;
            const double G0 = c.G[0], G1 = c.G[1], G2 = c.G[2], g0 = c.g0;
            const double bsum = c.bsum;
            const int kind = c.kind, prim = c.prim_id;
            // ---- cull (explicit FMAs: this part only has to be conservative) ----
            const double g = fma(gdz, n2, fma(gdy, n1, gdx * n0));
            const double on = fma(goz, n2, fma(goy, n1, gox * n0));
            double rg = __builtin_amdgcn_rcp(g);
            rg = fma(fma(-g, rg, 1.0), rg, rg);       // one Newton step: ~2^-50 relative
            const double tc = (dpl - on) * rg;
            const double px = fma(gdx, tc, gox), py = fma(gdy, tc, goy), pz = fma(gdz, tc, goz);
            const double be = fma(pz, B2, fma(py, B1, fma(px, B0, b0)));
            const double ga = fma(pz, G2, fma(py, G1, fma(px, G0, g0)));
            const double m = fma(r.mscale, bsum, 1e-6);
            // a record whose cull data are unusable (kind 1) carries a zero normal: g = 0, never trusted.  Also catches NaN.
            const bool untrusted = !(fabs(g) >= r.gmin);
            // every vote below is over ONE compare (its result register is the vote); the conditions meet as scalar algebra
            const unsigned long long inside_mask = ballot64(tc > kEps - r.tlow * fabs(rg)) & ballot64(tc < ghi) & ballot64(be > -m) & ballot64(ga > -m) &
                                                   ballot64(be + ga < 1.0 + m);
            // ---- events back to the owners, ascending lane = ascending record per ray ----
            unsigned long long ev = valid_mask & ballot64(kind != 2) & (ballot64(untrusted) | inside_mask);
            asm("" : "+s"(ev));        // (ev is finished in SGPRs before the per-lane AND below: left alone, the compiler ANDs every factor into the lane pattern separately)
            // Every owner takes ITS events together with the other owners: the events of the ray of rank k sit in lanes k, k + n, k + 2n, ...
            // (ascending lane = ascending record), so an owner masks ev with that stride pattern and pulls the (t', prim, untrusted) of its
            // lowest event from the lane that holds it with ds_bpermute; one round per event of the busiest ray of the step (mostly one),
            // where the former loop took one round per event of the step (~5).
            unsigned long long mine = ev & my_events;
            const unsigned long long tbits = __double_as_longlong(tc);
            const int pack = prim | (untrusted ? (int)0x80000000 : 0);
#ifdef RT_TIMERS
            tm_[15] += __popcll(ev);
#endif
            while (ballot64(mine != 0ull) != 0ull) {
              RT_TM_CNT(12)
              const bool has = mine != 0ull;
              const int L = has ? (int)__builtin_ctzll(mine) : 0;
              mine &= mine - 1ull;
              const int epack = __builtin_amdgcn_ds_bpermute(L << 2, pack);
              const unsigned int ehi = (unsigned int)__builtin_amdgcn_ds_bpermute(L << 2, (int)(tbits >> 32));
              const unsigned int elo = (unsigned int)__builtin_amdgcn_ds_bpermute(L << 2, (int)(unsigned)tbits);
              const double etc = __longlong_as_double(((unsigned long long)ehi << 32) | (unsigned long long)elo);
              const int eprim = epack & 0x7FFFFFFF;
              const bool eun = epack < 0;
              const bool open = has && !(ANY && found);
              if (open && eun) D.redo = true;
              const bool cand = open && !eun && etc < r.bound + r.tmargin;      // the owner's current bound (the gathered one may be older)
              if (ANY) {
                // may this survivor occlude?  f(t) = Dot(o - X, ref - X), X = o + d t (Scene.h:449)
                bool maybe = cand;
                if (has_ref) {
                  V3 X = mk(fma(r.d.x, etc, r.o.x), fma(r.d.y, etc, r.o.y), fma(r.d.z, etc, r.o.z));
                  maybe = cand && dot(r.o - X, ref - X) < -kEps + fmargin;
                }
                if (maybe) { found = true; D.c1_id = eprim; D.c1_t = etc; }
                open_mask &= ~ballot64(maybe);
              } else if (cand && eprim != D.c1_id && eprim != D.c2_id) {
                // the two nearest survivors stay sorted, the nearest discarded one is remembered (selects, no nested branches)
                const bool lt1 = etc < D.c1_t, lt2 = etc < D.c2_t;
                D.c3_t = fmin(D.c3_t, lt2 ? D.c2_t : etc);
                D.c2_t = lt1 ? D.c1_t : (lt2 ? etc : D.c2_t); D.c2_id = lt1 ? D.c1_id : (lt2 ? eprim : D.c2_id);
                D.c1_t = lt1 ? etc : D.c1_t; D.c1_id = lt1 ? eprim : D.c1_id;
                r.bound = fmin(r.bound, D.c1_t + 2.0 * r.tmargin);
                lds->rays[lane].bound = r.bound;
              }
            }
          }
          while (other) {                             // sphere etc.: exact test only (wave-uniform, rare)
            const int i = __builtin_ctz(other); other &= other - 1u;
            const int prim = uni(recs[i].prim_id);
            const bool lane_on = want && !(ANY && found);
            if (STATS && lane_on) st.exact_tests++;
            if constexpr (GQ) {
              const int type = uni((int)sc.obj_type[prim]);
              if (type == 5 || type == 6) {
                const int pidx = uni((int)sc.obj_index[prim]) + (type == 6 ? sc.n_rot : 0);
                if (lane_on) r.pq_mask |= 1ull << pidx;
                continue;
              }
            }
            if constexpr (PARAM) {
              const int type = uni((int)sc.obj_type[prim]);
              if (type == 5 || type == 6) {
                // the mailbox remembers the object this ray queued last: the reference re-solves an object in every
                // leaf that lists it (Scene.h:352), always with the same result
                bool push = lane_on && r.mb_id != prim;
                if (push) r.mb_id = prim;
                if (type == 5) { const DRotational &R = sc.rotationals[sc.obj_index[prim]]; push = push && !ref_box_miss(R.bmin, R.bmax, r.o, r.d) && param_may_matter(r, R.pmin, R.pmax, r.bound) && rot_shells_may_matter(R, r.o, r.d, r.bound); }
                else { const DBezier &Z = sc.beziers[sc.obj_index[prim]]; push = push && bez_may_matter(r, Z, r.bound); }
                nq = pairs_push(PQ, nq, push, prim);
                solve_pairs(64);
                continue;
              }
            }
            const double bt = r.best_t;
            double hu = kInfty, hv = kInfty;
            const double ht = lane_on ? leaf_other_hit<PARAM>(sc, r, r.o, r.d, prim, record, hu, hv) : kMiss;
            confirmed(ht, prim, lane_on, hu, hv);
            if (!ANY && r.best_t != bt) lds->rays[lane].bound = r.bound;
          }
        } else {
        unsigned other = 0u;                       // records of kind 2 (spheres ...): exact test after the loop
#ifdef RT_UNROLL
#pragma unroll RT_UNROLL
#endif
        for (int i = 0; i < nrec; i++) {
          const LeafRec &c = recs[i];
          const bool lane_on = want && !(ANY && found);
          if (ANY && __ballot(lane_on) == 0ull) break;        // every interested ray of the wave is occluded
          if (STATS) { if (lane_on) st.prim_tests++; if ((threadIdx.x & 63) == 0) st.wave_prims++; }
          // all cull fields in one batch of LDS reads (no short-circuit: each would cost another round trip)
          const double n0 = c.n[0], n1 = c.n[1], n2 = c.n[2], dpl = c.dpl;
          const double B0 = c.B[0], B1 = c.B[1], B2 = c.B[2], b0 = c.b0;
          const double G0 = c.G[0], G1 = c.G[1], G2 = c.G[2], g0 = c.g0;
          const double bsum = c.bsum;
          const int kind = uni(c.kind), prim = uni(c.prim_id);     // same LDS batch as the cull fields
          other |= (kind == 2) ? (1u << i) : 0u;
          // ---- cull (explicit FMAs: this part only has to be conservative) ----
          const double g = fma(r.d.z, n2, fma(r.d.y, n1, r.d.x * n0));
          const double on = fma(r.o.z, n2, fma(r.o.y, n1, r.o.x * n0));
          double rg = __builtin_amdgcn_rcp(g);
          rg = fma(fma(-g, rg, 1.0), rg, rg);       // one Newton step: ~2^-50 relative
          const double tc = (dpl - on) * rg;
          const double px = fma(r.d.x, tc, r.o.x), py = fma(r.d.y, tc, r.o.y), pz = fma(r.d.z, tc, r.o.z);
          const double be = fma(pz, B2, fma(py, B1, fma(px, B0, b0)));
          const double ga = fma(pz, G2, fma(py, G1, fma(px, G0, g0)));
          const double m = fma(r.mscale, bsum, 1e-6);
          const bool tri = lane_on & (kind != 2);
          const bool untrusted = !(fabs(g) >= r.gmin) | (kind == 1);      // also catches NaN
          const bool inside = (tc > kEps - r.tlow * fabs(rg)) & (tc < r.bound + r.tmargin) & (be > -m) & (ga > -m) & (be + ga < 1.0 + m);
          if (DEFER) {
            if (tri & untrusted) D.redo = true;
            if (tri & inside & !untrusted) {
              if (ANY) {
                // may this survivor occlude?  f(t) = Dot(o - X, ref - X), X = o + d t (Scene.h:449)
                bool maybe = true;
                if (has_ref) { V3 X = mk(px, py, pz); maybe = dot(r.o - X, ref - X) < -kEps + fmargin; }
                if (maybe) { found = true; D.c1_id = prim; D.c1_t = tc; }
              } else if (prim != D.c1_id && prim != D.c2_id) {
                if (tc < D.c1_t) { D.c3_t = fmin(D.c3_t, D.c2_t); D.c2_t = D.c1_t; D.c2_id = D.c1_id; D.c1_t = tc; D.c1_id = prim; }
                else if (tc < D.c2_t) { D.c3_t = fmin(D.c3_t, D.c2_t); D.c2_t = tc; D.c2_id = prim; }
                else D.c3_t = fmin(D.c3_t, tc);
                r.bound = fmin(r.bound, D.c1_t + 2.0 * r.tmargin);
              }
            }
          } else {
            const bool pass = tri & (untrusted | inside);
            if (pass) {
              if (STATS) st.exact_tests++;
              const TriExact &T = sc.tri_exact[uni(c.tri_idx)];
              double t = tri_exact_t(mk(T.v0[0], T.v0[1], T.v0[2]), mk(T.v1[0], T.v1[1], T.v1[2]), mk(T.v2[0], T.v2[1], T.v2[2]),
                                     mk(T.nrm[0], T.nrm[1], T.nrm[2]), r.o, r.d);
              confirmed(t, prim, true);
            }
          }
        }
        while (other) {                             // sphere etc.: exact test only (wave-uniform, rare)
          const int i = __builtin_ctz(other); other &= other - 1u;
          const int prim = uni(recs[i].prim_id);
          const bool lane_on = want && !(ANY && found);
          if (STATS && lane_on) st.exact_tests++;
          if constexpr (GQ) {
            const int type = uni((int)sc.obj_type[prim]);
            if (type == 5 || type == 6) {
              const int pidx = uni((int)sc.obj_index[prim]) + (type == 6 ? sc.n_rot : 0);
              if (lane_on) r.pq_mask |= 1ull << pidx;
              continue;
            }
          }
          double hu = kInfty, hv = kInfty;
          const double ht = lane_on ? leaf_other_hit<PARAM>(sc, r, r.o, r.d, prim, record, hu, hv) : kMiss;
          confirmed(ht, prim, lane_on, hu, hv);
        }
        }
        // the next chunk overwrites the LDS image: all reads above are complete (values consumed)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RT_TM(4)
#ifdef RT_TIMERS
        tm_[7] += nrec;
#endif
      }
      // what a leaf can change for the node steps: the bounds, and (occlusion) who still walks
      bound_f = bound_up_f(r.bound);
      if (ANY) live_mask = ballot64(active && !found);
    }
    if (sp == 0 || (ANY && live_mask == 0ull)) break;
    --sp;
    cur.i0 = __builtin_amdgcn_readlane(stk0, sp); cur.i1 = __builtin_amdgcn_readlane(stk1, sp);
  }
  solve_pairs(1);
  RT_TM_FLUSH
}

// Closest hit of one ray per lane (CScene::FirstInterPt_BSPTree, Scene.h:85-91; FirstInterPt, Scene.h:62-83,
// when the scene has no tree).  Result in r.best_t / r.best_id.  All lanes of the wave call this together.
// GQ (with PARAM = false): the Newton-solved objects are not solved here; r.pq_mask names the ones whose (ray, object) pair the caller
// queues with gq_push_closest, and r.best_* is the closest hit among all other objects (with a tree; without one the caller folds in
// object order, k_closest_p3).
template <bool STATS, bool PARAM, bool GQ = false>
__device__ __forceinline__ void trace_closest(const DevScene &sc, RayState &r, bool active, WaveLds *lds, Stats &st, bool record = true)
{
  auto take = [&](double t, int id, double hu = kInfty, double hv = kInfty) {   // the update rule of Scene.h:74-79 (objects come in id order)
    if (t < kMiss && (r.best_id < 0 || t < r.best_t - kEps)) { r.best_t = t; r.best_id = id; if (PARAM) { r.best_u = hu; r.best_v = hv; } }
  };
  if (!sc.has_tree) {
    if constexpr (PARAM) {
      PairQueue &Q = wave_pair_queue();
      const int lane = threadIdx.x & 63;
      { RayLds &me = lds->rays[lane]; me.o[0] = r.o.x; me.o[1] = r.o.y; me.o[2] = r.o.z; me.d[0] = r.d.x; me.d[1] = r.d.y; me.d[2] = r.d.z; }
      __builtin_amdgcn_wave_barrier();
      int nq = 0;
      // the owners take the results of a solved batch, in queue (= object) order
      auto fold = [&](int nb) {
        for (int q = 0; q < nb; q++)
          if ((int)Q.owner[q] == lane) take(Q.t[q], Q.obj[q], Q.u[q], Q.v[q]);
        __builtin_amdgcn_wave_barrier();
      };
      auto drain = [&](int keep_below) {              // solve until fewer than keep_below pairs wait
        while (nq >= keep_below && nq > 0) { const int nb = pairs_solve(sc, lds, Q, nq, record); fold(nb); pairs_shift(Q, nq, nb); nq -= nb; }
      };
      for (int id = 0; id < sc.n_objects; id++) {
        const int type = uni((int)sc.obj_type[id]);
        if (type == 5) {                              // surface of revolution: box test now (Object.h:836-838), Newton later
          const DRotational &R = sc.rotationals[sc.obj_index[id]];
          // (a candidate replaces the best so far only if it is nearer by eps, Scene.h:74: hits beyond best_t cannot matter)
          const bool want = active && !ref_box_miss(R.bmin, R.bmax, r.o, r.d) && param_may_matter(r, R.pmin, R.pmax, r.best_id >= 0 ? r.best_t : kMiss) && rot_shells_may_matter(R, r.o, r.d, r.best_id >= 0 ? r.best_t : kMiss);
          if (STATS && active) st.exact_tests++;
          nq = pairs_push(Q, nq, want, id);
          drain(64);
        } else {
          drain(1);                                   // the candidate order is the object order: settle the queue first
          double hu, hv;
          double t = active ? object_hit<PARAM>(sc, id, r.o, r.d, record, hu, hv) : kMiss;
          if (STATS && active) st.exact_tests++;
          if (active) take(t, id, hu, hv);
        }
      }
      drain(1);
      return;
    } else {
    for (int id = 0; id < sc.n_objects; id++) {
      if constexpr (GQ) {
        // a candidate replaces the best so far only if it is nearer by eps (Scene.h:74): solves whose hits lie beyond the best of the objects
        // BEFORE this one cannot matter.  (The best so far lacks the Newton-solved objects before this one: fewer solves are skipped than
        // in the in-place form above, the result is the same.)
        const int type = uni((int)sc.obj_type[id]);
        if (type == 5 || type == 6) {
          const int k = uni((int)sc.obj_index[id]);
          const double t_max = r.best_id >= 0 ? r.best_t : kMiss;
          bool want;
          if (type == 5) { const DRotational &R = sc.rotationals[k]; want = active && !ref_box_miss(R.bmin, R.bmax, r.o, r.d) && param_may_matter(r, R.pmin, R.pmax, t_max) && rot_shells_may_matter(R, r.o, r.d, t_max); }
          else { const DBezier &Z = sc.beziers[k]; want = active && bez_may_matter(r, Z, t_max); }
          if (STATS && active) st.exact_tests++;
          if (want) r.pq_mask |= 1ull << (k + (type == 6 ? sc.n_rot : 0));
          continue;
        }
      }
      double hu, hv;
      double t = active ? object_hit<PARAM>(sc, id, r.o, r.d, record, hu, hv) : kMiss;
      if (STATS && active) st.exact_tests++;
      if (active) take(t, id, hu, hv);
    }
    return;
    }
  }
  // root's infinite objects (Scene.h:329-346, 369-386) first: their hit bounds the walk.  They are applied
  // AFTER the finite objects below, as in the reference's leaf loop, so that finite objects win ties.
  double inf_t = kMiss; int inf_id = -1;
  for (int i = 0; i < sc.n_root_objs; i++) {
    const int id = uni(sc.root_objs[i]);
    double t = active ? object_t(sc, id, r.o, r.d) : kMiss;
    if (STATS && active) st.exact_tests++;
    if (t < kMiss && (inf_id < 0 || t < inf_t - kEps)) { inf_t = t; inf_id = id; }
  }
  if (inf_id >= 0) r.bound = inf_t + r.tmargin;
  Deferred D; D.c1_t = D.c2_t = D.c3_t = kMiss; D.c1_id = D.c2_id = -1; D.redo = false;
  bool dummy = false;
  walk<false, true, STATS, PARAM, GQ>(sc, r, active, false, mk(0, 0, 0), 0.0, lds, st, D, dummy, record);
  // ---- confirm the survivors, all lanes together ----
  const bool h1 = active && D.c1_id >= 0, h2 = active && D.c2_id >= 0;
  double t1 = kMiss, t2 = kMiss;
#pragma unroll 1
  for (int k = 0; k < 2; k++) {
    const bool hk = k ? h2 : h1;
    const int idk = k ? D.c2_id : D.c1_id;
    double tk = hk ? confirm_triangle(sc, idk, r.o, r.d) : kMiss;
    if (k) t2 = tk; else t1 = tk;
  }
  if (STATS) st.exact_tests += (h1 ? 1 : 0) + (h2 ? 1 : 0);
  // not conclusive: the nearest survivor was not a hit after all, or a third one is as near as the first
  bool redo = active && (D.redo || (h1 && !(t1 < kMiss)) || (D.c3_t < kMiss && D.c3_t <= D.c1_t + 4.0 * r.tmargin));
#ifndef RT_NO_REDO
  if (__ballot(redo) != 0ull) {
    if (STATS) { if (redo) st.pad++; if ((threadIdx.x & 63) == 0) st.pad += (1ull << 32); }
    RayState q = r;
    q.best_t = kMiss; q.best_id = -1; q.bound = (inf_id >= 0) ? inf_t + r.tmargin : kMiss;
    q.mb_id = -1;                            // the fast walk used the mailbox as a 'queued' mark only
    walk<false, false, STATS, PARAM, GQ>(sc, q, redo, false, mk(0, 0, 0), 0.0, lds, st, D, dummy, record);
    if (redo) { r.best_t = q.best_t; r.best_id = q.best_id; if (PARAM) { r.best_u = q.best_u; r.best_v = q.best_v; } if (GQ) r.pq_mask |= q.pq_mask; }
  }
#endif
  if (active && !redo) {
    // spheres etc. confirmed during the walk sit in r.best_*; merge the triangle survivors in object order
    double e_t = r.best_t; int e_id = r.best_id; const double e_u = r.best_u, e_v = r.best_v;
    r.best_t = kMiss; r.best_id = -1;
    const bool first1 = !(h2 && t2 < kMiss) || D.c1_id < D.c2_id;
    if (first1) { take(t1, D.c1_id); take(t2, D.c2_id); } else { take(t2, D.c2_id); take(t1, D.c1_id); }
    take(e_t, e_id, e_u, e_v);
  }
  if (active) take(inf_t, inf_id);
}

// Occlusion query of one ray per lane: is there a hit X with occludes() true (Scene.h:449, Light.h:66,91,147)?
// t_limit: no hit beyond it can satisfy the predicate (kMiss if unbounded).  Shadow tests go through
// FirstInterPt_Unrecorded (Light.h:65, 90, 146; Scene.h:448 record = false).
// GQ (with PARAM = false): the answer covers every object but the Newton-solved ones; r.pq_mask names those whose pairs the caller queues
// with gq_push_occluded for the rays that come back unoccluded.
template <bool STATS, bool PARAM, bool ROOT_TEST = true, bool GQ = false>
__device__ __forceinline__ bool trace_occluded(const DevScene &sc, RayState &r, bool active, bool has_ref, V3 ref, double t_limit,
                                               WaveLds *lds, Stats &st)
{
  bool found = false;
  if (!sc.has_tree) {
    if constexpr (PARAM) {
      PairQueue &Q = wave_pair_queue();
      const int lane = threadIdx.x & 63;
      { RayLds &me = lds->rays[lane]; me.o[0] = r.o.x; me.o[1] = r.o.y; me.o[2] = r.o.z; me.d[0] = r.d.x; me.d[1] = r.d.y; me.d[2] = r.d.z; }
      __builtin_amdgcn_wave_barrier();
      int nq = 0;
      auto fold = [&](int nb) {
        for (int q = 0; q < nb; q++)
          if ((int)Q.owner[q] == lane && Q.t[q] < kMiss && occludes(r.o, r.d, Q.t[q], has_ref, ref)) found = true;
        __builtin_amdgcn_wave_barrier();
      };
      auto drain = [&](int keep_below) {
        while (nq >= keep_below && nq > 0) { const int nb = pairs_solve(sc, lds, Q, nq, false); fold(nb); pairs_shift(Q, nq, nb); nq -= nb; }
      };
      for (int id = 0; id < sc.n_objects; id++) {
        const bool lane_on = active && !found;
        if (__ballot(lane_on) == 0ull) { nq = 0; break; }     // every ray is settled: what still waits in the queue is moot
        const int type = uni((int)sc.obj_type[id]);
        if (type == 5) {
          const DRotational &R = sc.rotationals[sc.obj_index[id]];
          const bool want = lane_on && !ref_box_miss(R.bmin, R.bmax, r.o, r.d) && param_may_matter(r, R.pmin, R.pmax, t_limit) && rot_shells_may_matter(R, r.o, r.d, t_limit);
          if (STATS && lane_on) st.exact_tests++;
          nq = pairs_push(Q, nq, want, id);
          drain(64);
        } else {
          double hu, hv;
          double t = lane_on ? object_hit<PARAM>(sc, id, r.o, r.d, false, hu, hv) : kMiss;
          if (STATS && lane_on) st.exact_tests++;
          if (lane_on && t < kMiss && occludes(r.o, r.d, t, has_ref, ref)) found = true;
        }
      }
      drain(1);
      return found;
    } else {
    for (int id = 0; id < sc.n_objects; id++) {
      const bool lane_on = active && !found;
      if (__ballot(lane_on) == 0ull) break;
      if constexpr (GQ) { const int type = uni((int)sc.obj_type[id]); if (type == 5 || type == 6) continue; }   // (any object may occlude: the order does not matter)
      double hu, hv;
      double t = lane_on ? object_hit<PARAM>(sc, id, r.o, r.d, false, hu, hv) : kMiss;
      if (STATS && lane_on) st.exact_tests++;
      if (lane_on && t < kMiss && occludes(r.o, r.d, t, has_ref, ref)) found = true;
    }
    if constexpr (GQ) { const int np = sc.n_rot + sc.n_bez; if (active && !found) r.pq_mask = np >= 64 ? ~0ull : ((1ull << np) - 1ull); }
    return found;
    }
  }
  for (int i = 0; i < sc.n_root_objs; i++) {
    const int id = uni(sc.root_objs[i]);
    const bool lane_on = active && !found;
    double t = lane_on ? object_t(sc, id, r.o, r.d) : kMiss;
    if (STATS && lane_on) st.exact_tests++;
    if (lane_on && t < kMiss && occludes(r.o, r.d, t, has_ref, ref)) found = true;
  }
  r.bound = t_limit;
  const bool pre = found;
  // every finite object lies in the root box: a ray that misses it (same conservative FP32 test as the node steps) has nothing to
  // meet in the tree, and a wave of such rays -- most shadow rays of an open scene -- skips the walk and its set-up altogether
  // (C3 +4 %, C4 +2 %; not in the one-light frame kernel, whose shadow rays start on the geometry inside the box: C2 -0.5 % with it)
  const bool walk_on = active && !pre && (!ROOT_TEST || slab_f(r, bound_up_f(t_limit), sc.rbf_min, sc.rbf_max));
  if (ROOT_TEST && __ballot(walk_on) == 0ull) return found;
  const double fmargin = 3.0 * (len(ref - r.o) + 1.0) * r.tmargin;
  Deferred D; D.c1_t = D.c2_t = D.c3_t = kMiss; D.c1_id = D.c2_id = -1; D.redo = false;
  walk<true, true, STATS, PARAM, GQ>(sc, r, walk_on, has_ref, ref, fmargin, lds, st, D, found, false);
  // ---- confirm the survivor that stopped the walk ----
  const bool h1 = active && !pre && found && D.c1_id >= 0;
  bool ok = false;
  if (h1) { double t = confirm_triangle(sc, D.c1_id, r.o, r.d); ok = (t < kMiss) && occludes(r.o, r.d, t, has_ref, ref); }
  if (STATS && h1) st.exact_tests++;
  // (found && !h1) = a sphere confirmed inline; (!found) = nothing may occlude; both are final unless flagged
  bool redo = active && !pre && ((h1 && !ok) || (D.redo && !(h1 && ok)));
#ifndef RT_NO_REDO
  if (__ballot(redo) != 0ull) {
    if (STATS) { if (redo) st.pad++; if ((threadIdx.x & 63) == 0) st.pad += (1ull << 32); }
    bool f2 = false;
    RayState q = r; q.bound = t_limit; q.mb_id = -1;
    walk<true, false, STATS, PARAM, GQ>(sc, q, redo, has_ref, ref, fmargin, lds, st, D, f2, false);
    if (redo) { found = f2; if (GQ) r.pq_mask |= q.pq_mask; }
  }
#endif
  return found;
}

// ---- deferred solves: the global (ray, object) queue (rt_defer.inc) -----------------------------------------------------------------
// Every object has its own stretch of the queue, so that a solver launch works on ONE object (its data are wave-uniform: scalar loads, no
// per-lane tables).  Returns the pair's slot (the new head of the ray's list) or `head`.
__device__ __forceinline__ uint32_t gq_push(const DevScene &sc, bool push, int pidx, uint32_t ray, uint32_t tag, uint32_t head)
{
  const unsigned long long m = ballot64(push);
  if (m == 0ull) return head;
  const unsigned int lane = threadIdx.x & 63;
  unsigned int base = 0;
  if (lane == (unsigned)__builtin_ctzll(m)) base = atomicAdd(&sc.gq_count[pidx], (unsigned int)__popcll(m));
  base = (unsigned int)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const unsigned int idx = base + (unsigned int)__popcll(m & below);
  if (!push || idx >= sc.gq_region) return head;                 // (a launch is sized so that its rays cannot fill a stretch: never taken)
  const unsigned int slot = (unsigned int)pidx * sc.gq_region + idx;
  PairRec p; p.ray = ray; p.tag = tag; p.next = head; p.n_seeds = 0;
  sc.gq_pairs[slot] = p;
  return slot;
}
// the pairs of a closest-hit query (after trace_closest<.., GQ>).  With a tree the solves that cannot matter are skipped here, against the
// final closest hit of the other objects: a Newton hit beyond it (by more than the tie window) changes nothing.
__device__ __forceinline__ uint32_t gq_push_closest(const DevScene &sc, const RayState &r, bool active, uint32_t ray)
{
  uint32_t head = 0xFFFFFFFFu;
  const int np = sc.n_rot + sc.n_bez;
  const double t_max = r.best_id >= 0 ? r.best_t + r.tmargin : kMiss;
  for (int pidx = 0; pidx < np; pidx++) {
    bool want = active && ((r.pq_mask >> pidx) & 1ull);
    if (ballot64(want) == 0ull) continue;
    const bool bez = pidx >= sc.n_rot;
    if (sc.has_tree) {
      if (!bez) { const DRotational &R = sc.rotationals[pidx]; want = want && !ref_box_miss(R.bmin, R.bmax, r.o, r.d) && param_may_matter(r, R.pmin, R.pmax, t_max) && rot_shells_may_matter(R, r.o, r.d, t_max); }
      else { const DBezier &Z = sc.beziers[pidx - sc.n_rot]; want = want && bez_may_matter(r, Z, t_max); }
    }
    head = gq_push(sc, want, pidx, ray, (uint32_t)pidx, head);
  }
  return head;
}
// the pairs of an occlusion query that came back unoccluded (after trace_occluded<.., GQ>); tag_hi = sample << 8 | light << 20
__device__ __forceinline__ void gq_push_occluded(const DevScene &sc, const RayState &r, bool open, double t_limit, uint32_t ray, uint32_t tag_hi)
{
  const int np = sc.n_rot + sc.n_bez;
  for (int pidx = 0; pidx < np; pidx++) {
    bool want = open && ((r.pq_mask >> pidx) & 1ull);
    if (ballot64(want) == 0ull) continue;
    const bool bez = pidx >= sc.n_rot;
    if (!bez) { const DRotational &R = sc.rotationals[pidx]; want = want && !ref_box_miss(R.bmin, R.bmax, r.o, r.d) && param_may_matter(r, R.pmin, R.pmax, t_limit) && rot_shells_may_matter(R, r.o, r.d, t_limit); }
    else { const DBezier &Z = sc.beziers[pidx - sc.n_rot]; want = want && bez_may_matter(r, Z, t_limit); }
    (void)gq_push(sc, want, pidx, ray, tag_hi | (uint32_t)pidx, 0xFFFFFFFFu);
  }
}

// ---- shading ------------------------------------------------------------------------------------
// CBaseObject::Norm without bump map (Object.h:79-83) -> InitialNorm of the object type
// (u, v): surface parameters of the hit for Bezier / rotational objects.  The reference reads them from the object's
// hidden cur_u / cur_v state, which a later intersection can overwrite before ReflectRay runs; here
// the hit record carries them, so the normal always belongs to the hit being shaded.
template <bool PARAM = false>
__device__ __forceinline__ V3 object_initial_norm(const DevScene &sc, int id, V3 pt, double hu = kInfty, double hv = kInfty)
{
  const int type = sc.obj_type[id];
  const uint32_t k = sc.obj_index[id];
  if constexpr (PARAM) {
    if (type == 6) return bez_norm(sc.beziers[k], hu, hv);            // Object.h:1049
    if (type == 5) return rot_norm(sc.rotationals[k], hu, hv);        // Object.h:799
  }
  if (type == 1) {                                    // CTriangle::InitialNorm, Object.h:448-457
    const TriExact &T = sc.tri_exact[k];
    if (!sc.tri_shade) return mk(T.nrm[0], T.nrm[1], T.nrm[2]);      // no triangle of the scene has vertex normals
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

// ---- textures (Object.h:79-92, 186-196; per type :280-319, 470-486, 725-771, 810-813, 1051-1054) ----
// texture->at<cv::Vec3b>(row, col) with the indices clamped into the image: the reference reads out of bounds when an
// index reaches rows / cols (e.g. u0 == 1 on a sphere's pole).  The angle of a hit on a surface of revolution is
// whatever real number Newton's iteration ended on (also negative, also beyond 2 PI): its column wraps around.
struct Texel { double b, g, r; };
__device__ __forceinline__ Texel texel_at(const DTexture &T, int row, int col)
{
  row = row < 0 ? 0 : (row >= T.rows ? T.rows - 1 : row);
  col = col < 0 ? 0 : (col >= T.cols ? T.cols - 1 : col);
  const uint8_t *q = T.bgr + ((size_t)row * T.cols + col) * 3;
  Texel t; t.b = q[0]; t.g = q[1]; t.r = q[2];
  return t;
}
__device__ __forceinline__ int to_int(double x)      // int(double) of the reference, defined for every input
{
  if (!(x > -2147483648.0)) return -2147483647 - 1;
  if (!(x < 2147483647.0)) return 2147483647;
  return (int)x;
}
// CInftyPlane's in-plane axis u0 (Object.h:727-734, 746-753, 762-770)
__device__ __forceinline__ V3 plane_u0(V3 n)
{
  const V3 a = mk(1, 0, 0), b = mk(0, 1, 0), c = mk(0, 0, 1);
  if (fabs(dot(a, n)) < 0.5) return unitize(a - n * dot(a, n));
  if (fabs(dot(b, n)) < 0.5) return unitize(b - n * dot(b, n));
  return unitize(c - n * dot(c, n));
}
// CTriangle::UV_colour / UV_norm (Object.h:470-486): the same area-weighted textureCoord
__device__ __forceinline__ void tri_uv_at(const DevScene &sc, uint32_t k, V3 pt, double &u, double &v)
{
  const TriExact &T = sc.tri_exact[k];
  V3 v0 = mk(T.v0[0], T.v0[1], T.v0[2]), v1 = mk(T.v1[0], T.v1[1], T.v1[2]), v2 = mk(T.v2[0], T.v2[1], T.v2[2]);
  double S0 = len(cross(v1 - pt, v2 - pt));
  double S1 = len(cross(v0 - pt, v2 - pt));
  double S2 = len(cross(v0 - pt, v1 - pt));
  double Sa = len(cross(v1 - v0, v2 - v0));
  const double *c = sc.tri_uv + 6 * (size_t)k;
  // CPoint2D * double, + (Geometry.h:149-151), left to right
  u = c[0] * (S0 / Sa) + c[2] * (S1 / Sa) + c[4] * (S2 / Sa);
  v = c[1] * (S0 / Sa) + c[3] * (S1 / Sa) + c[5] * (S2 / Sa);
}
// CSphere::UV_norm (Object.h:289-296); UV_colour scales it by rows / cols (:280-287)
__device__ __forceinline__ void sphere_uv(const DSphere &S, V3 pt, double &u0, double &v0)
{
  if (pt.x == 0 && pt.y == 0) { u0 = 0; v0 = 0; return; }
  const V3 q = pt - mk(S.c[0], S.c[1], S.c[2]);
  v0 = (atan2(q.y, q.x) + kPiRef) / (2 * kPiRef);
  u0 = (asin(q.z / S.r) + kPiRef / 2) / kPiRef;
}

// CBaseObject::colour (Object.h:186-196)
template <bool PARAM = false>
__device__ __forceinline__ Col object_colour(const DevScene &sc, int id, V3 pt, double hu = kInfty, double hv = kInfty)
{
  const int ti = sc.obj_tex ? sc.obj_tex[id] : -1;
  if (ti < 0) return mkc(sc.obj_colour[3 * id], sc.obj_colour[3 * id + 1], sc.obj_colour[3 * id + 2]);
  const DTexture T = sc.textures[ti];
  const int type = sc.obj_type[id];
  const uint32_t k = sc.obj_index[id];
  double u = 0, v = 0;                                  // CBaseObject::UV_colour default: CPoint2D() (Object.h:61)
  if (type == 0) {                                      // Object.h:280-287
    double u0, v0; sphere_uv(sc.spheres[k], pt, u0, v0);
    u = to_int(u0 * T.rows); v = to_int(v0 * T.cols);
    if (pt.x == 0 && pt.y == 0) { u = 0; v = 0; }
  } else if (type == 1) {                               // Object.h:470-477
    if (sc.tri_uv) tri_uv_at(sc, k, pt, u, v);
  } else if (type == 4) {                               // Object.h:725-742
    const DPlane &P = sc.planes[k];
    const V3 n = mk(P.n[0], P.n[1], P.n[2]), u0 = plane_u0(n), v0 = cross(n, u0);
    const V3 rel = pt - mk(P.p0[0], P.p0[1], P.p0[2]);
    const double scale = sc.obj_tex_scale[2 * id];
    int up = to_int(dot(rel, u0) * scale) % T.rows, vp = to_int(dot(rel, v0) * scale) % T.cols;
    up = (up + T.rows) % T.rows; vp = (vp + T.cols) % T.cols;
    u = up; v = vp;
  } else if (PARAM && type == 5) {                      // Object.h:810; theta of the Newton root is any real: column taken modulo cols
    u = to_int(hu * T.rows);
    int vc = to_int(hv * T.cols / (2 * kPiRef)) % T.cols;
    v = (vc + T.cols) % T.cols;
  } else if (PARAM && type == 6) {                      // Object.h:1051
    u = to_int(hu * T.rows); v = to_int(hv * T.cols);
  }
  const Texel c = texel_at(T, to_int(u), to_int(v));    // Object.h:193
  return mkc(c.r, c.g, c.b);                            // Object.h:194
}

// CBaseObject::Norm (Object.h:79-92): InitialNorm, perturbed by the bump texture when the object has one
template <bool PARAM = false>
__device__ __forceinline__ V3 object_norm(const DevScene &sc, int id, V3 pt, double hu = kInfty, double hv = kInfty)
{
  const V3 N = object_initial_norm<PARAM>(sc, id, pt, hu, hv);
  const int bi = sc.obj_bump ? sc.obj_bump[id] : -1;
  if (bi < 0) return N;
  const DTexture T = sc.textures[bi];
  const int type = sc.obj_type[id];
  const uint32_t k = sc.obj_index[id];
  double u = 0, v = 0;                                  // CBaseObject::UV_norm / Tangent / Bitangent defaults (Object.h:62-64)
  V3 tan = mk(0, 0, 0), bit = mk(0, 0, 0);
  if (type == 0) {                                      // Object.h:289-319
    const DSphere &S = sc.spheres[k];
    sphere_uv(S, pt, u, v);
    tan = unitize(mk(-2 * kPiRef * S.r * sin(2 * kPiRef * v - kPiRef) * cos(kPiRef * u - kPiRef / 2),
                     2 * kPiRef * S.r * cos(2 * kPiRef * v - kPiRef) * cos(kPiRef * u - kPiRef / 2), 0));
    bit = unitize(mk(-kPiRef * S.r * cos(2 * kPiRef * v - kPiRef) * sin(kPiRef * u - kPiRef / 2),
                     -kPiRef * S.r * sin(2 * kPiRef * v - kPiRef) * sin(kPiRef * u - kPiRef / 2),
                     kPiRef * S.r * cos(kPiRef * u - kPiRef / 2)));
  } else if (type == 1) {                               // Object.h:479-486; no Tangent / Bitangent override
    if (sc.tri_uv) tri_uv_at(sc, k, pt, u, v);
  } else if (type == 4) {                               // Object.h:744-771, 700
    const DPlane &P = sc.planes[k];
    const V3 n = mk(P.n[0], P.n[1], P.n[2]), u0 = plane_u0(n), v0 = cross(n, u0);
    const V3 rel = pt - mk(P.p0[0], P.p0[1], P.p0[2]);
    const double scale = sc.obj_tex_scale[2 * id + 1];
    const double a = dot(rel, u0) / scale, b = dot(rel, v0) / scale;
    u = a - floor(a); v = b - floor(b);
    tan = u0; bit = cross(n, u0);
  } else if (PARAM && type == 5) {                      // Object.h:811-813
    const DRotational &R = sc.rotationals[k];
    u = hu; v = hv / (2 * kPiRef); v = v - floor(v);    // theta of the Newton root is any real: one turn = one image width
    tan = unitize(rot_ds_dth(R, hu, hv)); bit = unitize(rot_ds_dt(R, hu, hv));
  } else if (PARAM && type == 6) {                      // Object.h:1052-1054
    V3 S, du, dv; bez_eval(sc.beziers[k], hu, hv, S, du, dv);
    u = hu; v = hv;
    tan = unitize(du); bit = unitize(dv);
  }
  const Texel c = texel_at(T, to_int(u * T.rows), to_int(v * T.cols));     // Object.h:87
  return unitize(bit * (c.r / 127.5 - 1) + tan * (c.g / 127.5 - 1) + N * (c.b / 127.5 - 1));   // Object.h:88-90
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
// (Scene.h:436-455).  Every shadow query is a wave-packet occlusion walk; point and parallel lights are
// the one-sample case of the same loop.
// Can NO shadow ray from pt towards an area light's samples be occluded?  Conservative (true = certainly none), from pt and the light's
// rectangle alone -- every sample target lies in R = [lp.x -+ (sn + 1) rr] x [lp.y -+ (sn + 1) rr] x {lp.z} (Light.h:126-140: cell corner + random(rr)):
//  * finite objects: all of them lie in the boxes of a cut through the top of the hierarchy (DevScene::cover_*, at most 16; the root box when
//    the root is a leaf).  A plane through pt that has R on its non-negative side and a whole box below -delta separates every ray
//    pt + s (q - pt), s >= 0, q in R, from that box: tried, box by box, for the four planes through pt and an edge of R, and for the plane z = pt.z.  delta = 1e-6 S^2 (S = 1 + the largest coordinate in play) is ten orders above the rounding of these products and far
//    above the eps = 1e-8 by which the reference's triangle test reaches beyond a triangle;
//  * the infinite planes (the root's own objects), CInftyPlane::FirstInterPt as plane_t restates it: t = a / (n.d), a = -(D + n.pt).  With
//    g = n.(q - pt) of one strict sign over R: a of the other sign (or zero) gives t <= 0 -- no hit; a of the same sign but so small that
//    |a| / min|n.d| < eps / 4 gives t < eps -- no hit (the point lies ON the plane: the floor under a floor point); a ray that rounds to
//    parallel returns 0 or nothing, and t = 0 never satisfies the predicate of Scene.h:449.  Anything else: not clear.
// A wave whose points are all clear adds up the light's samples without making a single ray (C3: the floor and the wall away from the solid).
__device__ __forceinline__ bool area_light_all_clear(const DevScene &sc, V3 lp, int sn, double rr, V3 pt)
{
  const double hs = (sn + 2) * rr;
  const double cx0 = lp.x - hs - pt.x, cx1 = lp.x + hs - pt.x, cy0 = lp.y - hs - pt.y, cy1 = lp.y + hs - pt.y, dz = lp.z - pt.z;
  const double S = 1.0 + sc.coord_scale + dmax(dmax(fabs(pt.x), fabs(pt.y)), fabs(pt.z)) + dmax(dmax(fabs(lp.x), fabs(lp.y)), fabs(lp.z)) + hs;
  const double delta = 1e-6 * S * S;
  const double adz = fabs(dz), sg = (dz >= 0.0) ? 1.0 : -1.0;
  const double kx0 = -sg * cx0, kx1 = sg * cx1, ky0 = -sg * cy0, ky1 = sg * cy1;
  // every box of the cover (sc.n_cover boxes that hold all finite objects between them) must be cut off by one of the five planes
  for (int c = 0; c < sc.n_cover; c++) {
    const double bx0 = (double)sc.cover_min[c][0] - pt.x, bx1 = (double)sc.cover_max[c][0] - pt.x, by0 = (double)sc.cover_min[c][1] - pt.y, by1 = (double)sc.cover_max[c][1] - pt.y;
    const double bz0 = (double)sc.cover_min[c][2] - pt.z, bz1 = (double)sc.cover_max[c][2] - pt.z;
    bool clear = adz * bx1 + dmax(kx0 * bz0, kx0 * bz1) < -delta;                                      // normal  sg (dz, 0, -cx0): R at x >= its low edge
    clear = clear || (-adz * bx0 + dmax(kx1 * bz0, kx1 * bz1) < -delta);                              // normal -sg (dz, 0, -cx1)
    clear = clear || (adz * by1 + dmax(ky0 * bz0, ky0 * bz1) < -delta);
    clear = clear || (-adz * by0 + dmax(ky1 * bz0, ky1 * bz1) < -delta);
    clear = clear || ((dz >= 0.0 ? bz1 : -bz0) < -1e-6 * S);                                          // the box wholly on the other side of z = pt.z
    if (!clear) return false;
  }
  const double lmax = sqrt(dmax(cx0 * cx0, cx1 * cx1) + dmax(cy0 * cy0, cy1 * cy1) + dz * dz);
  for (int i = 0; i < sc.n_root_objs; i++) {
    const int id = uni(sc.root_objs[i]);
    if (uni((int)sc.obj_type[id]) != 4) return false;
    const DPlane &P = sc.planes[sc.obj_index[id]];
    const V3 p0 = mk(P.p0[0], P.p0[1], P.p0[2]), n = mk(P.n[0], P.n[1], P.n[2]);
    const double D = -dot(p0, n), a = -(D + dot(n, pt));                                                // as plane_t has them
    const double gz = n.z * dz;
    const double gmin = gz + dmin(n.x * cx0, n.x * cx1) + dmin(n.y * cy0, n.y * cy1), gmax = gz + dmax(n.x * cx0, n.x * cx1) + dmax(n.y * cy0, n.y * cy1);
    const double gm = 1e-9 * S * (fabs(n.x) + fabs(n.y) + fabs(n.z));
    double sgn, gl;
    if (gmin > gm) { sgn = 1.0; gl = gmin; } else if (gmax < -gm) { sgn = -1.0; gl = -gmax; } else return false;
    if (a * sgn > 0.0 && !(fabs(a) * lmax < 0.25 * kEps * gl)) return false;
  }
  return true;
}

template <bool STATS, bool PARAM, bool ONE = false>
__device__ __forceinline__ double light_luminosity(const DevScene &sc, const DLight &L, int li, V3 pt, bool active,
                                                   const RngKey &key, WaveLds *lds, Stats &st, unsigned long long *proved = nullptr)
{
  const V3 lp = mk(L.p[0], L.p[1], L.p[2]);
  const int type = uni(L.type);
  const bool area = (type == 102);
  const int sn = ONE ? 1 : (area ? uni(L.sample_num) : 1);      // ONE: the caller has checked that this light has one sample
  const double rr = area ? (L.len / double(sn)) / 2 : 0.0;
  const V3 O = mk(lp.x - sn * rr, lp.y - sn * rr, lp.z);
  const double per = area ? L.lum_per_sample : L.lum;
  double lum = 0;
  if constexpr (!ONE && !STATS && !PARAM) {
    if (area && sn > 1 && sc.has_tree && sc.lit_skip && !sc.has_param) {
      const bool clear = !active || area_light_all_clear(sc, lp, sn, rr, pt);
      if (proved && active && clear) *proved += (unsigned long long)(sn * sn);                        // census: counted as shadow rays, not traced
      if (__ballot(!clear) == 0ull) { for (int s = 0; s < sn * sn; s++) lum += per; return lum; }      // (the same sum, sample by sample)
      active = active && !clear;     // in a mixed wave the clear points sit out the walks (every sample counts for them: `occ` stays false)
    }
  }
  const double scale = ray_origin_scale(pt, sc.coord_scale);      // every shadow ray of this point starts at pt
  for (int s = 0; s < sn * sn; s++) {
    const int i = s / sn, j = s % sn;
    V3 dir;
    if (area) {
      double x0 = O.x + i * rr * 2, y0 = O.y + j * rr * 2, z0 = O.z;
      double rx = ref_random(rr, 1000, key.seed, key.pixel, key.sample, key.path, (uint32_t)li, (uint32_t)s, 0);
      double ry = ref_random(rr, 1000, key.seed, key.pixel, key.sample, key.path, (uint32_t)li, (uint32_t)s, 1);
      dir = mk(x0 + rx, y0 + ry, z0) - pt;
    } else dir = (type == 100) ? (lp - pt) : neg(lp);
    RayState r; ray_setup_scaled(r, pt, unitize(dir), scale);
    // hits beyond the projection of the light position onto the ray can never satisfy the predicate
    const double t_limit = (type == 101) ? kMiss : fabs(dot(lp - pt, r.d)) + 1e-6 * (1.0 + sc.coord_scale);
    if (STATS && active) { st.rays++; st.shadow_rays++; }
    bool occ = trace_occluded<STATS, PARAM, !ONE>(sc, r, active, type != 101, lp, t_limit, lds, st);
    if (!occ) lum += per;
  }
  return lum;
}

// CScene::LocalIllumination, Scene.h:93-122, in two pieces so that the shadow pass can sum it light by light:
// shade_setup = what does not depend on the light (normal, flipped normal, reflected view direction, material,
// object colour), light_term = one light's contribution given its luminosity at pt.
template <bool PARAM = false>
__device__ __forceinline__ V3 shade_setup(const DevScene &sc, V3 d, int id, V3 pt, double hu, double hv, V3 &norm, V3 &refl, DMaterial &M, Col &oc)
{
  V3 N = object_norm<PARAM>(sc, id, pt, hu, hv);
  norm = N;
  if (dot(d, norm) > kEps) norm = neg(norm);
  M = sc.materials[sc.obj_mat[id]];
  oc = object_colour<PARAM>(sc, id, pt, hu, hv);         // Scene.h:101
  refl = reflect_dir(N, d);
  return N;
}
__device__ __forceinline__ Col light_term(const DevScene &sc, const DMaterial &M, Col oc, V3 norm, V3 refl, const DLight &L, V3 pt, double l)
{
  V3 ldir = (L.type == 101) ? mk(L.p[0], L.p[1], L.p[2]) : unitize(pt - mk(L.p[0], L.p[1], L.p[2]));
  double diffuse = M.diff * dmax(0.0, -dot(norm, ldir) * l);
  double specular = M.spec * l * pow(dmax(0.0, -dot(ldir, refl)), M.shiness);
  return cmul(oc, M.emissive + sc.ambient * M.ambient + diffuse + specular);
}
// lum[li] = luminosity of light li at pt, computed beforehand by the shadow pass (light_luminosity)
template <bool PARAM = false>
__device__ __forceinline__ Col local_illumination(const DevScene &sc, V3 d, int id, V3 pt, const double *lum, V3 &N_out,
                                                  double hu = kInfty, double hv = kInfty)
{
  Col colour = mkc(0, 0, 0), oc; V3 norm, refl; DMaterial M;
  N_out = shade_setup<PARAM>(sc, d, id, pt, hu, hv, norm, refl, M, oc);
  for (int li = 0; li < sc.n_lights; li++) colour = cadd(colour, light_term(sc, M, oc, norm, refl, sc.lights[li], pt, lum[li]));
  return colour;
}

}  // namespace rt
