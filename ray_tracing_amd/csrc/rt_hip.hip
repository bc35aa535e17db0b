// rt_hip.hip -- gfx950 kernels and the C ABI of include/rt_hip.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see __graft_entry__.build()).
// -ffp-contract=off keeps the reference's rounding in every formula that is restated from it; the
// triangle cull uses explicit fma() where speed matters and exactness does not.
//
// There is no CPU fallback in this file: every entry point needs a HIP device.
#include "../../include/rt_hip.h"
#include "rt_device.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <atomic>
#include <memory>
#include <mutex>
#include <vector>

using namespace rt;

// ================================================================================================
// errors
// ================================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(RT_HIP_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
  } while (0)

extern "C" const char *rt_hip_last_error(void) { return g_err.c_str(); }

// ================================================================================================
// host-side objects
// ================================================================================================
struct DevBuf {
  void *p = nullptr; size_t bytes = 0;
  int ensure(size_t need)
  {
    if (need <= bytes) return RT_HIP_OK;
    if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
    size_t cap = need + need / 4 + 4096;
    hipError_t e = hipMalloc(&p, cap);
    if (e != hipSuccess) { p = nullptr; return fail(RT_HIP_ENOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    bytes = cap;
    return RT_HIP_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// Contexts and scenes are reference counted: a scene keeps its context, a photon map / god ray / communicator keeps what it was
// made from, so that handles may be destroyed in any order (a destroy call drops the caller's reference; the memory goes
// with the last one).
struct rt_hip_ctx {
  int device = 0;
  Stats *d_stats = nullptr;      // accumulated counters
  std::atomic<int> refs{1};
};
static void ctx_retain(rt_hip_ctx *c) { if (c) c->refs.fetch_add(1); }
static void ctx_release(rt_hip_ctx *c)
{
  if (!c || c->refs.fetch_sub(1) != 1) return;
  if (hipSetDevice(c->device) != hipSuccess) (void)hipGetLastError();      // never leave a sticky error behind
  if (c->d_stats) (void)hipFree(c->d_stats);
  delete c;
}

// one node of the ray tree of a pixel sample (level arrays of the wavefront Whitted integrator)
struct alignas(16) PathNode {
  double o[3], d[3];
  double t, weight;
  double col[3];
  double w_refr, w_refl;
  int32_t id; uint32_t pixel; uint32_t sample; uint32_t path;
  int32_t child_refr, child_refl;   // >= 0 index in the next level, -1 none, -2 background (weight cut, RayTracer.h:11)
};
static_assert(sizeof(PathNode) == 128, "PathNode layout");

struct DevCamera {
  V3 view_pt, start, left, up;
  int res, W, H, depth;
};

struct rt_hip_godray;
struct rt_hip_scene {
  std::atomic<int> refs{1};        // the caller's handle + one per photon map / god ray made from it
  rt_hip_ctx *ctx = nullptr;
  rt_hip_godray *god = nullptr;    // set for the duration of rt_hip_render_godray: run_levels adds the light-shaft radiance
  DevScene dev{};
  // scene arrays in HBM; shared (read-only) by the handles made with rt_hip_scene_share
  struct Geometry { std::vector<void *> allocs; ~Geometry() { for (void *p : allocs) (void)hipFree(p); } };
  std::shared_ptr<Geometry> geo = std::make_shared<Geometry>();
  // render workspace (grow-only, reused between calls)
  static constexpr int kMaxLevels = 12;   // = the ray-tree path bits of rng_counter (rt_math.hpp): level L numbers its rays below 2^(L+1); the reference's layouts use depth <= 8
  DevBuf levels[kMaxLevels];
  DevBuf order;                   // heavy-first wave order lists + flags
  DevBuf lum;                     // per node, per light luminosity (shadow pass -> shade pass)
  DevBuf wave_cost;               // k_frame_direct: run time of every wave of the last depth-1 frame (k_cost_order)
  uint64_t cost_sig[6] = {0, 0, 0, 0, 0, 0};   // the tiling those costs belong to (0: none)
  DevBuf counter;                 // queue counter + per-render ray counters
  DevBuf staging;                 // host-pointer entry points
  uint64_t last_rays = 0, last_shadow = 0;
  const void *last_counters = nullptr;   // RenderCounters of the last render call (the scene's own, or the allocation-free entry point's)
  unsigned int last_n_level[kMaxLevels] = {0}; int last_level = 0;   // sizes of the level arrays of the last run: bounds until level_counts() settles them
  bool last_exact[kMaxLevels] = {false};
  unsigned int fan = 2;            // children a ray-tree node can have: 2 when some material both reflects and refracts, else 1
  bool keep_uv = false;            // photon mapping eye pass: keep the (u, v) of every node's hit (k_shade overwrites them)
  bool one_sample = false;         // exactly one light and it takes one shadow ray per point: k_frame_direct has a loop-free form for it
  DevBuf uv[kMaxLevels];
  // rt_hip_render_whitted_batch_dev: the frames' cameras go to the device through a ring of pinned slots (a copy from pageable memory would make
  // every call wait for the stream); a slot is reused only after the launch that read it has finished (its event)
  static constexpr int kBatchSlots = 8, kBatchMax = 64;
  DevBuf batch_cams; DevCamera *batch_pinned = nullptr; hipEvent_t batch_ev[kBatchSlots] = {}; bool batch_ev_used[kBatchSlots] = {}; unsigned int batch_next = 0;
  // deferred Newton solves (rt_defer.inc): the global pair queue and the solver's per-batch buffers
  DevBuf gq_pairs, gq_counts, gq_res, gq_rays, gq_uvh, gq_seeds, gq_seed_mask, gq_occ;
  bool defer_fits = false;         // every Newton-solved object, profile sample and light sample has a bit in the masks of rt_defer.inc
  unsigned int max_seeds = 1, shadow_samples = 1;   // largest uSampleNum + 1; shadow rays per shaded point (all lights)
  uint64_t last_pairs = 0;         // (ray, object) pairs the last render queued (diagnostic)
  DevBuf aa_maps, aa_edges, aa_mask;   // anti-aliasing drivers: G-buffer maps, list of masked pixels, the mask when the caller keeps none
  uint64_t last_aa_edges = 0;
  // optional per-kernel timing (rt_hip_set_profiling)
  bool prof_on = false;
  struct EvPair { hipEvent_t a, b; int k; };
  std::vector<EvPair> ev_pool; size_t ev_used = 0;
  rt_hip_profile prof{};
  // tree statistics after refinement (for tests / DESIGN)
  uint32_t n_nodes = 0, n_leaves = 0, n_cull = 0, max_depth = 0;
};

// level_n[lv]: nodes of ray-tree level lv of the render in flight, counted on the device (k_shade appends level lv + 1) so that the host
// never waits for a level to learn the size of the next
struct RenderCounters { unsigned int queue_count; unsigned int pad; unsigned long long rays; unsigned long long shadow_rays; unsigned int level_n[16];
                        unsigned long long shadow_rays_proved; };   // of shadow_rays: not traced, their points proved unoccluded (area_light_all_clear)
// number of nodes a level kernel works on: the host's bound (cap), cut to the device's count when one is given
__device__ __forceinline__ unsigned int level_count(const unsigned int *n_dev, unsigned int cap)
{ if (!n_dev) return cap; const unsigned int n = *n_dev; return n < cap ? n : cap; }

// ================================================================================================
// scene preparation (host)
// ================================================================================================
namespace {
// Leaf size of the library's own hierarchy (leaves hold fewer objects than this).  Sweeps of round 2 with the FP32 box tests: C2 (1 M
// triangles) 12: -2 %, 16: 0, 18-20: +1.4 %, 24: 0, 32: -6 %, 62: -27 %; C4 (11.6 k objects) flat from 8 to 24; C3 (a 1,369-triangle solid, 100
// shadow rays per sample: the records culled per ray are what its time goes into) 4: +28 %, 6-8: +33 %, 12: +25 %, 20: +13 % against 24.
// A node step on a scene whose pairs stay in the scalar cache is cheap, on a large one it is a memory round trip: small scenes get small leaves.
constexpr uint32_t kOwnLeaf = 20, kOwnLeafSmall = 8, kSmallScene = 4096;
static uint32_t own_leaf_default(uint32_t n_objects) { return n_objects <= kSmallScene ? kOwnLeafSmall : kOwnLeaf; }

struct Box { double mn[3], mx[3]; };

#ifndef RT_TEST_NO_HULL_BOXES
#define RT_TEST_NO_HULL_BOXES 0   /* 1: a build WITHOUT the hull boxes below, to see tests/test_gpu_parity.py::test_rotational_overshoot_in_tree fail on it */
#endif
struct Builder {
  const rt_hip_scene_desc *d;
  uint32_t leaf_refine;
  std::vector<Box> aabb;            // per object (finite ones)
  std::vector<uint8_t> finite;
  // Surfaces of revolution whose surface leaves their m_AABB: CRotationalObject::InitializeAABB takes z from the profile's two END points only
  // (Object.h:954-957), and the reference uses that box as a ray filter and for the tree's lists, not as a bound of the surface -- a profile whose
  // inner control points overshoot in height is hit OUTSIDE it.  The walk prunes by leaf boxes, so every leaf that lists such an object has its box
  // grown (unclipped) to the object's control-hull box (rt_param.hpp: param_prune_box): the hit lies in the leaf's box whatever the tree.
  std::vector<uint8_t> loose;       // per object: 1 = hull[] holds a box that reaches beyond aabb[]
  std::vector<Box> hull;
  std::vector<DPair> pairs;
  std::vector<unsigned char> stream;   // leaf headers + records
  std::vector<TriExact> tri_exact;
  uint32_t n_leaves = 0, max_depth = 0, n_recs = 0;
  std::string err;
  struct Sub { ChildRef ref; Box box; bool empty; };

  static void rot_box(const rt_hip_rotational &z, double *mn, double *mx)
  {
    const int n = z.n_profile;
    double r = 0;
    for (int k = 0; k < n; k++) if (z.profile[k][0] > r) r = z.profile[k][0];
    mn[0] = z.center[0] - r - kEps; mx[0] = z.center[0] + r + kEps;
    mn[1] = z.center[1] - r - kEps; mx[1] = z.center[1] + r + kEps;
    mn[2] = z.center[2] + std::min(z.profile[0][1], z.profile[n - 1][1]) - kEps;
    mx[2] = z.center[2] + std::max(z.profile[0][1], z.profile[n - 1][1]) + kEps;
  }
  // object AABBs as the reference builds them: CTriangle::InitializeAABB (Object.h:459-468),
  // CSphere ctor (Object.h:234); planes and circles keep the infinite default (Object.h:34)
  void object_boxes()
  {
    aabb.resize(d->n_objects); finite.assign(d->n_objects, 0); loose.assign(d->n_objects, 0); hull.resize(d->n_objects);
    for (uint32_t i = 0; i < d->n_objects; i++) {
      uint32_t k = d->obj_index[i];
      Box b;
      if (d->obj_type[i] == RT_HIP_TRIANGLE) {
        const rt_hip_triangle &t = d->triangles[k];
        for (int a = 0; a < 3; a++) {
          b.mn[a] = std::min(std::min(t.v[0][a], t.v[1][a]), t.v[2][a]) - kEps;
          b.mx[a] = std::max(std::max(t.v[0][a], t.v[1][a]), t.v[2][a]) + kEps;
        }
        finite[i] = 1;
      } else if (d->obj_type[i] == RT_HIP_SPHERE) {
        const rt_hip_sphere &s = d->spheres[k];
        for (int a = 0; a < 3; a++) { b.mn[a] = s.c[a] - s.r; b.mx[a] = s.c[a] + s.r; }
        finite[i] = 1;
      } else if (d->obj_type[i] == RT_HIP_BEZIER) {          // CBezierObject::InitializeAABB, Object.h:1204-1227
        const rt_hip_bezier &z = d->beziers[k];
        for (int a = 0; a < 3; a++) {
          double lo = kInfty, hi = -kInfty;
          for (int p = 0; p < 4; p++) for (int q = 0; q < 4; q++) { lo = std::min(lo, z.ctrl[p][q][a]); hi = std::max(hi, z.ctrl[p][q][a]); }
          b.mn[a] = lo + z.center[a]; b.mx[a] = hi + z.center[a];
        }
        finite[i] = 1;
      } else if (d->obj_type[i] == RT_HIP_ROTATIONAL) {      // CRotationalObject::InitializeAABB, Object.h:954-970
        const rt_hip_rotational &z = d->rotationals[k];
        rot_box(z, b.mn, b.mx);
        finite[i] = 1;
        DRotational o; std::memset(&o, 0, sizeof o);
        std::memcpy(o.prof, z.profile, sizeof o.prof); std::memcpy(o.center, z.center, 24);
        o.u_range = z.u_range; o.n_prof = z.n_profile; o.u_sample_num = z.u_sample_num;
        rot_box(z, o.bmin, o.bmax);
        param_prune_box(o);                                   // pmin / pmax: where an accepted hit can lie
        for (int a = 0; a < 3; a++) {
          hull[i].mn[a] = std::min(b.mn[a], o.pmin[a]); hull[i].mx[a] = std::max(b.mx[a], o.pmax[a]);
          if (o.pmin[a] < b.mn[a] || o.pmax[a] > b.mx[a]) loose[i] = 1;
        }
      } else {
        for (int a = 0; a < 3; a++) { b.mn[a] = kInfty; b.mx[a] = kInfty; }
      }
      aabb[i] = b;
    }
  }
  // CBaseObject::PointInBox, Object.h:212-222
  bool overlaps(uint32_t id, const Box &box) const
  {
    const Box &b = aabb[id];
    for (int a = 0; a < 3; a++)
      if (b.mn[a] > box.mx[a] + kEps || b.mx[a] < box.mn[a] - kEps) return false;
    return true;
  }

  LeafRec make_rec(uint32_t prim) const
  {
    LeafRec c; std::memset(&c, 0, sizeof c);
    c.prim_id = (int32_t)prim;
    if (d->obj_type[prim] != RT_HIP_TRIANGLE) { c.kind = 2; return c; }
    c.tri_idx = (int32_t)d->obj_index[prim];
    const TriExact &T = tri_exact[d->obj_index[prim]];
    c.kind = 1;                      // until the cull data prove usable: the normal stays zero, so that d.n = 0 marks the record untrusted by itself
    V3 v0 = mk(T.v0[0], T.v0[1], T.v0[2]), v1 = mk(T.v1[0], T.v1[1], T.v1[2]), v2 = mk(T.v2[0], T.v2[1], T.v2[2]);
    V3 n = mk(T.nrm[0], T.nrm[1], T.nrm[2]);
    V3 e1 = v1 - v0, e2 = v2 - v0;
    V3 N = cross(e1, e2);
    double nn = dot(N, N);
    double l1 = len(e1), l2 = len(e2);
    bool ok = std::isfinite(nn) && nn > 0 && !is_infty_pt(n) && std::isfinite(n.x) && l1 > 0 && l2 > 0;
    double sinv = ok ? (l1 * l2) / std::sqrt(nn) : 1e30;   // 1/sin(angle at v0)
    if (!ok || sinv > 1e3) return c;
    // dual basis: beta = dot(P - v0, B), gamma = dot(P - v0, G) with B = (e2 x N)/|N|^2, G = (N x e1)/|N|^2
    V3 B = cross(e2, N) / nn, G = cross(N, e1) / nn;
    double bsum = std::fabs(B.x) + std::fabs(B.y) + std::fabs(B.z) + std::fabs(G.x) + std::fabs(G.y) + std::fabs(G.z) + sinv;
    double b0 = -dot(v0, B), g0 = -dot(v0, G), dpl = dot(v0, n);
    if (!std::isfinite(bsum) || !std::isfinite(b0) || !std::isfinite(g0) || !std::isfinite(dpl)) return c;
    c.dpl = dpl;
    c.B[0] = B.x; c.B[1] = B.y; c.B[2] = B.z; c.b0 = b0;
    c.G[0] = G.x; c.G[1] = G.y; c.G[2] = G.z; c.g0 = g0;
    std::memcpy(c.n, T.nrm, 24);     // m_norm
    c.bsum = bsum; c.kind = 0;
    return c;
  }

  // FP32 box coordinates, rounded away from the box's inside
  static float f32_down(double x) { float f = (float)x; return ((double)f > x) ? std::nextafterf(f, -INFINITY) : f; }
  static float f32_up(double x) { float f = (float)x; return ((double)f < x) ? std::nextafterf(f, INFINITY) : f; }
  // emits a leaf for the object list `objs` in kd cell `cell`
  Sub emit_leaf(const std::vector<uint32_t> &objs, const Box &cell, uint32_t depth)
  {
    Sub out; out.empty = objs.empty(); out.ref = ChildRef{0, 0}; out.box = cell;
    if (out.empty) return out;
    if (objs.size() > 0x7FFFFFF0u) { err = "leaf too large"; out.empty = true; return out; }
    Box u; for (int a = 0; a < 3; a++) { u.mn[a] = 1e300; u.mx[a] = -1e300; }
    for (uint32_t id : objs)
      for (int a = 0; a < 3; a++) { u.mn[a] = std::min(u.mn[a], aabb[id].mn[a]); u.mx[a] = std::max(u.mx[a], aabb[id].mx[a]); }
    LeafHeader H; std::memset(&H, 0, sizeof H);
    for (int a = 0; a < 3; a++) {
      // every hit point of a referenced object inside this cell lies in cell /\ union(AABB)
      out.box.mn[a] = std::max(cell.mn[a], u.mn[a]) - kBoxPad;
      out.box.mx[a] = std::min(cell.mx[a], u.mx[a]) + kBoxPad;
      H.bmin[a] = f32_down(out.box.mn[a]); H.bmax[a] = f32_up(out.box.mx[a]);
    }
    for (uint32_t id : objs)
      if (loose[id] && !RT_TEST_NO_HULL_BOXES)
        for (int a = 0; a < 3; a++) { out.box.mn[a] = std::min(out.box.mn[a], hull[id].mn[a] - kBoxPad); out.box.mx[a] = std::max(out.box.mx[a], hull[id].mx[a] + kBoxPad); }
    for (int a = 0; a < 3; a++) { H.bmin[a] = f32_down(out.box.mn[a]); H.bmax[a] = f32_up(out.box.mx[a]); }
    H.count = (int32_t)objs.size();
    size_t off = stream.size();
    if (off / 64 > 0x7FFFFFFFu) { err = "leaf stream too large"; out.empty = true; return out; }
    stream.resize(off + 64 + objs.size() * sizeof(LeafRec));
    std::memcpy(&stream[off], &H, 64);
    LeafRec *recs = (LeafRec *)&stream[off + 64];
    for (size_t i = 0; i < objs.size(); i++) recs[i] = make_rec(objs[i]);
    out.ref.i0 = -1 - (int32_t)objs.size(); out.ref.i1 = (int32_t)(off / 64);
    n_leaves++; n_recs += (uint32_t)objs.size(); max_depth = std::max(max_depth, depth);
    return out;
  }
  Sub emit_inner(int axis, int slot, const Sub &l, const Sub &r, uint32_t depth)
  {
    // `slot` was reserved before the children were built (parents precede children in memory)
    if (l.empty || r.empty) {
      // the reserved slot stays unused when a side is empty: mark it harmless
      std::memset(&pairs[slot], 0, sizeof(DPair));
      return l.empty ? r : l;
    }
    DPair P; std::memset(&P, 0, sizeof P);
    for (int a = 0; a < 3; a++) { P.lmin[a] = l.box.mn[a]; P.lmax[a] = l.box.mx[a]; P.rmin[a] = r.box.mn[a]; P.rmax[a] = r.box.mx[a]; }
    P.l = l.ref; P.r = r.ref; P.axis = axis;
    pairs[slot] = P;
    Sub out; out.empty = false; out.ref = ChildRef{slot, axis};
    for (int a = 0; a < 3; a++) { out.box.mn[a] = std::min(l.box.mn[a], r.box.mn[a]); out.box.mx[a] = std::max(l.box.mx[a], r.box.mx[a]); }
    max_depth = std::max(max_depth, depth);
    return out;
  }
  // Splits a cell below a reference leaf (leaf_refine = n, rt_hip.h).  The closest hit does not depend on where the
  // planes lie -- every object is listed in every cell its box overlaps (the rule of Object.h:212-222) and a leaf's
  // hits are compared as in Scene.h:349-387 -- so the plane is chosen for speed: of 15 equidistant planes per axis
  // across the content's bounds inside the cell, the one with the least surface-area cost
  // area(L) n(L) + area(R) n(R).  (The reference's rule, the cell's middle on axis depth % 3, spends a third of its
  // planes across the thin side of flat content: C2 ran 15 % slower on it.)
  bool choose_plane(const std::vector<uint32_t> &objs, const Box &cell, int &axis, double &mid) const
  {
    Box t; for (int a = 0; a < 3; a++) { t.mn[a] = 1e300; t.mx[a] = -1e300; }
    for (uint32_t id : objs)
      for (int a = 0; a < 3; a++) { t.mn[a] = std::min(t.mn[a], aabb[id].mn[a]); t.mx[a] = std::max(t.mx[a], aabb[id].mx[a]); }
    for (int a = 0; a < 3; a++) { t.mn[a] = std::max(t.mn[a], cell.mn[a]); t.mx[a] = std::min(t.mx[a], cell.mx[a]); }
    constexpr int NB = 16;
    double best = 1e300; bool found = false;
    for (int a = 0; a < 3; a++) {
      const double ext = t.mx[a] - t.mn[a];
      if (!(ext > 0) || !std::isfinite(ext)) continue;
      const double step = ext / NB;
      // object -> first plane it is left of (counts in L from there on), last plane it is right of (counts in R up to there)
      size_t first_l[NB + 2] = {0}, last_r[NB + 2] = {0};
      for (uint32_t id : objs) {
        double kl = std::ceil((aabb[id].mn[a] - kEps - t.mn[a]) / step), kr = std::floor((aabb[id].mx[a] + kEps - t.mn[a]) / step);
        int il = kl < 1 ? 1 : (kl > NB ? NB : (int)kl), ir = kr > NB - 1 ? NB - 1 : (kr < 0 ? 0 : (int)kr);
        first_l[il]++; last_r[ir]++;
      }
      size_t nl = 0, nr_after[NB + 2] = {0};
      for (int k = NB - 1; k >= 1; k--) nr_after[k] = nr_after[k + 1] + last_r[k];
      double e[3]; for (int q = 0; q < 3; q++) e[q] = t.mx[q] - t.mn[q];
      for (int k = 1; k < NB; k++) {
        nl += first_l[k];
        const size_t nr = nr_after[k];
        const double wl = step * k, wr = ext - wl;
        double f[3] = {e[0], e[1], e[2]};
        f[a] = wl; const double al = f[0] * f[1] + f[1] * f[2] + f[0] * f[2];
        f[a] = wr; const double ar = f[0] * f[1] + f[1] * f[2] + f[0] * f[2];
        const double cost = al * (double)nl + ar * (double)nr;
        if (cost < best) { best = cost; axis = a; mid = t.mn[a] + wl; found = true; }
      }
    }
    return found;
  }
  // a subtree over `objs` in `cell`, down to leaf_refine objects per leaf
  Sub refine(std::vector<uint32_t> &objs, const Box &cell, uint32_t depth)
  {
    if (leaf_refine == 0 || objs.size() < leaf_refine || depth >= 56) return emit_leaf(objs, cell, depth);
    int axis = 0; double mid = 0;
    if (!choose_plane(objs, cell, axis, mid)) return emit_leaf(objs, cell, depth);
    Box L = cell, R = cell;
    L.mx[axis] = mid; R.mn[axis] = mid;
    std::vector<uint32_t> lo, ro;
    for (uint32_t id : objs) { if (overlaps(id, L)) lo.push_back(id); if (overlaps(id, R)) ro.push_back(id); }
    // stop when the split no longer separates anything (all objects straddle the plane)
    if (lo.size() == objs.size() && ro.size() == objs.size()) return emit_leaf(objs, cell, depth);
    std::vector<uint32_t>().swap(objs);
    int slot = (int)pairs.size(); pairs.emplace_back();
    Sub l = refine(lo, L, depth + 1);
    Sub r = refine(ro, R, depth + 1);
    return emit_inner(axis, slot, l, r, depth);
  }
  // the checks of walk() without building anything (the library's own tree still wants a well-formed description)
  bool validate(uint32_t ni, uint32_t depth)
  {
    if (ni >= d->n_nodes) { err = "node index out of range"; return false; }
    const rt_hip_node &n = d->nodes[ni];
    if (depth > 56) { err = "tree deeper than 56 levels"; return false; }
    if (n.split_axis < 0) {
      if (n.a < 0 || n.b < 0 || (uint64_t)n.a + (uint64_t)n.b > d->n_refs) { err = "leaf ref range out of bounds"; return false; }
      for (int i = 0; i < n.b; i++) if (d->refs[n.a + i] >= d->n_objects) { err = "leaf ref out of range"; return false; }
      return true;
    }
    if (n.split_axis > 2 || n.a <= (int)ni || n.b <= (int)ni) { err = "bad inner node"; return false; }
    return validate((uint32_t)n.a, depth + 1) && validate((uint32_t)n.b, depth + 1);
  }
  // The library's own hierarchy (leaf_refine = DEFAULT or OWN | n): an object partition instead of the reference's
  // space partition.  Every object goes to ONE side (by the centre of its box; of 15 cuts per axis the one with the
  // least surface-area cost), the children's boxes are the bounds of their objects and may overlap.  No object is
  // listed twice, so the leaf stream is as short as it can be (C2: 1.0 M records instead of 1.8 M) and a ray meets
  // each object at most once.  The walk prunes by box and bound in any order, and the candidates' comparison is the
  // same, so the hits are the same (tests/test_gpu_parity.py runs every mode).
  static double half_area(const Box &b)
  {
    const double x = b.mx[0] - b.mn[0], y = b.mx[1] - b.mn[1], z = b.mx[2] - b.mn[2];
    return x * y + y * z + x * z;
  }
  void grow(Box &b, uint32_t id) const
  {
    for (int a = 0; a < 3; a++) { b.mn[a] = std::min(b.mn[a], aabb[id].mn[a]); b.mx[a] = std::max(b.mx[a], aabb[id].mx[a]); }
  }
  Sub bvh(std::vector<uint32_t> &objs, const Box &cell, uint32_t depth)
  {
    if (objs.size() < leaf_refine || depth >= 56) return emit_leaf(objs, cell, depth);
    constexpr int NB = 16;
    const Box none = {{1e300, 1e300, 1e300}, {-1e300, -1e300, -1e300}};
    double cmn[3] = {1e300, 1e300, 1e300}, cmx[3] = {-1e300, -1e300, -1e300};
    for (uint32_t id : objs)
      for (int a = 0; a < 3; a++) { const double c = (aabb[id].mn[a] + aabb[id].mx[a]) / 2; cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c); }
    double best = 1e300; int axis = -1, cut = 0;
    for (int a = 0; a < 3; a++) {
      const double ext = cmx[a] - cmn[a];
      if (!(ext > 0) || !std::isfinite(ext)) continue;
      Box bb[NB]; size_t bn[NB] = {0};
      for (int k = 0; k < NB; k++) bb[k] = none;
      for (uint32_t id : objs) {
        int k = (int)(NB * (((aabb[id].mn[a] + aabb[id].mx[a]) / 2 - cmn[a]) / ext)); k = k < 0 ? 0 : (k > NB - 1 ? NB - 1 : k);
        bn[k]++; grow(bb[k], id);
      }
      Box rb[NB]; size_t rn[NB]; Box acc = none; size_t n = 0;
      for (int k = NB - 1; k >= 1; k--) { if (bn[k]) { for (int q = 0; q < 3; q++) { acc.mn[q] = std::min(acc.mn[q], bb[k].mn[q]); acc.mx[q] = std::max(acc.mx[q], bb[k].mx[q]); } } n += bn[k]; rb[k] = acc; rn[k] = n; }
      acc = none; n = 0;
      for (int k = 1; k < NB; k++) {
        if (bn[k - 1]) { for (int q = 0; q < 3; q++) { acc.mn[q] = std::min(acc.mn[q], bb[k - 1].mn[q]); acc.mx[q] = std::max(acc.mx[q], bb[k - 1].mx[q]); } }
        n += bn[k - 1];
        if (n == 0 || rn[k] == 0) continue;
        const double cost = half_area(acc) * (double)n + half_area(rb[k]) * (double)rn[k];
        if (cost < best) { best = cost; axis = a; cut = k; }
      }
    }
    std::vector<uint32_t> lo, ro;
    if (axis >= 0) {
      const double ext = cmx[axis] - cmn[axis];
      for (uint32_t id : objs) {
        int k = (int)(NB * (((aabb[id].mn[axis] + aabb[id].mx[axis]) / 2 - cmn[axis]) / ext)); k = k < 0 ? 0 : (k > NB - 1 ? NB - 1 : k);
        (k < cut ? lo : ro).push_back(id);
      }
    } else {                   // all centres coincide: halve the list
      axis = 0;
      lo.assign(objs.begin(), objs.begin() + objs.size() / 2); ro.assign(objs.begin() + objs.size() / 2, objs.end());
    }
    std::vector<uint32_t>().swap(objs);
    int slot = (int)pairs.size(); pairs.emplace_back();
    Sub l = bvh(lo, cell, depth + 1);
    Sub r = bvh(ro, cell, depth + 1);
    return emit_inner(axis, slot, l, r, depth);
  }
  // the library's own tree over the reference's root cell
  Sub rebuild()
  {
    Sub bad; bad.empty = true; bad.ref = ChildRef{0, 0};
    if (!validate(0, 0)) return bad;
    std::vector<uint32_t> all;
    for (uint32_t i = 0; i < d->n_objects; i++) if (finite[i]) all.push_back(i);
    Box cell; for (int a = 0; a < 3; a++) { cell.mn[a] = d->nodes[0].box_min[a]; cell.mx[a] = d->nodes[0].box_max[a]; }
    bad.box = cell;
    return bvh(all, cell, 0);
  }
  // walks the reference tree of the description
  Sub walk(uint32_t ni, uint32_t depth)
  {
    Sub bad; bad.empty = true; bad.ref = ChildRef{0, 0};
    if (ni >= d->n_nodes) { err = "node index out of range"; return bad; }
    const rt_hip_node &n = d->nodes[ni];
    Box cell; for (int a = 0; a < 3; a++) { cell.mn[a] = n.box_min[a]; cell.mx[a] = n.box_max[a]; }
    bad.box = cell;
    if (depth > 56) { err = "tree deeper than 56 levels"; return bad; }
    if (n.split_axis < 0) {
      if (n.a < 0 || n.b < 0 || (uint64_t)n.a + (uint64_t)n.b > d->n_refs) { err = "leaf ref range out of bounds"; return bad; }
      std::vector<uint32_t> objs;
      for (int i = 0; i < n.b; i++) {
        uint32_t id = d->refs[n.a + i];
        if (id >= d->n_objects) { err = "leaf ref out of range"; return bad; }
        if (finite[id]) objs.push_back(id);   // infinite objects only reach the device through root_objects
      }
      return refine(objs, cell, depth);
    }
    if (n.split_axis > 2 || n.a <= (int)ni || n.b <= (int)ni) { err = "bad inner node"; return bad; }
    int slot = (int)pairs.size(); pairs.emplace_back();
    Sub l = walk((uint32_t)n.a, depth + 1);
    if (!err.empty()) return bad;
    Sub r = walk((uint32_t)n.b, depth + 1);
    if (!err.empty()) return bad;
    return emit_inner(n.split_axis, slot, l, r, depth);
  }
};

template <class T>
int upload(rt_hip_scene *s, const std::vector<T> &v, const T **out)
{
  *out = nullptr;
  if (v.empty()) return RT_HIP_OK;
  void *p = nullptr;
  HIP_TRY(hipMalloc(&p, v.size() * sizeof(T)));
  s->geo->allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T *)p;
  return RT_HIP_OK;
}

}  // namespace

// ================================================================================================
// kernels
// ================================================================================================
#ifndef RT_BLOCK
#define RT_BLOCK 64     /* one wave per workgroup: waves of very different cost pack the CUs best (measured) */
#endif
#ifndef RT_MINWAVES
#define RT_MINWAVES 4   /* __launch_bounds__ second argument: 4 waves per SIMD = a 128-VGPR budget (no spills at 125) */
#endif
#define RT_WAVES (RT_BLOCK / 64)
#ifndef RT_PARAM_MINWAVES
#define RT_PARAM_MINWAVES 2   /* kernels with the Newton solvers compiled in: two waves per SIMD (256 registers per lane, some spills) measured faster than one (512, none) and than three */
#endif

// diagnostic only (collect_stats builds of the kernels): start/end wall-clock stamp of every wave
#define RT_MAX_STAMPS 65536
__device__ unsigned long long g_stamps[4 * RT_MAX_STAMPS];   // [kernel: 0 k_shadow, 1 k_primary][wave][start, end]
__device__ __forceinline__ void stamp(unsigned int wave, int which, int kernel = 0)
{
  if ((threadIdx.x & 63) == 0 && wave < RT_MAX_STAMPS) g_stamps[2 * RT_MAX_STAMPS * kernel + 2 * wave + which] = wall_clock64();
}
#ifdef RT_SOLVE_STATS
extern "C" int rt_hip_debug_read_solve_steps(unsigned long long *out, int reset)
{
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_solve_steps), 16 * 8, 0, hipMemcpyDeviceToHost));
  if (reset) { unsigned long long z[16] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rt::g_solve_steps), z, sizeof z, 0, hipMemcpyHostToDevice)); }
  return RT_HIP_OK;
}
#endif
#ifdef RT_TIMERS
extern "C" int rt_hip_debug_read_timers(unsigned long long *out, int reset)
{
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_timers), 32 * 8, 0, hipMemcpyDeviceToHost));
  if (reset) { unsigned long long z[32] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(rt::g_timers), z, sizeof z, 0, hipMemcpyHostToDevice)); }
  return 0;
}
#endif
extern "C" int rt_hip_debug_read_stamps(unsigned long long *out, size_t n_waves)
{
  if (n_waves > RT_MAX_STAMPS) n_waves = RT_MAX_STAMPS;
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), n_waves * 16, 0, hipMemcpyDeviceToHost));
  return RT_HIP_OK;
}
extern "C" int rt_hip_debug_read_stamps2(unsigned long long *out, size_t n_waves, int kernel)
{
  if (n_waves > RT_MAX_STAMPS) n_waves = RT_MAX_STAMPS;
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), n_waves * 16, (size_t)kernel * 16 * RT_MAX_STAMPS, hipMemcpyDeviceToHost));
  return RT_HIP_OK;
}


__device__ __forceinline__ void stats_flush(Stats *g, const Stats &s)
{
  if (!g) return;
  // per-lane counters -> one atomic per lane is too many: reduce over the wave first
  unsigned long long v[8] = {s.rays, s.shadow_rays, s.node_visits, s.prim_tests, s.exact_tests, s.wave_nodes, s.wave_prims, s.pad};
#pragma unroll
  for (int k = 0; k < 8; k++) {
    unsigned long long x = v[k];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0 && x) atomicAdd(((unsigned long long *)g) + k, x);
  }
}

// XCD-aware block remap.  Consecutive hardware block ids go round-robin over the 8 XCDs (each with its
// own L2).  Work items are handed out in chunks of RT_XCD_CHUNK consecutive blocks: chunk c goes to the
// blocks with id % 8 == c % 8, so that one XCD's L2 serves spatially adjacent work while all XCDs walk
// the image together (a contiguous 1/8 of the frame per XCD would tie the kernel to its most expensive
// eighth).  The tail that does not fill 8 whole chunks keeps its identity mapping (bijective).
#define RT_XCD_CHUNK (128 / RT_WAVES)   /* one 128-wave super-tile per XCD turn */
__device__ __forceinline__ unsigned int xcd_remap(unsigned int b, unsigned int nb)
{
  const unsigned int group = 8u * RT_XCD_CHUNK;
  const unsigned int full = (nb / group) * group;
  if (b >= full) return b;
  const unsigned int g = b / group, r = b % group;
  const unsigned int x = r % 8, i = r / 8;            // x: which XCD's turn, i: position inside its chunk
  return g * group + x * RT_XCD_CHUNK + i;
}

// rt_hip_trace_closest: CScene::FirstInterPt_BSPTree / FirstInterPt for a list of rays
template <bool STATS, bool PARAM>
__global__ void __launch_bounds__(RT_BLOCK, PARAM ? RT_PARAM_MINWAVES : RT_MINWAVES) k_trace_closest(DevScene sc, const rt_hip_ray *rays, size_t n, rt_hip_hit *hits, Stats *gstats, int record)
{
  __shared__ WaveLds s_lds[RT_WAVES];
  WaveLds *stack = &s_lds[threadIdx.x >> 6];
  const size_t i = (size_t)xcd_remap(blockIdx.x, gridDim.x) * RT_BLOCK + threadIdx.x;
  const bool active = i < n;
  Stats st{};
  V3 o = mk(0, 0, 0), d = mk(1, 0, 0);
  if (active) { const rt_hip_ray q = rays[i]; o = mk(q.o[0], q.o[1], q.o[2]); d = unitize(mk(q.d[0], q.d[1], q.d[2])); }
  RayState r; ray_setup(r, o, d, sc.coord_scale);
  if (STATS && active) st.rays++;
  trace_closest<STATS, PARAM>(sc, r, active, stack, st, record != 0);
  if (active) {
    rt_hip_hit h; h.prim_id = r.best_id; h.reserved = 0; h.u = kInfty; h.v = kInfty;
    h.t = (r.best_id >= 0) ? r.best_t : kInfty;
    if (PARAM && r.best_id >= 0) { h.u = r.best_u; h.v = r.best_v; }
    hits[i] = h;
  }
  if (STATS) stats_flush(gstats, st);
}

template <bool STATS, bool PARAM>
__global__ void __launch_bounds__(RT_BLOCK, PARAM ? RT_PARAM_MINWAVES : RT_MINWAVES) k_trace_occluded(DevScene sc, const rt_hip_ray *rays, const double *ref_pts, size_t n,
                                                              uint8_t *occluded, Stats *gstats)
{
  __shared__ WaveLds s_lds[RT_WAVES];
  WaveLds *stack = &s_lds[threadIdx.x >> 6];
  const size_t i = (size_t)xcd_remap(blockIdx.x, gridDim.x) * RT_BLOCK + threadIdx.x;
  const bool active = i < n;
  Stats st{};
  V3 o = mk(0, 0, 0), d = mk(1, 0, 0), ref = mk(0, 0, 0);
  if (active) {
    const rt_hip_ray q = rays[i]; o = mk(q.o[0], q.o[1], q.o[2]); d = unitize(mk(q.d[0], q.d[1], q.d[2]));
    if (ref_pts) ref = mk(ref_pts[3 * i], ref_pts[3 * i + 1], ref_pts[3 * i + 2]);
  }
  RayState r; ray_setup(r, o, d, sc.coord_scale);
  const double t_limit = ref_pts ? fabs(dot(ref - o, d)) + 1e-6 * (1.0 + sc.coord_scale) : kMiss;
  if (STATS && active) { st.rays++; st.shadow_rays++; }
  bool occ = trace_occluded<STATS, PARAM>(sc, r, active, ref_pts != nullptr, ref, t_limit, stack, st);
  if (active) occluded[i] = occ ? 1 : 0;
  if (STATS) stats_flush(gstats, st);
}

struct RenderArgs {
  DevCamera cam;
  double bg[3];
  double weight_threshold;
  uint64_t seed;
  uint64_t pixel_first, pixel_count;
  uint32_t v_lo, tiles_x, tiles_y;   // 8x8 tiling of the rows that hold the shard
  uint32_t sample_first, n_samples;   // samples handled by this launch
  double lens_radius; int focusing;
  // interleaved shard of the 128x64-pixel super-tiles (rt_hip_params.tile_offset / tile_stride): this call renders the
  // super-tiles tile_offset, tile_offset + tile_stride, ... (n_st of them) and its outputs hold them one after the other
  uint32_t tile_offset, tile_stride, n_st;
  uint32_t out_bgr8;                  // the frame goes out as the reference's Vec3b picture (Camera.h:69) instead of f64 R,G,B
  uint64_t n_out, n_primary;          // entries of the output arrays; camera rays of this call
  unsigned int *wave_cost;            // k_frame_direct: where a wave leaves its run time in cycles (NULL: nowhere), see k_cost_order
  // rt_hip_render_whitted_batch_dev: n_samples frames in one launch -- the "sample" a wave belongs to is its frame; cameras and output pitches per frame
  const DevCamera *batch_cams; uint64_t batch_rgb_pitch, batch_id_pitch;      // pitches in bytes / in entries
};

// ---- heavy-first work order --------------------------------------------------------------------
// Waves whose rays can meet the tree cost 10-50x more than waves that only see the infinite planes.  A
// grid in image order leaves the expensive tiles of the lower image rows for the end, and the kernel
// then drains at low occupancy for a third of its run time.  k_order lists the waves flagged heavy
// first (in their original, spatially coherent order), the light ones after them.  Results do not
// depend on the order.
__device__ __forceinline__ bool ray_meets_box(V3 o, V3 d, const double *bmin, const double *bmax)
{
  RayState r; r.o = o; r.d = d; r.inv = mk(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z)); r.oi = mk(o.x * r.inv.x, o.y * r.inv.y, o.z * r.inv.z); r.bound = kMiss;
  return slab(r, bmin, bmax);
}

// (BATCH, rt_hip_render_whitted_batch_dev: s_local = the frame, frame-major -- every frame's tiles together, so that one frame's slow last tiles
// run beside the next frame's first ones.  Measured against it: the frames' copies of a super-tile side by side on one XCD, to share its L2 --
// no gain at N = 1, 10 % slower on half and quarter frames: the slow tiles of all frames then end the launch together.)
template <bool BATCH = false>
__device__ __forceinline__ void tile_of_wave(const RenderArgs &A, unsigned int wave, unsigned int &s_local, unsigned int &tx, unsigned int &ty)
{
  const unsigned int sx = (A.tiles_x + 15) / 16;
  const unsigned int per_sample = A.n_st * 128;
  s_local = wave / per_sample;
  const unsigned int wi = wave % per_sample, st_i = A.tile_offset + (wi / 128) * A.tile_stride, in_st = wi % 128;
  tx = (st_i % sx) * 16 + (in_st % 16); ty = (st_i / sx) * 8 + (in_st / 16);
}
// where pixel (h, v) goes in the output arrays: raster order relative to the pixel shard, or -- with a tile shard -- the
// call's super-tiles one after the other, 128 x 64 entries each in row-major order (entries outside the image stay unwritten)
__device__ __forceinline__ uint64_t out_slot(const RenderArgs &A, unsigned int h, unsigned int v)
{
  if (A.tile_stride > 1) {
    const unsigned int sx = (A.tiles_x + 15) / 16;
    const unsigned int st_local = ((v / 64) * sx + h / 128 - A.tile_offset) / A.tile_stride;
    return (uint64_t)st_local * 8192 + (v % 64) * 128 + (h % 128);
  }
  return (uint64_t)v * A.cam.W + h - A.pixel_first;
}
// Camera.h:69: picture.at<Vec3b>(v, h) = (uchar)B, (uchar)G, (uchar)R -- or the doubles before that store
__device__ __forceinline__ void store_colour(const RenderArgs &A, double *rgb, uint64_t q, Col c, bool stream_out)
{
  if (A.out_bgr8) {
    unsigned char *o = (unsigned char *)rgb + 3 * q;
    o[0] = (unsigned char)(int)c.b; o[1] = (unsigned char)(int)c.g; o[2] = (unsigned char)(int)c.r;
  } else if (stream_out) {
    __builtin_nontemporal_store(c.r, &rgb[3 * q]); __builtin_nontemporal_store(c.g, &rgb[3 * q + 1]); __builtin_nontemporal_store(c.b, &rgb[3 * q + 2]);
  } else { rgb[3 * q] = c.r; rgb[3 * q + 1] = c.g; rgb[3 * q + 2] = c.b; }
}

// one thread per wave of k_primary: heavy iff one of the tile's corner rays meets the root box
__global__ void __launch_bounds__(256) k_classify(DevScene sc, RenderArgs A, unsigned int n_waves, unsigned char *flags)
{
  const unsigned int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= n_waves) return;
  unsigned int s_local, tx, ty; tile_of_wave(A, w, s_local, tx, ty);
  bool heavy = false;
  if (tx < A.tiles_x && ty < A.tiles_y) {
    for (int c = 0; c < 4; c++) {
      const unsigned int h = tx * 8 + ((c & 1) ? 7 : 0), v = A.v_lo + ty * 8 + ((c & 2) ? 7 : 0);
      V3 cur = A.cam.start - A.cam.left * (double(h) / double(A.cam.res)) - A.cam.up * (double(v) / double(A.cam.res));
      heavy = heavy || ray_meets_box(A.cam.view_pt, unitize(cur - A.cam.view_pt), sc.rbmin, sc.rbmax);
    }
  }
  flags[w] = heavy ? 1 : 0;
}

// stable partition of the wave indices by flag, 1024 indices per block, two small launches:
//   k_order_count  : heavy entries per block
//   k_order        : every block sums the counts of the blocks before it (a few dozen numbers), then places its own
//                    entries -- heavy ones from its heavy base, light ones after ALL heavy ones from its light base
__global__ void __launch_bounds__(1024) k_order_count(const unsigned char *flags, unsigned int n, unsigned int *counts)
{
  __shared__ unsigned int s_w[16];
  const unsigned int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const unsigned int i = blockIdx.x * 1024 + t;
  const unsigned long long mask = __ballot(i < n && flags[i]);
  if (lane == 0) s_w[w] = (unsigned int)__popcll(mask);
  __syncthreads();
  if (t == 0) { unsigned int c = 0; for (int k = 0; k < 16; k++) c += s_w[k]; counts[blockIdx.x] = c; }
}
__global__ void __launch_bounds__(1024) k_order(const unsigned char *flags, unsigned int n, const unsigned int *counts, unsigned int *order)
{
  __shared__ unsigned int s_w[16];
  __shared__ unsigned int s_base[2];
  const unsigned int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (w == 0) {                                   // heavy entries before this block, and in all blocks
    unsigned int before = 0, total = 0;
    for (unsigned int b = lane; b < gridDim.x; b += 64) { const unsigned int c = counts[b]; total += c; if (b < blockIdx.x) before += c; }
    for (int off = 32; off > 0; off >>= 1) { before += __shfl_down(before, off, 64); total += __shfl_down(total, off, 64); }
    if (lane == 0) { s_base[0] = before; s_base[1] = total + (blockIdx.x * 1024u - before); }   // light base: after all heavy ones
  }
  const unsigned int i = blockIdx.x * 1024 + t;
  const bool valid = i < n;
  const bool f = valid && flags[i];
  const unsigned long long mask = __ballot(f);
  if (lane == 0) s_w[w] = (unsigned int)__popcll(mask);
  __syncthreads();
  unsigned int before_w = 0;
  for (unsigned int k = 0; k < w; k++) before_w += s_w[k];
  const unsigned int hrank = before_w + (unsigned int)__popcll(mask & below);     // heavy entries of this block before me
  if (valid) order[f ? (s_base[0] + hrank) : (s_base[1] + (t - hrank))] = i;
}

// level 0: camera / lens ray generation (Camera.h:65-67, 189-190) + closest hit, one 8x8 pixel tile per wave
template <bool STATS, bool PARAM>
// gen_only: the rays are made and stored, their closest hits are found afterwards (closest_deferred, rt_defer.inc)
__global__ void __launch_bounds__(RT_BLOCK, PARAM ? RT_PARAM_MINWAVES : RT_MINWAVES) k_primary(DevScene sc, RenderArgs A, PathNode *nodes, const unsigned int *order,
                                                                    unsigned char *shadow_flags, Stats *gstats, int gen_only)
{
  __shared__ WaveLds s_lds[RT_WAVES];
  WaveLds *stack = &s_lds[threadIdx.x >> 6];
  const unsigned int slot = xcd_remap(blockIdx.x, gridDim.x) * RT_WAVES + (threadIdx.x >> 6);
  const unsigned int wave = order ? (unsigned int)uni((int)order[slot]) : slot;
  const unsigned int lane = threadIdx.x & 63;
  // 8x8-pixel tile per wave; tiles are numbered super-tile by super-tile (16x8 tiles = 128x64 pixels =
  // 128 waves = one XCD chunk of blocks), so that an XCD's L2 sees a compact 2-D patch of the image
  unsigned int s_local, tx, ty; tile_of_wave(A, wave, s_local, tx, ty);
  const unsigned int h = tx * 8 + (lane & 7);
  const unsigned int v = A.v_lo + ty * 8 + (lane >> 3);
  const uint64_t p = (uint64_t)v * A.cam.W + h;
  const bool active = (s_local < A.n_samples) && (tx < A.tiles_x) && (ty < A.tiles_y) && (h < (unsigned)A.cam.W) && (v < (unsigned)A.cam.H) &&
                      (p >= A.pixel_first) && (p < A.pixel_first + A.pixel_count);
  Stats st{};
#ifdef RT_STAMPS
  stamp(wave, 0, 1);
#endif
  const uint32_t sample = A.sample_first + s_local;
  // curPt = startPt - left*(h/res) - up*(v/res), Camera.h:65
  V3 cur = A.cam.start - A.cam.left * (double(h) / double(A.cam.res)) - A.cam.up * (double(v) / double(A.cam.res));
  V3 o = A.cam.view_pt;
  if (A.focusing) {   // Camera.h:189: lens point, draws in x,y,z order
    double jx = ref_random(A.lens_radius, 1000, A.seed, (uint32_t)p, sample, 0, 0, 0, 0);
    double jy = ref_random(A.lens_radius, 1000, A.seed, (uint32_t)p, sample, 0, 0, 0, 1);
    double jz = ref_random(A.lens_radius, 1000, A.seed, (uint32_t)p, sample, 0, 0, 0, 2);
    o = mk(o.x + jx, o.y + jy, o.z + jz);
  }
  V3 d = unitize(unitize(cur - o));     // Camera.h:67 + CRay ctor (Geometry.h:253-254)
  RayState r; ray_setup(r, o, d, sc.coord_scale);
  if (STATS && active && !gen_only) st.rays++;
  if (!gen_only) trace_closest<STATS, PARAM>(sc, r, active, stack, st);
  if (active) {
    PathNode &nd = nodes[(uint64_t)wave * 64 + lane];     // level 0 is stored in tile order (see node_of_pixel)
    nd.o[0] = o.x; nd.o[1] = o.y; nd.o[2] = o.z; nd.d[0] = d.x; nd.d[1] = d.y; nd.d[2] = d.z;
    nd.t = r.best_t; nd.weight = 1.0; nd.id = r.best_id;
    nd.pixel = (uint32_t)p; nd.sample = sample; nd.path = 1;
    nd.child_refr = -1; nd.child_refl = -1; nd.w_refr = 0; nd.w_refl = 0;
    if (PARAM) { nd.col[0] = r.best_u; nd.col[1] = r.best_v; }      // (u, v) of the hit, until k_shade replaces them by the colour
  }
#ifdef RT_STAMPS
  stamp(wave, 1, 1);
#endif
  if (shadow_flags) {
    // will this wave's shadow pass meet the tree?  (first light only: it orders work, nothing else)
    bool heavy = false;
    if (active && r.best_id >= 0 && sc.n_lights > 0) {
      const DLight &L0 = sc.lights[0];
      const V3 pt = o + d * r.best_t, lp = mk(L0.p[0], L0.p[1], L0.p[2]);
      heavy = ray_meets_box(pt, unitize(L0.type == 101 ? neg(lp) : lp - pt), sc.rbmin, sc.rbmax);
    }
    const unsigned long long hm = __ballot(heavy);
    if (lane == 0) shadow_flags[wave] = hm ? 1 : 0;
  }
  if (!active) {
    PathNode &nd = nodes[(uint64_t)wave * 64 + lane];     // padding lane of an edge tile: a miss nobody reads
    nd.id = -1; nd.t = kMiss; nd.weight = 0; nd.pixel = 0; nd.sample = 0; nd.path = 1;
    nd.child_refr = -1; nd.child_refl = -1; nd.w_refr = 0; nd.w_refl = 0;
    for (int k = 0; k < 3; k++) { nd.o[k] = 0; nd.d[k] = (k == 0); }
  }
  if (STATS) stats_flush(gstats, st);
}

// A depth-1 frame in ONE launch (Camera.h:53-75 with depth == 1): camera ray + closest hit as in k_primary, the
// luminosities as in k_shadow, the local illumination, the frame.  Nothing goes through a level array, and the
// low-occupancy tail of the primary pass disappears into the shadow walks of the same waves.
// BATCH: the tiles of several frames in one launch (rt_hip_render_whitted_batch_dev): a wave's frame is what tile_of_wave calls its sample.
template <bool STATS, bool PARAM, bool ONE = false, bool BATCH = false>
__global__ void __launch_bounds__(RT_BLOCK, PARAM ? RT_PARAM_MINWAVES : RT_MINWAVES) k_frame_direct(DevScene sc, RenderArgs A, const unsigned int *order, double *lum,
                                                                          RenderCounters *cnt, Stats *gstats, double *rgb, int32_t *prim_id)
{
  __shared__ WaveLds s_lds[RT_WAVES];
  WaveLds *stack = &s_lds[threadIdx.x >> 6];
  const unsigned int slot = xcd_remap(blockIdx.x, gridDim.x) * RT_WAVES + (threadIdx.x >> 6);
  const unsigned int wave = order ? (unsigned int)uni((int)order[slot]) : slot;
  const unsigned int lane = threadIdx.x & 63;
  if (A.wave_cost && lane == 0) stack->t_start = __builtin_readcyclecounter();     // kept in LDS: no register lives across the walks for it
  unsigned int s_local, tx, ty; tile_of_wave<BATCH>(A, wave, s_local, tx, ty);
  const unsigned int h = tx * 8 + (lane & 7);
  const unsigned int v = A.v_lo + ty * 8 + (lane >> 3);
  const uint64_t p = (uint64_t)v * A.cam.W + h;
  const bool active = (s_local < A.n_samples) && (tx < A.tiles_x) && (ty < A.tiles_y) && (h < (unsigned)A.cam.W) && (v < (unsigned)A.cam.H) &&
                      (p >= A.pixel_first) && (p < A.pixel_first + A.pixel_count);
  Stats st{};
  // Camera.h:65-67; the direction waits in LDS across the walks (not in registers: they are what the walks are short of)
  auto camera_ray = [&](V3 &o, V3 &d) {
    unsigned int s2, tx2, ty2; tile_of_wave<BATCH>(A, wave, s2, tx2, ty2);
    const unsigned int h = tx2 * 8 + (lane & 7), v = A.v_lo + ty2 * 8 + (lane >> 3);
    const DevCamera &cam = BATCH ? A.batch_cams[uni((int)(s2 < A.n_samples ? s2 : 0u))] : A.cam;
    V3 cur = cam.start - cam.left * (double(h) / double(cam.res)) - cam.up * (double(v) / double(cam.res));
    o = cam.view_pt;
    d = unitize(unitize(cur - o));
  };
  int id; double t, hu = kInfty, hv = kInfty; V3 pt;
  {
    V3 o, d; camera_ray(o, d);
    stack->cam_d[lane][0] = d.x; stack->cam_d[lane][1] = d.y; stack->cam_d[lane][2] = d.z;
    RayState r; ray_setup(r, o, d, sc.coord_scale);
    if (STATS && active) st.rays++;
    trace_closest<STATS, PARAM>(sc, r, active, stack, st);
    id = r.best_id; t = r.best_t;
    if (PARAM) { hu = r.best_u; hv = r.best_v; }
    pt = o + d * t;
  }
  const bool hit = active && id >= 0;
  if (!hit) pt = mk(0, 0, 0);
  // what lives across the light walks is pt, id and the pixel number; the rest is made again where it is used
  const uint32_t pix = (uint32_t)p;
  double l_one = 0;                           // ONE: the one luminosity stays in a register
  unsigned long long my_proved = 0;
  const int nl = ONE ? 1 : sc.n_lights;      // ONE: one light of one sample, so neither loop exists and nothing is hoisted across the walk
  for (int li = 0; li < nl; li++) {
    const DLight &L = sc.lights[li];
    RngKey key; key.seed = A.seed; key.pixel = pix; key.sample = A.sample_first + (BATCH ? 0u : s_local); key.path = 1;
    double l = light_luminosity<STATS, PARAM, ONE>(sc, L, li, pt, hit, key, stack, st, ONE ? nullptr : &my_proved);
    if (ONE) l_one = l;
    else if (hit) lum[((size_t)wave * 64 + lane) * nl + li] = l;
  }
  if (active) {
    Col c = mkc(A.bg[0], A.bg[1], A.bg[2]);
    if (hit) {
      V3 N; const V3 d = mk(stack->cam_d[lane][0], stack->cam_d[lane][1], stack->cam_d[lane][2]);
      c = local_illumination<PARAM>(sc, d, id, pt, ONE ? &l_one : lum + ((size_t)wave * 64 + lane) * nl, N, hu, hv);
    }
    // the frame is written once and read by nobody on this GPU: streamed past the L2 lines that hold the hierarchy
    const uint64_t q = out_slot(A, pix % (unsigned)A.cam.W, pix / (unsigned)A.cam.W);
    if (BATCH) {
      unsigned int s2, tx2, ty2; tile_of_wave<BATCH>(A, wave, s2, tx2, ty2);
      store_colour(A, (double *)((unsigned char *)rgb + (uint64_t)s2 * A.batch_rgb_pitch), q, c, true);
      if (prim_id) __builtin_nontemporal_store(id, &prim_id[(uint64_t)s2 * A.batch_id_pitch + q]);
    } else {
    store_colour(A, rgb, q, c, true);
    if (prim_id) __builtin_nontemporal_store(id, &prim_id[q]);
    }
  }
  // shadow rays of this wave: the count per shaded point does not depend on the walks
  unsigned long long per_hit = 0;
  for (int li = 0; li < sc.n_lights; li++) {
    const DLight &L = sc.lights[li];
    per_hit += (L.type == 102) ? (unsigned long long)L.sample_num * L.sample_num : 1ull;
  }
  const unsigned long long my_shadow = per_hit * (unsigned long long)__popcll(__ballot(hit));
  if (lane == 0 && my_shadow) atomicAdd(&cnt->shadow_rays, my_shadow);
  if (!ONE) {
    if (__ballot(my_proved != 0ull) != 0ull) {
      for (int off = 32; off > 0; off >>= 1) my_proved += __shfl_down(my_proved, off, 64);
      if (lane == 0) atomicAdd(&cnt->shadow_rays_proved, my_proved);
    }
  }
  if (STATS) stats_flush(gstats, st);
  if (A.wave_cost && lane == 0) {
    const unsigned long long dt = __builtin_readcyclecounter() - stack->t_start;
    A.wave_cost[wave] = dt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned int)dt;
  }
}

// index of the level-0 node of raster pixel p (inverse of k_primary's tile numbering)
__device__ __forceinline__ uint64_t node_of_pixel(const RenderArgs &A, uint64_t p, uint32_t s_local)
{
  const unsigned int h = (unsigned int)(p % A.cam.W), v = (unsigned int)(p / A.cam.W) - A.v_lo;
  const unsigned int tx = h / 8, ty = v / 8;
  const unsigned int sx = (A.tiles_x + 15) / 16;
  const unsigned int st_local = ((ty / 8) * sx + tx / 16 - A.tile_offset) / A.tile_stride, in_st = (ty % 8) * 16 + (tx % 16);
  const uint64_t wave = (uint64_t)s_local * (A.n_st * 128) + (uint64_t)st_local * 128 + in_st;
  return wave * 64 + (v % 8) * 8 + (h % 8);
}

// levels >= 1: closest hit for queued secondary rays
template <bool STATS, bool PARAM>
__global__ void __launch_bounds__(RT_BLOCK, PARAM ? RT_PARAM_MINWAVES : RT_MINWAVES) k_trace_nodes(DevScene sc, PathNode *nodes, unsigned int n_cap, const unsigned int *n_dev, Stats *gstats)
{
  __shared__ WaveLds s_lds[RT_WAVES];
  WaveLds *stack = &s_lds[threadIdx.x >> 6];
  const unsigned int n = level_count(n_dev, n_cap);
  if (blockIdx.x * RT_BLOCK >= n) return;                     // the grid is sized for the bound: blocks beyond the level's real size leave at once
  const unsigned int i = blockIdx.x * RT_BLOCK + threadIdx.x;
  const bool active = i < n;
  Stats st{};
  V3 o = mk(0, 0, 0), d = mk(1, 0, 0);
  if (active) { o = mk(nodes[i].o[0], nodes[i].o[1], nodes[i].o[2]); d = mk(nodes[i].d[0], nodes[i].d[1], nodes[i].d[2]); }
  RayState r; ray_setup(r, o, d, sc.coord_scale);
  if (STATS && active) st.rays++;
  trace_closest<STATS, PARAM>(sc, r, active, stack, st);
  if (active) { nodes[i].t = r.best_t; nodes[i].id = r.best_id; if (PARAM) { nodes[i].col[0] = r.best_u; nodes[i].col[1] = r.best_v; } }
  if (STATS) stats_flush(gstats, st);
}

// Shadow pass: luminosity of every light at every hit node (Scene.h:104-109), one wave-packet
// occlusion walk per light sample.  Kept apart from k_shade so that the walk runs at full occupancy.
template <bool STATS, bool PARAM>
__global__ void __launch_bounds__(RT_BLOCK, PARAM ? RT_PARAM_MINWAVES : RT_MINWAVES) k_shadow(DevScene sc, RenderArgs A, const PathNode *nodes, unsigned int n_cap, const unsigned int *n_dev, double *lum,
                                                      const unsigned int *order, RenderCounters *cnt, Stats *gstats)
{
  __shared__ WaveLds s_lds[RT_WAVES];
  WaveLds *stack = &s_lds[threadIdx.x >> 6];
  const unsigned int n = level_count(n_dev, n_cap);
  if (!order && blockIdx.x * RT_BLOCK >= n) return;           // (with an order list -- level 0 only -- n is the host's exact count)
  const unsigned int slot = (order ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x) * RT_WAVES + (threadIdx.x >> 6);
  const unsigned int i = (order ? (unsigned int)uni((int)order[slot]) : slot) * 64 + (threadIdx.x & 63);
  const bool in_range = i < n;
  Stats st{};
#ifdef RT_STAMPS
  stamp(i >> 6, 0);
#endif
  int id = -1; double t = kMiss; V3 o = mk(0, 0, 0), d = mk(1, 0, 0);
  RngKey key; key.seed = A.seed; key.pixel = 0; key.sample = 0; key.path = 1;
  if (in_range) {
    const PathNode &nd = nodes[i];
    id = nd.id; t = nd.t; o = mk(nd.o[0], nd.o[1], nd.o[2]); d = mk(nd.d[0], nd.d[1], nd.d[2]);
    key.pixel = nd.pixel; key.sample = nd.sample; key.path = nd.path;
  }
  const bool hit = in_range && id >= 0;
  const V3 pt = hit ? (o + d * t) : mk(0, 0, 0);     // ray.beginPt() + ray.direction() * t
  unsigned long long my_shadow = 0, my_proved = 0;
  for (int li = 0; li < sc.n_lights; li++) {
    const DLight &L = sc.lights[li];
    double l = light_luminosity<STATS, PARAM>(sc, L, li, pt, hit, key, stack, st, &my_proved);
    if (hit) { lum[(size_t)i * sc.n_lights + li] = l; my_shadow += (L.type == 102) ? (unsigned long long)L.sample_num * L.sample_num : 1ull; }
  }
  // ray census (always on: it is what rt_hip_last_ray_count reports)
  for (int off = 32; off > 0; off >>= 1) my_shadow += __shfl_down(my_shadow, off, 64);
  if ((threadIdx.x & 63) == 0 && my_shadow) atomicAdd(&cnt->shadow_rays, my_shadow);
  if (__ballot(my_proved != 0ull) != 0ull) {
    for (int off = 32; off > 0; off >>= 1) my_proved += __shfl_down(my_proved, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt->shadow_rays_proved, my_proved);
  }
#ifdef RT_STAMPS
  stamp(i >> 6, 1);
#endif
  if (STATS) stats_flush(gstats, st);
}

// RayTracer (RayTracer.h:8-55) for one level of the ray tree: local illumination from the luminosities of
// the shadow pass and generation of the refracted / reflected child rays into the next level's queue.
// direct_rgb != NULL (level 0 of a depth-1 frame, one sample per pixel): the colour is final, so it goes straight
// to the frame (Camera.h:69) and the separate resolve pass over the level array is skipped.
template <bool PARAM>
__global__ void __launch_bounds__(RT_BLOCK) k_shade(DevScene sc, RenderArgs A, PathNode *nodes, unsigned int n_cap, const unsigned int *n_dev, int depth_left,
                                                     const double *lum, PathNode *next, unsigned int next_cap, RenderCounters *cnt, int next_level,
                                                     double *direct_rgb, int32_t *direct_id, double *uv_out)
{
  const unsigned int n = level_count(n_dev, n_cap);
  if (blockIdx.x * RT_BLOCK >= n) return;
  const unsigned int i = blockIdx.x * RT_BLOCK + threadIdx.x;
  const bool in_range = i < n;
  PathNode nd;
  if (in_range) nd = nodes[i];
  else { nd.id = -1; nd.t = kMiss; nd.weight = 0; nd.pixel = 0; nd.sample = 0; nd.path = 1; for (int k = 0; k < 3; k++) { nd.o[k] = 0; nd.d[k] = (k == 0); } }
  const bool hit = in_range && nd.id >= 0;
  const V3 o = mk(nd.o[0], nd.o[1], nd.o[2]), d = mk(nd.d[0], nd.d[1], nd.d[2]);
  const V3 pt = hit ? (o + d * nd.t) : mk(0, 0, 0);     // ray.beginPt() + ray.direction() * t
  V3 N = mk(0, 0, 1);
  Col c = mkc(A.bg[0], A.bg[1], A.bg[2]);
  int child_refr = -1, child_refl = -1; double w_refr = 0, w_refl = 0;
  bool want_refr = false, want_refl = false; V3 dir_refr = mk(0, 0, 0), dir_refl = mk(0, 0, 0);
  if (hit) {
    c = local_illumination<PARAM>(sc, d, nd.id, pt, lum + (size_t)i * sc.n_lights, N, nd.col[0], nd.col[1]);   // col[0..1] = (u, v) until here
    if (PARAM && uv_out) { uv_out[2 * (size_t)i] = nd.col[0]; uv_out[2 * (size_t)i + 1] = nd.col[1]; }
    if (depth_left > 1) {
      const DMaterial M = sc.materials[sc.obj_mat[nd.id]];
      double refrW = M.refrW, reflW = M.reflW;
      if (!(M.reflW == 0 || M.refrW == 0)) {             // Fresnel, Object.h:130-131, 159-160
        double R = fresnel_R(N, d, M.RI);
        reflW = R; refrW = 1 - R;
      }
      bool total_reflect = false;
      if (refrW != 0) {
        V3 t = refract_dir(N, d, M.RI);
        if (!is_infty_pt(t)) { w_refr = refrW; dir_refr = t; if (nd.weight * refrW < A.weight_threshold) child_refr = -2; else want_refr = true; }
        else total_reflect = true;
      }
      if (reflW != 0) {
        dir_refl = reflect_dir(N, d);
        w_refl = total_reflect ? (reflW + refrW) : reflW;
        if (nd.weight * w_refl < A.weight_threshold) child_refl = -2; else want_refl = true;
      }
    }
  }
  // wave-aggregated append of the child rays to the next level
  const unsigned int lane = threadIdx.x & 63;
  const unsigned long long m1 = __ballot(want_refr), m2 = __ballot(want_refl);
  const unsigned int total = __popcll(m1) + __popcll(m2);
  unsigned int base = 0;
  if (total) {
    if (lane == 0) { base = atomicAdd(&cnt->level_n[next_level], total); atomicAdd(&cnt->rays, (unsigned long long)total); }   // census: every child is one closest-hit query
    base = __builtin_amdgcn_readfirstlane(base);
  }
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  if (want_refr) {
    unsigned int k = base + __popcll(m1 & below);
    if (k < next_cap) {
      PathNode &q = next[k];
      q.o[0] = pt.x; q.o[1] = pt.y; q.o[2] = pt.z; q.d[0] = dir_refr.x; q.d[1] = dir_refr.y; q.d[2] = dir_refr.z;
      q.weight = nd.weight * w_refr; q.pixel = nd.pixel; q.sample = nd.sample; q.path = nd.path * 2;
      q.id = -1; q.t = kMiss; q.child_refr = -1; q.child_refl = -1; q.w_refr = 0; q.w_refl = 0;
      child_refr = (int)k;
    }
  }
  if (want_refl) {
    unsigned int k = base + __popcll(m1) + __popcll(m2 & below);
    if (k < next_cap) {
      PathNode &q = next[k];
      q.o[0] = pt.x; q.o[1] = pt.y; q.o[2] = pt.z; q.d[0] = dir_refl.x; q.d[1] = dir_refl.y; q.d[2] = dir_refl.z;
      q.weight = nd.weight * w_refl; q.pixel = nd.pixel; q.sample = nd.sample; q.path = nd.path * 2 + 1;
      q.id = -1; q.t = kMiss; q.child_refr = -1; q.child_refl = -1; q.w_refr = 0; q.w_refl = 0;
      child_refl = (int)k;
    }
  }
  if (direct_rgb) {
    if (in_range && nd.weight > 0.0) {          // padding lanes of edge tiles carry weight 0
      const uint64_t q = out_slot(A, nd.pixel % (unsigned)A.cam.W, nd.pixel / (unsigned)A.cam.W);
      store_colour(A, direct_rgb, q, c, false);
      if (direct_id) direct_id[q] = nd.id;
    }
    return;
  }
  if (in_range) {
    PathNode &w = nodes[i];
    w.col[0] = c.r; w.col[1] = c.g; w.col[2] = c.b;
    w.w_refr = w_refr; w.w_refl = w_refl; w.child_refr = child_refr; w.child_refl = child_refl;
  }
}

// bottom-up composition with the reference's clamped colour arithmetic (RayTracer.h:33, 45, 50)
__global__ void __launch_bounds__(RT_BLOCK) k_combine(PathNode *nodes, unsigned int n_cap, const unsigned int *n_dev, const PathNode *next, RenderArgs A)
{
  const unsigned int i = blockIdx.x * RT_BLOCK + threadIdx.x;
  if (i >= level_count(n_dev, n_cap)) return;
  PathNode &w = nodes[i];
  Col c = mkc(w.col[0], w.col[1], w.col[2]);
  const Col bg = mkc(A.bg[0], A.bg[1], A.bg[2]);
  if (w.child_refr >= 0) { const PathNode &q = next[w.child_refr]; c = cadd(c, cmul(mkc(q.col[0], q.col[1], q.col[2]), w.w_refr)); }
  else if (w.child_refr == -2) c = cadd(c, cmul(bg, w.w_refr));
  if (w.child_refl >= 0) { const PathNode &q = next[w.child_refl]; c = cadd(c, cmul(mkc(q.col[0], q.col[1], q.col[2]), w.w_refl)); }
  else if (w.child_refl == -2) c = cadd(c, cmul(bg, w.w_refl));
  w.col[0] = c.r; w.col[1] = c.g; w.col[2] = c.b;
}

// Camera.h:69 up to (out_bgr8: including) the uchar store / Camera.h:191 accumulation over the samples of this launch
__global__ void __launch_bounds__(RT_BLOCK) k_resolve(const PathNode *nodes, RenderArgs A, double spp, int accumulate,
                                                      double *rgb, int32_t *prim_id)
{
  const uint64_t q = (uint64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
  if (q >= A.n_out) return;
  uint64_t p = A.pixel_first + q;                   // raster pixel of output entry q
  if (A.tile_stride > 1) {
    const unsigned int sx = (A.tiles_x + 15) / 16;
    const unsigned int st_i = A.tile_offset + (unsigned int)(q / 8192) * A.tile_stride, r = (unsigned int)(q % 8192);
    const unsigned int h = (st_i % sx) * 128 + r % 128, v = (st_i / sx) * 64 + r / 128;
    if (h >= (unsigned)A.cam.W || v >= (unsigned)A.cam.H) return;       // padding of an edge super-tile
    p = (uint64_t)v * A.cam.W + h;
  }
  if (!A.focusing) {
    const PathNode &w = nodes[node_of_pixel(A, p, 0)];
    store_colour(A, rgb, q, mkc(w.col[0], w.col[1], w.col[2]), false);
    if (prim_id) prim_id[q] = w.id;
    return;
  }
  double r = accumulate ? rgb[3 * q] : 0.0, g = accumulate ? rgb[3 * q + 1] : 0.0, b = accumulate ? rgb[3 * q + 2] : 0.0;
  for (uint32_t s = 0; s < A.n_samples; s++) {
    const PathNode &w = nodes[node_of_pixel(A, p, s)];
    Col c = cdiv(mkc(w.col[0], w.col[1], w.col[2]), spp);   // RayTracer(...) / sampleTimes
    r += c.r; g += c.g; b += c.b;
  }
  rgb[3 * q] = r; rgb[3 * q + 1] = g; rgb[3 * q + 2] = b;
}

// rt_hip_tiles_to_frame_dev: the all-gathered tile shards (rank-major: tile_stride blocks of per_rank super-tiles of
// 128 x 64 entries) back into the raster frame
__global__ void __launch_bounds__(256) k_tiles_to_frame(const unsigned char *tiles, unsigned int W, unsigned int H, unsigned int stride, uint64_t rank_pitch,
                                                        unsigned int elem, unsigned char *frame, uint64_t tiles_frame_pitch, uint64_t frame_pitch)
{
  const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (uint64_t)W * H) return;
  tiles += (uint64_t)blockIdx.y * tiles_frame_pitch; frame += (uint64_t)blockIdx.y * frame_pitch;       // blockIdx.y: frame of a batch
  const unsigned int h = (unsigned int)(p % W), v = (unsigned int)(p / W);
  const unsigned int sx = ((W + 7) / 8 + 15) / 16;
  const unsigned int st_i = (v / 64) * sx + h / 128;
  const unsigned char *blk = tiles + (uint64_t)(st_i % stride) * rank_pitch;          // the shard of the rank that rendered this super-tile
  const uint64_t src = (uint64_t)(st_i / stride) * 8192 + (v % 64) * 128 + (h % 128);
  if (elem == 24) { const double *a = (const double *)blk + 3 * src; double *o = (double *)frame + 3 * p; o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
  else if (elem == 4) ((unsigned int *)frame)[p] = ((const unsigned int *)blk)[src];
  else for (unsigned int k = 0; k < elem; k++) frame[p * elem + k] = blk[src * elem + k];
}

// ================================================================================================
// workspace of the allocation-free entry point (rt_hip_render_whitted_noalloc, below)
// ================================================================================================
// One fixed array per GPU, resident from the moment the library's code object is loaded on that GPU until the process
// ends.  In host code the NAME of the array is only a stand-in (it is not a device address): the address on each device
// is asked for once, with that device current, in rt_hip_context_create, and kept per device in g_scratch_ptr.
#define RT_NOALLOC_SCRATCH_BYTES (32u << 20)
#define RT_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
static unsigned char *g_scratch_ptr[RT_MAX_DEVICES] = {nullptr};
static std::mutex g_scratch_mutex;
// called with `device` current
static int scratch_setup(int device)
{
  if (device < 0 || device >= RT_MAX_DEVICES) return RT_HIP_OK;      // such a device has no allocation-free entry point (it says so when called)
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[device]) return RT_HIP_OK;
  void *ptr = nullptr;
  HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
  if (!ptr) return fail(RT_HIP_EHIP, "hipGetSymbolAddress(scratch) returned NULL");
  g_scratch_ptr[device] = (unsigned char *)ptr;
  return RT_HIP_OK;
}

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" int rt_hip_device_count(int *n)
{
  if (!n) return fail(RT_HIP_EINVAL, "n is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) { *n = 0; return fail(RT_HIP_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  *n = c;
  return RT_HIP_OK;
}

extern "C" int rt_hip_context_create(int device_id, rt_hip_ctx **out)
{
  if (!out) return fail(RT_HIP_EINVAL, "out is NULL");
  *out = nullptr;
  int c = 0;
  HIP_TRY(hipGetDeviceCount(&c));
  if (c <= 0) return fail(RT_HIP_EHIP, "no HIP device: this library has no CPU fallback");
  if (device_id < 0 || device_id >= c) return fail(RT_HIP_EINVAL, "device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  { int rc = scratch_setup(device_id); if (rc) return rc; }
  rt_hip_ctx *ctx = new rt_hip_ctx();
  ctx->device = device_id;
  hipError_t e = hipMalloc((void **)&ctx->d_stats, sizeof(Stats));
  if (e != hipSuccess) { delete ctx; return fail(RT_HIP_ENOMEM, std::string("hipMalloc stats: ") + hipGetErrorString(e)); }
  e = hipMemset(ctx->d_stats, 0, sizeof(Stats));
  if (e != hipSuccess) { (void)hipFree(ctx->d_stats); delete ctx; return fail(RT_HIP_EHIP, std::string("hipMemset: ") + hipGetErrorString(e)); }
  *out = ctx;
  return RT_HIP_OK;
}

extern "C" int rt_hip_context_destroy(rt_hip_ctx *ctx)
{
  ctx_release(ctx);
  return RT_HIP_OK;
}

extern "C" int rt_hip_get_stats(rt_hip_ctx *ctx, rt_hip_stats *out, int reset)
{
  if (!ctx || !out) return fail(RT_HIP_EINVAL, "NULL argument");
  HIP_TRY(hipSetDevice(ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  Stats s;
  HIP_TRY(hipMemcpy(&s, ctx->d_stats, sizeof s, hipMemcpyDeviceToHost));
  out->rays = s.rays; out->shadow_rays = s.shadow_rays; out->node_visits = s.node_visits; out->prim_tests = s.prim_tests;
  out->exact_tests = s.exact_tests; out->wave_node_visits = s.wave_nodes; out->wave_prim_tests = s.wave_prims; out->reserved = s.pad;
  if (reset) HIP_TRY(hipMemset(ctx->d_stats, 0, sizeof(Stats)));
  return RT_HIP_OK;
}

static void scene_retain(rt_hip_scene *s) { if (s) s->refs.fetch_add(1); }
extern "C" int rt_hip_scene_destroy(rt_hip_scene *s)
{
  if (!s || s->refs.fetch_sub(1) != 1) return RT_HIP_OK;      // a photon map or god ray still renders with it
  if (s->ctx && hipSetDevice(s->ctx->device) != hipSuccess) (void)hipGetLastError();
  s->geo.reset();
  for (auto &e : s->ev_pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  for (auto &b : s->levels) b.release();
  for (auto &b : s->uv) b.release();
  s->counter.release(); s->staging.release(); s->lum.release(); s->wave_cost.release(); s->order.release(); s->aa_maps.release(); s->aa_edges.release(); s->aa_mask.release();
  for (DevBuf *b : {&s->gq_pairs, &s->gq_counts, &s->gq_res, &s->gq_rays, &s->gq_uvh, &s->gq_seeds, &s->gq_seed_mask, &s->gq_occ, &s->batch_cams}) b->release();
  if (s->batch_pinned) { (void)hipHostFree(s->batch_pinned); for (hipEvent_t e : s->batch_ev) if (e) (void)hipEventDestroy(e); }
  ctx_release(s->ctx);
  delete s;
  return RT_HIP_OK;
}

extern "C" int rt_hip_scene_create(rt_hip_ctx *ctx, const rt_hip_scene_desc *d, uint32_t leaf_refine, rt_hip_scene **out)
{
  if (!ctx || !d || !out) return fail(RT_HIP_EINVAL, "NULL argument");
  *out = nullptr;
  if (d->struct_size != sizeof(rt_hip_scene_desc)) return fail(RT_HIP_EINVAL, "rt_hip_scene_desc.struct_size mismatch");
  if (d->n_objects == 0) return fail(RT_HIP_EINVAL, "scene has no objects");
  if (!d->obj_type || !d->obj_index || !d->obj_material || !d->obj_colour) return fail(RT_HIP_EINVAL, "object arrays missing");
  if ((d->n_triangles && !d->triangles) || (d->n_planes && !d->planes) || (d->n_spheres && !d->spheres) ||
      (d->n_circles && !d->circles) || (d->n_lights && !d->lights) || !d->materials || d->n_materials == 0)
    return fail(RT_HIP_EINVAL, "type arrays missing");
  if ((d->n_beziers && !d->beziers) || (d->n_rotationals && !d->rotationals)) return fail(RT_HIP_EINVAL, "type arrays missing");
  for (uint32_t k = 0; k < d->n_rotationals; k++) {
    const rt_hip_rotational &z = d->rotationals[k];
    if (z.n_profile < 2 || z.n_profile > 8) return fail(RT_HIP_EINVAL, "rotational object: 2..8 profile control points");
    if (z.u_sample_num < 1 || z.u_sample_num > 4096) return fail(RT_HIP_EINVAL, "rotational object: u_sample_num out of range [1, 4096]");
  }
  if (d->n_lights > 16) return fail(RT_HIP_EUNSUPPORTED, "more than 16 lights");
  if (d->n_textures) {
    if (!d->textures || !d->obj_texture || !d->obj_bump_texture) return fail(RT_HIP_EINVAL, "texture arrays missing");
    for (uint32_t k = 0; k < d->n_textures; k++)
      if (!d->textures[k].bgr || d->textures[k].rows == 0 || d->textures[k].cols == 0 || d->textures[k].rows > 32768 || d->textures[k].cols > 32768)
        return fail(RT_HIP_EINVAL, "texture " + std::to_string(k) + ": empty or larger than 32768 x 32768");
    for (uint32_t i = 0; i < d->n_objects; i++) {
      const int32_t t = d->obj_texture[i], b = d->obj_bump_texture[i];
      if (t < -1 || t >= (int32_t)d->n_textures || b < -1 || b >= (int32_t)d->n_textures) return fail(RT_HIP_EINVAL, "object texture index out of range");
      if (d->obj_type[i] == RT_HIP_INFTY_PLANE && (t >= 0 || b >= 0)) {
        if (!d->obj_texture_scale) return fail(RT_HIP_EINVAL, "obj_texture_scale missing for a textured plane");
        if (b >= 0 && !(d->obj_texture_scale[2 * i + 1] != 0)) return fail(RT_HIP_EINVAL, "bumpTextureScale of a plane must not be 0");
      }
    }
  }
  for (uint32_t i = 0; i < d->n_objects; i++) {
    uint32_t k = d->obj_index[i], lim = 0;
    switch (d->obj_type[i]) {
      case RT_HIP_TRIANGLE: lim = d->n_triangles; break;
      case RT_HIP_INFTY_PLANE: lim = d->n_planes; break;
      case RT_HIP_SPHERE: lim = d->n_spheres; break;
      case RT_HIP_CIRCLE: lim = d->n_circles; break;
      case RT_HIP_BEZIER: lim = d->n_beziers; break;
      case RT_HIP_ROTATIONAL: lim = d->n_rotationals; break;
      default: return fail(RT_HIP_EUNSUPPORTED, "object type " + std::to_string((int)d->obj_type[i]) + " is not supported");
    }
    if (k >= lim) return fail(RT_HIP_EINVAL, "obj_index out of range at object " + std::to_string(i));
    if (d->obj_material[i] >= d->n_materials) return fail(RT_HIP_EINVAL, "obj_material out of range");
  }
  for (uint32_t i = 0; i < d->n_lights; i++) {
    const rt_hip_light &L = d->lights[i];
    if (L.type != RT_HIP_POINT_LIGHT && L.type != RT_HIP_PARALLEL_LIGHT && L.type != RT_HIP_AREA_LIGHT)
      return fail(RT_HIP_EINVAL, "bad light type");
    if (L.type == RT_HIP_AREA_LIGHT && (L.sample_num < 1 || L.sample_num > 127)) return fail(RT_HIP_EINVAL, "area light sample_num out of range [1,127]");
  }
  if (d->n_nodes && (!d->nodes || (d->n_refs && !d->refs) || (d->n_root_objects && !d->root_objects)))
    return fail(RT_HIP_EINVAL, "tree arrays missing");
  for (uint32_t i = 0; i < d->n_root_objects; i++)
    if (d->root_objects[i] >= d->n_objects) return fail(RT_HIP_EINVAL, "root_objects out of range");

  HIP_TRY(hipSetDevice(ctx->device));
  rt_hip_scene *s = new rt_hip_scene();
  s->ctx = ctx; ctx_retain(ctx);
  Builder B; B.d = d;
  // rt_hip.h: DEFAULT = the library's own hierarchy with its tuned leaf size; NONE = the reference's tree as described;
  // n = the reference's tree with its leaves split down to n objects; RT_HIP_LEAF_REFINE_OWN | n = own hierarchy, leaf size n
  const bool own_tree = (leaf_refine == RT_HIP_LEAF_REFINE_DEFAULT) || (leaf_refine != RT_HIP_LEAF_REFINE_NONE && (leaf_refine & RT_HIP_LEAF_REFINE_OWN));
  B.leaf_refine = (leaf_refine == RT_HIP_LEAF_REFINE_DEFAULT) ? own_leaf_default(d->n_objects) : (leaf_refine == RT_HIP_LEAF_REFINE_NONE ? 0u : (leaf_refine & ~RT_HIP_LEAF_REFINE_OWN));
  if (own_tree && B.leaf_refine < 2) B.leaf_refine = own_leaf_default(d->n_objects);
  // triangles: m_norm = ((v1 - v0) ^ (v2 - v0)).Unitize(), Object.h:365
  B.tri_exact.resize(d->n_triangles);
  std::vector<TriShade> tri_shade(d->n_triangles);
  for (uint32_t k = 0; k < d->n_triangles; k++) {
    const rt_hip_triangle &t = d->triangles[k];
    V3 v0 = mk(t.v[0][0], t.v[0][1], t.v[0][2]), v1 = mk(t.v[1][0], t.v[1][1], t.v[1][2]), v2 = mk(t.v[2][0], t.v[2][1], t.v[2][2]);
    V3 n = unitize(cross(v1 - v0, v2 - v0));
    TriExact &T = B.tri_exact[k];
    T.v0[0] = v0.x; T.v0[1] = v0.y; T.v0[2] = v0.z; T.v1[0] = v1.x; T.v1[1] = v1.y; T.v1[2] = v1.z;
    T.v2[0] = v2.x; T.v2[1] = v2.y; T.v2[2] = v2.z; T.nrm[0] = n.x; T.nrm[1] = n.y; T.nrm[2] = n.z;
    TriShade &S = tri_shade[k];
    std::memcpy(S.vn, t.vn, sizeof S.vn); S.has_vn = t.has_vn; S.pad = 0;
  }
  B.object_boxes();
  double cs = 0;
  std::vector<int32_t> root_objs;
  ChildRef root_ref{0, 0};
  bool has_tree = d->n_nodes != 0;
  if (has_tree) {
    Builder::Sub root = own_tree ? B.rebuild() : B.walk(0, 0);
    if (!B.err.empty()) { rt_hip_scene_destroy(s); return fail(RT_HIP_EINVAL, "tree: " + B.err); }
    if (root.empty) {
      // a tree without finite content: one empty leaf keeps the "has tree" semantics (Scene.h:87)
      LeafHeader H; std::memset(&H, 0, sizeof H);
      for (int a = 0; a < 3; a++) { H.bmin[a] = 3e38f; H.bmax[a] = -3e38f; }
      size_t off = B.stream.size(); B.stream.resize(off + 64); std::memcpy(&B.stream[off], &H, 64);
      root_ref = ChildRef{-1, (int32_t)(off / 64)};
    } else {
      root_ref = root.ref;
      for (int a = 0; a < 3; a++) { s->dev.rbmin[a] = root.box.mn[a]; s->dev.rbmax[a] = root.box.mx[a]; s->dev.rbf_min[a] = Builder::f32_down(root.box.mn[a]); s->dev.rbf_max[a] = Builder::f32_up(root.box.mx[a]); }
      for (int a = 0; a < 3; a++) cs = std::max(cs, std::max(std::fabs(root.box.mn[a]), std::fabs(root.box.mx[a])));
    }
    // the cover for area_light_all_clear: start from the root box, open the inner node with the largest surface until RT_COVER_BOXES boxes are there
    {
      struct CB { double mn[3], mx[3]; ChildRef ref; };
      std::vector<CB> cover;
      if (!root.empty) {
        CB c0; for (int a = 0; a < 3; a++) { c0.mn[a] = root.box.mn[a]; c0.mx[a] = root.box.mx[a]; } c0.ref = root_ref; cover.push_back(c0);
        while (cover.size() < (size_t)RT_COVER_BOXES) {
          int pick = -1; double best = -1;
          for (size_t k = 0; k < cover.size(); k++) {
            if (cover[k].ref.i0 < 0 || (size_t)cover[k].ref.i0 >= B.pairs.size()) continue;
            const double ex = cover[k].mx[0] - cover[k].mn[0], ey = cover[k].mx[1] - cover[k].mn[1], ez = cover[k].mx[2] - cover[k].mn[2];
            const double area = ex * ey + ey * ez + ez * ex;
            if (area > best) { best = area; pick = (int)k; }
          }
          if (pick < 0) break;
          const DPair P = B.pairs[cover[pick].ref.i0];
          CB l, r; for (int a = 0; a < 3; a++) { l.mn[a] = P.lmin[a]; l.mx[a] = P.lmax[a]; r.mn[a] = P.rmin[a]; r.mx[a] = P.rmax[a]; } l.ref = P.l; r.ref = P.r;
          cover[pick] = l; cover.push_back(r);
        }
      }
      s->dev.n_cover = (int32_t)cover.size();
      for (size_t k = 0; k < cover.size(); k++) for (int a = 0; a < 3; a++) {
        s->dev.cover_min[k][a] = Builder::f32_down(cover[k].mn[a]); s->dev.cover_max[k][a] = Builder::f32_up(cover[k].mx[a]); }
    }
    if (B.pairs.empty()) B.pairs.emplace_back();
    // slack so that the fixed-size direct-to-LDS chunk loads never read past the allocation
    B.stream.resize(B.stream.size() + 2 * RT_CHUNK_BYTES, 0);
    // root->objects: when the reference root was never split it holds ALL objects (BSPTree.h:60); finite
    // ones are already in the root leaf, so only the infinite ones are kept for the per-ray list
    for (uint32_t i = 0; i < d->n_root_objects; i++) if (!B.finite[d->root_objects[i]]) root_objs.push_back((int32_t)d->root_objects[i]);
  } else {
    for (uint32_t i = 0; i < d->n_objects; i++) if (B.finite[i]) for (int a = 0; a < 3; a++)
      cs = std::max(cs, std::max(std::fabs(B.aabb[i].mn[a]), std::fabs(B.aabb[i].mx[a])));
  }
  std::vector<DPlane> planes(d->n_planes); std::vector<DSphere> spheres(d->n_spheres); std::vector<DCircle> circles(d->n_circles);
  for (uint32_t k = 0; k < d->n_planes; k++) { std::memcpy(planes[k].p0, d->planes[k].p0, 24); std::memcpy(planes[k].n, d->planes[k].n, 24); }
  for (uint32_t k = 0; k < d->n_spheres; k++) { std::memcpy(spheres[k].c, d->spheres[k].c, 24); spheres[k].r = d->spheres[k].r; }
  for (uint32_t k = 0; k < d->n_circles; k++) { std::memcpy(circles[k].p0, d->circles[k].p0, 24); std::memcpy(circles[k].n, d->circles[k].n, 24); circles[k].r = d->circles[k].r; }
  std::vector<DBezier> beziers(d->n_beziers); std::vector<DRotational> rots(d->n_rotationals);
  for (uint32_t k = 0; k < d->n_beziers; k++) { std::memcpy(beziers[k].c, d->beziers[k].ctrl, sizeof beziers[k].c); std::memcpy(beziers[k].center, d->beziers[k].center, 24); param_prune_box(beziers[k]); }
  for (uint32_t k = 0; k < d->n_rotationals; k++) {
    const rt_hip_rotational &z = d->rotationals[k]; DRotational &o = rots[k];
    std::memset(&o, 0, sizeof o);
    std::memcpy(o.prof, z.profile, sizeof o.prof); std::memcpy(o.center, z.center, 24);
    o.u_range = z.u_range; o.n_prof = z.n_profile; o.u_sample_num = z.u_sample_num;
    Builder::rot_box(z, o.bmin, o.bmax);
    param_prune_box(o); param_prune_shells(o);
  }
  std::vector<DMaterial> mats(d->n_materials);
  for (uint32_t k = 0; k < d->n_materials; k++) std::memcpy(&mats[k], &d->materials[k], sizeof(DMaterial));
  std::vector<DLight> lights(d->n_lights);
  for (uint32_t k = 0; k < d->n_lights; k++) {
    const rt_hip_light &L = d->lights[k]; DLight &o = lights[k];
    o.type = L.type; o.sample_num = L.sample_num; std::memcpy(o.p, L.pos_or_dir, 24); o.lum = L.luminosity; o.len = L.len;
    o.lum_per_sample = (L.type == RT_HIP_AREA_LIGHT) ? L.luminosity / (L.sample_num * L.sample_num) : 0.0;   // Light.h:107
  }
  s->fan = 1;
  for (uint32_t k = 0; k < d->n_materials; k++) if (d->materials[k].reflW != 0 && d->materials[k].refrW != 0) s->fan = 2;
  s->one_sample = d->n_lights == 1 && (d->lights[0].type != RT_HIP_AREA_LIGHT || d->lights[0].sample_num == 1);
  std::vector<uint8_t> types(d->obj_type, d->obj_type + d->n_objects);
  std::vector<uint32_t> index(d->obj_index, d->obj_index + d->n_objects), omat(d->obj_material, d->obj_material + d->n_objects);
  std::vector<double> colour(d->obj_colour, d->obj_colour + 3 * (size_t)d->n_objects);

  DevScene &D = s->dev;
  int rc = RT_HIP_OK;
  auto up = [&](int r) { if (rc == RT_HIP_OK) rc = r; };
  if (d->n_textures) {
    // cv::Mat images of Object.h:70-73, one allocation each; the per-object tables only exist for textured scenes
    std::vector<DTexture> tex(d->n_textures);
    for (uint32_t k = 0; k < d->n_textures && rc == RT_HIP_OK; k++) {
      const rt_hip_texture &t = d->textures[k];
      std::vector<uint8_t> px(t.bgr, t.bgr + (size_t)t.rows * t.cols * 3);
      up(upload(s, px, &tex[k].bgr));
      tex[k].rows = (int32_t)t.rows; tex[k].cols = (int32_t)t.cols;
    }
    std::vector<int32_t> otex(d->obj_texture, d->obj_texture + d->n_objects), obump(d->obj_bump_texture, d->obj_bump_texture + d->n_objects);
    std::vector<double> oscale(2 * (size_t)d->n_objects, 1.0);
    if (d->obj_texture_scale) oscale.assign(d->obj_texture_scale, d->obj_texture_scale + 2 * (size_t)d->n_objects);
    up(upload(s, tex, &D.textures)); up(upload(s, otex, &D.obj_tex)); up(upload(s, obump, &D.obj_bump)); up(upload(s, oscale, &D.obj_tex_scale));
    if (d->tri_texture_coord && d->n_triangles) {
      std::vector<double> tuv(d->tri_texture_coord, d->tri_texture_coord + 6 * (size_t)d->n_triangles);
      up(upload(s, tuv, &D.tri_uv));
    }
  }
  {
    // the walk's pairs: FP32 boxes rounded outward (the FP64 ones stay the builder's)
    std::vector<DPairF> pf(B.pairs.size());
    for (size_t i = 0; i < pf.size(); i++) {
      const DPair &P = B.pairs[i];
      for (int a = 0; a < 3; a++) {
        pf[i].lmin[a] = Builder::f32_down(P.lmin[a]); pf[i].lmax[a] = Builder::f32_up(P.lmax[a]);
        pf[i].rmin[a] = Builder::f32_down(P.rmin[a]); pf[i].rmax[a] = Builder::f32_up(P.rmax[a]);
      }
      pf[i].l = P.l; pf[i].r = P.r;
    }
    up(upload(s, pf, &D.pairs));
  }
  up(upload(s, B.stream, &D.leaf_stream));
  up(upload(s, root_objs, &D.root_objs)); up(upload(s, B.tri_exact, &D.tri_exact));
  {
    // vertex normals only where a mesh has them (Object.h:448-457): a flat-shaded scene keeps 80 bytes per triangle out of HBM
    // and one scattered line per shaded point out of the caches
    bool any_vn = false;
    for (const TriShade &S : tri_shade) if (S.has_vn) { any_vn = true; break; }
    D.tri_shade = nullptr;
    if (any_vn) up(upload(s, tri_shade, &D.tri_shade));
  }
  up(upload(s, planes, &D.planes)); up(upload(s, spheres, &D.spheres)); up(upload(s, circles, &D.circles));
  up(upload(s, beziers, &D.beziers)); up(upload(s, rots, &D.rotationals));
  D.has_param = (d->n_beziers || d->n_rotationals) ? 1 : 0;
  D.n_rot = (int32_t)d->n_rotationals; D.n_bez = (int32_t)d->n_beziers; D.param_obj = nullptr; D.gq_pairs = nullptr; D.gq_count = nullptr; D.gq_region = 0;
  if (D.has_param) {
    std::vector<int32_t> pobj((size_t)D.n_rot + D.n_bez, -1);
    for (uint32_t id = 0; id < d->n_objects; id++) {
      if (types[id] == 5) pobj[index[id]] = (int32_t)id;
      else if (types[id] == 6) pobj[(size_t)D.n_rot + index[id]] = (int32_t)id;
    }
    up(upload(s, pobj, &D.param_obj));
    s->defer_fits = D.n_rot + D.n_bez <= 64 && d->n_lights < 4096;
    s->max_seeds = 1; s->shadow_samples = 0;
    for (uint32_t k = 0; k < d->n_rotationals; k++) {
      if (d->rotationals[k].u_sample_num > 63) s->defer_fits = false;
      s->max_seeds = std::max(s->max_seeds, (unsigned int)d->rotationals[k].u_sample_num + 1u);
    }
    for (uint32_t k = 0; k < d->n_lights; k++) {
      const unsigned int sn = d->lights[k].type == RT_HIP_AREA_LIGHT ? (unsigned int)d->lights[k].sample_num : 1u;
      if (sn * sn > 128u || sn > 11u) s->defer_fits = false;
      s->shadow_samples += sn * sn;
    }
    for (int32_t v : pobj) if (v < 0) s->defer_fits = false;     // a surface no object uses: keep the in-place form
  }
  up(upload(s, types, &D.obj_type)); up(upload(s, index, &D.obj_index)); up(upload(s, omat, &D.obj_mat));
  up(upload(s, colour, &D.obj_colour)); up(upload(s, mats, &D.materials)); up(upload(s, lights, &D.lights));
  if (rc != RT_HIP_OK) { std::string m = g_err; rt_hip_scene_destroy(s); return fail(rc, m); }
  D.root = root_ref; D.has_tree = has_tree ? 1 : 0;
  D.n_root_objs = (int32_t)root_objs.size(); D.n_objects = (int32_t)d->n_objects; D.n_lights = (int32_t)d->n_lights;
  D.ambient = d->ambient; D.coord_scale = cs;
  { const char *e = std::getenv("RT_HIP_LIT_SKIP"); D.lit_skip = (e && e[0] == '0') ? 0 : 1; }
  s->n_nodes = (uint32_t)B.pairs.size(); s->n_leaves = B.n_leaves; s->n_cull = B.n_recs; s->max_depth = B.max_depth;
  *out = s;
  return RT_HIP_OK;
}

// A second handle on the same scene: the geometry in HBM is shared, the render workspace (level arrays,
// luminosities, work order, counters) is its own, so that the two handles can render at the same time on
// two streams ("frames in flight": the draining tail of one frame's kernels is filled by the next frame).
extern "C" int rt_hip_scene_share(rt_hip_scene *src, rt_hip_scene **out)
{
  if (!src || !out) return fail(RT_HIP_EINVAL, "NULL argument");
  rt_hip_scene *s = new rt_hip_scene();
  s->ctx = src->ctx; ctx_retain(s->ctx); s->dev = src->dev; s->geo = src->geo;
  s->n_nodes = src->n_nodes; s->n_leaves = src->n_leaves; s->n_cull = src->n_cull; s->max_depth = src->max_depth;
  s->one_sample = src->one_sample; s->fan = src->fan;
  s->defer_fits = src->defer_fits; s->max_seeds = src->max_seeds; s->shadow_samples = src->shadow_samples;
  *out = s;
  return RT_HIP_OK;
}

// brackets one kernel launch with events when profiling is on
static int prof_flush(rt_hip_scene *s)
{
  if (s->ev_used == 0) return RT_HIP_OK;
  HIP_TRY(hipDeviceSynchronize());
  for (size_t i = 0; i < s->ev_used; i++) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_pool[i].a, s->ev_pool[i].b));
    s->prof.ms[s->ev_pool[i].k] += ms; s->prof.launches[s->ev_pool[i].k]++;
  }
  s->ev_used = 0;
  return RT_HIP_OK;
}
struct ProfScope {
  rt_hip_scene *s; hipStream_t st; rt_hip_scene::EvPair *e = nullptr;
  ProfScope(rt_hip_scene *s_, int k, hipStream_t st_) : s(s_), st(st_)
  {
    if (!s || !s->prof_on) return;
    if (s->ev_used == s->ev_pool.size()) {
      if (s->ev_pool.size() >= 8192) { if (prof_flush(s) != RT_HIP_OK) return; }
      else {
        rt_hip_scene::EvPair p; p.k = 0;
        if (hipEventCreate(&p.a) != hipSuccess) return;
        if (hipEventCreate(&p.b) != hipSuccess) { (void)hipEventDestroy(p.a); return; }
        s->ev_pool.push_back(p);
      }
    }
    e = &s->ev_pool[s->ev_used++]; e->k = k;
    (void)hipEventRecord(e->a, st);
  }
  ~ProfScope() { if (e) (void)hipEventRecord(e->b, st); }
};

extern "C" int rt_hip_set_profiling(rt_hip_scene *s, int on)
{
  if (!s) return fail(RT_HIP_EINVAL, "NULL argument");
  HIP_TRY(hipSetDevice(s->ctx->device));
  if (!on) { int rc = prof_flush(s); if (rc) return rc; }
  s->prof_on = on != 0;
  return RT_HIP_OK;
}
extern "C" int rt_hip_get_profile(rt_hip_scene *s, rt_hip_profile *out, int reset)
{
  if (!s || !out) return fail(RT_HIP_EINVAL, "NULL argument");
  HIP_TRY(hipSetDevice(s->ctx->device));
  int rc = prof_flush(s); if (rc) return rc;
  *out = s->prof;
  if (reset) std::memset(&s->prof, 0, sizeof s->prof);
  return RT_HIP_OK;
}

// launch the (STATS, PARAM) instantiation of a walk kernel
#ifdef RT_MINIMAL   /* experiment builds (tools/build_variant.sh): only the plain instantiations, no drivers beyond rt_hip_render_whitted */
#define RT_LAUNCH_SP(K, stats, param, grid, st, ...) hipLaunchKernelGGL((K<false, false>), grid, dim3(RT_BLOCK), 0, st, __VA_ARGS__)
#else
#define RT_LAUNCH_SP(K, stats, param, grid, st, ...)                                                            \
  do {                                                                                                          \
    if (param) { if (stats) hipLaunchKernelGGL((K<true, true>), grid, dim3(RT_BLOCK), 0, st, __VA_ARGS__);      \
                 else hipLaunchKernelGGL((K<false, true>), grid, dim3(RT_BLOCK), 0, st, __VA_ARGS__); }         \
    else       { if (stats) hipLaunchKernelGGL((K<true, false>), grid, dim3(RT_BLOCK), 0, st, __VA_ARGS__);     \
                 else hipLaunchKernelGGL((K<false, false>), grid, dim3(RT_BLOCK), 0, st, __VA_ARGS__); }        \
  } while (0)
#endif

static inline unsigned int blocks_for(size_t n) { return (unsigned int)((n + RT_BLOCK - 1) / RT_BLOCK); }

extern "C" int rt_hip_trace_closest_dev(rt_hip_scene *s, const rt_hip_ray *d_rays, size_t n, int use_tree, int record,
                                         rt_hip_hit *d_hits, uint32_t collect_stats, void *stream)
{
  (void)record;   // identical for sphere / triangle / circle / plane (Object.h:238, 333, 659, 692)
  if (!s || (n && (!d_rays || !d_hits))) return fail(RT_HIP_EINVAL, "NULL argument");
  if (n == 0) return RT_HIP_OK;
  if (n > 0xFFFFFFFFull * 64) return fail(RT_HIP_EINVAL, "too many rays");
  HIP_TRY(hipSetDevice(s->ctx->device));
  DevScene sc = s->dev; if (!use_tree) sc.has_tree = 0;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(s, RT_HIP_K_TRACE_CLOSEST, st);
  RT_LAUNCH_SP(k_trace_closest, collect_stats != 0, sc.has_param != 0, dim3(blocks_for(n)), st, sc, d_rays, n, d_hits,
               collect_stats ? s->ctx->d_stats : (Stats *)nullptr, record);
  HIP_TRY(hipGetLastError());
  return RT_HIP_OK;
}

extern "C" int rt_hip_trace_closest(rt_hip_scene *s, const rt_hip_ray *rays, size_t n, int use_tree, int record, rt_hip_hit *hits)
{
  if (!s || (n && (!rays || !hits))) return fail(RT_HIP_EINVAL, "NULL argument");
  if (n == 0) return RT_HIP_OK;
  HIP_TRY(hipSetDevice(s->ctx->device));
  size_t rb = n * sizeof(rt_hip_ray), hb = n * sizeof(rt_hip_hit);
  int rc = s->staging.ensure(rb + hb); if (rc) return rc;
  rt_hip_ray *dr = (rt_hip_ray *)s->staging.p; rt_hip_hit *dh = (rt_hip_hit *)((char *)s->staging.p + rb);
  HIP_TRY(hipMemcpy(dr, rays, rb, hipMemcpyHostToDevice));
  rc = rt_hip_trace_closest_dev(s, dr, n, use_tree, record, dh, 0, nullptr); if (rc) return rc;
  HIP_TRY(hipMemcpy(hits, dh, hb, hipMemcpyDeviceToHost));
  return RT_HIP_OK;
}

extern "C" int rt_hip_trace_occluded_dev(rt_hip_scene *s, const rt_hip_ray *d_rays, const double *d_ref_pts, size_t n, int use_tree, uint8_t *d_occluded,
                                          uint32_t collect_stats, void *stream)
{
  if (!s || (n && (!d_rays || !d_occluded))) return fail(RT_HIP_EINVAL, "NULL argument");
  if (n == 0) return RT_HIP_OK;
  if (n > 0xFFFFFFFFull * 64) return fail(RT_HIP_EINVAL, "too many rays");
  HIP_TRY(hipSetDevice(s->ctx->device));
  DevScene sc = s->dev; if (!use_tree) sc.has_tree = 0;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(s, RT_HIP_K_TRACE_OCCLUDED, st);
  RT_LAUNCH_SP(k_trace_occluded, collect_stats != 0, sc.has_param != 0, dim3(blocks_for(n)), st, sc, d_rays, d_ref_pts, n, d_occluded,
               collect_stats ? s->ctx->d_stats : (Stats *)nullptr);
  HIP_TRY(hipGetLastError());
  return RT_HIP_OK;
}

extern "C" int rt_hip_trace_occluded(rt_hip_scene *s, const rt_hip_ray *rays, const double *ref_pts, size_t n, int use_tree, uint8_t *occluded)
{
  if (!s || (n && (!rays || !occluded))) return fail(RT_HIP_EINVAL, "NULL argument");
  if (n == 0) return RT_HIP_OK;
  HIP_TRY(hipSetDevice(s->ctx->device));
  size_t rb = n * sizeof(rt_hip_ray), pb = ref_pts ? n * 24 : 0, ob = (n + 15) & ~(size_t)15;
  int rc = s->staging.ensure(rb + pb + ob); if (rc) return rc;
  char *base = (char *)s->staging.p;
  HIP_TRY(hipMemcpy(base, rays, rb, hipMemcpyHostToDevice));
  if (ref_pts) HIP_TRY(hipMemcpy(base + rb, ref_pts, pb, hipMemcpyHostToDevice));
  rc = rt_hip_trace_occluded_dev(s, (const rt_hip_ray *)base, ref_pts ? (const double *)(base + rb) : (const double *)nullptr, n, use_tree,
                                 (uint8_t *)(base + rb + pb), 0, nullptr);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(occluded, base + rb + pb, n, hipMemcpyDeviceToHost));
  return RT_HIP_OK;
}

// CCamera constructor + the loop-invariant part of Render_RayTracer (Camera.h:17-28, 55-57)
static int camera_setup(const rt_hip_camera *c, DevCamera *k)
{
  if (!c) return fail(RT_HIP_EINVAL, "camera is NULL");
  if (c->resolution <= 0) return fail(RT_HIP_EINVAL, "camera resolution must be positive");
  V3 v_eye = mk(c->eye_dir[0], c->eye_dir[1], c->eye_dir[2]), v_left = mk(c->left[0], c->left[1], c->left[2]);
  V3 view = mk(c->view_pt[0], c->view_pt[1], c->view_pt[2]);
  V3 eye = unitize(v_eye);
  V3 up = unitize(cross(v_left, v_eye));
  V3 left = unitize(cross(up, eye));
  V3 center = view + eye * c->dist_of_plane;
  k->view_pt = view; k->left = left; k->up = up;
  k->res = c->resolution; k->depth = c->depth;
  k->W = (int)(c->hor_range * c->resolution); k->H = (int)(c->ver_range * c->resolution);
  k->start = center + left * (c->hor_range / 2) + up * (c->ver_range / 2);
  if (k->W <= 0 || k->H <= 0) return fail(RT_HIP_EINVAL, "empty image");
  if ((uint64_t)k->W * k->H > 0xFFFFFFFFull) return fail(RT_HIP_EINVAL, "image too large");
  return RT_HIP_OK;
}

extern "C" int rt_hip_camera_size(const rt_hip_camera *cam, int *w, int *h)
{
  DevCamera k; int rc = camera_setup(cam, &k); if (rc) return rc;
  if (w) *w = k.W; if (h) *h = k.H;
  return RT_HIP_OK;
}

static int godray_level(rt_hip_scene *s, PathNode *nodes, unsigned int n, const unsigned int *n_dev, hipStream_t st);   // rt_godray.inc

#include "rt_defer.inc"

// Levels of the ray tree once level 0 holds its n0 nodes with their closest hits (k_primary / k_primary_list):
// shadow pass, shading + child rays, closest hit of the children, ... and the bottom-up composition.
// order1: optional heavy-first order of the level-0 waves for the shadow pass.
static int run_levels_from(rt_hip_scene *s, const RenderArgs &A, bool stats, hipStream_t st, uint64_t n0, const unsigned int *order1,
                           uint64_t primary_rays, double *direct_rgb, int32_t *direct_id)
{
  int rc;
  RenderCounters *cnt = (RenderCounters *)s->counter.p;
  Stats *gs = stats ? s->ctx->d_stats : nullptr;
  const bool param = s->dev.has_param != 0;
  const bool defer = defer_ok(s);      // Newton solves as kernels of their own (rt_defer.inc): every level's size is fetched, the solver's launches depend on counts anyway
  // cap[lv]: the host's bound on the size of level lv.  Level 0 is exact.  Below it the sizes are only known on the device
  // (cnt->level_n): level lv + 1 holds at most `fan` children per node of level lv (2 when some material both reflects and
  // refracts, else 1), the kernels are launched for that bound and their surplus blocks leave at once -- no read-back, no
  // wait per level.  Only when the bound would outgrow kLevelBudget (deep trees through glass at full resolution) the real
  // size of that level is fetched, as every level's was before.
  unsigned int cap[rt_hip_scene::kMaxLevels] = {0};
  bool exact[rt_hip_scene::kMaxLevels] = {false};
  cap[0] = (unsigned int)n0; exact[0] = true;
  // (2 GiB: a handle's grow-only workspace is the sum of its level arrays, and every shared handle in flight owns one -- a 1080p depth-8 glass
  // frame ran blind up to 8 GiB per handle under the former 6 GiB bound; beyond the budget a level costs one 4-byte read-back)
  constexpr uint64_t kLevelBudget = 2ull << 30;
  int depth = A.cam.depth < 1 ? 1 : A.cam.depth;
  if (depth > rt_hip_scene::kMaxLevels) return fail(RT_HIP_EUNSUPPORTED, "camera depth > 12: the sampler keys 12 bits of ray-tree path (rng_counter), deeper rays would share random streams");
  HIP_TRY(hipMemsetAsync(cnt->level_n, 0, sizeof cnt->level_n, st));
  int last = 0;
  for (int lv = 0; lv < depth; lv++) {
    const unsigned int n = cap[lv];
    if (n == 0) break;
    last = lv;
    const unsigned int *n_dev = exact[lv] ? nullptr : &cnt->level_n[lv];
    const int depth_left = depth - lv;
    unsigned int next_cap = 0; PathNode *next = nullptr;
    bool fetch_next = false;
    if (depth_left > 1) {
      uint64_t bound = (uint64_t)s->fan * n;
      if (bound * sizeof(PathNode) > kLevelBudget) { bound = 2ull * n; fetch_next = true; }   // too large to run blind: learn the real size after this level
      if (defer) fetch_next = true;
      if (bound > 0x7FFFFFFFull) return fail(RT_HIP_ENOMEM, "ray tree level too large");
      if (fetch_next && !exact[lv]) {
        // this level's own size is still a bound: settle it first so that the next level's array is sized from the real count
        unsigned int real = 0;
        HIP_TRY(hipMemcpyAsync(&real, &cnt->level_n[lv], sizeof real, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        cap[lv] = std::min(real, cap[lv]); exact[lv] = true;
        lv--; continue;
      }
      rc = s->levels[lv + 1].ensure(bound * sizeof(PathNode)); if (rc) return rc;
      next = (PathNode *)s->levels[lv + 1].p; next_cap = (unsigned int)bound;
    }
    PathNode *cur = (PathNode *)s->levels[lv].p;
    if (lv > 0 && defer) { rc = closest_deferred(s, path_view(cur), n, stats, true, st); if (rc) return rc; }
    else if (lv > 0) {
      ProfScope ps(s, RT_HIP_K_TRACE_NODES, st);
      RT_LAUNCH_SP(k_trace_nodes, stats, param, dim3(blocks_for(n)), st, s->dev, cur, n, n_dev, gs);
    }
    rc = s->lum.ensure((size_t)n * std::max(1, s->dev.n_lights) * sizeof(double)); if (rc) return rc;
    double *lum = (double *)s->lum.p;
    const bool direct = direct_rgb && lv == 0 && depth == 1;
    if (s->dev.n_lights > 0 && defer) { rc = shadow_deferred(s, A, cur, n, lum, stats, st); if (rc) return rc; }
    else if (s->dev.n_lights > 0) {
      ProfScope ps(s, RT_HIP_K_SHADOW, st);
      const unsigned int *ord = (lv == 0) ? (const unsigned int *)order1 : nullptr;
      RT_LAUNCH_SP(k_shadow, stats, param, dim3(blocks_for(n)), st, s->dev, A, (const PathNode *)cur, n, n_dev, lum, ord, cnt, gs);
    }
    {
      ProfScope ps(s, RT_HIP_K_SHADE, st);
      double *uv_out = nullptr;
      if (s->keep_uv && param) { rc = s->uv[lv].ensure((size_t)n * 16); if (rc) return rc; uv_out = (double *)s->uv[lv].p; }
      const int next_level = (lv + 1 < rt_hip_scene::kMaxLevels) ? lv + 1 : lv;
      if (param) hipLaunchKernelGGL(k_shade<true>, dim3(blocks_for(n)), dim3(RT_BLOCK), 0, st, s->dev, A, cur, n, n_dev, depth_left, (const double *)lum, next, next_cap, cnt, next_level,
                                    direct ? direct_rgb : (double *)nullptr, direct ? direct_id : (int32_t *)nullptr, uv_out);
      else hipLaunchKernelGGL(k_shade<false>, dim3(blocks_for(n)), dim3(RT_BLOCK), 0, st, s->dev, A, cur, n, n_dev, depth_left, (const double *)lum, next, next_cap, cnt, next_level,
                              direct ? direct_rgb : (double *)nullptr, direct ? direct_id : (int32_t *)nullptr, (double *)nullptr);
    }
    if (s->god) { rc = godray_level(s, cur, n, n_dev, st); if (rc) return rc; }
    HIP_TRY(hipGetLastError());
    if (depth_left > 1) {
      cap[lv + 1] = next_cap;
      if (fetch_next) {
        unsigned int qn = 0;
        HIP_TRY(hipMemcpyAsync(&qn, &cnt->level_n[lv + 1], sizeof qn, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        cap[lv + 1] = std::min(qn, next_cap); exact[lv + 1] = true;
      }
    }
  }
  for (int lv = last - 1; lv >= 0; lv--) {
    ProfScope ps(s, RT_HIP_K_COMBINE, st);
    hipLaunchKernelGGL(k_combine, dim3(blocks_for(cap[lv])), dim3(RT_BLOCK), 0, st, (PathNode *)s->levels[lv].p, cap[lv], exact[lv] ? (const unsigned int *)nullptr : &cnt->level_n[lv],
                       (const PathNode *)s->levels[lv + 1].p, A);
    HIP_TRY(hipGetLastError());
  }
  // what the host knows of this ray tree: bounds (exact for level 0 and for fetched levels); level_counts() settles them when somebody
  // needs the real sizes (photon-map eye pass, ray census)
  for (int lv = 0; lv < rt_hip_scene::kMaxLevels; lv++) { s->last_n_level[lv] = (lv <= last) ? cap[lv] : 0; s->last_exact[lv] = (lv <= last) ? exact[lv] : true; }
  s->last_level = last;
  s->last_rays += primary_rays;                 // the rays of the deeper levels are counted on the device (cnt->rays)
  return RT_HIP_OK;
}

// settles s->last_n_level to the real sizes of the last ray tree (waits for the stream's work)
static int level_counts(rt_hip_scene *s)
{
  bool all = true; for (int lv = 0; lv <= s->last_level; lv++) all = all && s->last_exact[lv];
  if (all || !s->last_counters) return RT_HIP_OK;
  HIP_TRY(hipDeviceSynchronize());
  RenderCounters c{};
  HIP_TRY(hipMemcpy(&c, s->last_counters, sizeof c, hipMemcpyDeviceToHost));
  int last = 0;
  for (int lv = 0; lv <= s->last_level; lv++) {
    if (!s->last_exact[lv]) { s->last_n_level[lv] = std::min(c.level_n[lv], s->last_n_level[lv]); s->last_exact[lv] = true; }
    if (s->last_n_level[lv]) last = lv;
  }
  s->last_level = last;
  return RT_HIP_OK;
}


// bytes of the heavy-first order workspace for n_waves level-0 waves: two order lists, the per-block counts, two flag arrays
static inline size_t order_bytes(uint64_t n_waves) { return (size_t)n_waves * 10 + (size_t)((n_waves + 1023) / 1024) * 128 + 64; }      // (x 128: k_cost_count's 32 classes per block)
struct OrderBufs { unsigned int *order0 = nullptr, *order1 = nullptr, *ocounts = nullptr; unsigned char *flags0 = nullptr, *flags1 = nullptr; unsigned int oblocks = 0; };
// classification of the level-0 waves + their heavy-first order (k_classify, k_order_count, k_order) in `mem` (order_bytes(n_waves))
static int order_waves(rt_hip_scene *s, const RenderArgs &A, uint64_t n_waves, void *mem, hipStream_t st, OrderBufs &O, bool prof)
{
  O.oblocks = (unsigned int)((n_waves + 1023) / 1024);
  O.order0 = (unsigned int *)mem; O.order1 = O.order0 + n_waves; O.ocounts = O.order1 + n_waves;
  O.flags0 = (unsigned char *)(O.ocounts + O.oblocks); O.flags1 = O.flags0 + n_waves;
  rt_hip_scene *ps = prof ? s : nullptr;
  ProfScope scope(ps, RT_HIP_K_ORDER, st);
  hipLaunchKernelGGL(k_classify, dim3((unsigned int)((n_waves + 255) / 256)), dim3(256), 0, st, s->dev, A, (unsigned int)n_waves, O.flags0);
  hipLaunchKernelGGL(k_order_count, dim3(O.oblocks), dim3(1024), 0, st, (const unsigned char *)O.flags0, (unsigned int)n_waves, O.ocounts);
  hipLaunchKernelGGL(k_order, dim3(O.oblocks), dim3(1024), 0, st, (const unsigned char *)O.flags0, (unsigned int)n_waves, (const unsigned int *)O.ocounts, O.order0);
  HIP_TRY(hipGetLastError());
  return RT_HIP_OK;
}
// ---- longest-first order from the previous frame's wave costs --------------------------------------------------------------
// Heavy-first knows two classes.  Within the heavy class wave costs still differ 20x (a tile at the horizon walks ten times the leaves of a
// tile at the viewer's feet), and the last long waves of a launch run on an almost empty GPU.  A renderer's frames come in sequences: the
// same handle renders the same tiling again with a camera that has moved a little, so the time each wave took LAST time (k_frame_direct
// leaves it in wave_cost) is a good estimate of what it will take now.  k_cost_count / k_cost_order list the waves by cost class, longest first,
// 32 classes half an octave apart, stably (neighbouring tiles stay neighbours inside a class: they share leaves in L2).  Results do not depend
// on the order; a stale or missing cost array only costs time.
// cost class: 0 = the longest; half an octave per class over 2^24 .. 2^8 cycles (absolute, so that blocks need not agree on a maximum first)
__device__ __forceinline__ unsigned int cost_class(unsigned int c)
{
  if (c == 0) return 31u;
  const int k = 48 - (int)(2.0f * __log2f((float)c));
  return (unsigned int)(k < 0 ? 0 : (k > 31 ? 31 : k));
}
// two small launches, 1024 waves per block, like k_order_count / k_order:
//   k_cost_count : entries per class and block
//   k_cost_order : every block sums what lies before its entries of each class (all blocks' lower classes + earlier blocks' same class) and places them
__global__ void __launch_bounds__(1024) k_cost_count(const unsigned int *cost, unsigned int n, unsigned int *counts)
{
  __shared__ unsigned int s_h[32];
  const unsigned int t = threadIdx.x, i = blockIdx.x * 1024 + t;
  if (t < 32) s_h[t] = 0;
  __syncthreads();
  if (i < n) atomicAdd(&s_h[cost_class(cost[i])], 1u);
  __syncthreads();
  if (t < 32) counts[blockIdx.x * 32 + t] = s_h[t];
}
__global__ void __launch_bounds__(1024) k_cost_order(const unsigned int *cost, unsigned int n, const unsigned int *counts, unsigned int *order)
{
  __shared__ unsigned int s_tot[32], s_before[32], s_base[32], s_wc[16][32];
  const unsigned int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t < 32) {                                   // class t: entries in all blocks, and in the blocks before this one
    unsigned int tot = 0, bef = 0;
    for (unsigned int b = 0; b < gridDim.x; b++) { const unsigned int c = counts[b * 32 + t]; tot += c; if (b < blockIdx.x) bef += c; }
    s_tot[t] = tot; s_before[t] = bef;
  }
  __syncthreads();
  if (t == 0) { unsigned int a = 0; for (int k = 0; k < 32; k++) { s_base[k] = a + s_before[k]; a += s_tot[k]; } }
  const unsigned int i = blockIdx.x * 1024 + t;
  const bool valid = i < n;
  const unsigned int c = valid ? cost_class(cost[i]) : 32u;
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  unsigned int my_rank = 0;
  for (unsigned int b = 0; b < 32; b++) {
    const unsigned long long m = __ballot(c == b);
    if (lane == 0) s_wc[w][b] = (unsigned int)__popcll(m);
    if (c == b) my_rank = (unsigned int)__popcll(m & below);
  }
  __syncthreads();
  if (valid) {
    unsigned int before = 0;
    for (unsigned int k = 0; k < w; k++) before += s_wc[k][c];
    order[s_base[c] + before + my_rank] = i;
  }
}

// does the depth-1 frame take the loop-free kernel (one light of one sample: the luminosity stays in a register, no lum array)?
static inline bool frame_direct_one(const rt_hip_scene *s, bool stats) { return !stats && !s->dev.has_param && s->one_sample; }
// a depth-1 frame in one launch (k_frame_direct); lum: n0 * n_lights doubles, needed unless frame_direct_one()
static int launch_frame_direct(rt_hip_scene *s, const RenderArgs &A, bool stats, hipStream_t st, uint64_t n_waves, const unsigned int *order0,
                               double *lum, RenderCounters *cnt, double *rgb, int32_t *ids, bool prof)
{
  Stats *gs = stats ? s->ctx->d_stats : nullptr;
  const bool param = s->dev.has_param != 0;
  const unsigned int pblocks = (unsigned int)(n_waves / RT_WAVES);
  {
    ProfScope ps(prof ? s : nullptr, RT_HIP_K_FRAME_DIRECT, st);
    if (!frame_direct_one(s, stats) && !lum) return fail(RT_HIP_EINVAL, "internal: luminosity array missing");
    if (frame_direct_one(s, stats))
      hipLaunchKernelGGL((k_frame_direct<false, false, true>), dim3(pblocks), dim3(RT_BLOCK), 0, st, s->dev, A, order0, lum, cnt, gs, rgb, ids);
    else
      RT_LAUNCH_SP(k_frame_direct, stats, param, dim3(pblocks), st, s->dev, A, order0, lum, cnt, gs, rgb, ids);
  }
  HIP_TRY(hipGetLastError());
  for (int lv = 0; lv < rt_hip_scene::kMaxLevels; lv++) s->last_n_level[lv] = 0;
  s->last_level = 0;
  s->last_rays += A.n_primary;
  return RT_HIP_OK;
}
// RT_HIP_COST_ORDER in the environment (read per call): 0 keeps the two-class heavy-first order for every frame (A/B runs), 2 uses the
// last frame's costs whatever the size of the launch (tests on small frames); otherwise the size rule below decides
static inline int cost_order_mode() { const char *e = std::getenv("RT_HIP_COST_ORDER"); return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 1; }
static inline uint64_t level0_waves(const RenderArgs &A) { return (uint64_t)A.n_st * 128 * A.n_samples; }

// the wavefront Whitted integrator for one batch of (pixels x samples); leaves final colours in level 0
static int run_levels(rt_hip_scene *s, const RenderArgs &A, bool stats, hipStream_t st, double *direct_rgb = nullptr, int32_t *direct_id = nullptr)
{
  const uint64_t n_waves = level0_waves(A);
  const uint64_t n0 = n_waves * 64;      // level 0 in tile order, edge tiles padded
  if (A.pixel_count == 0 || n0 > 0x7FFFFFFFull) return fail(RT_HIP_EINVAL, "batch size out of range");
  int rc = s->counter.ensure(sizeof(RenderCounters)); if (rc) return rc;
  Stats *gs = stats ? s->ctx->d_stats : nullptr;
  const bool param = s->dev.has_param != 0;
  const unsigned int pblocks = (unsigned int)(n_waves / RT_WAVES);
  // heavy-first order of the level-0 waves (only worth it with a tree)
  OrderBufs O;
  const int cam_depth = A.cam.depth < 1 ? 1 : A.cam.depth;
  const bool defer = defer_ok(s);
  const bool direct = direct_rgb && cam_depth == 1 && s->dev.n_lights > 0 && !s->god && !s->keep_uv && !A.focusing && A.n_samples == 1 && !defer;
  // a depth-1 frame of a tiling this handle has rendered before: longest-first by the waves' last run times (k_cost_order)
  const uint64_t sig[6] = {n_waves, ((uint64_t)A.cam.W << 32) | (uint64_t)A.cam.H, ((uint64_t)A.tile_offset << 32) | A.tile_stride, A.pixel_first, A.pixel_count, (uint64_t)A.v_lo + 1};
  // (worth it from about six rounds of the GPU's 4,096 wave slots on: measured on C2, the full frame's 32,640 waves gain 3 %, a half frame's
  // 16,320 lose 4 % -- fewer rounds leave less of a tail to fill, and the cost order runs fewer neighbouring tiles together)
  const int cost_mode = cost_order_mode();
  const bool cost_sized = cost_mode != 0 && (cost_mode == 2 || n_waves >= 6u * 4096u) && n_waves <= (1u << 18);
  const bool by_cost = direct && s->dev.has_tree && !stats && cost_sized && std::memcmp(sig, s->cost_sig, sizeof sig) == 0;
  // a launch that fits the wave slots twice over starts (nearly) all its waves at once: no order to compute
  const bool ordered = !defer && s->dev.has_tree && (by_cost || !(direct && n_waves <= 2u * 4096u));
  if (ordered) {
    rc = s->order.ensure(order_bytes(n_waves)); if (rc) return rc;
    if (by_cost) {
      O.order0 = (unsigned int *)s->order.p;
      ProfScope scope(s, RT_HIP_K_ORDER, st);
      const unsigned int cb = (unsigned int)((n_waves + 1023) / 1024);
      unsigned int *ccounts = O.order0 + 2 * n_waves;               // (the place of the heavy-first lists' counts and flags: n_waves * 2 bytes + ...)
      hipLaunchKernelGGL(k_cost_count, dim3(cb), dim3(1024), 0, st, (const unsigned int *)s->wave_cost.p, (unsigned int)n_waves, ccounts);
      hipLaunchKernelGGL(k_cost_order, dim3(cb), dim3(1024), 0, st, (const unsigned int *)s->wave_cost.p, (unsigned int)n_waves, (const unsigned int *)ccounts, O.order0);
      HIP_TRY(hipGetLastError());
    } else { rc = order_waves(s, A, n_waves, s->order.p, st, O, true); if (rc) return rc; }
  }
  if (direct) {
    if (!frame_direct_one(s, stats)) { rc = s->lum.ensure((size_t)n0 * s->dev.n_lights * sizeof(double)); if (rc) return rc; }
    RenderArgs B = A;
    if (s->dev.has_tree && !stats && cost_sized) {
      rc = s->wave_cost.ensure((size_t)n_waves * sizeof(unsigned int)); if (rc) return rc;
      B.wave_cost = (unsigned int *)s->wave_cost.p;
      std::memcpy(s->cost_sig, sig, sizeof sig);
    }
    return launch_frame_direct(s, B, stats, st, n_waves, O.order0, (double *)s->lum.p, (RenderCounters *)s->counter.p, direct_rgb, direct_id, true);
  }
  rc = s->levels[0].ensure(n0 * sizeof(PathNode)); if (rc) return rc;
  PathNode *L0 = (PathNode *)s->levels[0].p;
  if (defer) {
    {
      ProfScope ps(s, RT_HIP_K_PRIMARY, st);
      hipLaunchKernelGGL((k_primary<false, false>), dim3(pblocks), dim3(RT_BLOCK), 0, st, s->dev, A, L0, (const unsigned int *)nullptr, (unsigned char *)nullptr, gs, 1);
      HIP_TRY(hipGetLastError());
    }
    rc = closest_deferred(s, path_view(L0), (unsigned int)n0, stats, true, st); if (rc) return rc;
    return run_levels_from(s, A, stats, st, n0, nullptr, A.n_primary * A.n_samples, direct_rgb, direct_id);
  }
  {
    ProfScope ps(s, RT_HIP_K_PRIMARY, st);
    RT_LAUNCH_SP(k_primary, stats, param, dim3(pblocks), st, s->dev, A, L0, (const unsigned int *)O.order0, O.flags1, gs, 0);
  }
  unsigned int *order1 = nullptr;
  if (s->dev.has_tree && s->dev.n_lights > 0) {
    ProfScope ps(s, RT_HIP_K_ORDER, st);
    hipLaunchKernelGGL(k_order_count, dim3(O.oblocks), dim3(1024), 0, st, (const unsigned char *)O.flags1, (unsigned int)n_waves, O.ocounts);
    hipLaunchKernelGGL(k_order, dim3(O.oblocks), dim3(1024), 0, st, (const unsigned char *)O.flags1, (unsigned int)n_waves, (const unsigned int *)O.ocounts, O.order1);
    order1 = O.order1;
  }
  HIP_TRY(hipGetLastError());
  return run_levels_from(s, A, stats, st, n0, order1, A.n_primary * A.n_samples, direct_rgb, direct_id);
}

static int fill_args(rt_hip_scene *s, const rt_hip_camera *cam, const rt_hip_params *P, RenderArgs *A)
{
  if (!s || !P) return fail(RT_HIP_EINVAL, "NULL argument");
  if (P->struct_size != sizeof(rt_hip_params)) return fail(RT_HIP_EINVAL, "rt_hip_params.struct_size mismatch");
  A->wave_cost = nullptr; A->batch_cams = nullptr; A->batch_rgb_pitch = 0; A->batch_id_pitch = 0;
  int rc = camera_setup(cam, &A->cam); if (rc) return rc;
  const uint64_t total = (uint64_t)A->cam.W * A->cam.H;
  A->pixel_first = P->pixel_count ? P->pixel_first : 0;
  A->pixel_count = P->pixel_count ? P->pixel_count : total;
  if (A->pixel_first >= total || A->pixel_count > total - A->pixel_first) return fail(RT_HIP_EINVAL, "pixel shard outside the image");
  for (int k = 0; k < 3; k++) A->bg[k] = P->background[k];
  A->weight_threshold = (P->weight_threshold > 0) ? P->weight_threshold : 0.1;   // RayTracer.h:7
  A->seed = P->seed;
  A->v_lo = (uint32_t)(A->pixel_first / A->cam.W);
  const uint32_t v_hi = (uint32_t)((A->pixel_first + A->pixel_count - 1) / A->cam.W);
  A->tiles_x = (uint32_t)(A->cam.W + 7) / 8; A->tiles_y = (v_hi - A->v_lo + 8) / 8;
  A->sample_first = 0; A->n_samples = 1; A->lens_radius = 0; A->focusing = 0;
  const uint32_t sx = (A->tiles_x + 15) / 16, sy = (A->tiles_y + 7) / 8;
  A->tile_offset = 0; A->tile_stride = 1; A->n_st = sx * sy; A->n_out = A->pixel_count; A->n_primary = A->pixel_count;
  A->out_bgr8 = 0;
  if (P->frame_format != RT_HIP_FRAME_RGB_F64 && P->frame_format != RT_HIP_FRAME_BGR8) return fail(RT_HIP_EINVAL, "rt_hip_params.frame_format unknown");
  if (P->tile_stride > 1) {
    if (P->pixel_count) return fail(RT_HIP_EINVAL, "a tile shard (tile_stride > 1) and a pixel shard (pixel_count) exclude each other");
    if (P->tile_offset >= P->tile_stride) return fail(RT_HIP_EINVAL, "tile_offset must be below tile_stride");
    A->tile_offset = P->tile_offset; A->tile_stride = P->tile_stride;
    const uint32_t total = sx * sy;
    A->n_st = (total > P->tile_offset) ? (total - P->tile_offset + P->tile_stride - 1) / P->tile_stride : 0;
    A->n_out = (uint64_t)A->n_st * 8192;
    uint64_t np = 0;                    // pixels of the image inside this call's super-tiles
    for (uint32_t k = 0; k < A->n_st; k++) {
      const uint32_t st_i = A->tile_offset + k * A->tile_stride;
      const int64_t w = std::min<int64_t>(128, (int64_t)A->cam.W - (int64_t)(st_i % sx) * 128), h = std::min<int64_t>(64, (int64_t)A->cam.H - (int64_t)(st_i / sx) * 64);
      if (w > 0 && h > 0) np += (uint64_t)(w * h);
    }
    A->n_primary = np;
  } else if (P->tile_offset) return fail(RT_HIP_EINVAL, "tile_offset without tile_stride");
  return RT_HIP_OK;
}
static bool tile_sharded(const RenderArgs &A) { return A.tile_stride > 1; }
// A copy of a call's arguments over another range of whole pixel rows [v_first, v_last] (the anti-aliasing drivers' G-buffer with its halo rows):
// everything fill_args derives from the rows -- the 8x8 tiling, the super-tile count that sizes level 0 and the launches, the output and
// camera-ray counts -- is derived again, so that no user of the copy meets the original's figures.  Pixel shards only.
static RenderArgs args_over_rows(const RenderArgs &A, uint32_t v_first, uint32_t v_last)
{
  RenderArgs G = A;
  G.pixel_first = (uint64_t)v_first * A.cam.W; G.pixel_count = (uint64_t)(v_last - v_first + 1) * A.cam.W;
  G.v_lo = v_first; G.tiles_y = (v_last - v_first + 8) / 8;
  G.n_st = ((G.tiles_x + 15) / 16) * ((G.tiles_y + 7) / 8);
  G.n_out = G.pixel_count; G.n_primary = G.pixel_count;
  return G;
}

extern "C" int rt_hip_render_whitted_dev(rt_hip_scene *s, const rt_hip_camera *cam, const rt_hip_params *P, double *d_rgb,
                                          int32_t *d_prim_id, void *stream)
{
  RenderArgs A; int rc = fill_args(s, cam, P, &A); if (rc) return rc;
  if (!d_rgb) return fail(RT_HIP_EINVAL, "rgb is NULL");
  HIP_TRY(hipSetDevice(s->ctx->device));
  hipStream_t st = (hipStream_t)stream;
  rc = s->counter.ensure(sizeof(RenderCounters)); if (rc) return rc;
  HIP_TRY(hipMemsetAsync(s->counter.p, 0, sizeof(RenderCounters), st));
  s->last_rays = 0; s->last_shadow = 0; s->last_counters = s->counter.p;
  A.out_bgr8 = (P->frame_format == RT_HIP_FRAME_BGR8) ? 1u : 0u;
  if (A.n_st == 0) return RT_HIP_OK;        // a tile shard beyond the image's last super-tile: nothing to render
  const bool direct = A.cam.depth <= 1;     // a depth-1 frame is final after level 0: k_shade writes it
  rc = run_levels(s, A, P->collect_stats != 0, st, direct ? d_rgb : nullptr, direct ? d_prim_id : nullptr); if (rc) return rc;
  if (!direct) {
    ProfScope ps(s, RT_HIP_K_RESOLVE, st);
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for(A.n_out)), dim3(RT_BLOCK), 0, st, (const PathNode *)s->levels[0].p, A, 1.0, 0, d_rgb, d_prim_id);
  }
  HIP_TRY(hipGetLastError());
  return RT_HIP_OK;
}

// Several depth-1 frames of ONE tiling in one launch: frame k from cams[k] to d_rgb + k * rgb_pitch_bytes (and d_prim_id + k * id_pitch).  A rank of an
// N-GPU run renders 1/N of every frame -- an eighth of a 1080p frame is 4,080 waves, one round of the GPU's wave slots, so a launch per frame lasts as long
// as its slowest tile and throughput has to come from launches in flight; the shares of a sub-batch of frames in one launch fill the GPU like a whole frame.
extern "C" int rt_hip_render_whitted_batch_dev(rt_hip_scene *s, const rt_hip_camera *cams, uint32_t n_frames, const rt_hip_params *P, void *d_rgb,
                                                uint64_t rgb_pitch_bytes, int32_t *d_prim_id, uint64_t id_pitch, void *stream)
{
  if (!cams || n_frames == 0 || n_frames > 64) return fail(RT_HIP_EINVAL, "1 to 64 frames per batch");
  RenderArgs A; int rc = fill_args(s, &cams[0], P, &A); if (rc) return rc;
  if (!d_rgb) return fail(RT_HIP_EINVAL, "rgb is NULL");
  if (P->collect_stats) return fail(RT_HIP_EUNSUPPORTED, "no statistics build of the batched frame kernel");
  if (s->dev.has_param || s->dev.n_lights <= 0 || s->god || s->keep_uv)
    return fail(RT_HIP_EUNSUPPORTED, "batched frames: scenes of triangles / planes / spheres / circles with at least one light (use rt_hip_render_whitted_dev per frame)");
  A.out_bgr8 = (P->frame_format == RT_HIP_FRAME_BGR8) ? 1u : 0u;
  const uint64_t frame_bytes = A.n_out * (A.out_bgr8 ? 3u : 24u);
  if (n_frames > 1 && (rgb_pitch_bytes < frame_bytes || (d_prim_id && id_pitch < A.n_out))) return fail(RT_HIP_EINVAL, "frame pitch smaller than a frame");
  HIP_TRY(hipSetDevice(s->ctx->device));
  if (!s->batch_pinned) {
    HIP_TRY(hipHostMalloc((void **)&s->batch_pinned, sizeof(DevCamera) * rt_hip_scene::kBatchSlots * rt_hip_scene::kBatchMax, hipHostMallocDefault));
    for (int k = 0; k < rt_hip_scene::kBatchSlots; k++) HIP_TRY(hipEventCreateWithFlags(&s->batch_ev[k], hipEventDisableTiming));
  }
  rc = s->batch_cams.ensure(sizeof(DevCamera) * rt_hip_scene::kBatchSlots * rt_hip_scene::kBatchMax); if (rc) return rc;
  const unsigned int slot = s->batch_next++ % rt_hip_scene::kBatchSlots;
  if (s->batch_ev_used[slot]) HIP_TRY(hipEventSynchronize(s->batch_ev[slot]));
  DevCamera *host_cams = s->batch_pinned + (size_t)slot * rt_hip_scene::kBatchMax;
  DevCamera *dev_cams = (DevCamera *)s->batch_cams.p + (size_t)slot * rt_hip_scene::kBatchMax;
  for (uint32_t k = 0; k < n_frames; k++) {
    DevCamera &c = host_cams[k];
    rc = camera_setup(&cams[k], &c); if (rc) return rc;
    if (c.depth > 1) return fail(RT_HIP_EUNSUPPORTED, "batched frames are depth-1 frames (Camera.h:53-75 with depth == 1)");
    if (c.W != A.cam.W || c.H != A.cam.H || c.res != A.cam.res) return fail(RT_HIP_EINVAL, "the cameras of a batch share one picture size");
  }
  hipStream_t st = (hipStream_t)stream;
  rc = s->counter.ensure(sizeof(RenderCounters)); if (rc) return rc;
  HIP_TRY(hipMemsetAsync(s->counter.p, 0, sizeof(RenderCounters), st));
  HIP_TRY(hipMemcpyAsync(dev_cams, host_cams, n_frames * sizeof(DevCamera), hipMemcpyHostToDevice, st));
  s->last_rays = 0; s->last_shadow = 0; s->last_counters = s->counter.p;
  if (A.n_st == 0) return RT_HIP_OK;
  A.n_samples = n_frames; A.batch_cams = dev_cams; A.batch_rgb_pitch = rgb_pitch_bytes; A.batch_id_pitch = id_pitch;
  const uint64_t n_waves = level0_waves(A);
  if (n_waves * 64 > 0x7FFFFFFFull) return fail(RT_HIP_EINVAL, "batch size out of range");
  const bool one = frame_direct_one(s, false);
  if (!one) { rc = s->lum.ensure((size_t)n_waves * 64 * s->dev.n_lights * sizeof(double)); if (rc) return rc; }
  {
    ProfScope ps(s, RT_HIP_K_FRAME_DIRECT, st);
    const unsigned int pblocks = (unsigned int)(n_waves / RT_WAVES);
    if (one) hipLaunchKernelGGL((k_frame_direct<false, false, true, true>), dim3(pblocks), dim3(RT_BLOCK), 0, st, s->dev, A, (const unsigned int *)nullptr, (double *)nullptr,
                                (RenderCounters *)s->counter.p, (Stats *)nullptr, (double *)d_rgb, d_prim_id);
    else hipLaunchKernelGGL((k_frame_direct<false, false, false, true>), dim3(pblocks), dim3(RT_BLOCK), 0, st, s->dev, A, (const unsigned int *)nullptr, (double *)s->lum.p,
                            (RenderCounters *)s->counter.p, (Stats *)nullptr, (double *)d_rgb, d_prim_id);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(s->batch_ev[slot], st)); s->batch_ev_used[slot] = true;
  for (int lv = 0; lv < rt_hip_scene::kMaxLevels; lv++) s->last_n_level[lv] = 0;
  s->last_level = 0;
  s->last_rays += A.n_primary * n_frames;
  return RT_HIP_OK;
}

// rt_hip_render_whitted_dev for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid
// hipMalloc): the depth-1 frame in one launch (k_frame_direct), with the counters, the heavy-first wave order and the
// per-light luminosities in the per-GPU `scratch` array instead of the scene handle's grow-only buffers.  No hipMalloc,
// hipFree, event or stream creation anywhere below.  Every call on a GPU uses that GPU's one array: calls that overlap
// on one GPU race; what must start at zero (the counters) is cleared on the stream first.
extern "C" int rt_hip_render_whitted_noalloc(rt_hip_scene *s, const rt_hip_camera *cam, const rt_hip_params *P, double *d_rgb,
                                              int32_t *d_prim_id, void *stream)
{
  RenderArgs A; int rc = fill_args(s, cam, P, &A); if (rc) return rc;
  if (!d_rgb) return fail(RT_HIP_EINVAL, "rgb is NULL");
  if (A.cam.depth > 1) return fail(RT_HIP_EUNSUPPORTED, "rt_hip_render_whitted_noalloc: camera depth must be 1 (deeper ray trees need level arrays that grow with the frame)");
  if (s->dev.n_lights <= 0) return fail(RT_HIP_EUNSUPPORTED, "rt_hip_render_whitted_noalloc: the scene needs at least one light");
  const int device = s->ctx->device;
  if (device < 0 || device >= RT_MAX_DEVICES || !g_scratch_ptr[device]) return fail(RT_HIP_EUNSUPPORTED, "rt_hip_render_whitted_noalloc: no scratch array on this device");
  HIP_TRY(hipSetDevice(device));
  unsigned char *ptr = g_scratch_ptr[device];          // this GPU's copy of `scratch`
  hipStream_t st = (hipStream_t)stream;
  A.out_bgr8 = (P->frame_format == RT_HIP_FRAME_BGR8) ? 1u : 0u;
  if (A.n_st == 0) return RT_HIP_OK;
  const uint64_t n_waves = level0_waves(A), n0 = n_waves * 64;
  if (A.pixel_count == 0 || n0 > 0x7FFFFFFFull) return fail(RT_HIP_EINVAL, "batch size out of range");
  // layout: [0, 256) counters | wave order workspace | luminosities (only for scenes that are not one light of one sample)
  const size_t ord_b = s->dev.has_tree ? ((order_bytes(n_waves) + 255) & ~(size_t)255) : 0;
  const size_t lum_b = frame_direct_one(s, P->collect_stats != 0) ? 0 : (size_t)n0 * s->dev.n_lights * sizeof(double);
  const size_t need = 256 + ord_b + lum_b;
  if (need > RT_NOALLOC_SCRATCH_BYTES)
    return fail(RT_HIP_EINVAL, "rt_hip_render_whitted_noalloc: frame needs " + std::to_string(need) + " bytes of scratch, the fixed array holds " +
                                   std::to_string((size_t)RT_NOALLOC_SCRATCH_BYTES) + " (use rt_hip_render_whitted_dev, or shard the pixels)");
  RenderCounters *cnt = (RenderCounters *)ptr;
  HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(RenderCounters), st));
  s->last_rays = 0; s->last_shadow = 0; s->last_counters = cnt;
  OrderBufs O;
  if (s->dev.has_tree) { rc = order_waves(s, A, n_waves, ptr + 256, st, O, false); if (rc) return rc; }
  return launch_frame_direct(s, A, P->collect_stats != 0, st, n_waves, O.order0, (double *)(ptr + 256 + ord_b), cnt, d_rgb, d_prim_id, false);
}

extern "C" int rt_hip_tile_shard_info(const rt_hip_camera *cam, uint32_t tile_stride, uint32_t *super_tiles, uint32_t *per_rank)
{
  DevCamera k; int rc = camera_setup(cam, &k); if (rc) return rc;
  if (tile_stride == 0) tile_stride = 1;
  const uint32_t sx = (((uint32_t)k.W + 7) / 8 + 15) / 16, sy = (((uint32_t)k.H + 7) / 8 + 7) / 8;
  if (super_tiles) *super_tiles = sx * sy;
  if (per_rank) *per_rank = (sx * sy + tile_stride - 1) / tile_stride;
  return RT_HIP_OK;
}
extern "C" int rt_hip_tiles_to_frames_dev(const rt_hip_camera *cam, uint32_t tile_stride, const void *d_tiles, uint64_t rank_pitch_bytes, uint32_t elem_bytes,
                                          uint32_t n_frames, uint64_t tiles_frame_pitch_bytes, void *d_frames, uint64_t frame_pitch_bytes, void *stream)
{
  DevCamera k; int rc = camera_setup(cam, &k); if (rc) return rc;
  if (!d_tiles || !d_frames || elem_bytes == 0 || elem_bytes > 256 || n_frames == 0 || n_frames > 65535) return fail(RT_HIP_EINVAL, "rt_hip_tiles_to_frames_dev: bad argument");
  if (tile_stride == 0) tile_stride = 1;
  uint32_t total = 0, per_rank = 0;
  rc = rt_hip_tile_shard_info(cam, tile_stride, &total, &per_rank); if (rc) return rc;
  const uint64_t npix = (uint64_t)k.W * k.H, shard = (uint64_t)per_rank * 8192 * elem_bytes;
  if (tiles_frame_pitch_bytes == 0) tiles_frame_pitch_bytes = shard;
  if (frame_pitch_bytes == 0) frame_pitch_bytes = npix * elem_bytes;
  if (rank_pitch_bytes == 0) rank_pitch_bytes = shard * n_frames;
  const bool wide = elem_bytes % 4 == 0;
  if (rank_pitch_bytes < tiles_frame_pitch_bytes * (n_frames - 1) + shard || tiles_frame_pitch_bytes < shard || frame_pitch_bytes < npix * elem_bytes ||
      (wide && (rank_pitch_bytes % 8 || tiles_frame_pitch_bytes % 8 || frame_pitch_bytes % 8)))
    return fail(RT_HIP_EINVAL, "rt_hip_tiles_to_frames_dev: a pitch is too small or misaligned");
  hipLaunchKernelGGL(k_tiles_to_frame, dim3((unsigned int)((npix + 255) / 256), n_frames), dim3(256), 0, (hipStream_t)stream, (const unsigned char *)d_tiles,
                     (unsigned int)k.W, (unsigned int)k.H, tile_stride, rank_pitch_bytes, elem_bytes, (unsigned char *)d_frames, tiles_frame_pitch_bytes, frame_pitch_bytes);
  HIP_TRY(hipGetLastError());
  return RT_HIP_OK;
}
extern "C" int rt_hip_tiles_to_frame_dev(const rt_hip_camera *cam, uint32_t tile_stride, const void *d_tiles, uint64_t rank_pitch_bytes, uint32_t elem_bytes,
                                         void *d_frame, void *stream)
{
  if (rank_pitch_bytes == 0) {              // tightly packed single frame
    uint32_t total = 0, per_rank = 0;
    int rc = rt_hip_tile_shard_info(cam, tile_stride ? tile_stride : 1, &total, &per_rank); if (rc) return rc;
    rank_pitch_bytes = (uint64_t)per_rank * 8192 * elem_bytes;
  }
  return rt_hip_tiles_to_frames_dev(cam, tile_stride, d_tiles, rank_pitch_bytes, elem_bytes, 1, 0, d_frame, 0, stream);
}

// Camera.h:187-192 for the sample shard [s_first, s_first + s_count) of spp: sum of sample / spp into d_rgb
static int focusing_batches(rt_hip_scene *s, RenderArgs A, bool stats, uint32_t spp, uint32_t s_first, uint32_t s_count, double *d_rgb, hipStream_t st)
{
  int rc;
  // batch the samples so that one level-0 array stays near 8 GiB (RT_HIP_DOF_BATCH_GIB: another budget).  With the all-clear test most waves of
  // a launch cost next to nothing and the launch lasts as long as the waves around the geometry: C3 at 1024 spp 3.96 s with 1 GiB batches (4 samples),
  // 3.34 / 2.85 / 2.75 s with 2 / 4 / 8 GiB, no further gain beyond
  uint64_t budget = 8ull << 30;
  { const char *e = std::getenv("RT_HIP_DOF_BATCH_GIB"); if (e && std::atoi(e) > 0) budget = (uint64_t)std::min(64, std::atoi(e)) << 30; }
  uint64_t per = budget / ((uint64_t)((A.tiles_x + 15) / 16) * ((A.tiles_y + 7) / 8) * 128 * 64 * sizeof(PathNode));
  uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(per, 64));
  for (uint32_t done = 0; done < s_count; done += batch) {
    A.sample_first = s_first + done; A.n_samples = std::min(batch, s_count - done);
    rc = run_levels(s, A, stats, st); if (rc) return rc;
    { ProfScope ps(s, RT_HIP_K_RESOLVE, st);
    hipLaunchKernelGGL(k_resolve, dim3(blocks_for(A.pixel_count)), dim3(RT_BLOCK), 0, st, (const PathNode *)s->levels[0].p, A, (double)spp,
                       done ? 1 : 0, d_rgb, (int32_t *)nullptr); }
    HIP_TRY(hipGetLastError());
  }
  return RT_HIP_OK;
}

extern "C" int rt_hip_render_focusing_dev(rt_hip_scene *s, const rt_hip_camera *cam, const rt_hip_params *P, double *d_rgb, void *stream)
{
  RenderArgs A; int rc = fill_args(s, cam, P, &A); if (rc) return rc;
  if (!d_rgb) return fail(RT_HIP_EINVAL, "rgb is NULL");
  HIP_TRY(hipSetDevice(s->ctx->device));
  hipStream_t st = (hipStream_t)stream;
  const uint32_t spp = P->spp ? P->spp : 100;       // Camera.h:171
  uint32_t s_first = P->sample_count ? P->sample_first : 0, s_count = P->sample_count ? P->sample_count : spp;
  if (s_first >= spp || s_count > spp - s_first) return fail(RT_HIP_EINVAL, "sample shard outside [0, spp)");
  if (tile_sharded(A) || P->frame_format != RT_HIP_FRAME_RGB_F64) return fail(RT_HIP_EUNSUPPORTED, "Render_Focusing shards by samples or pixel ranges and returns f64 sums (tile shards / BGR8 frames: rt_hip_render_whitted)");
  A.focusing = 1; A.lens_radius = P->lens_radius;
  rc = s->counter.ensure(sizeof(RenderCounters)); if (rc) return rc;
  HIP_TRY(hipMemsetAsync(s->counter.p, 0, sizeof(RenderCounters), st));
  s->last_rays = 0; s->last_shadow = 0; s->last_counters = s->counter.p;
  return focusing_batches(s, A, P->collect_stats != 0, spp, s_first, s_count, d_rgb, st);
}

static int render_host(rt_hip_scene *s, const rt_hip_camera *cam, const rt_hip_params *P, double *rgb, int32_t *prim_id, bool focusing)
{
  RenderArgs A; int rc = fill_args(s, cam, P, &A); if (rc) return rc;
  if (!rgb) return fail(RT_HIP_EINVAL, "rgb is NULL");
  HIP_TRY(hipSetDevice(s->ctx->device));
  const size_t cb = A.n_out * ((!focusing && P->frame_format == RT_HIP_FRAME_BGR8) ? 3 : 24), ib = A.n_out * 4;
  const size_t id_off = (cb + 255) & ~(size_t)255;                             // the ids behind the colours, aligned
  if (A.n_out == 0) return RT_HIP_OK;
  rc = s->staging.ensure(id_off + ib); if (rc) return rc;
  if (tile_sharded(A)) HIP_TRY(hipMemset(s->staging.p, 0, id_off + ib));      // entries of edge super-tiles outside the image
  double *d_rgb = (double *)s->staging.p; int32_t *d_id = (int32_t *)((char *)s->staging.p + id_off);
  rc = focusing ? rt_hip_render_focusing_dev(s, cam, P, d_rgb, nullptr) : rt_hip_render_whitted_dev(s, cam, P, d_rgb, prim_id ? d_id : nullptr, nullptr);
  if (rc) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(rgb, d_rgb, cb, hipMemcpyDeviceToHost));
  if (prim_id && !focusing) HIP_TRY(hipMemcpy(prim_id, d_id, ib, hipMemcpyDeviceToHost));
  return RT_HIP_OK;
}

extern "C" int rt_hip_render_whitted(rt_hip_scene *s, const rt_hip_camera *cam, const rt_hip_params *P, double *rgb, int32_t *prim_id)
{ return render_host(s, cam, P, rgb, prim_id, false); }

extern "C" int rt_hip_render_focusing(rt_hip_scene *s, const rt_hip_camera *cam, const rt_hip_params *P, double *rgb)
{ return render_host(s, cam, P, rgb, nullptr, true); }

extern "C" int rt_hip_last_ray_count(rt_hip_scene *s, uint64_t *rays, uint64_t *shadow_rays)
{
  if (!s) return fail(RT_HIP_EINVAL, "NULL argument");
  HIP_TRY(hipSetDevice(s->ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  RenderCounters c{};
  const void *src = s->last_counters ? s->last_counters : s->counter.p;
  if (src) HIP_TRY(hipMemcpy(&c, src, sizeof c, hipMemcpyDeviceToHost));
  if (rays) *rays = s->last_rays + c.rays + c.shadow_rays;
  if (shadow_rays) *shadow_rays = c.shadow_rays;
  return RT_HIP_OK;
}

extern "C" int rt_hip_last_proved_shadow_rays(rt_hip_scene *s, uint64_t *n)
{
  if (!s || !n) return fail(RT_HIP_EINVAL, "NULL argument");
  HIP_TRY(hipSetDevice(s->ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  RenderCounters c{};
  const void *src = s->last_counters ? s->last_counters : s->counter.p;
  if (src) HIP_TRY(hipMemcpy(&c, src, sizeof c, hipMemcpyDeviceToHost));
  *n = c.shadow_rays_proved;
  return RT_HIP_OK;
}

extern "C" int rt_hip_last_wave_costs(rt_hip_scene *s, uint32_t *out, uint64_t cap, uint64_t *n_waves)
{
  if (!s || !n_waves) return fail(RT_HIP_EINVAL, "NULL argument");
  *n_waves = s->cost_sig[0];
  if (!out || s->cost_sig[0] == 0) return RT_HIP_OK;
  if (cap < s->cost_sig[0]) return fail(RT_HIP_EINVAL, "rt_hip_last_wave_costs: room for " + std::to_string(cap) + " entries, the frame had " + std::to_string(s->cost_sig[0]) + " waves");
  HIP_TRY(hipSetDevice(s->ctx->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, s->wave_cost.p, (size_t)s->cost_sig[0] * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return RT_HIP_OK;
}

#ifndef RT_MINIMAL
#include "rt_aa.inc"
#include "rt_ppm.inc"
#include "rt_godray.inc"
#include "rt_comm.inc"
#else
static int godray_level(rt_hip_scene *, PathNode *, unsigned int, const unsigned int *, hipStream_t) { return RT_HIP_OK; }
#endif
