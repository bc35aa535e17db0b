/*
 * rt_oracle.c -- CPU restatement of the reference hot path (see rt_oracle.h for the
 * role of this file: TEST INFRASTRUCTURE, parity pinned only by SURVEY.md 8(c) figures).
 *
 * Plain C99, IEEE double, build with -O2 -ffp-contract=off (oracle/Makefile).
 * Reference paths are relative to /root/reference/RayTracing/.
 *
 * Deliberate, documented departures from the reference text:
 *  - unary minus returns by value (the reference returns a reference to a local,
 *    Geometry.h:90-95, which is UB; intended semantics restated);
 *  - rand() is replaced by a counter-based generator keyed by
 *    (pixel, sample, path, light, light-sample, dimension) so that results do not
 *    depend on evaluation order or thread count;
 *  - CScene is passed by pointer, not copied per RayTracer call (RayTracer.h:8) --
 *    a cost difference only.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ constants */
static const double EPS = 1e-8;          /* Geometry.h:12 */
static const double PI_REF = 3.1415926535; /* Geometry.h:13 */
static const double INFTY = 1e+8;        /* Geometry.h:48 */

/* ------------------------------------------------------------------ CPoint3D (Geometry.h:16-138) */
typedef struct { double x, y, z; } v3;
typedef struct { double u, v; } v2;
typedef struct { v3 o, d; } ray_t;       /* CRay, Geometry.h:249-263 */
typedef struct { v3 mn, mx; } box_t;     /* CBoundingBox, Geometry.h:285-300 */

static inline v3 V3(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 v_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }       /* :24 */
static inline v3 v_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }       /* :25 */
static inline v3 v_mul(v3 a, double l) { return V3(a.x * l, a.y * l, a.z * l); }         /* :26 */
static inline v3 v_div(v3 a, double l) { return V3(a.x / l, a.y / l, a.z / l); }         /* :27 */
static inline v3 v_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }                            /* :90-95, by value */
static inline v3 v_cross(v3 a, v3 b)                                                      /* :51-56 */
{ return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline double v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }     /* :121-124 */
static inline double v_dist2(v3 a, v3 b)                                                  /* :133-138 */
{ return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y) + (a.z - b.z) * (a.z - b.z); }
static inline double v_dist(v3 a, v3 b) { return sqrt(v_dist2(a, b)); }                  /* :126-131 */
static inline double v_len(v3 a) { return v_dist(a, V3(0, 0, 0)); }                       /* :36 */
static inline double v_len2(v3 a) { return v_dist2(a, V3(0, 0, 0)); }                     /* :37 */
static inline int v_eq(v3 a, v3 b)                                                        /* :97-103 */
{ return fabs(a.x - b.x) < EPS && fabs(a.y - b.y) < EPS && fabs(a.z - b.z) < EPS; }
static inline v3 infty_pt(void) { return V3(INFTY, INFTY, INFTY); }                       /* :49 */
static inline int is_infty_pt(v3 a) { return v_eq(a, infty_pt()); }
static inline v3 v_unitize(v3 a)                                                          /* :113-119 */
{
  if (v_eq(a, V3(0, 0, 0)) || is_infty_pt(a)) return infty_pt();
  return V3(a.x / v_len(a), a.y / v_len(a), a.z / v_len(a));
}
static inline ray_t mk_ray(v3 o, v3 d) { ray_t r; r.o = o; r.d = v_unitize(d); return r; } /* :253-254 */

static inline int v2_eq(v2 a, v2 b) { return fabs(a.u - b.u) < EPS && fabs(a.v - b.v) < EPS; } /* :206-212 */
static inline v2 infty_pt2(void) { v2 r = {INFTY, INFTY}; return r; }                     /* :169 */

/* ------------------------------------------------------------------ CBoundingBox (Geometry.h:302-368) */
static double box_plane_t(ray_t ray, v3 p0, v3 norm)                                      /* :335-342 */
{
  if (fabs(v_dot(norm, ray.d)) < EPS) return INFTY;
  double D = -v_dot(p0, norm);
  return -(D + v_dot(norm, ray.o)) / v_dot(norm, ray.d);
}
static inline double dmin(double a, double b) { return (b < a) ? b : a; }  /* std::min */
static inline double dmax(double a, double b) { return (a < b) ? b : a; }  /* std::max */

static v2 box_collide(box_t b, ray_t ray)                                                 /* :302-333 */
{
  double t1, t2;
  t1 = box_plane_t(ray, b.mn, V3(1, 0, 0)); t2 = box_plane_t(ray, b.mx, V3(1, 0, 0));
  double t_x_min = dmin(t1, t2), t_x_max = dmax(t1, t2);
  t1 = box_plane_t(ray, b.mn, V3(0, 1, 0)); t2 = box_plane_t(ray, b.mx, V3(0, 1, 0));
  double t_y_min = dmin(t1, t2), t_y_max = dmax(t1, t2);
  t1 = box_plane_t(ray, b.mn, V3(0, 0, 1)); t2 = box_plane_t(ray, b.mx, V3(0, 0, 1));
  double t_z_min = dmin(t1, t2), t_z_max = dmax(t1, t2);
  double t_max = dmin(dmin(t_x_max, t_y_max), t_z_max);
  t_x_min = (t_x_min == INFTY) ? -INFTY : t_x_min;
  t_y_min = (t_y_min == INFTY) ? -INFTY : t_y_min;
  t_z_min = (t_z_min == INFTY) ? -INFTY : t_z_min;
  double t_min = dmax(dmax(t_x_min, t_y_min), t_z_min);
  v2 r;
  if (t_min < t_max + EPS) { r.u = t_min; r.v = t_max; return r; }
  return infty_pt2();
}
static inline int box_eq(box_t a, box_t b) { return v_eq(a.mn, b.mn) && v_eq(a.mx, b.mx); } /* :344-350 */
static inline box_t infty_box(void) { box_t b; b.mn = infty_pt(); b.mx = infty_pt(); return b; }

/* ------------------------------------------------------------------ CColour (Colour.h:7-63) */
typedef struct { double r, g, b; } col_t;
static inline col_t C3(double r, double g, double b) { col_t c = {r, g, b}; return c; }
static inline col_t c_add(col_t a, col_t b)                                               /* :26-29, :41-47 */
{ return C3(dmin(255.0, a.r + b.r), dmin(255.0, a.g + b.g), dmin(255.0, a.b + b.b)); }
static inline col_t c_mul(col_t a, double w)                                              /* :31-34 */
{ return C3(dmin(255.0, a.r * w), dmin(255.0, a.g * w), dmin(255.0, a.b * w)); }
static inline col_t c_div(col_t a, double w)                                              /* :36-39 */
{ return C3(dmin(255.0, a.r / w), dmin(255.0, a.g / w), dmin(255.0, a.b / w)); }

/* ------------------------------------------------------------------ CMaterial (Material.h:13-124) */
typedef struct { double reflW, refrW, RI, diff, spec, shiness, emissive, ambient; } mat_t;

void orc_material_preset(int name, double o[8])                                           /* :45-124 */
{
  double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  switch (name) {
    case ORC_MIRROR:    { double t[8] = {1, 0, 0, 0, 1, 2, 0, 0}; memcpy(m, t, sizeof m); } break;
    case ORC_GLASS:     { double t[8] = {0.05, 0.95, 1.52, 0, 0.05, 20, 0, 0}; memcpy(m, t, sizeof m); } break;
    case ORC_ICE:       { double t[8] = {0.05, 0.95, 1.31, 0, 0.05, 20, 0, 1}; memcpy(m, t, sizeof m); } break;
    case ORC_METAL:     { double t[8] = {0.7, 0, 0, 0.2, 0.8, 20, 0, 1}; memcpy(m, t, sizeof m); } break;
    case ORC_PORCELAIN: { double t[8] = {0.1, 0, 0, 1, 0.1, 2, 0, 1}; memcpy(m, t, sizeof m); } break;
    case ORC_RUBBER:    { double t[8] = {0, 0, 0, 0.3, 0.4, 1, 0, 1}; memcpy(m, t, sizeof m); } break;
    case ORC_PLASTIC:   { double t[8] = {0, 0, 0, 0.6, 0.0, 1, 0, 1}; memcpy(m, t, sizeof m); } break;
    default: break;
  }
  memcpy(o, m, sizeof m);
}

/* ------------------------------------------------------------------ objects (Object.h) */
typedef struct tex_s { int rows, cols; unsigned char *bgr; } tex_t;   /* cv::Mat CV_8UC3 */

typedef struct {
  int type;                 /* m_type */
  mat_t mat;
  col_t colour;
  box_t aabb;               /* m_AABB, infinite by default (Object.h:34) */
  /* sphere: c, r ; plane/circle: p0, n (unitized), r */
  v3 c; double r;
  v3 p0, n;
  /* triangle (Object.h:340-344) */
  v3 vtx[3], vnorm[3], m_norm; int have_vn;
  /* bezier patch (Object.h:1056-1062) */
  v3 ctrl[4][4]; v3 center;
  /* rotational (Object.h:800-808) */
  v2 prof[8]; int n_prof; int u_sample_num; double u_range;
  /* textures (Object.h:69-74, 342, 704-705) */
  const struct tex_s *texture, *bump_texture;   /* NULL = haveTexture / haveBumpTexture false */
  double texture_scale, bump_texture_scale;
  v2 tex_coord[3];
} obj_t;

typedef struct { int type; v3 pos_or_dir; double lum; double len; int sample_num; double lum_per_sample; } light_t;

typedef struct bsp_node {                 /* BSPNode, BSPTree.h:20-38 */
  box_t box;
  int *objs; int n_objs, cap_objs;
  int split_axis;
  struct bsp_node *left, *right;
} bsp_node;

struct orc_scene {                         /* CScene, Scene.h:12-60 */
  obj_t *objs; int n_objs, cap_objs;
  light_t *lights; int n_lights, cap_lights;
  double ambient;
  bsp_node *root;
  int max_obj_num, max_depth;
  tex_t **tex; int n_tex;                  /* images shared between objects (cv::Mat *, Scene.h:243) */
};

typedef struct { int64_t rays, nodes, prims, shadow; } counters_t;
static _Thread_local counters_t *tl_cnt = NULL;

/* ------------------------------------------------------------------ RNG (replaces rand(), Light.h:11) */
static inline uint64_t mix64(uint64_t z)
{
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
uint32_t orc_rng_u32(uint64_t seed, uint32_t key0, uint32_t key1, uint32_t counter)
{
  uint64_t x = mix64(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)key0 + 1));
  x = mix64(x ^ (0xD1B54A32D192ED03ULL * ((uint64_t)key1 + 1)));
  x = mix64(x ^ (0x8CB92BA72F3D8DD7ULL * ((uint64_t)counter + 1)));
  return (uint32_t)(x >> 32);
}
typedef struct { uint64_t seed; uint32_t pixel, sample, path; } rng_ctx;
/* counter layout shared with the device code: path(12) | light(4) | light-sample(14) | dim(2) */
static inline uint32_t rng_counter(uint32_t path, uint32_t light, uint32_t lsample, uint32_t dim)
{ return (path << 20) | ((light & 15u) << 16) | ((lsample & 0x3FFFu) << 2) | (dim & 3u); }
/* random(r, x) = 2 r (rand() % x) / x - r, Light.h:11; rand() := 31-bit draw */
static inline double ref_random(double r, int x, const rng_ctx *g, uint32_t light, uint32_t lsample, uint32_t dim)
{
  uint32_t k = orc_rng_u32(g->seed, g->pixel, g->sample, rng_counter(g->path, light, lsample, dim)) >> 1;
  return 2 * r * (double)(k % (uint32_t)x) / (double)x - r;
}

/* ------------------------------------------------------------------ scene building */
orc_scene *orc_scene_new(double ambient)
{
  orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
  s->ambient = ambient;
  s->max_obj_num = 100; s->max_depth = 40;     /* BSPTree.h:56-57 */
  return s;
}
static void free_node(bsp_node *n)
{
  if (!n) return;
  free_node(n->left); free_node(n->right);
  free(n->objs); free(n);
}
void orc_scene_free(orc_scene *s)
{
  if (!s) return;
  for (int i = 0; i < s->n_tex; i++) { free(s->tex[i]->bgr); free(s->tex[i]); }
  free(s->tex);
  free_node(s->root); free(s->objs); free(s->lights); free(s);
}
static obj_t *new_obj(orc_scene *s, int type, const double mat[8], const double col[3])
{
  if (s->n_objs == s->cap_objs) {
    s->cap_objs = s->cap_objs ? s->cap_objs * 2 : 1024;
    s->objs = (obj_t *)realloc(s->objs, (size_t)s->cap_objs * sizeof(obj_t));
  }
  obj_t *o = &s->objs[s->n_objs++];
  memset(o, 0, sizeof *o);
  o->type = type;
  memcpy(&o->mat, mat, sizeof(mat_t));
  o->colour = C3(col[0], col[1], col[2]);
  o->aabb = infty_box();                        /* Object.h:34 */
  return o;
}
int orc_num_objects(const orc_scene *s) { return s->n_objs; }

int orc_add_sphere(orc_scene *s, const double c[3], double r, const double mat[8], const double col[3])
{                                               /* Object.h:229-235 */
  obj_t *o = new_obj(s, ORC_SPHERE, mat, col);
  o->c = V3(c[0], c[1], c[2]); o->r = r;
  o->aabb.mn = v_sub(o->c, V3(r, r, r)); o->aabb.mx = v_add(o->c, V3(r, r, r));
  return s->n_objs - 1;
}
int orc_add_plane(orc_scene *s, const double p0[3], const double n[3], const double mat[8], const double col[3])
{                                               /* Object.h:685-689 */
  obj_t *o = new_obj(s, ORC_INFTY_PLANE, mat, col);
  o->p0 = V3(p0[0], p0[1], p0[2]); o->n = v_unitize(V3(n[0], n[1], n[2]));
  return s->n_objs - 1;
}
int orc_add_circle(orc_scene *s, const double p0[3], const double n[3], double r, const double mat[8], const double col[3])
{                                               /* Object.h:651-656: no AABB -> infinite */
  obj_t *o = new_obj(s, ORC_CIRCLE, mat, col);
  o->p0 = V3(p0[0], p0[1], p0[2]); o->n = v_unitize(V3(n[0], n[1], n[2])); o->r = r;
  return s->n_objs - 1;
}
static void tri_init_aabb(obj_t *o)              /* Object.h:459-468 */
{
  double x_min = dmin(dmin(o->vtx[0].x, o->vtx[1].x), o->vtx[2].x);
  double x_max = dmax(dmax(o->vtx[0].x, o->vtx[1].x), o->vtx[2].x);
  double y_min = dmin(dmin(o->vtx[0].y, o->vtx[1].y), o->vtx[2].y);
  double y_max = dmax(dmax(o->vtx[0].y, o->vtx[1].y), o->vtx[2].y);
  double z_min = dmin(dmin(o->vtx[0].z, o->vtx[1].z), o->vtx[2].z);
  double z_max = dmax(dmax(o->vtx[0].z, o->vtx[1].z), o->vtx[2].z);
  o->aabb.mn = V3(x_min - EPS, y_min - EPS, z_min - EPS);
  o->aabb.mx = V3(x_max + EPS, y_max + EPS, z_max + EPS);
}
int orc_add_triangle(orc_scene *s, const double v[9], const double vn[9], const double mat[8], const double col[3])
{                                               /* Object.h:360-384 */
  obj_t *o = new_obj(s, ORC_TRIANGLE, mat, col);
  for (int i = 0; i < 3; i++) o->vtx[i] = V3(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
  o->m_norm = v_unitize(v_cross(v_sub(o->vtx[1], o->vtx[0]), v_sub(o->vtx[2], o->vtx[0])));
  if (vn) { for (int i = 0; i < 3; i++) o->vnorm[i] = V3(vn[3 * i], vn[3 * i + 1], vn[3 * i + 2]); o->have_vn = 1; }
  else    { for (int i = 0; i < 3; i++) o->vnorm[i] = o->m_norm; o->have_vn = 0; }
  tri_init_aabb(o);
  return s->n_objs - 1;
}
int orc_get_triangle(const orc_scene *s, int id, double v[9], double vn[9])
{
  if (id < 0 || id >= s->n_objs || s->objs[id].type != ORC_TRIANGLE) return -1;
  const obj_t *o = &s->objs[id];
  for (int i = 0; i < 3; i++) {
    v[3 * i] = o->vtx[i].x; v[3 * i + 1] = o->vtx[i].y; v[3 * i + 2] = o->vtx[i].z;
    vn[3 * i] = o->vnorm[i].x; vn[3 * i + 1] = o->vnorm[i].y; vn[3 * i + 2] = o->vnorm[i].z;
  }
  return o->have_vn;
}

/* Parser.h:35-82 token loop + Scene.h:150-166 */
int orc_read_obj_triangle(orc_scene *s, const char *path, const double center[3], double scale,
                          const double mat[8], const double col[3])
{
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  size_t np = 0, cp = 0, nn = 0, cn = 0;
  v3 *pts = NULL, *nrm = NULL;
  int added = 0;
  char flag[4096]; flag[0] = 0;
  v3 ctr = V3(center[0], center[1], center[2]);
  for (;;) {
    /* "for (; objStream.good();) { objStream >> objFlag;": a failed extraction leaves objFlag unchanged */
    int got = fscanf(f, "%4095s", flag);
    int at_end = (got != 1);
    if (strcmp(flag, "v") == 0 || strcmp(flag, "vn") == 0) {
      double x = 0, y = 0, z = 0;
      if (!at_end) { if (fscanf(f, "%lf", &x) == 1) { if (fscanf(f, "%lf", &y) == 1) { if (fscanf(f, "%lf", &z) != 1) z = 0; } } }
      if (flag[1] == 0) { if (np == cp) { cp = cp ? cp * 2 : 1024; pts = (v3 *)realloc(pts, cp * sizeof(v3)); } pts[np++] = V3(x, y, z); }
      else              { if (nn == cn) { cn = cn ? cn * 2 : 1024; nrm = (v3 *)realloc(nrm, cn * sizeof(v3)); } nrm[nn++] = V3(x, y, z); }
    } else if (strcmp(flag, "vt") == 0) {
      double u = 0, v = 0;
      if (!at_end) { if (fscanf(f, "%lf", &u) == 1) { if (fscanf(f, "%lf", &v) != 1) v = 0; } }
    } else if (strcmp(flag, "f") == 0 && !at_end) {
      char line[2048];
      if (!fgets(line, sizeof line, f)) line[0] = 0;
      int vi[64], ni[64], cnt = 0, pos = 0, a, b, c, rd;
      while (cnt < 64 && sscanf(line + pos, "%i/%i/%i%n", &a, &b, &c, &rd) == 3) {   /* Parser.h:68 */
        pos += rd; vi[cnt] = a - 1; ni[cnt] = c - 1; cnt++;
      }
      if (cnt != 0) {
        /* Scene.h:153-165: every face becomes a CTriangle of its first three tuples (4-arg ctor, Object.h:373) */
        double v[9], vn[9];
        for (int k = 0; k < 3; k++) {
          v3 p = v_add(v_mul(pts[vi[k]], scale), ctr);
          v3 n = nrm[ni[k]];
          v[3 * k] = p.x; v[3 * k + 1] = p.y; v[3 * k + 2] = p.z;
          vn[3 * k] = n.x; vn[3 * k + 1] = n.y; vn[3 * k + 2] = n.z;
        }
        orc_add_triangle(s, v, vn, mat, col);
        added++;
      }
    }
    if (at_end) break;
  }
  fclose(f); free(pts); free(nrm);
  return added;
}

void orc_add_point_light(orc_scene *s, const double pos[3], double lum)
{
  if (s->n_lights == s->cap_lights) { s->cap_lights = s->cap_lights ? 2 * s->cap_lights : 8; s->lights = (light_t *)realloc(s->lights, (size_t)s->cap_lights * sizeof(light_t)); }
  light_t *l = &s->lights[s->n_lights++]; memset(l, 0, sizeof *l);
  l->type = ORC_POINT_LIGHT; l->pos_or_dir = V3(pos[0], pos[1], pos[2]); l->lum = lum;      /* Light.h:51-52 */
}
void orc_add_parallel_light(orc_scene *s, const double dir[3], double lum)
{
  if (s->n_lights == s->cap_lights) { s->cap_lights = s->cap_lights ? 2 * s->cap_lights : 8; s->lights = (light_t *)realloc(s->lights, (size_t)s->cap_lights * sizeof(light_t)); }
  light_t *l = &s->lights[s->n_lights++]; memset(l, 0, sizeof *l);
  l->type = ORC_PARALLEL_LIGHT; l->pos_or_dir = v_unitize(V3(dir[0], dir[1], dir[2])); l->lum = lum; /* Light.h:76-77 */
}
void orc_add_area_light(orc_scene *s, const double pos[3], double lum, double len, int sample_num)
{
  if (s->n_lights == s->cap_lights) { s->cap_lights = s->cap_lights ? 2 * s->cap_lights : 8; s->lights = (light_t *)realloc(s->lights, (size_t)s->cap_lights * sizeof(light_t)); }
  light_t *l = &s->lights[s->n_lights++]; memset(l, 0, sizeof *l);
  l->type = ORC_AREA_LIGHT; l->pos_or_dir = V3(pos[0], pos[1], pos[2]); l->lum = lum;       /* Light.h:101-108 */
  l->len = len; l->sample_num = sample_num;
  l->lum_per_sample = lum / (sample_num * sample_num);
}

/* ------------------------------------------------------------------ primitive intersection */
/* Eigen 3.x fixed-size 3x3 determinant (bruteforce_det3_helper), rows m[r][c] */
static inline double det3(double m00, double m01, double m02,
                          double m10, double m11, double m12,
                          double m20, double m21, double m22)
{
  return m00 * (m11 * m22 - m12 * m21) - m01 * (m10 * m22 - m12 * m20) + m02 * (m10 * m21 - m11 * m20);
}

typedef struct { v3 pt; double u, v; } hit_t;    /* point + surface parameters of the recorded hit */

static v3 tri_first_inter_pt(const obj_t *o, ray_t ray)     /* Object.h:416-446 */
{
  if (fabs(v_dot(ray.d, o->m_norm)) < EPS) return infty_pt();
  v3 E1 = v_sub(o->vtx[0], o->vtx[1]);
  v3 E2 = v_sub(o->vtx[0], o->vtx[2]);
  v3 S = v_sub(o->vtx[0], ray.o);
  double dA1 = det3(S.x, E1.x, E2.x, S.y, E1.y, E2.y, S.z, E1.z, E2.z);
  double dA2 = det3(ray.d.x, S.x, E2.x, ray.d.y, S.y, E2.y, ray.d.z, S.z, E2.z);
  double dA3 = det3(ray.d.x, E1.x, S.x, ray.d.y, E1.y, S.y, ray.d.z, E1.z, S.z);
  double dB = det3(ray.d.x, E1.x, E2.x, ray.d.y, E1.y, E2.y, ray.d.z, E1.z, E2.z);
  double t = dA1 / dB;
  double belta = dA2 / dB;
  double gamma = dA3 / dB;
  if (t > EPS && belta > -EPS && belta < 1 + EPS && gamma > -EPS && gamma < 1 + EPS && belta + gamma < 1 + EPS)
    return v_add(ray.o, v_mul(ray.d, t));
  return infty_pt();
}

static v3 sphere_first_inter_pt(const obj_t *o, ray_t ray)  /* Object.h:250-278 */
{
  v3 l = v_sub(o->c, ray.o);
  double tp = v_dot(l, ray.d);
  double d2 = v_len2(l) - tp * tp;
  double radius = o->r;
  if (d2 > radius * radius + EPS) return infty_pt();
  double t_prime = sqrt(radius * radius - d2);
  if (v_len2(l) < radius * radius - EPS) {
    double t = tp + t_prime;
    return v_add(ray.o, v_mul(ray.d, t));
  } else if (v_len2(l) > radius * radius + EPS) {
    double t = tp - t_prime;
    if (t > EPS) return v_add(ray.o, v_mul(ray.d, t));
    return infty_pt();
  } else {
    if (tp < EPS) return infty_pt();
    return v_add(ray.o, v_mul(ray.d, tp * 2));
  }
}

static v3 plane_first_inter_pt(const obj_t *o, ray_t ray)   /* Object.h:708-723 */
{
  if (fabs(v_dot(o->n, ray.d)) < EPS) {
    if (fabs(v_dot(v_sub(o->p0, ray.o), o->n)) < EPS) return ray.o;
    return infty_pt();
  }
  double D = -v_dot(o->p0, o->n);
  double t = -(D + v_dot(o->n, ray.o)) / v_dot(o->n, ray.d);
  if (t < EPS) return infty_pt();
  return v_add(ray.o, v_mul(ray.d, t));
}

static v3 circle_first_inter_pt(const obj_t *o, ray_t ray)  /* Object.h:667-678 */
{
  if (fabs(v_dot(o->n, ray.d)) < EPS) return infty_pt();
  double D = -v_dot(o->p0, o->n);
  double t = -(D + v_dot(o->n, ray.o)) / v_dot(o->n, ray.d);
  v3 pt = v_add(ray.o, v_mul(ray.d, t));
  if (t < EPS || v_dist(pt, o->p0) > o->r - EPS) return infty_pt();
  return pt;
}


/* ------------------------------------------------------------------ Bezier curve / surfaces (Bezier.h) */
static inline v2 p2_add(v2 a, v2 b) { v2 r = {a.u + b.u, a.v + b.v}; return r; }       /* Geometry.h:149 */
static inline v2 p2_sub(v2 a, v2 b) { v2 r = {a.u - b.u, a.v - b.v}; return r; }       /* :150 */
static inline v2 p2_mul(v2 a, double l) { v2 r = {a.u * l, a.v * l}; return r; }       /* :151 */
static inline v2 p2_unitize(v2 a)                                                      /* :219-225 */
{
  v2 z = {0, 0};
  if (v2_eq(a, z) || v2_eq(a, infty_pt2())) return infty_pt2();
  double len = sqrt((a.u - 0) * (a.u - 0) + (a.v - 0) * (a.v - 0));                    /* Len() = Dist(*this, (0,0)), :232-236 */
  v2 r = {a.u / len, a.v / len};
  return r;
}
/* CBezierCurve::Point2D(t, i, deg), de Casteljau by recursion (Bezier.h:28-34) */
static v2 curve_pt(const obj_t *o, double t, int i, int deg)
{
  if (deg == 0) return o->prof[i];
  return p2_add(p2_mul(curve_pt(o, t, i, deg - 1), 1 - t), p2_mul(curve_pt(o, t, i + 1, deg - 1), t));
}
static v2 curve_point2d(const obj_t *o, double t) { return curve_pt(o, t, 0, o->n_prof - 1); }       /* :16 */
static v2 curve_tan(const obj_t *o, double t)                                                        /* :36-39 */
{ int deg = o->n_prof - 1; return p2_mul(p2_sub(curve_pt(o, t, 1, deg - 1), curve_pt(o, t, 0, deg - 1)), deg); }
static v2 curve_norm(const obj_t *o, double t)                                                       /* :41-45 */
{ v2 tv = curve_tan(o, t); v2 n = {tv.v, -tv.u}; return p2_unitize(n); }
/* CRotationalSurface (Bezier.h:65-87) */
/* sin / cos / acos of the surfaces of revolution: NOT libm's.  The angle is unbounded (Object.h:864, 870 commented out), two Newton seeds can reach
   the same point 2 pi k apart and the nearer root is decided in the last bits of t' -- so the checker and the kernels must round the angle's
   sine and cosine alike.  Same operations as ray_tracing_amd/csrc/rt_math.hpp: rt_sincos / rt_acos (+ - x / floor sqrt only; Cody-Waite
   reduction by a three-part pi/2, the polynomial kernels of fdlibm's k_sin.c / k_cos.c / e_acos.c).  <= 2 ulp from glibc's. */
static void det_sincos(double x, double *s_out, double *c_out)
{
  const double n = floor(x * 6.36619772367581382433e-01 + 0.5);
  double r = ((x - n * 1.57079632673412561417e+00) - n * 6.07710050630396597660e-11) - n * 2.02226624871116645580e-21;
  r = r - n * 8.47842766036889956997e-32;
  const double z = r * r;
  const double s = r + r * z * (-1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                   z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
  const double c = 1.0 - 0.5 * z + z * z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 +
                   z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double q = n - 4.0 * floor(n * 0.25);
  if (q == 0.0) { *s_out = s; *c_out = c; }
  else if (q == 1.0) { *s_out = c; *c_out = -s; }
  else if (q == 2.0) { *s_out = -s; *c_out = -c; }
  else { *s_out = -c; *c_out = s; }
}
static double det_sin(double x) { double s, c; det_sincos(x, &s, &c); return s; }
static double det_cos(double x) { double s, c; det_sincos(x, &s, &c); return c; }
static double det_acos_ratio(double z)
{
  const double p = z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 + z * (2.01212532134862925881e-01 + z * (-4.00555345006794114027e-02 +
                   z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
  const double q = 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 + z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
  return p / q;
}
static double det_acos(double x)
{
  const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
  const double ax = fabs(x);
  if (!(ax < 1.0)) {
    if (x == 1.0) return 0.0;
    if (x == -1.0) return pi + 2.0 * pio2_lo;
    return (x - x) / (x - x);
  }
  if (ax < 0.5) {
    if (ax < 6.9388939039072284e-18) return pio2_hi + pio2_lo;
    return pio2_hi - (x - (pio2_lo - x * det_acos_ratio(x * x)));
  }
  if (x < 0) {
    const double z = (1.0 + x) * 0.5, s = sqrt(z), w = det_acos_ratio(z) * s - pio2_lo;
    return pi - 2.0 * (s + w);
  }
  { const double z = (1.0 - x) * 0.5, s = sqrt(z), w = det_acos_ratio(z) * s; return 2.0 * (s + w); }
}
static v3 rot_point(const obj_t *o, double t, double th) { v2 p = curve_point2d(o, t); return V3(p.u * det_cos(th), p.u * det_sin(th), p.v); }
static v3 rot_norm(const obj_t *o, double t, double th) { v2 r = curve_norm(o, t); return v_unitize(V3(r.u * det_cos(th), r.u * det_sin(th), r.v)); }
static v3 rot_ds_dt(const obj_t *o, double t, double th) { v2 d = curve_tan(o, t); return V3(det_cos(th) * d.u, det_sin(th) * d.u, d.v); }
static v3 rot_ds_dth(const obj_t *o, double t, double th) { v2 p = curve_point2d(o, t); return V3(-(p.u) * det_sin(th), p.u * det_cos(th), 0); }
/* CBezierSurface::Point3D(u, v, i, j, k, l), m = n = 3 (Bezier.h:116-128) */
static v3 surf_pt(const obj_t *o, double u, double v, int i, int j, int k, int l)
{
  const int m = 3, n = 3;
  if (k == 0 && l == 0) return o->ctrl[i][j];
  else if (l == 0) return v_add(v_mul(surf_pt(o, u, v, i, j, k - 1, 0), 1 - u), v_mul(surf_pt(o, u, v, i + 1, j, k - 1, 0), u));
  else if (k == m) return v_add(v_mul(surf_pt(o, u, v, 0, j, m, l - 1), 1 - v), v_mul(surf_pt(o, u, v, 0, j + 1, m, l - 1), v));
  else if (k == 0) return v_add(v_mul(surf_pt(o, u, v, i, j, 0, l - 1), 1 - v), v_mul(surf_pt(o, u, v, i, j + 1, 0, l - 1), v));
  else if (l == n) return v_add(v_mul(surf_pt(o, u, v, i, 0, k - 1, n), 1 - u), v_mul(surf_pt(o, u, v, i + 1, 0, k - 1, n), u));
  return infty_pt();   /* not reached for m = n = 3 */
}
static v3 surf_point(const obj_t *o, double u, double v) { return surf_pt(o, u, v, 0, 0, 3, 3); }     /* :98 */
static v3 surf_du(const obj_t *o, double u, double v)                                                 /* :130-133 */
{ return v_mul(v_sub(surf_pt(o, u, v, 1, 0, 2, 3), surf_pt(o, u, v, 0, 0, 2, 3)), 3); }
static v3 surf_dv(const obj_t *o, double u, double v)                                                 /* :135-138 */
{ return v_mul(v_sub(surf_pt(o, u, v, 0, 1, 3, 2), surf_pt(o, u, v, 0, 0, 3, 2)), 3); }
static v3 surf_norm(const obj_t *o, double u, double v) { return v_unitize(v_cross(surf_du(o, u, v), surf_dv(o, u, v))); }  /* :99 */

static const int MAX_STEP_NUM = 10;           /* Object.h:831 */
static const double MIN_STEP_LEN = 1e-8;      /* :832 (= eps) */
static const double MAX_DEVIATION = 0.3;      /* :833 */

/* CRotationalObject::t_v_Initial (Object.h:972-1037): seed (t, theta) from the plane / sphere through the profile point */
static v2 rot_t_v_initial(const obj_t *o, double u, ray_t ray)
{
  v2 pt2 = curve_point2d(o, u);
  v3 p0 = V3(o->center.x, o->center.y, o->center.z + pt2.v);
  v3 n = V3(0, 0, 1);
  double v = 0;
  if (fabs(v_dot(n, ray.d)) < 0.1) {
    v3 p = V3(0, 0, 0);
    double t = 0;
    v3 l = v_sub(p0, ray.o);
    double tp = v_dot(l, ray.d);
    double d2 = v_len2(l) - tp * tp;
    if (d2 > pt2.u * pt2.u + EPS) return infty_pt2();
    double t_prime = sqrt(pt2.u * pt2.u - d2);
    if (v_len2(l) < pt2.u * pt2.u - EPS * 1000) { t = tp + t_prime; p = v_add(ray.o, v_mul(ray.d, t)); }
    else if (v_len2(l) > pt2.u * pt2.u + EPS * 1000) {
      t = tp - t_prime;
      if (t > EPS) p = v_add(ray.o, v_mul(ray.d, t));
      else return infty_pt2();
    } else {
      if (tp < EPS * 1000) return infty_pt2();
      t = tp * 2; p = v_add(ray.o, v_mul(ray.d, tp * 2));
    }
    double cos_theta = v_sub(p, p0).x / v_len(v_sub(p, p0));
    double theta = det_acos(cos_theta);
    if (p.y > p0.y || (p.y == p0.y && p.x > p0.x)) v = theta; else v = 2 * PI_REF - theta;
    v2 r = {t, v};
    return r;
  }
  double D = -v_dot(p0, n);
  double t = -(D + v_dot(n, ray.o)) / v_dot(n, ray.d);
  v3 p = v_add(ray.o, v_mul(ray.d, t));
  if (t < EPS) return infty_pt2();
  double cos_theta = v_sub(p, p0).x / v_len(v_sub(p, p0));
  double theta = det_acos(cos_theta);
  if (p.y > p0.y || (p.y == p0.y && p.x > p0.x)) v = theta; else v = 2 * PI_REF - theta;
  v2 r = {t, v};
  return r;
}

/* CRotationalObject::FirstInterPt / _Unrecorded (Object.h:834-952); they differ in the t' threshold only */
static hit_t rot_first_inter_pt(const obj_t *o, ray_t ray, int record)
{
  hit_t h; h.pt = infty_pt(); h.u = INFTY; h.v = INFTY;
  if (v2_eq(box_collide(o->aabb, ray), infty_pt2())) return h;
  double cur_u = INFTY, cur_v = INFTY, cur_t = INFTY;
  int have = 0;
  const double t_thr = record ? EPS * 10 : EPS * 5000;
  double u_int = o->u_range / (double)o->u_sample_num;
  for (int i = 0; i <= o->u_sample_num; i++) {
    double u = i * u_int;
    v2 tv = rot_t_v_initial(o, u, ray);
    if (v2_eq(tv, infty_pt2())) continue;
    double v = tv.v, t = tv.u;
    v3 w = ray.d;
    for (int step = 0; step < MAX_STEP_NUM; step++) {
      v3 dPu = rot_ds_dt(o, u, v), dPv = rot_ds_dth(o, u, v);
      double D = v_dot(w, v_cross(dPu, dPv));
      v3 df = v_sub(v_add(o->center, rot_point(o, u, v)), v_add(ray.o, v_mul(ray.d, t)));
      double t_p = t + v_dot(dPu, v_cross(dPv, df)) / D;
      double u_p = u + v_dot(w, v_cross(dPv, df)) / D;
      double v_p = v - v_dot(w, v_cross(dPu, df)) / D;
      if (u_p < -MAX_DEVIATION || u_p > 1 + MAX_DEVIATION || t_p < -MAX_DEVIATION) break;
      if (v_len2(V3(t_p - t, u_p - u, v_p - v)) < MIN_STEP_LEN && t_p > t_thr && 0 < u_p && u_p < o->u_range) {
        have = 1;
        if (t_p < cur_t) { cur_u = u_p; cur_v = v_p; cur_t = t_p; break; }
      }
      u = u_p; v = v_p; t = t_p;
    }
  }
  if (have) { h.pt = v_add(o->center, rot_point(o, cur_u, cur_v)); h.u = cur_u; h.v = cur_v; }
  return h;
}

/* CBezierObject::FirstInterPt / _Unrecorded (Object.h:1087-1202): identical arithmetic */
static hit_t bezier_first_inter_pt(const obj_t *o, ray_t ray)
{
  hit_t h; h.pt = infty_pt(); h.u = INFTY; h.v = INFTY;
  double cur_u = INFTY, cur_v = INFTY, cur_t = INFTY;
  int have = 0;
  double u_int = 1 / (double)4, v_int = 1 / (double)4;       /* uSampleNum2 = vSampleNum2 = 4, :1085-1086 */
  for (int i = 0; i <= 4; i++) {
    for (int j = 0; j <= 4; j++) {
      double u = i * u_int, v = j * v_int;
      double t = v_dist(v_add(surf_point(o, u, v), o->center), ray.o);
      v3 w = ray.d;
      for (int step = 0; step < MAX_STEP_NUM; step++) {
        v3 dPu = surf_du(o, u, v), dPv = surf_dv(o, u, v);
        double D = v_dot(w, v_cross(dPu, dPv));
        v3 df = v_sub(v_add(o->center, surf_point(o, u, v)), v_add(ray.o, v_mul(ray.d, t)));
        double t_p = t + v_dot(dPu, v_cross(dPv, df)) / D;
        double u_p = u + v_dot(w, v_cross(dPv, df)) / D;
        double v_p = v - v_dot(w, v_cross(dPu, df)) / D;
        if (u_p < -MAX_DEVIATION || u_p > 1 + MAX_DEVIATION || v_p < -MAX_DEVIATION || v_p > 1 + MAX_DEVIATION || t_p < -MAX_DEVIATION) break;
        if (v_len2(V3(t_p - t, u_p - u, v_p - v)) < MIN_STEP_LEN && t_p > EPS && 0 < u_p && u_p < 1 && 0 < v_p && v_p < 1) {
          have = 1;
          if (t_p < cur_t - EPS) { cur_u = u_p; cur_v = v_p; cur_t = t_p; break; }
        }
        u = u_p; v = v_p; t = t_p;
      }
    }
  }
  if (have) { h.pt = v_add(o->center, surf_point(o, cur_u, cur_v)); h.u = cur_u; h.v = cur_v; }
  return h;
}

int orc_add_bezier(orc_scene *s, const double ctrl[48], const double center[3], const double mat[8], const double col[3])
{                                               /* Object.h:1075-1083, AABB :1204-1227 */
  obj_t *o = new_obj(s, ORC_BEZIER, mat, col);
  o->center = V3(center[0], center[1], center[2]);
  v3 mn = V3(INFTY, INFTY, INFTY), mx = V3(-INFTY, -INFTY, -INFTY);
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
    v3 c = V3(ctrl[(i * 4 + j) * 3], ctrl[(i * 4 + j) * 3 + 1], ctrl[(i * 4 + j) * 3 + 2]);
    o->ctrl[i][j] = c;
    mn = V3(dmin(mn.x, c.x), dmin(mn.y, c.y), dmin(mn.z, c.z));
    mx = V3(dmax(mx.x, c.x), dmax(mx.y, c.y), dmax(mx.z, c.z));
  }
  o->aabb.mn = v_add(mn, o->center); o->aabb.mx = v_add(mx, o->center);
  return s->n_objs - 1;
}

int orc_add_rotational(orc_scene *s, const double *prof, int n_prof, const double center[3], int u_sample_num, double u_range,
                       const double mat[8], const double col[3])
{                                               /* Object.h:818-829, AABB :954-970 */
  if (n_prof < 2 || n_prof > 8 || u_sample_num < 1) return -1;
  obj_t *o = new_obj(s, ORC_ROTATIONAL, mat, col);
  o->center = V3(center[0], center[1], center[2]);
  o->n_prof = n_prof; o->u_sample_num = u_sample_num; o->u_range = u_range;
  double r = 0;
  for (int i = 0; i < n_prof; i++) { o->prof[i].u = prof[2 * i]; o->prof[i].v = prof[2 * i + 1]; if (o->prof[i].u > r) r = o->prof[i].u; }
  double z_max = o->center.z + dmax(o->prof[0].v, o->prof[n_prof - 1].v);
  double z_min = o->center.z + dmin(o->prof[0].v, o->prof[n_prof - 1].v);
  o->aabb.mn = V3(o->center.x - r - EPS, o->center.y - r - EPS, z_min - EPS);
  o->aabb.mx = V3(o->center.x + r + EPS, o->center.y + r + EPS, z_max + EPS);
  return s->n_objs - 1;
}

/* virtual FirstInterPt / FirstInterPt_Unrecorded dispatch (Object.h:37-38) */
static hit_t obj_first_inter_pt(const obj_t *o, ray_t ray, int record)
{
  hit_t h; h.u = INFTY; h.v = INFTY;
  if (tl_cnt) tl_cnt->prims++;
  switch (o->type) {
    case ORC_BEZIER:      return bezier_first_inter_pt(o, ray);
    case ORC_ROTATIONAL:  return rot_first_inter_pt(o, ray, record);
    case ORC_TRIANGLE:    h.pt = tri_first_inter_pt(o, ray); break;
    case ORC_SPHERE:      h.pt = sphere_first_inter_pt(o, ray); break;
    case ORC_INFTY_PLANE: h.pt = plane_first_inter_pt(o, ray); break;
    case ORC_CIRCLE:      h.pt = circle_first_inter_pt(o, ray); break;
    default:              h.pt = infty_pt(); break;
  }
  return h;
}

/* ------------------------------------------------------------------ normals, reflect/refract, Fresnel */
static v3 obj_initial_norm(const obj_t *o, v3 pt, double u, double v)
{
  switch (o->type) {
    case ORC_BEZIER: return surf_norm(o, u, v);                                           /* Object.h:1049 */
    case ORC_ROTATIONAL: return rot_norm(o, u, v);                                        /* Object.h:799 */
    case ORC_SPHERE: return v_unitize(v_sub(pt, o->c));                                   /* Object.h:239 */
    case ORC_TRIANGLE: {                                                                  /* Object.h:448-457 */
      if (!o->have_vn) return o->m_norm;
      double S0 = v_len(v_cross(v_sub(o->vtx[1], pt), v_sub(o->vtx[2], pt)));
      double S1 = v_len(v_cross(v_sub(o->vtx[0], pt), v_sub(o->vtx[2], pt)));
      double S2 = v_len(v_cross(v_sub(o->vtx[0], pt), v_sub(o->vtx[1], pt)));
      double S = v_len(v_cross(v_sub(o->vtx[1], o->vtx[0]), v_sub(o->vtx[2], o->vtx[0])));
      return v_add(v_add(v_mul(o->vnorm[0], S0 / S), v_mul(o->vnorm[1], S1 / S)), v_mul(o->vnorm[2], S2 / S));
    }
    case ORC_CIRCLE: case ORC_INFTY_PLANE: return o->n;                                   /* Object.h:660,693 */
    default: return infty_pt();
  }
}
/* ---- textures (Object.h:79-92, 186-196 and the per-type UV_colour / UV_norm / Tangent / Bitangent) ----
 * texture->at<cv::Vec3b>(row, col): cv::Mat::at does not check bounds, and the reference's indices do leave the image
 * (row == rows at a sphere's pole since PI is 3.1415926535): clamped here.  On a surface of revolution the angle of a
 * hit is whatever real number Newton's iteration ended on (negative, beyond 2 PI): its column wraps around instead. */
static const unsigned char *texel_at(const tex_t *t, int row, int col)
{
  row = row < 0 ? 0 : (row >= t->rows ? t->rows - 1 : row);
  col = col < 0 ? 0 : (col >= t->cols ? t->cols - 1 : col);
  return t->bgr + ((size_t)row * t->cols + col) * 3;
}
static int to_int(double x)          /* int(double), defined for every input */
{
  if (!(x > -2147483648.0)) return -2147483647 - 1;
  if (!(x < 2147483647.0)) return 2147483647;
  return (int)x;
}
static v3 plane_u0(v3 n)                                                                  /* Object.h:727-734 */
{
  v3 a = V3(1, 0, 0), b = V3(0, 1, 0), c = V3(0, 0, 1);
  if (fabs(v_dot(a, n)) < 0.5) return v_unitize(v_sub(a, v_mul(n, v_dot(a, n))));
  if (fabs(v_dot(b, n)) < 0.5) return v_unitize(v_sub(b, v_mul(n, v_dot(b, n))));
  return v_unitize(v_sub(c, v_mul(n, v_dot(c, n))));
}
static v2 tri_uv(const obj_t *o, v3 pt)                                                   /* Object.h:470-486 */
{
  double S0 = v_len(v_cross(v_sub(o->vtx[1], pt), v_sub(o->vtx[2], pt)));
  double S1 = v_len(v_cross(v_sub(o->vtx[0], pt), v_sub(o->vtx[2], pt)));
  double S2 = v_len(v_cross(v_sub(o->vtx[0], pt), v_sub(o->vtx[1], pt)));
  double S = v_len(v_cross(v_sub(o->vtx[1], o->vtx[0]), v_sub(o->vtx[2], o->vtx[0])));
  return p2_add(p2_add(p2_mul(o->tex_coord[0], S0 / S), p2_mul(o->tex_coord[1], S1 / S)), p2_mul(o->tex_coord[2], S2 / S));
}
static v2 sphere_uv(const obj_t *o, v3 pt)                                                /* Object.h:289-296 */
{
  v2 r = {0, 0};
  if (pt.x == 0 && pt.y == 0) return r;
  v3 q = v_sub(pt, o->c);
  r.v = (atan2(q.y, q.x) + PI_REF) / (2 * PI_REF);
  r.u = (asin(q.z / o->r) + PI_REF / 2) / PI_REF;
  return r;
}
static col_t obj_colour(const obj_t *o, v3 pt, double hu, double hv)                      /* Object.h:186-196 */
{
  if (!o->texture) return o->colour;
  const tex_t *T = o->texture;
  v2 uv = {0, 0};                                                                         /* Object.h:61 */
  switch (o->type) {
    case ORC_SPHERE: { v2 n = sphere_uv(o, pt); uv.u = to_int(n.u * T->rows); uv.v = to_int(n.v * T->cols); } break;   /* :280-287 */
    case ORC_TRIANGLE: uv = tri_uv(o, pt); break;
    case ORC_INFTY_PLANE: {                                                               /* :725-742 */
      v3 u0 = plane_u0(o->n), v0 = v_cross(o->n, u0), rel = v_sub(pt, o->p0);
      int up = to_int(v_dot(rel, u0) * o->texture_scale) % T->rows, vp = to_int(v_dot(rel, v0) * o->texture_scale) % T->cols;
      up = (up + T->rows) % T->rows; vp = (vp + T->cols) % T->cols;
      uv.u = up; uv.v = vp;
    } break;
    case ORC_ROTATIONAL: {                                                                /* :810; theta is any real: the column wraps */
      int vc = to_int(hv * T->cols / (2 * PI_REF)) % T->cols;
      uv.u = to_int(hu * T->rows); uv.v = (vc + T->cols) % T->cols;
    } break;
    case ORC_BEZIER: uv.u = to_int(hu * T->rows); uv.v = to_int(hv * T->cols); break;                  /* :1051 */
    default: break;
  }
  const unsigned char *c = texel_at(T, to_int(uv.u), to_int(uv.v));                       /* :193 */
  return C3(c[2], c[1], c[0]);                                                            /* :194 */
}
/* CBaseObject::Norm, Object.h:79-92 */
static v3 obj_norm(const obj_t *o, v3 pt, double u, double v)
{
  v3 N = obj_initial_norm(o, pt, u, v);
  if (!o->bump_texture) return N;
  const tex_t *T = o->bump_texture;
  v2 uv = {0, 0}; v3 tan = V3(0, 0, 0), bit = V3(0, 0, 0);                                /* Object.h:62-64 */
  switch (o->type) {
    case ORC_SPHERE: {                                                                    /* :289-319 */
      uv = sphere_uv(o, pt);
      double uu = uv.u, vv = uv.v, r = o->r;
      tan = v_unitize(V3(-2 * PI_REF * r * sin(2 * PI_REF * vv - PI_REF) * cos(PI_REF * uu - PI_REF / 2), 2 * PI_REF * r * cos(2 * PI_REF * vv - PI_REF) * cos(PI_REF * uu - PI_REF / 2), 0));
      bit = v_unitize(V3(-PI_REF * r * cos(2 * PI_REF * vv - PI_REF) * sin(PI_REF * uu - PI_REF / 2), -PI_REF * r * sin(2 * PI_REF * vv - PI_REF) * sin(PI_REF * uu - PI_REF / 2),
                         PI_REF * r * cos(PI_REF * uu - PI_REF / 2)));
    } break;
    case ORC_TRIANGLE: uv = tri_uv(o, pt); break;                                         /* :479-486 */
    case ORC_INFTY_PLANE: {                                                               /* :744-771, 700 */
      v3 u0 = plane_u0(o->n), v0 = v_cross(o->n, u0), rel = v_sub(pt, o->p0);
      double a = v_dot(rel, u0) / o->bump_texture_scale, b = v_dot(rel, v0) / o->bump_texture_scale;
      uv.u = a - floor(a); uv.v = b - floor(b);
      tan = u0; bit = v_cross(o->n, u0);
    } break;
    case ORC_ROTATIONAL:                                                                  /* :811-813 */
      uv.u = u; uv.v = v / (2 * PI_REF); uv.v = uv.v - floor(uv.v);                        /* theta is any real: one turn = one image width */
      tan = v_unitize(rot_ds_dth(o, u, v)); bit = v_unitize(rot_ds_dt(o, u, v));
      break;
    case ORC_BEZIER:                                                                      /* :1052-1054 */
      uv.u = u; uv.v = v;
      tan = v_unitize(surf_du(o, u, v)); bit = v_unitize(surf_dv(o, u, v));
      break;
    default: break;
  }
  const unsigned char *c = texel_at(T, to_int(uv.u * T->rows), to_int(uv.v * T->cols));   /* :87 */
  return v_unitize(v_add(v_add(v_mul(bit, c[2] / 127.5 - 1), v_mul(tan, c[1] / 127.5 - 1)), v_mul(N, c[0] / 127.5 - 1)));   /* :88-90 */
}

/* registering images and attaching them (Object.h:198-210, 773-787; Scene.h:243) */
int orc_add_texture_image(orc_scene *s, int rows, int cols, const unsigned char *bgr)
{
  if (rows <= 0 || cols <= 0 || !bgr) return -1;
  tex_t *t = (tex_t *)malloc(sizeof *t);
  t->rows = rows; t->cols = cols; t->bgr = (unsigned char *)malloc((size_t)rows * cols * 3);
  memcpy(t->bgr, bgr, (size_t)rows * cols * 3);
  s->tex = (tex_t **)realloc(s->tex, (size_t)(s->n_tex + 1) * sizeof *s->tex);
  s->tex[s->n_tex] = t;
  return s->n_tex++;
}
int orc_obj_add_texture(orc_scene *s, int obj, int tex, double scale)
{
  if (obj < 0 || obj >= s->n_objs || tex < 0 || tex >= s->n_tex) return -1;
  s->objs[obj].texture = s->tex[tex]; s->objs[obj].texture_scale = scale;
  return 0;
}
int orc_obj_add_bump_texture(orc_scene *s, int obj, int tex, double scale)
{
  if (obj < 0 || obj >= s->n_objs || tex < 0 || tex >= s->n_tex) return -1;
  s->objs[obj].bump_texture = s->tex[tex]; s->objs[obj].bump_texture_scale = scale;
  return 0;
}
int orc_triangle_set_tex_coord(orc_scene *s, int obj, const double tc[6])
{
  if (obj < 0 || obj >= s->n_objs || s->objs[obj].type != ORC_TRIANGLE) return -1;
  for (int i = 0; i < 3; i++) { s->objs[obj].tex_coord[i].u = tc[2 * i]; s->objs[obj].tex_coord[i].v = tc[2 * i + 1]; }
  return 0;
}

static ray_t obj_reflect_ray(const obj_t *o, ray_t ray, v3 pt, double u, double v)        /* Object.h:94-101 */
{
  v3 N = obj_norm(o, pt, u, v);
  if (v_dot(ray.d, N) > EPS) N = v_neg(N);
  v3 dir = v_sub(ray.d, v_mul(N, 2 * v_dot(ray.d, N)));
  return mk_ray(pt, v_unitize(dir));
}
static ray_t obj_refract_ray(const obj_t *o, ray_t ray, v3 pt, double u, double v)        /* Object.h:103-126 */
{
  v3 N = obj_norm(o, pt, u, v);
  v3 T = infty_pt();
  if (v_dot(ray.d, N) < EPS) {
    double ni_nt = 1 / o->mat.RI;
    double cos_theta_i2 = v_dot(ray.d, N) * v_dot(ray.d, N);
    double cos_theta_t = sqrt(1 - ni_nt * ni_nt * (1 - cos_theta_i2));
    T = v_add(v_mul(N, ni_nt * v_dot(v_neg(ray.d), N) - cos_theta_t), v_mul(ray.d, ni_nt));
  } else {
    N = v_neg(N);
    double ni_nt = o->mat.RI;
    double cos_theta_i2 = v_dot(ray.d, N) * v_dot(ray.d, N);
    if (1 - ni_nt * ni_nt * (1 - cos_theta_i2) > EPS) {
      double cos_theta_t = sqrt(1 - ni_nt * ni_nt * (1 - cos_theta_i2));
      T = v_add(v_mul(N, ni_nt * v_dot(v_neg(ray.d), N) - cos_theta_t), v_mul(ray.d, ni_nt));
    }
  }
  return mk_ray(pt, v_unitize(T));
}
/* common part of reflW/refrW(ray, pt), Object.h:128-184: returns (r_p^2 + r_v^2)/2 */
static double fresnel_R(const obj_t *o, ray_t ray, v3 pt, double u, double v)
{
  v3 N = obj_norm(o, pt, u, v);
  double ni_nt = 0, cos_theta_i = 0, cos_theta_t = 0;
  if (v_dot(ray.d, N) < EPS) {
    ni_nt = 1 / o->mat.RI;
    double cos_theta_i2 = v_dot(ray.d, N) * v_dot(ray.d, N);
    cos_theta_i = sqrt(cos_theta_i2);
    cos_theta_t = sqrt(1 - ni_nt * ni_nt * (1 - cos_theta_i2));
  } else {
    N = v_neg(N);
    ni_nt = o->mat.RI;
    double cos_theta_i2 = v_dot(ray.d, N) * v_dot(ray.d, N);
    cos_theta_i = sqrt(cos_theta_i2);
    if (1 - ni_nt * ni_nt * (1 - cos_theta_i2) > EPS)
      cos_theta_t = sqrt(1 - ni_nt * ni_nt * (1 - cos_theta_i2));
  }
  double r_p = (ni_nt * cos_theta_i - cos_theta_t) / (ni_nt * cos_theta_i + cos_theta_t);
  double r_v = (cos_theta_i - ni_nt * cos_theta_t) / (cos_theta_i + ni_nt * cos_theta_t);
  return (r_p * r_p + r_v * r_v) / 2;
}
static double obj_reflW(const obj_t *o, ray_t ray, v3 pt, double u, double v)             /* Object.h:128-155 */
{
  if (o->mat.reflW == 0 || o->mat.refrW == 0) return o->mat.reflW;
  return fresnel_R(o, ray, pt, u, v);
}
static double obj_refrW(const obj_t *o, ray_t ray, v3 pt, double u, double v)             /* Object.h:157-184 */
{
  if (o->mat.reflW == 0 || o->mat.refrW == 0) return o->mat.refrW;
  return 1 - fresnel_R(o, ray, pt, u, v);
}

/* ------------------------------------------------------------------ BSP tree build (BSPTree.h:58-109) */
static int obj_point_in_box(const obj_t *o, box_t box)       /* Object.h:212-222 */
{
  if (box_eq(box, infty_box())) return 1;
  box_t b = o->aabb;
  if (b.mn.x > box.mx.x + EPS || b.mx.x < box.mn.x - EPS ||
      b.mn.y > box.mx.y + EPS || b.mx.y < box.mn.y - EPS ||
      b.mn.z > box.mx.z + EPS || b.mx.z < box.mn.z - EPS)
    return 0;
  return 1;
}
static bsp_node *new_node(box_t box)
{
  bsp_node *n = (bsp_node *)calloc(1, sizeof *n);
  n->box = box; n->split_axis = -1;
  return n;
}
static void node_push(bsp_node *n, int id)
{
  if (n->n_objs == n->cap_objs) { n->cap_objs = n->cap_objs ? n->cap_objs * 2 : 16; n->objs = (int *)realloc(n->objs, (size_t)n->cap_objs * sizeof(int)); }
  n->objs[n->n_objs++] = id;
}
static void subdivide(orc_scene *s, bsp_node *node, int depth, int cur_axis)
{
  if (node->n_objs < s->max_obj_num || depth == s->max_depth) return;       /* :60 */
  node->split_axis = cur_axis;
  int next_axis = 0;
  box_t left, right;
  v3 mn = node->box.mn, mx = node->box.mx;
  if (cur_axis == 0) {                                                          /* :69-76 */
    left.mn = mn; left.mx = V3((mn.x + mx.x) / 2, mx.y, mx.z);
    right.mn = V3((mn.x + mx.x) / 2, mn.y, mn.z); right.mx = mx;
    next_axis = 1;
  } else if (cur_axis == 1) {                                                   /* :77-84 */
    left.mn = mn; left.mx = V3(mx.x, (mn.y + mx.y) / 2, mx.z);
    right.mn = V3(mn.x, (mn.y + mx.y) / 2, mn.z); right.mx = mx;
    next_axis = 2;
  } else {                                                                      /* :85-92 */
    left.mn = mn; left.mx = V3(mx.x, mx.y, (mn.z + mx.z) / 2);
    right.mn = V3(mn.x, mn.y, (mn.z + mx.z) / 2); right.mx = mx;
    next_axis = 0;
  }
  node->left = new_node(left); node->right = new_node(right);
  int *infty_objs = NULL; int n_inf = 0, cap_inf = 0;
  for (int i = 0; i < node->n_objs; i++) {                                      /* :94-102 */
    int id = node->objs[i];
    const obj_t *o = &s->objs[id];
    if (obj_point_in_box(o, left)) node_push(node->left, id);
    if (obj_point_in_box(o, right)) node_push(node->right, id);
    if (node == s->root && box_eq(o->aabb, infty_box())) {
      if (n_inf == cap_inf) { cap_inf = cap_inf ? cap_inf * 2 : 8; infty_objs = (int *)realloc(infty_objs, (size_t)cap_inf * sizeof(int)); }
      infty_objs[n_inf++] = id;
    }
  }
  free(node->objs); node->objs = NULL; node->n_objs = node->cap_objs = 0;      /* :103 */
  if (node == s->root) { node->objs = infty_objs; node->n_objs = node->cap_objs = n_inf; } /* :104-105 */
  else free(infty_objs);
  subdivide(s, node->left, depth + 1, next_axis);
  subdivide(s, node->right, depth + 1, next_axis);
}
static box_t biggest_aabb(const orc_scene *s)                /* Scene.h:295-317 */
{
  double x_min = INFTY, x_max = -INFTY, y_min = INFTY, y_max = -INFTY, z_min = INFTY, z_max = -INFTY;
  for (int i = 0; i < s->n_objs; i++) {
    box_t b = s->objs[i].aabb;
    if (box_eq(b, infty_box())) continue;
    if (b.mn.x < x_min) x_min = b.mn.x;
    if (b.mn.y < y_min) y_min = b.mn.y;
    if (b.mn.z < z_min) z_min = b.mn.z;
    if (b.mx.x > x_max) x_max = b.mx.x;
    if (b.mx.y > y_max) y_max = b.mx.y;
    if (b.mx.z > z_max) z_max = b.mx.z;
  }
  if (x_min != INFTY && y_min != INFTY && z_min != INFTY && x_max != -INFTY && y_max != -INFTY && z_max != INFTY) {
    box_t r; r.mn = V3(x_min, y_min, z_min); r.mx = V3(x_max, y_max, z_max); return r;
  }
  return infty_box();
}
void orc_construct_bsp(orc_scene *s, int max_obj_num, int max_depth)  /* Scene.h:286-293 */
{
  free_node(s->root); s->root = NULL;
  s->max_obj_num = max_obj_num > 0 ? max_obj_num : 100;
  s->max_depth = max_depth > 0 ? max_depth : 40;
  box_t big = biggest_aabb(s);
  if (box_eq(big, infty_box())) return;
  s->root = new_node(big);
  for (int i = 0; i < s->n_objs; i++) node_push(s->root, i);
  subdivide(s, s->root, 0, 0);
}
static void stats_rec(const bsp_node *n, int depth, int64_t st[8], int is_root)
{
  st[0]++;
  if (depth > st[4]) st[4] = depth;
  if (!n->left && !n->right) {
    st[1]++;
    if (n->n_objs == 0) st[2]++;
    st[3] += n->n_objs;
    if (n->n_objs > st[5]) st[5] = n->n_objs;
  } else {
    (void)is_root;
    stats_rec(n->left, depth + 1, st, 0); stats_rec(n->right, depth + 1, st, 0);
  }
}
void orc_tree_stats(const orc_scene *s, int64_t st[8], double root_box[6])
{
  memset(st, 0, 8 * sizeof(int64_t));
  if (!s->root) return;
  st[7] = 1;
  stats_rec(s->root, 0, st, 1);
  st[6] = (s->root->left || s->root->right) ? s->root->n_objs : 0;
  if (root_box) {
    root_box[0] = s->root->box.mn.x; root_box[1] = s->root->box.mn.y; root_box[2] = s->root->box.mn.z;
    root_box[3] = s->root->box.mx.x; root_box[4] = s->root->box.mx.y; root_box[5] = s->root->box.mx.z;
  }
}

/* CScene::RotationalSurfaceToMesh (Scene.h:168-213) and, with a texture, RotationalSurfaceToMesh_Texture (:215-266) */
int orc_rotational_surface_to_mesh(orc_scene *s, const double *prof, int n_prof, int u_num, int v_num, int tex,
                                   const double center[3], const double mat[8], const double col[3])
{
  if (n_prof < 2 || n_prof > 8 || u_num < 1 || v_num < 1 || tex >= s->n_tex) return -1;
  obj_t surf; memset(&surf, 0, sizeof surf);
  surf.n_prof = n_prof;
  for (int i = 0; i < n_prof; i++) { surf.prof[i].u = prof[2 * i]; surf.prof[i].v = prof[2 * i + 1]; }
  const v3 c = V3(center[0], center[1], center[2]);
  const double du = 1 / (double)u_num, dv = 2 * PI_REF / (double)v_num;
  const double rows = tex >= 0 ? s->tex[tex]->rows : 0, cols = tex >= 0 ? s->tex[tex]->cols : 0;
  size_t cap = (size_t)(u_num + 3) * (size_t)(v_num + 3), n_pts = 0;
  v3 *points = (v3 *)malloc(cap * sizeof(v3)), *normals = (v3 *)malloc(cap * sizeof(v3));
  long pt_id = 0; int added = 0;
  for (double u = 0, i = 0; u < 1 + EPS; u += du, i++) {
    if (tex < 0 && fabs(u - 1) < 100 * EPS) u = 1;                                        /* :177-178 */
    for (double v = 0, j = 0; v < 2 * PI_REF + EPS; v += dv, j++) {
      if (tex < 0 && fabs(v - 2 * PI_REF) < 100 * EPS) v = 2 * PI_REF;                    /* :181-182 */
      if (n_pts == cap) { cap *= 2; points = (v3 *)realloc(points, cap * sizeof(v3)); normals = (v3 *)realloc(normals, cap * sizeof(v3)); }
      /* (the mesh generator keeps the C library's sin / cos, like the host mirror it is compared with bit for bit: its output is input geometry) */
      { v2 pp = curve_point2d(&surf, u), rr = curve_norm(&surf, u);
        points[n_pts] = V3(pp.u * cos(v), pp.u * sin(v), pp.v); normals[n_pts] = v_unitize(V3(rr.u * cos(v), rr.u * sin(v), rr.v)); n_pts++; }
      if (i != 0 && j != 0 && pt_id - v_num - 2 >= 0) {
        const long a = pt_id, b = pt_id - 1, cc = pt_id - v_num - 1, d = pt_id - v_num - 2;
        const long idx[2][3] = {{a, b, cc}, {b, d, cc}};
        const double tc[2][6] = {
          {u * rows, v * cols / (2 * PI_REF), u * rows, (v - dv) * cols / (2 * PI_REF), (u - du) * rows, v * cols / (2 * PI_REF)},
          {u * rows, (v - dv) * cols / (2 * PI_REF), (u - du) * rows, (v - dv) * cols / (2 * PI_REF), (u - du) * rows, v * cols / (2 * PI_REF)}};
        for (int k = 0; k < 2; k++) {
          double vv[9], vn[9];
          for (int m = 0; m < 3; m++) {
            v3 q = v_add(points[idx[k][m]], c), nq = normals[idx[k][m]];
            vv[3 * m] = q.x; vv[3 * m + 1] = q.y; vv[3 * m + 2] = q.z; vn[3 * m] = nq.x; vn[3 * m + 1] = nq.y; vn[3 * m + 2] = nq.z;
          }
          int id = orc_add_triangle(s, vv, vn, mat, col);
          if (tex >= 0) { orc_triangle_set_tex_coord(s, id, tc[k]); orc_obj_add_texture(s, id, tex, 1); }
          added++;
        }
      }
      pt_id++;
    }
  }
  free(points); free(normals);
  return added;
}

/* ------------------------------------------------------------------ closest hit (Scene.h:62-91, 319-432) */
typedef struct { v3 pt; int obj; double u, v; } inter_t;     /* CIntersection, Scene.h:23-28 */
static inline inter_t no_inter(void) { inter_t r; r.pt = infty_pt(); r.obj = -1; r.u = r.v = INFTY; return r; }

/* the candidate-update block repeated at Scene.h:65-81, 329-346, 352-368, 369-386 */
static inline void consider(orc_scene *s, int id, ray_t ray, int record, inter_t *best)
{
  hit_t h = obj_first_inter_pt(&s->objs[id], ray, record);
  if (!is_infty_pt(h.pt)) {
    if (is_infty_pt(best->pt) || v_dist(h.pt, ray.o) < v_dist(best->pt, ray.o) - EPS) {
      best->pt = h.pt; best->obj = id; best->u = h.u; best->v = h.v;
    }
  }
}
static inter_t first_inter_pt(orc_scene *s, ray_t ray, int record)     /* Scene.h:62-83 */
{
  inter_t best = no_inter();
  for (int i = 0; i < s->n_objs; i++) consider(s, i, ray, record, &best);
  return best;
}
static double scene_plane_t(ray_t ray, v3 p0, v3 norm)                /* Scene.h:421-432 */
{
  if (fabs(v_dot(norm, ray.d)) < EPS) {
    if (fabs(v_dot(v_sub(p0, ray.o), norm)) <= EPS) return 0;
    return INFTY;
  }
  double D = -v_dot(p0, norm);
  return -(D + v_dot(norm, ray.o)) / v_dot(norm, ray.d);
}
static inter_t ray_tree_intersect(orc_scene *s, ray_t ray, bsp_node *node, int record)   /* Scene.h:319-419 */
{
  if (tl_cnt) tl_cnt->nodes++;
  v2 tmm = box_collide(node->box, ray);
  double t_min = tmm.u, t_max = tmm.v;
  if (v2_eq(tmm, infty_pt2())) {                                         /* :326-348 */
    inter_t best = no_inter();
    for (int i = 0; i < s->root->n_objs; i++) consider(s, s->root->objs[i], ray, record, &best);
    return best;
  }
  if (!node->left && !node->right) {                                     /* :349-388 */
    inter_t best = no_inter();
    for (int i = 0; i < node->n_objs; i++) consider(s, node->objs[i], ray, record, &best);
    for (int i = 0; i < s->root->n_objs; i++) consider(s, s->root->objs[i], ray, record, &best);
    return best;
  }
  v3 p0 = v_div(v_add(node->box.mn, node->box.mx), 2);                   /* :389 */
  v3 normal = V3(0, 0, 0);
  if (node->split_axis == 0) normal.x = 1;
  if (node->split_axis == 1) normal.y = 1;
  if (node->split_axis == 2) normal.z = 1;
  double dist = scene_plane_t(ray, p0, normal);
  bsp_node *near_n, *far_n;
  if (v_dot(v_sub(p0, ray.o), normal) > 0) { near_n = node->left; far_n = node->right; }
  else { near_n = node->right; far_n = node->left; }
  if (dist > t_max + EPS || dist < -EPS)                                  /* :406 */
    return ray_tree_intersect(s, ray, near_n, record);
  else if (dist < t_min - EPS && fabs(dist) > EPS)                        /* :408 */
    return ray_tree_intersect(s, ray, far_n, record);
  else {
    inter_t i1 = ray_tree_intersect(s, ray, near_n, record);
    inter_t i2 = ray_tree_intersect(s, ray, far_n, record);
    if (v_dist2(i1.pt, ray.o) < v_dist2(i2.pt, ray.o) - EPS) return i1;   /* :414 */
    return i2;
  }
}
static inter_t first_inter_pt_bsp(orc_scene *s, ray_t ray, int record)  /* Scene.h:85-91 */
{
  if (tl_cnt) tl_cnt->rays++;
  if (!s->root) return first_inter_pt(s, ray, record);
  return ray_tree_intersect(s, ray, s->root, record);
}

/* ------------------------------------------------------------------ lights (Light.h, Scene.h:436-455) */
static v3 light_direction(const light_t *l, v3 pt)
{
  if (l->type == ORC_PARALLEL_LIGHT) return l->pos_or_dir;               /* Light.h:80 */
  return v_unitize(v_sub(pt, l->pos_or_dir));                            /* Light.h:55,111 */
}
static double light_luminosity(orc_scene *s, const light_t *l, int light_idx, v3 pt, const rng_ctx *g)
{
  if (l->type == ORC_POINT_LIGHT) {                                      /* Light.h:34-44 + 62-69 */
    for (int i = 0; i < s->n_objs; i++) {
      ray_t ray = mk_ray(pt, v_sub(l->pos_or_dir, pt));
      if (tl_cnt) tl_cnt->shadow++;
      v3 ip = obj_first_inter_pt(&s->objs[i], ray, 0).pt;
      if (!is_infty_pt(ip) && v_dot(v_sub(pt, ip), v_sub(l->pos_or_dir, ip)) < -EPS) return 0;
    }
    return l->lum;
  }
  if (l->type == ORC_PARALLEL_LIGHT) {                                   /* Light.h:34-44 + 87-94 */
    for (int i = 0; i < s->n_objs; i++) {
      ray_t ray = mk_ray(pt, v_neg(l->pos_or_dir));
      if (tl_cnt) tl_cnt->shadow++;
      v3 ip = obj_first_inter_pt(&s->objs[i], ray, 0).pt;
      if (!is_infty_pt(ip)) return 0;
    }
    return l->lum;
  }
  /* area light: Scene.h:436-455 (tree) or Light.h:126-162 (no tree) */
  double L = 0;
  int sn = l->sample_num;
  double r = (l->len / (double)sn) / 2;
  v3 O = V3(l->pos_or_dir.x - sn * r, l->pos_or_dir.y - sn * r, l->pos_or_dir.z);
  for (int i = 0; i < sn; i++) {
    for (int j = 0; j < sn; j++) {
      double x0 = O.x + i * r * 2, y0 = O.y + j * r * 2, z0 = O.z;
      uint32_t ls = (uint32_t)(i * sn + j);
      /* draw order pinned: x first, then y */
      double rx = ref_random(r, 1000, g, (uint32_t)light_idx, ls, 0);
      double ry = ref_random(r, 1000, g, (uint32_t)light_idx, ls, 1);
      v3 sample_pt = V3(x0 + rx, y0 + ry, z0);
      ray_t ray = mk_ray(pt, v_sub(sample_pt, pt));
      if (tl_cnt) tl_cnt->shadow++;
      if (s->root) {                                                      /* Scene.h:448-451 */
        v3 ip = first_inter_pt_bsp(s, ray, 0).pt;
        if (!is_infty_pt(ip) && v_dot(v_sub(pt, ip), v_sub(l->pos_or_dir, ip)) < -EPS) continue;
        L += l->lum_per_sample;
      } else {                                                            /* Light.h:152-162 + 143-150 */
        double Ls = l->lum_per_sample;
        for (int k = 0; k < s->n_objs; k++) {
          v3 ip = obj_first_inter_pt(&s->objs[k], ray, 0).pt;
          if (!is_infty_pt(ip) && v_dot(v_sub(pt, ip), v_sub(l->pos_or_dir, ip)) < -EPS) { Ls = 0; break; }
        }
        L += Ls;
      }
    }
  }
  return L;
}

/* ------------------------------------------------------------------ shading (Scene.h:93-122, RayTracer.h:8-55) */
static col_t local_illumination(orc_scene *s, ray_t ray, inter_t in, const rng_ctx *g)
{
  const obj_t *o = &s->objs[in.obj];
  col_t colour = C3(0, 0, 0);
  v3 norm = obj_norm(o, in.pt, in.u, in.v);
  if (v_dot(ray.d, norm) > EPS) norm = v_neg(norm);
  col_t obj_colour_ = obj_colour(o, in.pt, in.u, in.v);                  /* Scene.h:101 */
  for (int li = 0; li < s->n_lights; li++) {
    const light_t *l = &s->lights[li];
    double L = light_luminosity(s, l, li, in.pt, g);
    ray_t reflect = obj_reflect_ray(o, ray, in.pt, in.u, in.v);
    double emissive = o->mat.emissive;
    v3 ldir = light_direction(l, in.pt);
    double diffuse = o->mat.diff * dmax(0.0, -v_dot(norm, ldir) * L);
    double specular = o->mat.spec * L * pow(dmax(0.0, -v_dot(ldir, reflect.d)), o->mat.shiness);
    colour = c_add(colour, c_mul(obj_colour_, emissive + s->ambient * o->mat.ambient + diffuse + specular));
  }
  return colour;
}

/* CHitpoint, PhotonMapper.h:14-26 (+ the position in the ray tree, used as a key by the tests) */
typedef struct { v3 pt, norm, omega; int h, v; double wgt, R, N, M; col_t tau; uint32_t pixel, path; } hitpoint_t;
typedef struct { hitpoint_t *a; size_t n, cap; int W; int key_sample; } hp_sink_t;   /* key_sample: path key = (sample << 12) | path (anti-aliased eye pass) */
static _Thread_local hp_sink_t *tl_hp = NULL;      /* non-NULL: ray_tracer is CPhotonMapper::RayTracingPass (PhotonMapper.h:159-220) */

struct orc_godray;
static _Thread_local const struct orc_godray *tl_god = NULL;   /* non-NULL: ray_tracer is CGodRay::RayTracingPass (GodRay.h:142-191) */
static col_t god_radiance(const struct orc_godray *g, ray_t ray);

static const double THRESHOLD = 0.1;                                      /* RayTracer.h:7 */
static col_t ray_tracer(orc_scene *s, ray_t ray, int depth, double weight, col_t bg, rng_ctx g,
                        int32_t *first_id)
{
  col_t colour = C3(0, 0, 0);
  if (weight < THRESHOLD) return bg;
  inter_t in = first_inter_pt_bsp(s, ray, 1);
  if (first_id) *first_id = in.obj;
  if (in.obj < 0) return bg;
  const obj_t *o = &s->objs[in.obj];
  if (tl_god) colour = c_add(colour, god_radiance(tl_god, ray));           /* GodRay.h:155 */
  colour = c_add(colour, local_illumination(s, ray, in, &g));            /* GodRay.h:157 / RayTracer.h:21 (0 + x: the clamp is the same) */
  if (tl_hp && o->mat.reflW == 0 && o->mat.refrW == 0) {                  /* PhotonMapper.h:174-186 */
    hp_sink_t *k = tl_hp;
    if (k->n == k->cap) { k->cap = k->cap ? k->cap * 2 : 1024; k->a = (hitpoint_t *)realloc(k->a, k->cap * sizeof(hitpoint_t)); }
    hitpoint_t *hp = &k->a[k->n++];
    memset(hp, 0, sizeof *hp);
    hp->pt = in.pt;
    hp->norm = obj_norm(o, in.pt, in.u, in.v);
    if (v_dot(ray.d, hp->norm) > EPS) hp->norm = v_neg(hp->norm);
    hp->omega = ray.d;
    hp->h = (int)(g.pixel % (uint32_t)k->W); hp->v = (int)(g.pixel / (uint32_t)k->W);
    hp->wgt = weight; hp->R = 1; hp->N = 0; hp->M = 0; hp->tau = C3(0, 0, 0);      /* R_initial, :10, :25 */
    hp->pixel = g.pixel; hp->path = k->key_sample ? ((g.sample << 12) | g.path) : g.path;
  }
  if (depth > 1) {
    int total_reflect = 0;
    double refrW = obj_refrW(o, ray, in.pt, in.u, in.v);
    double reflW = obj_reflW(o, ray, in.pt, in.u, in.v);
    if (refrW != 0) {
      ray_t rr = obj_refract_ray(o, ray, in.pt, in.u, in.v);
      if (!is_infty_pt(rr.d)) {
        rng_ctx g2 = g; g2.path = g.path * 2;
        col_t rc = ray_tracer(s, rr, depth - 1, weight * refrW, bg, g2, NULL);
        colour = c_add(colour, c_mul(rc, refrW));
      } else total_reflect = 1;
    }
    if (reflW != 0) {
      ray_t rl = obj_reflect_ray(o, ray, in.pt, in.u, in.v);
      rng_ctx g2 = g; g2.path = g.path * 2 + 1;
      if (!total_reflect) {
        col_t rc = ray_tracer(s, rl, depth - 1, weight * reflW, bg, g2, NULL);
        colour = c_add(colour, c_mul(rc, reflW));
      } else {
        col_t rc = ray_tracer(s, rl, depth - 1, weight * (reflW + refrW), bg, g2, NULL);
        colour = c_add(colour, c_mul(rc, reflW + refrW));
      }
    }
  }
  return colour;
}

/* ------------------------------------------------------------------ camera (Camera.h:17-28, 53-75, 171-199) */
typedef struct { v3 view_pt, eye, center, up, left, start; double hor, ver; int res, depth, W, H; } cam_t;
static cam_t cam_init(const orc_camera *c)
{
  cam_t k;
  v3 v_eye = V3(c->eye_dir[0], c->eye_dir[1], c->eye_dir[2]);
  v3 v_left = V3(c->left_arg[0], c->left_arg[1], c->left_arg[2]);
  k.view_pt = V3(c->view_pt[0], c->view_pt[1], c->view_pt[2]);
  k.eye = v_unitize(v_eye);
  k.up = v_unitize(v_cross(v_left, v_eye));                               /* :19, un-normalised eye */
  k.left = v_unitize(v_cross(k.up, k.eye));                               /* :21 */
  k.hor = c->hor_range; k.ver = c->ver_range; k.res = c->resolution; k.depth = c->depth;
  k.center = v_add(k.view_pt, v_mul(k.eye, c->dist_of_plane));            /* :27 */
  k.W = (int)(k.hor * k.res); k.H = (int)(k.ver * k.res);                 /* :55-56 */
  k.start = v_add(v_add(k.center, v_mul(k.left, k.hor / 2)), v_mul(k.up, k.ver / 2)); /* :57 */
  return k;
}
static inline v3 cam_cur_pt(const cam_t *k, int h, int v)                 /* :65 */
{
  return v_sub(v_sub(k->start, v_mul(k->left, (double)h / (double)k->res)), v_mul(k->up, (double)v / (double)k->res));
}
void orc_camera_size(const orc_camera *c, int *w, int *h) { cam_t k = cam_init(c); *w = k.W; *h = k.H; }
void orc_camera_rays(const orc_camera *c, double *rays)
{
  cam_t k = cam_init(c);
  for (int v = 0; v < k.H; v++)
    for (int h = 0; h < k.W; h++) {
      v3 cur = cam_cur_pt(&k, h, v);
      ray_t r = mk_ray(k.view_pt, v_unitize(v_sub(cur, k.view_pt)));       /* :67 */
      double *o = rays + ((size_t)v * k.W + h) * 6;
      o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; o[3] = r.d.x; o[4] = r.d.y; o[5] = r.d.z;
    }
}

int orc_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static void add_counters(int64_t *dst, const counters_t *c)
{
  if (!dst) return;
#ifdef _OPENMP
#pragma omp critical
#endif
  { dst[0] += c->rays; dst[1] += c->nodes; dst[2] += c->prims; dst[3] += c->shadow; }
}

void orc_render_whitted(orc_scene *s, const orc_camera *c, const double bg[3], uint64_t seed,
                        double *rgb, int32_t *prim_id, size_t first, size_t count, int threads,
                        int64_t *counters)
{
  cam_t k = cam_init(c);
  size_t total = (size_t)k.W * k.H;
  if (count == 0) { first = 0; count = total; }
  col_t bgc = C3(bg[0], bg[1], bg[2]);
  if (counters) memset(counters, 0, 4 * sizeof(int64_t));
#ifdef _OPENMP
  int nt = threads > 0 ? threads : omp_get_max_threads();
#pragma omp parallel num_threads(nt)
#endif
  {
    counters_t cnt = {0, 0, 0, 0};
    tl_cnt = counters ? &cnt : NULL;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 256)
#endif
    for (long long p = (long long)first; p < (long long)(first + count); p++) {
      int v = (int)(p / k.W), h = (int)(p % k.W);
      v3 cur = cam_cur_pt(&k, h, v);
      ray_t ray = mk_ray(k.view_pt, v_unitize(v_sub(cur, k.view_pt)));     /* Camera.h:67 */
      rng_ctx g; g.seed = seed; g.pixel = (uint32_t)p; g.sample = 0; g.path = 1;
      int32_t id = -1;
      col_t col = ray_tracer(s, ray, k.depth, 1, bgc, g, &id);             /* :68 */
      size_t q = (size_t)p - first;
      rgb[3 * q] = col.r; rgb[3 * q + 1] = col.g; rgb[3 * q + 2] = col.b;
      if (prim_id) prim_id[q] = id;
    }
    tl_cnt = NULL;
    if (counters) add_counters(counters, &cnt);
  }
  (void)threads;
}

void orc_render_focusing(orc_scene *s, const orc_camera *c, double radius, int spp, const double bg[3],
                         uint64_t seed, double *rgb, size_t first, size_t count, int threads,
                         int64_t *counters)
{
  cam_t k = cam_init(c);
  size_t total = (size_t)k.W * k.H;
  if (count == 0) { first = 0; count = total; }
  col_t bgc = C3(bg[0], bg[1], bg[2]);
  if (counters) memset(counters, 0, 4 * sizeof(int64_t));
#ifdef _OPENMP
  int nt = threads > 0 ? threads : omp_get_max_threads();
#pragma omp parallel num_threads(nt)
#endif
  {
    counters_t cnt = {0, 0, 0, 0};
    tl_cnt = counters ? &cnt : NULL;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
    for (long long p = (long long)first; p < (long long)(first + count); p++) {
      int v = (int)(p / k.W), h = (int)(p % k.W);
      v3 cur = cam_cur_pt(&k, h, v);
      col_t colour = C3(0, 0, 0);
      for (int i = 0; i < spp; i++) {                                      /* Camera.h:187-192 */
        rng_ctx g; g.seed = seed; g.pixel = (uint32_t)p; g.sample = (uint32_t)i; g.path = 0;
        /* lens draws: path 0, dims 0,1,2 = x,y,z (draw order pinned x,y,z) */
        double jx = ref_random(radius, 1000, &g, 0, 0, 0);
        double jy = ref_random(radius, 1000, &g, 0, 0, 1);
        double jz = ref_random(radius, 1000, &g, 0, 0, 2);
        v3 pnt = V3(k.view_pt.x + jx, k.view_pt.y + jy, k.view_pt.z + jz);
        ray_t ray = mk_ray(pnt, v_unitize(v_sub(cur, pnt)));
        g.path = 1;
        col_t sc = ray_tracer(s, ray, k.depth, 1, bgc, g, NULL);
        colour = c_add(colour, c_div(sc, spp));
      }
      size_t q = (size_t)p - first;
      rgb[3 * q] = colour.r; rgb[3 * q + 1] = colour.g; rgb[3 * q + 2] = colour.b;
    }
    tl_cnt = NULL;
    if (counters) add_counters(counters, &cnt);
  }
  (void)threads;
}

/* ------------------------------------------------------------------ adaptive anti-aliasing (Camera.h:77-169, 201-289) */
/* cv::filter2D(mat, grad, CV_64F, [0 1 0; 1 -4 1; 0 1 0]) with its default border, BORDER_REFLECT_101 (OpenCV 2.4.9 is
 * not in the image: restated from its documentation; products accumulated in the kernel's row-major order) */
static double laplace_at(const double *m, int W, int H, int v, int h)       /* Camera.h:282-289 */
{
  int vu = (v == 0) ? ((H > 1) ? 1 : 0) : v - 1, vd = (v == H - 1) ? ((H > 1) ? H - 2 : 0) : v + 1;
  int hl = (h == 0) ? ((W > 1) ? 1 : 0) : h - 1, hr = (h == W - 1) ? ((W > 1) ? W - 2 : 0) : h + 1;
  double s = m[(size_t)vu * W + h];
  s += m[(size_t)v * W + hl]; s += m[(size_t)v * W + h] * -4.0; s += m[(size_t)v * W + hr]; s += m[(size_t)vd * W + h];
  return s;
}

/* aa[6] = oversampleTimes, r0, depth / norm / type / colour thresholds (Camera.h:77-82).  focusing = 0:
 * Render_RayTracer_Antialiasing (Camera.h:83-169); focusing = 1: Render_Focusing_Antialiasing (Camera.h:201-280) with
 * spp lens samples of the given radius.  A primary ray that hits nothing (NULL dereference in the reference,
 * Camera.h:109) counts with depth = Dist(inftyPt, viewPt), a zero normal and type -1.
 * RNG keys: lens point (pixel, lens sample i, path 0, stream 0, dims 0..2); picture-plane jitter (pixel,
 * ray index i * oversampleTimes + j, path 0, stream 1, dims 0..2); the ray tree below uses (pixel, ray index). */
/* the maps of Camera.h:89-121 (207-232, 321-352, 456-487): depth, normal, type (and, unless focusing, the RayTracer colour) per pixel */
static double *aa_maps(orc_scene *s, const cam_t *kp, int focusing, col_t bgc, uint64_t seed, int threads)
{
  const cam_t k = *kp;
  const int W = k.W, H = k.H;
  const size_t total = (size_t)W * H;
  double *maps = (double *)malloc(total * 8 * sizeof(double));
#ifdef _OPENMP
  int nt = threads > 0 ? threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 256) num_threads(nt)
#endif
  for (long long p = 0; p < (long long)total; p++) {                      /* Camera.h:98-121, 213-232 */
    int v = (int)(p / W), h = (int)(p % W);
    v3 cur = cam_cur_pt(&k, h, v);
    ray_t ray = mk_ray(k.view_pt, v_unitize(v_sub(cur, k.view_pt)));
    inter_t in = first_inter_pt(s, ray, 1);                              /* :106 */
    v3 norm = V3(0, 0, 0); double t, type = -1;
    if (in.obj >= 0) {
      norm = obj_norm(&s->objs[in.obj], in.pt, in.u, in.v);              /* :109 */
      type = (double)s->objs[in.obj].type;                                /* :115 */
    }
    t = v_dist(in.pt, ray.o);                                             /* :110 */
    maps[p] = t; maps[total + p] = norm.x; maps[2 * total + p] = norm.y; maps[3 * total + p] = norm.z; maps[4 * total + p] = type;
    if (!focusing) {
      rng_ctx g; g.seed = seed; g.pixel = (uint32_t)p; g.sample = 0; g.path = 1;
      const struct orc_godray *god = tl_god; hp_sink_t *sink = tl_hp;
      tl_god = NULL; tl_hp = NULL;                                        /* the maps come from the plain RayTracer (:116, :348, :483) */
      col_t c0 = ray_tracer(s, ray, k.depth, 1, bgc, g, NULL);           /* :116 */
      tl_god = god; tl_hp = sink;
      maps[5 * total + p] = c0.r; maps[6 * total + p] = c0.g; maps[7 * total + p] = c0.b;
    }
  }
  (void)H;
  return maps;
}
/* the test of Camera.h:137-146 (246-251, 373-382, 504-513) */
static int aa_edge(const double *maps, int W, int H, int v, int h, const double aa[6], int focusing)
{
  const size_t total = (size_t)W * H;
  int edge = fabs(laplace_at(maps, W, H, v, h)) > aa[2] ||
             fabs(laplace_at(maps + total, W, H, v, h)) > aa[3] ||
             fabs(laplace_at(maps + 2 * total, W, H, v, h)) > aa[3] ||
             fabs(laplace_at(maps + 3 * total, W, H, v, h)) > aa[3] ||
             fabs(laplace_at(maps + 4 * total, W, H, v, h)) > aa[4];
  if (!focusing)
    edge = edge || fabs(laplace_at(maps + 5 * total, W, H, v, h)) > aa[5] ||
           fabs(laplace_at(maps + 6 * total, W, H, v, h)) > aa[5] ||
           fabs(laplace_at(maps + 7 * total, W, H, v, h)) > aa[5];
  return edge;
}
/* one picture-plane-jittered ray of a masked pixel: ray index `idx` (Camera.h:152-156); returns RayTracer / RayTracingPass of it */
static col_t aa_jittered_sample(orc_scene *s, const cam_t *k, v3 pnt, v3 cur, uint32_t p, uint32_t idx, double r0, col_t bgc, uint64_t seed)
{
  rng_ctx g; g.seed = seed; g.pixel = p; g.sample = idx; g.path = 0;
  double kx = ref_random(r0, 1000, &g, 1, 0, 0);
  double ky = ref_random(r0, 1000, &g, 1, 0, 1);
  double kz = ref_random(r0, 1000, &g, 1, 0, 2);
  v3 cur1 = V3(cur.x + kx, cur.y + ky, cur.z + kz);
  ray_t ray1 = mk_ray(pnt, v_unitize(v_sub(cur1, pnt)));
  g.path = 1;
  return ray_tracer(s, ray1, k->depth, 1, bgc, g, NULL);
}

void orc_render_antialiasing(orc_scene *s, const orc_camera *c, const double aa[6], int focusing, double radius, int spp,
                             const double bg[3], uint64_t seed, double *rgb, unsigned char *edge_mask, int threads)
{
  cam_t k = cam_init(c);
  const int W = k.W, H = k.H;
  const size_t total = (size_t)W * H;
  const int os = (int)aa[0];
  const double r0 = aa[1];
  col_t bgc = C3(bg[0], bg[1], bg[2]);
  double *maps = aa_maps(s, &k, focusing, bgc, seed, threads);
#ifdef _OPENMP
  int nt = threads > 0 ? threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 16) num_threads(nt)
#endif
  for (long long p = 0; p < (long long)total; p++) {                      /* Camera.h:134-165, 243-276 */
    int v = (int)(p / W), h = (int)(p % W);
    v3 cur = cam_cur_pt(&k, h, v);
    int edge = aa_edge(maps, W, H, v, h, aa, focusing);
    if (edge_mask) edge_mask[p] = (unsigned char)edge;
    col_t colour = C3(0, 0, 0);
    const int n_lens = focusing ? spp : 1;
    for (int i = 0; i < n_lens; i++) {
      v3 pnt = k.view_pt;
      if (focusing) {                                                     /* :256 */
        rng_ctx g; g.seed = seed; g.pixel = (uint32_t)p; g.sample = (uint32_t)i; g.path = 0;
        double jx = ref_random(radius, 1000, &g, 0, 0, 0);
        double jy = ref_random(radius, 1000, &g, 0, 0, 1);
        double jz = ref_random(radius, 1000, &g, 0, 0, 2);
        pnt = V3(k.view_pt.x + jx, k.view_pt.y + jy, k.view_pt.z + jz);
      }
      if (edge) {
        for (int j = 0; j < os; j++) {                                    /* :152-157, 264-269 */
          col_t sc = aa_jittered_sample(s, &k, pnt, cur, (uint32_t)p, (uint32_t)(i * os + j), r0, bgc, seed);
          colour = c_add(colour, c_div(sc, focusing ? (double)(spp * os) : (double)os));
        }
        continue;
      }
      if (focusing) {                                                     /* :272 */
        rng_ctx g; g.seed = seed; g.pixel = (uint32_t)p; g.sample = (uint32_t)i; g.path = 1;
        ray_t ray = mk_ray(pnt, v_unitize(v_sub(cur, pnt)));
        colour = c_add(colour, c_div(ray_tracer(s, ray, k.depth, 1, bgc, g, NULL), spp));
      } else {
        colour = C3(maps[5 * total + p], maps[6 * total + p], maps[7 * total + p]);   /* :161-162: the same ray again */
      }
    }
    rgb[3 * p] = colour.r; rgb[3 * p + 1] = colour.g; rgb[3 * p + 2] = colour.b;
  }
  free(maps);
  (void)threads;
}

/* ------------------------------------------------------------------ batch queries for tests */

/* ------------------------------------------------------------------ progressive photon mapping (PhotonMapper.h, Camera.h:291-313) */
static const double R_INITIAL = 1, PPM_ALPHA = 0.8;                       /* PhotonMapper.h:10-11 */
typedef struct kd_node { hitpoint_t hp; int split_axis; struct kd_node *l, *r; } kd_node;
typedef struct { v3 pt, norm; col_t colour; } photon_hp_t;                /* CPhotonHP, :34-40 */
struct orc_ppm {
  orc_scene *s; cam_t cam; uint64_t seed; int photon_num, times; long long n_emitted;
  hitpoint_t *hps; size_t n_hp;
  kd_node *root;
  unsigned char *picture;                                                 /* CV_8UC3, stored R,G,B here */
  int64_t photon_segments, photon_hits;
  /* Render_PhotonMapper_Antialiasing (Camera.h:315-423): the passes add into a CV_32SC3 picture, `picture` is picture_ori */
  int aa_os; unsigned char *aa_mask; int *sum32;
};

static int cmp_axis;
static int cmp_hp(const void *a, const void *b)
{
  const hitpoint_t *x = (const hitpoint_t *)a, *y = (const hitpoint_t *)b;
  double p = cmp_axis == 0 ? x->pt.x : (cmp_axis == 1 ? x->pt.y : x->pt.z), q = cmp_axis == 0 ? y->pt.x : (cmp_axis == 1 ? y->pt.y : y->pt.z);
  return (p < q) ? -1 : (p > q ? 1 : 0);
}
/* KdTree::BuildTree, PhotonMapper.h:76-129 (max-variance axis, median; std::sort there, qsort here: ties may order
 * differently, which changes the shape of the tree but not what a range query returns) */
static kd_node *kd_build(hitpoint_t *h, size_t n)
{
  if (n == 0) return NULL;
  kd_node *node = (kd_node *)calloc(1, sizeof *node);
  if (n == 1) { node->hp = h[0]; node->split_axis = 0; return node; }
  double mx = 0, my = 0, mz = 0, vx = 0, vy = 0, vz = 0;
  for (size_t i = 0; i < n; i++) { mx += h[i].pt.x; my += h[i].pt.y; mz += h[i].pt.z; }
  mx /= (double)n; my /= (double)n; mz /= (double)n;
  for (size_t i = 0; i < n; i++) {
    vx += (h[i].pt.x - mx) * (h[i].pt.x - mx); vy += (h[i].pt.y - my) * (h[i].pt.y - my); vz += (h[i].pt.z - mz) * (h[i].pt.z - mz);
  }
  if (vx >= vy && vx >= vz) node->split_axis = 0; else if (vy >= vx && vy >= vz) node->split_axis = 1; else node->split_axis = 2;
  cmp_axis = node->split_axis;
  qsort(h, n, sizeof *h, cmp_hp);
  node->hp = h[n / 2];
  node->l = kd_build(h, n / 2);
  if (n >= 3) node->r = kd_build(h + n / 2 + 1, n - n / 2 - 1);
  return node;
}
static void kd_free(kd_node *n) { if (!n) return; kd_free(n->l); kd_free(n->r); free(n); }

/* random(r, x) for photons: draw number `d` of photon `photon` in pass `pass` */
typedef struct { uint64_t seed; uint32_t pass, photon, draw; } prng_t;
static inline double photon_random(double r, int x, prng_t *g)
{
  uint32_t k = orc_rng_u32(g->seed ^ 0x50484F544F4E53ULL, g->pass, g->photon, g->draw++) >> 1;
  return 2 * r * (double)(k % (uint32_t)x) / (double)x - r;
}
/* CPhotonMapper::RandomDirection, :146; draws in x, y, z order (pinned) */
static v3 random_direction(prng_t *g)
{
  double x = photon_random(1000, 1000, g), y = photon_random(1000, 1000, g), z = photon_random(1000, 1000, g);
  return v_unitize(V3(x, y, z));
}

/* CPhotonMapper::AccumulatePhotons, :334-355 */
static void kd_accumulate(orc_ppm *m, kd_node *node, const photon_hp_t *ph)
{
  if (!node) return;
  if (v_dist(node->hp.pt, ph->pt) < node->hp.R + EPS && v_dot(node->hp.norm, ph->norm) > EPS * 100) {
    node->hp.M++;
    node->hp.tau = c_add(node->hp.tau, c_div(ph->colour, (PI_REF * node->hp.R * node->hp.R * (double)m->n_emitted) / 3000));
  }
  v3 normal = V3(0, 0, 0);
  if (node->split_axis == 0) normal.x = 1;
  if (node->split_axis == 1) normal.y = 1;
  if (node->split_axis == 2) normal.z = 1;
  double l = v_dot(v_sub(ph->pt, node->hp.pt), normal);
  if (l < EPS || fabs(l) < R_INITIAL + EPS) kd_accumulate(m, node->l, ph);
  if (l > -EPS || fabs(l) < R_INITIAL + EPS) kd_accumulate(m, node->r, ph);
}
static inline unsigned char sat_u8(double x)       /* cv::saturate_cast<uchar>(double): round to nearest even, clamp */
{ double r = nearbyint(x); return (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r)); }
/* CPhotonMapper::RadianceEvaluation, :357-376, CV_8UC3 picture */
static void kd_radiance(orc_ppm *m, kd_node *node)
{
  if (!node) return;
  hitpoint_t *hp = &node->hp;
  unsigned char *px = &m->picture[3 * ((size_t)hp->v * m->cam.W + hp->h)];
  const double t[3] = {hp->tau.r, hp->tau.g, hp->tau.b};
  if (m->sum32) {                                                         /* :365-367: Vec3i(tau...) * wgt added to the CV_32SC3 picture */
    int *q = &m->sum32[3 * ((size_t)hp->v * m->cam.W + hp->h)];
    for (int c = 0; c < 3; c++) q[c] += (int)nearbyint((double)(int)t[c] * hp->wgt);     /* int truncation, then saturate_cast<int>(x * wgt) = cvRound */
  } else
  for (int c = 0; c < 3; c++) {
    unsigned char tb = (unsigned char)t[c];                               /* Vec3b(tau...): double -> uchar truncation */
    unsigned char add = sat_u8((double)tb * hp->wgt);                     /* Vec3b * double */
    int sum = (int)px[c] + (int)add;                                      /* Vec3b += : saturating */
    px[c] = (unsigned char)(sum > 255 ? 255 : sum);
  }
  hp->tau = C3(0, 0, 0);
  if (hp->M > 0) hp->R *= sqrt((hp->N + hp->M * PPM_ALPHA) / (hp->N + hp->M));
  hp->N += hp->M * PPM_ALPHA;
  hp->M = 0;
  kd_radiance(m, node->l); kd_radiance(m, node->r);
}

static int cmp_key(const void *a, const void *b)
{
  const hitpoint_t *x = (const hitpoint_t *)a, *y = (const hitpoint_t *)b;
  if (x->pixel != y->pixel) return x->pixel < y->pixel ? -1 : 1;
  return x->path < y->path ? -1 : (x->path > y->path ? 1 : 0);
}

/* Camera.h:291-308: eye pass (RayTracingPass per pixel) + ConstructKdTree */
orc_ppm *orc_ppm_begin(orc_scene *s, const orc_camera *c, const double bg[3], uint64_t seed, int photon_num, int times)
{
  orc_ppm *m = (orc_ppm *)calloc(1, sizeof *m);
  m->s = s; m->cam = cam_init(c); m->seed = seed; m->photon_num = photon_num > 0 ? photon_num : 50000; m->times = times;
  m->n_emitted = (long long)times * s->n_lights * m->photon_num;          /* :225 */
  const cam_t k = m->cam;
  const size_t total = (size_t)k.W * k.H;
  m->picture = (unsigned char *)calloc(total, 3);
  col_t bgc = C3(bg[0], bg[1], bg[2]);
  hp_sink_t all = {NULL, 0, 0, k.W, 0};
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    hp_sink_t mine = {NULL, 0, 0, k.W, 0};
    tl_hp = &mine;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
    for (long long p = 0; p < (long long)total; p++) {
      int v = (int)(p / k.W), h = (int)(p % k.W);
      v3 cur = cam_cur_pt(&k, h, v);
      ray_t ray = mk_ray(k.view_pt, v_unitize(v_sub(cur, k.view_pt)));
      rng_ctx g; g.seed = seed; g.pixel = (uint32_t)p; g.sample = 0; g.path = 1;
      col_t col = ray_tracer(s, ray, k.depth, 1, bgc, g, NULL);
      m->picture[3 * p] = (unsigned char)col.r; m->picture[3 * p + 1] = (unsigned char)col.g; m->picture[3 * p + 2] = (unsigned char)col.b;   /* :306 */
    }
    tl_hp = NULL;
#ifdef _OPENMP
#pragma omp critical
#endif
    {
      all.a = (hitpoint_t *)realloc(all.a, (all.n + mine.n + 1) * sizeof(hitpoint_t));
      memcpy(all.a + all.n, mine.a, mine.n * sizeof(hitpoint_t)); all.n += mine.n;
    }
    free(mine.a);
  }
  qsort(all.a, all.n, sizeof(hitpoint_t), cmp_key);     /* a fixed order for the tests; the kd build re-sorts anyway */
  m->hps = all.a; m->n_hp = all.n;
  hitpoint_t *tmp = (hitpoint_t *)malloc((all.n + 1) * sizeof(hitpoint_t));
  memcpy(tmp, all.a, all.n * sizeof(hitpoint_t));
  m->root = kd_build(tmp, all.n);                        /* ConstructKdTree, :141 */
  free(tmp);
  return m;
}

/* Camera.h:315-399 Render_PhotonMapper_Antialiasing up to PhotonMappingPass: maps from the plain RayTracer, edge mask, then the eye
 * pass -- a masked pixel sends oversampleTimes jittered rays through RayTracingPass, all of them depositing hit points on (h, v)
 * (:384-389), the others one ray (:393-394); picture_ori = the clamped colours, truncated (:390, :395).
 * aa[6] as in orc_render_antialiasing.  Hit-point key: (pixel, (ray index << 12) | ray-tree path). */
orc_ppm *orc_ppm_begin_aa(orc_scene *s, const orc_camera *c, const double aa[6], const double bg[3], uint64_t seed, int photon_num, int times)
{
  orc_ppm *m = (orc_ppm *)calloc(1, sizeof *m);
  m->s = s; m->cam = cam_init(c); m->seed = seed; m->photon_num = photon_num > 0 ? photon_num : 50000; m->times = times;
  m->n_emitted = (long long)times * s->n_lights * m->photon_num;          /* :225 */
  const cam_t k = m->cam;
  const int W = k.W, H = k.H;
  const size_t total = (size_t)W * H;
  const int os = (int)aa[0];
  m->picture = (unsigned char *)calloc(total, 3);
  m->aa_os = os; m->aa_mask = (unsigned char *)calloc(total, 1); m->sum32 = (int *)calloc(total * 3, sizeof(int));
  col_t bgc = C3(bg[0], bg[1], bg[2]);
  double *maps = aa_maps(s, &k, 0, bgc, seed, 0);
  hp_sink_t all = {NULL, 0, 0, W, 1};
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    hp_sink_t mine = {NULL, 0, 0, W, 1};
    tl_hp = &mine;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (long long p = 0; p < (long long)total; p++) {
      int v = (int)(p / W), h = (int)(p % W);
      v3 cur = cam_cur_pt(&k, h, v);
      col_t colour = C3(0, 0, 0);
      if (aa_edge(maps, W, H, v, h, aa, 0)) {
        m->aa_mask[p] = 1;
        for (int j = 0; j < os; j++)                                      /* :384-389 */
          colour = c_add(colour, c_div(aa_jittered_sample(s, &k, k.view_pt, cur, (uint32_t)p, (uint32_t)j, aa[1], bgc, seed), (double)os));
      } else {
        ray_t ray = mk_ray(k.view_pt, v_unitize(v_sub(cur, k.view_pt)));  /* :393-394 */
        rng_ctx g; g.seed = seed; g.pixel = (uint32_t)p; g.sample = 0; g.path = 1;
        colour = ray_tracer(s, ray, k.depth, 1, bgc, g, NULL);
      }
      m->picture[3 * p] = (unsigned char)colour.r; m->picture[3 * p + 1] = (unsigned char)colour.g; m->picture[3 * p + 2] = (unsigned char)colour.b;
    }
    tl_hp = NULL;
#ifdef _OPENMP
#pragma omp critical
#endif
    {
      all.a = (hitpoint_t *)realloc(all.a, (all.n + mine.n + 1) * sizeof(hitpoint_t));
      memcpy(all.a + all.n, mine.a, mine.n * sizeof(hitpoint_t)); all.n += mine.n;
    }
    free(mine.a);
  }
  free(maps);
  qsort(all.a, all.n, sizeof(hitpoint_t), cmp_key);
  m->hps = all.a; m->n_hp = all.n;
  hitpoint_t *tmp = (hitpoint_t *)malloc((all.n + 1) * sizeof(hitpoint_t));
  memcpy(tmp, all.a, all.n * sizeof(hitpoint_t));
  m->root = kd_build(tmp, all.n);
  free(tmp);
  return m;
}
void orc_ppm_edge_mask(const orc_ppm *m, unsigned char *mask)
{ const size_t n = (size_t)m->cam.W * m->cam.H; if (m->aa_mask) memcpy(mask, m->aa_mask, n); else memset(mask, 0, n); }

/* one iteration of the loop of PhotonMappingPass (:226-238): GeneratePhotons, TracePhotons, AccumulatePhotons, RadianceEvaluation */
void orc_ppm_pass(orc_ppm *m, int pass_index)
{
  orc_scene *s = m->s;
  uint32_t photon = 0;
  for (int li = 0; li < s->n_lights; li++) {
    const light_t *L = &s->lights[li];
    if (L->type != ORC_POINT_LIGHT && L->type != ORC_AREA_LIGHT) continue;                 /* :245, :251 */
    const int sn = (L->type == ORC_AREA_LIGHT) ? L->sample_num : 1;
    for (int t = 0; t < m->photon_num; t++) for (int i = 0; i < sn; i++) for (int j = 0; j < sn; j++, photon++) {
      prng_t g = {m->seed, (uint32_t)pass_index, photon, 0};
      v3 origin; col_t colour;
      if (L->type == ORC_POINT_LIGHT) { origin = L->pos_or_dir; colour = c_mul(C3(255, 255, 255), L->lum); }           /* :249 */
      else {
        double r = (L->len / (double)L->sample_num) / 2;                                                               /* :256-265 */
        v3 O = V3(L->pos_or_dir.x - L->sample_num * r, L->pos_or_dir.y - L->sample_num * r, L->pos_or_dir.z);
        double x0 = O.x + i * r * 2, y0 = O.y + j * r * 2, z0 = O.z;
        double rx = photon_random(r, 1000, &g), ry = photon_random(r, 1000, &g);
        origin = V3(x0 + rx, y0 + ry, z0); colour = c_mul(C3(255, 255, 255), L->lum_per_sample);
      }
      ray_t path = mk_ray(origin, random_direction(&g));
      double weight = 1;
      /* TracePhotons, :273-332 */
      inter_t in = first_inter_pt_bsp(s, path, 1); m->photon_segments++;
      for (int b = 0; b < 10 && weight > 0.2 && in.obj >= 0; b++) {
        const obj_t *o = &s->objs[in.obj];
        v3 hp = in.pt;
        v3 norm = obj_norm(o, hp, in.u, in.v);
        if (o->mat.reflW == 0 && o->mat.refrW == 0) {
          if (v_dot(path.d, norm) > EPS) norm = v_neg(norm);
          photon_hp_t ph; ph.pt = hp; ph.norm = norm; ph.colour = c_mul(colour, weight);
          kd_accumulate(m, m->root, &ph); m->photon_hits++;               /* the photonHps stack only defers this (:231-236) */
          v3 dir = random_direction(&g);
          if (v_dot(dir, norm) < 0) dir = v_neg(dir);
          path = mk_ray(hp, dir);
          weight /= 4;
        } else if (o->mat.refrW > 0) {
          ray_t np = obj_refract_ray(o, path, hp, in.u, in.v);
          if (!is_infty_pt(np.d)) path = np; else path = obj_reflect_ray(o, path, hp, in.u, in.v);
        } else if (o->mat.reflW > 0) {
          path = obj_reflect_ray(o, path, hp, in.u, in.v);
        }
        col_t oc = obj_colour(o, hp, in.u, in.v);                         /* PhotonMapper.h:296, 312, 324 */
        double sc = (colour.r + colour.g + colour.b) / (oc.r + oc.g + oc.b);
        colour = c_div(c_add(colour, c_mul(oc, sc)), 2);
        in = first_inter_pt_bsp(s, path, 1); m->photon_segments++;
      }
    }
  }
  kd_radiance(m, m->root);
}

size_t orc_ppm_num_hitpoints(const orc_ppm *m) { return m->n_hp; }
static void kd_collect(const kd_node *n, hitpoint_t *out, size_t *k) { if (!n) return; out[(*k)++] = n->hp; kd_collect(n->l, out, k); kd_collect(n->r, out, k); }
/* hit points sorted by (pixel, path): 14 doubles each = pixel, path, pt(3), norm(3), wgt, R, N, tau(3) */
void orc_ppm_hitpoints(const orc_ppm *m, double *out)
{
  hitpoint_t *tmp = (hitpoint_t *)malloc((m->n_hp + 1) * sizeof(hitpoint_t));
  size_t k = 0; kd_collect(m->root, tmp, &k);
  qsort(tmp, k, sizeof(hitpoint_t), cmp_key);
  for (size_t i = 0; i < k; i++) {
    double *o = out + 14 * i; const hitpoint_t *h = &tmp[i];
    o[0] = h->pixel; o[1] = h->path; o[2] = h->pt.x; o[3] = h->pt.y; o[4] = h->pt.z; o[5] = h->norm.x; o[6] = h->norm.y; o[7] = h->norm.z;
    o[8] = h->wgt; o[9] = h->R; o[10] = h->N; o[11] = h->tau.r; o[12] = h->tau.g; o[13] = h->tau.b;
  }
  free(tmp);
}
void orc_ppm_picture(const orc_ppm *m, double *rgb)
{
  const size_t n = (size_t)m->cam.W * m->cam.H * 3;
  if (!m->sum32) { for (size_t i = 0; i < n; i++) rgb[i] = m->picture[i]; return; }
  for (size_t i = 0; i < n; i++) {                                        /* Camera.h:402-420 */
    int add = m->sum32[i];
    if (m->aa_mask[i / 3]) add = (int)nearbyint((double)add * (1. / m->aa_os));    /* Vec3i / int = a * (1. / alpha), saturate_cast<int> */
    int v = (int)m->picture[i] + add;                                     /* Vec3b += Vec3i: saturate_cast<uchar> */
    rgb[i] = (double)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}
void orc_ppm_counters(const orc_ppm *m, int64_t out[2]) { out[0] = m->photon_segments; out[1] = m->photon_hits; }
void orc_ppm_end(orc_ppm *m) { if (!m) return; kd_free(m->root); free(m->hps); free(m->picture); free(m->aa_mask); free(m->sum32); free(m); }


/* ------------------------------------------------------------------ god ray (GodRay.h, Camera.h:425-448) */
typedef struct { v3 pos; double energy; uint32_t idx; } god_photon_t;     /* CGodPhoton, :18-24 (+ creation index: sort tie-break) */
typedef struct god_node { god_photon_t ph; int split_axis; struct god_node *l, *r; } god_node;
typedef struct { v3 pos, dir; } god_source_t;                             /* CLightSource, :9-15 */
struct orc_godray {
  orc_scene *s; col_t light_colour; uint64_t seed;
  god_source_t *src; size_t n_src, cap_src; uint32_t n_areas;
  god_photon_t *ph; size_t n_ph, cap_ph;
  god_node *root;
};
static const double GOD_RADIUS = 0.1, GOD_INTERVAL = 0.05, GOD_DECAY = 0.95, GOD_E0 = 0.01;   /* :128-131 */

static inline double god_random(double r, int x, uint64_t seed, uint32_t k0, uint32_t k1, uint32_t draw)
{ uint32_t k = orc_rng_u32(seed, k0, k1, draw) >> 1; return 2 * r * (double)(k % (uint32_t)x) / (double)x - r; }

orc_godray *orc_godray_new(orc_scene *s, const double light_colour[3], uint64_t seed)
{
  orc_godray *g = (orc_godray *)calloc(1, sizeof *g);
  g->s = s; g->light_colour = C3(light_colour[0], light_colour[1], light_colour[2]); g->seed = seed;
  return g;
}
/* CGodRay::AddAreaLightSource_XZ, :193-200; jitter draws in x, z order (pinned) */
void orc_godray_add_area_xz(orc_godray *g, double y, const double mn[2], const double mx[2], double interval, const double dir[3])
{
  uint32_t k = 0;
  for (double x = mn[0]; x < mx[0]; x += interval)
    for (double z = mn[1]; z < mx[1]; z += interval, k++) {
      if (g->n_src == g->cap_src) { g->cap_src = g->cap_src ? g->cap_src * 2 : 1024; g->src = (god_source_t *)realloc(g->src, g->cap_src * sizeof(god_source_t)); }
      double jx = god_random(interval / 3, 10000, g->seed ^ 0x474F4453ULL, g->n_areas, k, 0);
      double jz = god_random(interval / 3, 10000, g->seed ^ 0x474F4453ULL, g->n_areas, k, 1);
      g->src[g->n_src].pos = V3(x + jx, y, z + jz); g->src[g->n_src].dir = V3(dir[0], dir[1], dir[2]);
      g->n_src++;
    }
  g->n_areas++;
}
static int god_axis;
static int cmp_god(const void *a, const void *b)
{
  const god_photon_t *x = (const god_photon_t *)a, *y = (const god_photon_t *)b;
  double p = god_axis == 0 ? x->pos.x : (god_axis == 1 ? x->pos.y : x->pos.z), q = god_axis == 0 ? y->pos.x : (god_axis == 1 ? y->pos.y : y->pos.z);
  if (p < q) return -1;
  if (q < p) return 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);   /* std::sort leaves ties unspecified: creation order here and in the library */
}
/* CPhotonTree::BuildTree, :52-105 */
static god_node *god_build(god_photon_t *h, size_t n)
{
  if (n == 0) return NULL;
  god_node *node = (god_node *)calloc(1, sizeof *node);
  if (n == 1) { node->ph = h[0]; node->split_axis = 0; return node; }
  double mx = 0, my = 0, mz = 0, vx = 0, vy = 0, vz = 0;
  for (size_t i = 0; i < n; i++) { mx += h[i].pos.x; my += h[i].pos.y; mz += h[i].pos.z; }
  mx /= (double)n; my /= (double)n; mz /= (double)n;
  for (size_t i = 0; i < n; i++) {
    vx += (h[i].pos.x - mx) * (h[i].pos.x - mx); vy += (h[i].pos.y - my) * (h[i].pos.y - my); vz += (h[i].pos.z - mz) * (h[i].pos.z - mz);
  }
  if (vx >= vy && vx >= vz) node->split_axis = 0; else if (vy >= vx && vy >= vz) node->split_axis = 1; else node->split_axis = 2;
  god_axis = node->split_axis;
  qsort(h, n, sizeof *h, cmp_god);
  node->ph = h[n / 2];
  node->l = god_build(h, n / 2);
  if (n >= 3) node->r = god_build(h + n / 2 + 1, n - n / 2 - 1);
  return node;
}
static void god_free(god_node *n) { if (!n) return; god_free(n->l); god_free(n->r); free(n); }

/* CGodRay::PhotonMappingPass = GeneratePhotons (:202-215) + ConstructPhotonTree */
void orc_godray_build(orc_godray *g)
{
  for (size_t i = 0; i < g->n_src; i++) {
    ray_t ray = mk_ray(g->src[i].pos, g->src[i].dir);
    v3 pt = first_inter_pt_bsp(g->s, ray, 0).pt;
    double t_range = dmin(v_dist(g->src[i].pos, pt), 10.0);
    uint32_t draw = 0;
    for (double t = 0, a = 1; t < t_range; t += GOD_INTERVAL, a *= GOD_DECAY) {
      t += god_random(GOD_INTERVAL / 3, 10000, g->seed ^ 0x474F4450ULL, (uint32_t)i, 0, draw++);
      if (g->n_ph == g->cap_ph) { g->cap_ph = g->cap_ph ? g->cap_ph * 2 : 4096; g->ph = (god_photon_t *)realloc(g->ph, g->cap_ph * sizeof(god_photon_t)); }
      g->ph[g->n_ph].pos = v_add(g->src[i].pos, v_mul(g->src[i].dir, t));      /* the direction is NOT unitized here (:212) */
      g->ph[g->n_ph].energy = GOD_E0 * a; g->ph[g->n_ph].idx = (uint32_t)g->n_ph;
      g->n_ph++;
    }
  }
  god_photon_t *tmp = (god_photon_t *)malloc((g->n_ph + 1) * sizeof(god_photon_t));
  memcpy(tmp, g->ph, g->n_ph * sizeof(god_photon_t));
  g->root = god_build(tmp, g->n_ph);
  free(tmp);
}
size_t orc_godray_num_photons(const orc_godray *g) { return g->n_ph; }
void orc_godray_photons(const orc_godray *g, double *out /* 4 per photon: x, y, z, energy, creation order */)
{ for (size_t i = 0; i < g->n_ph; i++) { out[4 * i] = g->ph[i].pos.x; out[4 * i + 1] = g->ph[i].pos.y; out[4 * i + 2] = g->ph[i].pos.z; out[4 * i + 3] = g->ph[i].energy; } }

/* CRay::Dist, Geometry.h:265-272: distance to the infinite LINE through the ray */
static inline double ray_line_dist(ray_t ray, v3 pt)
{ v3 l = v_sub(pt, ray.o); double tp = v_dot(l, ray.d); double d2 = v_len2(l) - tp * tp; return sqrt(d2); }
/* CGodRay::RadianceEvaluation, :217-245 */
static col_t god_eval(const orc_godray *g, const god_node *node, ray_t ray)
{
  if (!node) return C3(0, 0, 0);
  col_t colour = C3(0, 0, 0);
  if (ray_line_dist(ray, node->ph.pos) < GOD_RADIUS) colour = c_add(colour, c_mul(g->light_colour, node->ph.energy));
  v3 p0 = node->ph.pos;
  v3 normal = V3(0, 0, 0);
  if (node->split_axis == 0) normal.x = 1;
  if (node->split_axis == 1) normal.y = 1;
  if (node->split_axis == 2) normal.z = 1;
  const god_node *near_n, *far_n;
  if (v_dot(v_sub(p0, ray.o), normal) > 0) { near_n = node->l; far_n = node->r; }
  else { near_n = node->r; far_n = node->l; normal = v_neg(normal); }
  colour = c_add(colour, god_eval(g, near_n, ray));
  if (v_dot(ray.d, normal) > -EPS || fabs(v_dot(v_sub(p0, ray.o), normal)) < GOD_RADIUS + EPS)
    colour = c_add(colour, god_eval(g, far_n, ray));
  return colour;
}
static col_t god_radiance(const struct orc_godray *g, ray_t ray) { return god_eval(g, g->root, ray); }

/* Camera.h:425-448 Render_RayTracer_GodRay after PhotonMappingPass: rgb NOT truncated (like orc_render_whitted) */
void orc_render_godray(orc_godray *g, const orc_camera *c, const double bg[3], double *rgb, int32_t *prim_id)
{
  cam_t k = cam_init(c);
  const size_t total = (size_t)k.W * k.H;
  col_t bgc = C3(bg[0], bg[1], bg[2]);
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    tl_god = g;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 64)
#endif
    for (long long p = 0; p < (long long)total; p++) {
      int v = (int)(p / k.W), h = (int)(p % k.W);
      v3 cur = cam_cur_pt(&k, h, v);
      ray_t ray = mk_ray(k.view_pt, v_unitize(v_sub(cur, k.view_pt)));
      rng_ctx rg; rg.seed = g->seed; rg.pixel = (uint32_t)p; rg.sample = 0; rg.path = 1;
      int32_t id = -1;
      col_t col = ray_tracer(g->s, ray, k.depth, 1, bgc, rg, &id);
      rgb[3 * p] = col.r; rgb[3 * p + 1] = col.g; rgb[3 * p + 2] = col.b;
      if (prim_id) prim_id[p] = id;
    }
    tl_god = NULL;
  }
}
/* Camera.h:450-537 Render_RayTracer_GodRay_Antialiasing after PhotonMappingPass: maps from the plain RayTracer (:483), masked pixels
 * re-sampled with oversampleTimes jittered rays through CGodRay::RayTracingPass (:516-523), the others one such ray (:527-528).
 * rgb NOT truncated. */
void orc_render_godray_aa(orc_godray *g, const orc_camera *c, const double aa[6], const double bg[3], double *rgb, unsigned char *edge_mask)
{
  cam_t k = cam_init(c);
  const int W = k.W, H = k.H;
  const size_t total = (size_t)W * H;
  const int os = (int)aa[0];
  col_t bgc = C3(bg[0], bg[1], bg[2]);
  double *maps = aa_maps(g->s, &k, 0, bgc, g->seed, 0);
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    tl_god = g;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (long long p = 0; p < (long long)total; p++) {
      int v = (int)(p / W), h = (int)(p % W);
      v3 cur = cam_cur_pt(&k, h, v);
      col_t colour = C3(0, 0, 0);
      int edge = aa_edge(maps, W, H, v, h, aa, 0);
      if (edge_mask) edge_mask[p] = (unsigned char)edge;
      if (edge) {
        for (int j = 0; j < os; j++)
          colour = c_add(colour, c_div(aa_jittered_sample(g->s, &k, k.view_pt, cur, (uint32_t)p, (uint32_t)j, aa[1], bgc, g->seed), (double)os));
      } else {
        ray_t ray = mk_ray(k.view_pt, v_unitize(v_sub(cur, k.view_pt)));
        rng_ctx rg; rg.seed = g->seed; rg.pixel = (uint32_t)p; rg.sample = 0; rg.path = 1;
        colour = ray_tracer(g->s, ray, k.depth, 1, bgc, rg, NULL);
      }
      rgb[3 * p] = colour.r; rgb[3 * p + 1] = colour.g; rgb[3 * p + 2] = colour.b;
    }
    tl_god = NULL;
  }
  free(maps);
}
void orc_godray_free(orc_godray *g) { if (!g) return; god_free(g->root); free(g->src); free(g->ph); free(g); }

void orc_trace_closest(orc_scene *s, const double *rays, size_t n, int use_bsp, int record,
                       int32_t *out_id, double *out_pt, double *out_uv)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256)
#endif
  for (long long i = 0; i < (long long)n; i++) {
    const double *r = rays + 6 * i;
    ray_t ray = mk_ray(V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]));
    inter_t in = use_bsp ? first_inter_pt_bsp(s, ray, record) : first_inter_pt(s, ray, record);
    out_id[i] = in.obj;
    if (out_pt) { out_pt[3 * i] = in.pt.x; out_pt[3 * i + 1] = in.pt.y; out_pt[3 * i + 2] = in.pt.z; }
    if (out_uv) { out_uv[2 * i] = in.u; out_uv[2 * i + 1] = in.v; }
  }
}

void orc_object_first_inter_pt(orc_scene *s, int id, const double r[6], double out_pt[3], double out_uv[2])
{
  ray_t ray = mk_ray(V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]));
  hit_t h = obj_first_inter_pt(&s->objs[id], ray, 1);
  out_pt[0] = h.pt.x; out_pt[1] = h.pt.y; out_pt[2] = h.pt.z;
  if (out_uv) { out_uv[0] = h.u; out_uv[1] = h.v; }
}
