/*
 * rt_oracle.h -- CPU restatement of the reference renderer's ray-scene hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under ray_tracing_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker.
 *
 * PARITY PIN: the reference (FlyingGiraffe/ray_tracing) cannot be built in this
 * image (it needs Eigen and OpenCV 2.4.9, both absent; writing stand-ins for them
 * is not allowed), and it ships no tests or golden vectors.  The only outputs of
 * the reference itself that we hold are the figures recorded by the surveyor in
 * SURVEY.md section 8(c) (C1 hit-ID hash 83db7884e5084a37, BSP statistics of the
 * C1 scene).  tests/test_oracle_pins.py checks the oracle against those.  Every
 * result those figures do not cover (radiance, shadows, reflection/refraction,
 * sphere/circle, DOF sampling) is "parity unpinned": the oracle restates the
 * reference source line by line, but no reference output confirms it.
 *
 * All arithmetic is IEEE double, compiled with -ffp-contract=off.
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference/RayTracing/).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* object type tags, Object.h:17-23 */
enum { ORC_SPHERE = 0, ORC_TRIANGLE = 1, ORC_POLYGON = 2, ORC_CIRCLE = 3,
       ORC_INFTY_PLANE = 4, ORC_ROTATIONAL = 5, ORC_BEZIER = 6 };
/* light type tags, Light.h:7-9 */
enum { ORC_POINT_LIGHT = 100, ORC_PARALLEL_LIGHT = 101, ORC_AREA_LIGHT = 102 };
/* material presets, Material.h:4-10 */
enum { ORC_MIRROR = 200, ORC_GLASS = 201, ORC_ICE = 202, ORC_METAL = 203,
       ORC_PORCELAIN = 204, ORC_RUBBER = 205, ORC_PLASTIC = 206 };

/* material = {reflW, refrW, RI, diff, spec, shiness, emissive, ambient} (Material.h:33-42) */
void orc_material_preset(int name, double out8[8]);

orc_scene *orc_scene_new(double ambient);                  /* Scene.h:15-16 */
void orc_scene_free(orc_scene *s);

/* all return the object's index in CScene::objects (the "primitive id") */
int orc_add_sphere(orc_scene *s, const double c[3], double r, const double mat[8], const double col[3]);
int orc_add_plane(orc_scene *s, const double p0[3], const double n[3], const double mat[8], const double col[3]);
int orc_add_circle(orc_scene *s, const double p0[3], const double n[3], double r, const double mat[8], const double col[3]);
/* CBezierObject(ctrl[4][4], center, ...), Object.h:1075; ctrl row-major [i][j][xyz] */
int orc_add_bezier(orc_scene *s, const double ctrl[48], const double center[3], const double mat[8], const double col[3]);
/* CRotationalObject(curve, center, ..., uSampleNum, uRange), Object.h:818; profile = n_prof (2..8) control points (u = radius, v = height) */
int orc_add_rotational(orc_scene *s, const double *prof, int n_prof, const double center[3], int u_sample_num, double u_range,
                       const double mat[8], const double col[3]);
/* vn may be NULL (haveVertexNorm=false, Object.h:360-371) */
int orc_add_triangle(orc_scene *s, const double v[9], const double vn[9], const double mat[8], const double col[3]);
/* Scene.h:150-166 + Parser.h:35-82; returns number of triangles added or -1 */
int orc_read_obj_triangle(orc_scene *s, const char *path, const double center[3], double scale,
                          const double mat[8], const double col[3]);
int orc_num_objects(const orc_scene *s);
/* copies out triangle i's vertices (9) and vertex normals (9); returns haveVertexNorm, -1 if not a triangle */
int orc_get_triangle(const orc_scene *s, int id, double v[9], double vn[9]);

void orc_add_point_light(orc_scene *s, const double pos[3], double lum);       /* Scene.h:274 */
void orc_add_parallel_light(orc_scene *s, const double dir[3], double lum);    /* Scene.h:268 */
void orc_add_area_light(orc_scene *s, const double pos[3], double lum, double len, int sample_num); /* Scene.h:280 */

/* Scene.h:286-293.  max_obj_num/max_depth <= 0 select the reference constants 100/40 (BSPTree.h:56-57). */
void orc_construct_bsp(orc_scene *s, int max_obj_num, int max_depth);
/* stats[0..7] = nodes, leaves, empty leaves, leaf refs, max depth, max leaf size, root infinite objs, has_tree */
void orc_tree_stats(const orc_scene *s, int64_t stats[8], double root_box[6]);

/* Counter-based RNG that replaces rand(): the draw is a pure function of (seed, key0, key1, counter). */
uint32_t orc_rng_u32(uint64_t seed, uint32_t key0, uint32_t key1, uint32_t counter);

/* Closest hit for n rays {o[3], d[3]} (d is normalised by the CRay constructor, Geometry.h:253).
 * use_bsp: 1 = FirstInterPt_BSPTree (Scene.h:85), 0 = FirstInterPt (Scene.h:62).
 * out_id[i] = object index or -1; out_pt (3 per ray) = hit point or (1e8,1e8,1e8); out_uv optional (2 per ray). */
void orc_trace_closest(orc_scene *s, const double *rays, size_t n, int use_bsp, int record,
                       int32_t *out_id, double *out_pt, double *out_uv);

typedef struct {
  double view_pt[3], eye_dir[3], left_arg[3];
  double dist_of_plane, hor_range, ver_range;
  int32_t resolution, depth;
} orc_camera;   /* CCamera ctor arguments, Camera.h:17-28 */

void orc_camera_size(const orc_camera *c, int *w, int *h);    /* Camera.h:55-56 */
/* rays for pixel list (Camera.h:65-67): out 6 doubles per pixel, order h-major like the reference loop */
void orc_camera_rays(const orc_camera *c, double *rays /* W*H*6, index v*W+h */);

/* Camera.h:53-75 Render_RayTracer: rgb (3 doubles per pixel, index (v*W+h)*3, R,G,B order, NOT truncated),
 * prim_id optional (primary hit id), pixel subset [first, first+count) of the v*W+h raster when count>0.
 * threads<=0 -> all OpenMP threads.  counters (optional, 4 x int64): rays, node visits, prim tests, shadow rays. */
void orc_render_whitted(orc_scene *s, const orc_camera *c, const double bg[3], uint64_t seed,
                        double *rgb, int32_t *prim_id, size_t first, size_t count, int threads,
                        int64_t *counters);

/* Camera.h:171-199 Render_Focusing with spp samples per pixel (reference: sampleTimes=100), lens radius. */
void orc_render_focusing(orc_scene *s, const orc_camera *c, double radius, int spp, const double bg[3],
                         uint64_t seed, double *rgb, size_t first, size_t count, int threads,
                         int64_t *counters);

/* Textures (Object.h:69-74): an image (rows x cols x 3 bytes B,G,R = cv::Mat CV_8UC3) is registered once and attached
 * as texture (Object.h:198-203; scale = CInftyPlane textureScale, :773-779) or bump texture (:205-210, :781-787);
 * triangles take their textureCoord (Object.h:342) in texel units. */
int orc_add_texture_image(orc_scene *s, int rows, int cols, const unsigned char *bgr);
int orc_obj_add_texture(orc_scene *s, int obj, int tex, double scale);
int orc_obj_add_bump_texture(orc_scene *s, int obj, int tex, double scale);
int orc_triangle_set_tex_coord(orc_scene *s, int obj, const double tc[6]);
/* CScene::RotationalSurfaceToMesh (tex < 0, Scene.h:168-213) / RotationalSurfaceToMesh_Texture (Scene.h:215-266);
 * returns the number of triangles added */
int orc_rotational_surface_to_mesh(orc_scene *s, const double *prof, int n_prof, int u_num, int v_num, int tex,
                                   const double center[3], const double mat[8], const double col[3]);

/* Adaptive anti-aliasing: Camera.h:83-169 (focusing = 0) and Camera.h:201-280 (focusing = 1).
 * aa[6] = oversampleTimes, r0, depth / norm / type / colour thresholds (Camera.h:77-82). */
void orc_render_antialiasing(orc_scene *s, const orc_camera *c, const double aa[6], int focusing, double radius, int spp,
                             const double bg[3], uint64_t seed, double *rgb, unsigned char *edge_mask, int threads);

/* Progressive photon mapping: Camera.h:291-313 Render_PhotonMapper + PhotonMapper.h.  begin = eye pass
 * (RayTracingPass per pixel, hit points on diffuse surfaces) + hit-point kd-tree; pass = one iteration of
 * PhotonMappingPass (emit photon_num photons per light [x sampleNum^2 for an area light], trace, gather, update).
 * rand() is replaced by the counter generator keyed (seed, pass, photon index, draw number). */
typedef struct orc_ppm orc_ppm;
orc_ppm *orc_ppm_begin(orc_scene *s, const orc_camera *c, const double bg[3], uint64_t seed, int photon_num, int times);
/* CCamera::Render_PhotonMapper_Antialiasing (Camera.h:315-423) up to PhotonMappingPass; aa[6] as in orc_render_antialiasing */
orc_ppm *orc_ppm_begin_aa(orc_scene *s, const orc_camera *c, const double aa[6], const double bg[3], uint64_t seed, int photon_num, int times);
void orc_ppm_edge_mask(const orc_ppm *m, unsigned char *mask);
void orc_ppm_pass(orc_ppm *m, int pass_index);
size_t orc_ppm_num_hitpoints(const orc_ppm *m);
/* sorted by (pixel, ray-tree path); 14 doubles each: pixel, path, pt[3], norm[3], wgt, R, N, tau[3] */
void orc_ppm_hitpoints(const orc_ppm *m, double *out);
void orc_ppm_picture(const orc_ppm *m, double *rgb /* W*H*3, the CV_8UC3 picture as R,G,B */);
void orc_ppm_counters(const orc_ppm *m, int64_t out[2] /* photon path segments, photon hits on diffuse surfaces */);
void orc_ppm_end(orc_ppm *m);

/* God ray: GodRay.h + Camera.h:425-448.  add_area_xz = AddAreaLightSource_XZ, build = PhotonMappingPass (light-shaft photons
 * + their kd-tree), render = Render_RayTracer_GodRay (Whitted + the photons within 0.1 of the line of every ray). */
typedef struct orc_godray orc_godray;
orc_godray *orc_godray_new(orc_scene *s, const double light_colour[3], uint64_t seed);
void orc_godray_add_area_xz(orc_godray *g, double y, const double mn[2], const double mx[2], double interval, const double dir[3]);
void orc_godray_build(orc_godray *g);
size_t orc_godray_num_photons(const orc_godray *g);
void orc_godray_photons(const orc_godray *g, double *out /* 4 per photon: x, y, z, energy; creation order */);
void orc_render_godray(orc_godray *g, const orc_camera *c, const double bg[3], double *rgb, int32_t *prim_id);
/* CCamera::Render_RayTracer_GodRay_Antialiasing (Camera.h:450-537) after PhotonMappingPass */
void orc_render_godray_aa(orc_godray *g, const orc_camera *c, const double aa[6], const double bg[3], double *rgb, unsigned char *edge_mask);
void orc_godray_free(orc_godray *g);

/* single primitive known-answer helper: hit point of object id for one ray (recorded variant) */
void orc_object_first_inter_pt(orc_scene *s, int id, const double ray[6], double out_pt[3], double out_uv[2]);

int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
