/*
 * rt_hip.h -- C ABI of the MI355X (gfx950) ray-scene hot path.
 *
 * The reference renderer (FlyingGiraffe/ray_tracing) has no FFI: its seams are C++
 * member calls inside one translation unit.  This header is the boundary that the
 * reference-shaped C++ host layer (ray_tracing_amd/host/) calls instead of those
 * members.  Each entry point cites the reference call it replaces; paths are relative
 * to /root/reference/RayTracing/.
 *
 * Conventions
 *  - plain C, POD structs with explicit sizes, no C++ or torch types;
 *  - every function returns RT_HIP_OK (0) or a negative RT_HIP_E* code and never
 *    throws or aborts; rt_hip_last_error() gives the text for the calling thread;
 *  - host-pointer entry points are synchronous: inputs are copied during the call and
 *    outputs are complete on return;
 *  - *_dev entry points take DEVICE pointers plus a hipStream_t (passed as void*), enqueue
 *    their work on that stream and return; the caller synchronises.  They wait for the
 *    stream only where a size must be known on the host: the first call that needs a larger
 *    workspace (hipMalloc), the edge count of the anti-aliasing drivers, a ray-tree level
 *    whose bound exceeds 6 GiB (deep trees through glass at full resolution), photon-map begin;
 *  - one context = one GPU; one process per GPU is the intended multi-GPU model (the
 *    pixel / sample range arguments of the render calls are the sharding hooks; the
 *    exchange of the shards is done by the caller, e.g. RCCL all-reduce / all-gather);
 *  - a context and its scenes may be used by one thread at a time;
 *  - there is NO CPU fallback: without a usable HIP device every call fails with
 *    RT_HIP_EHIP.
 *
 * Numeric contract (all arithmetic IEEE double, constants of Geometry.h:12-13,48):
 *  - primitive ids are indices into CScene::objects (Scene.h:18), -1 = miss;
 *  - an accepted hit's t is computed with the reference's own operation order
 *    (e.g. Object.h:416-446 for triangles), so it is bit-identical to a CPU build of
 *    the reference formula; the BSP walk is front-to-back with early exit, which can
 *    differ from the reference's exhaustive walk (Scene.h:406-418) only in which of
 *    two hits closer than 1e-8 to each other is reported.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_HIP_OK            0
#define RT_HIP_EINVAL       (-1)  /* bad argument / inconsistent scene description */
#define RT_HIP_EHIP         (-2)  /* HIP runtime error or no device */
#define RT_HIP_ENOMEM       (-3)
#define RT_HIP_EUNSUPPORTED (-4)

/* object types, values of Object.h:17-23 */
#define RT_HIP_SPHERE      0
#define RT_HIP_TRIANGLE    1
#define RT_HIP_POLYGON     2   /* not supported: no layout uses CPolygon */
#define RT_HIP_CIRCLE      3
#define RT_HIP_INFTY_PLANE 4
#define RT_HIP_ROTATIONAL  5
#define RT_HIP_BEZIER      6
/* light types, values of Light.h:7-9 */
#define RT_HIP_POINT_LIGHT    100
#define RT_HIP_PARALLEL_LIGHT 101
#define RT_HIP_AREA_LIGHT     102

typedef struct rt_hip_ctx   rt_hip_ctx;
typedef struct rt_hip_scene rt_hip_scene;

/* CRay (Geometry.h:249-263).  d need not be unit length: like the CRay constructor the
 * library normalises it (Geometry.h:253-254). */
typedef struct { double o[3]; double d[3]; } rt_hip_ray;                 /* 48 B */

/* CScene::CIntersection (Scene.h:23-28) in parametric form: the hit point is
 * o + d*t for sphere/triangle/circle/plane.  u,v: surface parameters for
 * Bezier/rotational objects (cur_u/cur_v of Object.h:805-808), else 1e8. */
typedef struct { double t; double u; double v; int32_t prim_id; int32_t reserved; } rt_hip_hit;  /* 32 B */

typedef struct { double v[3][3]; double vn[3][3]; int32_t has_vn; int32_t reserved; } rt_hip_triangle; /* CTriangle, Object.h:340-344 */
typedef struct { double p0[3]; double n[3]; } rt_hip_plane;              /* CInftyPlane; n already unitized (Object.h:686) */
typedef struct { double c[3]; double r; } rt_hip_sphere;                 /* CSphere, Object.h:246-247 */
typedef struct { double p0[3]; double n[3]; double r; } rt_hip_circle;   /* CCircle; n already unitized (Object.h:652) */
/* CBezierObject (Object.h:1041-1063): bicubic patch, controlPts[i][j] (Bezier.h:96) + center */
typedef struct { double ctrl[4][4][3]; double center[3]; } rt_hip_bezier;                         /* 408 B */
/* CRotationalObject (Object.h:791-808): Bezier profile curve (u = radius, v = height; 2..8 control points,
 * Bezier.h:13), rotated about the z axis through center; uSampleNum Newton seeds over [0, uRange] */
typedef struct { double profile[8][2]; double center[3]; double u_range; int32_t n_profile; int32_t u_sample_num; } rt_hip_rotational; /* 168 B */
/* CMaterial, Material.h:33-42 */
typedef struct { double reflW, refrW, RI, diff, spec, shiness, emissive, ambient; } rt_hip_material;
/* CPointLight / CParallelLight / CAreaLight (Light.h:48-124).  pos_or_dir: position, or the
 * already unitized direction of a parallel light (Light.h:77). */
typedef struct { int32_t type; int32_t sample_num; double pos_or_dir[3]; double luminosity; double len; } rt_hip_light;

/* A texture or bump texture (cv::Mat CV_8UC3 as cv::imread returns it, Object.h:198-210):
 * bgr[(row * cols + col) * 3 + {0,1,2}] = B, G, R.  The library copies the pixels. */
typedef struct { uint32_t rows; uint32_t cols; const uint8_t *bgr; } rt_hip_texture;

/* Flattened BSPNode (BSPTree.h:20-38), 64 B.  split_axis = -1 for a leaf (then a = first index
 * into refs[], b = number of refs), else 0/1/2 (then a = left child, b = right child). */
typedef struct {
  double box_min[3]; double box_max[3];
  int32_t split_axis; int32_t a; int32_t b; int32_t reserved;
} rt_hip_node;

/* Everything CScene holds that the hot path reads (Scene.h:17-21), flattened by the host. */
typedef struct {
  uint32_t struct_size;            /* sizeof(rt_hip_scene_desc), for versioning */
  uint32_t n_objects;              /* CScene::objects.size(); prim id = index */
  const uint8_t  *obj_type;        /* [n_objects] RT_HIP_* type */
  const uint32_t *obj_index;       /* [n_objects] index into the array of that type */
  const uint32_t *obj_material;    /* [n_objects] index into materials */
  const double   *obj_colour;      /* [n_objects][3] m_colour R,G,B (Object.h:67) */
  uint32_t n_triangles; const rt_hip_triangle *triangles;
  uint32_t n_planes;    const rt_hip_plane    *planes;
  uint32_t n_spheres;   const rt_hip_sphere   *spheres;
  uint32_t n_circles;   const rt_hip_circle   *circles;
  uint32_t n_materials; const rt_hip_material *materials;
  uint32_t n_lights;    const rt_hip_light    *lights;
  double ambient;                  /* CScene::ambient */
  /* BSP tree as built by BSPTree::Subdivide; n_nodes == 0 means "no tree" (Scene.h:87-88) */
  uint32_t n_nodes;     const rt_hip_node *nodes;      /* nodes[0] = root */
  uint32_t n_refs;      const uint32_t    *refs;       /* leaf object lists, prim ids */
  uint32_t n_root_objects; const uint32_t *root_objects; /* bspTree.root->objects after the build (BSPTree.h:104-105) */
  uint32_t n_beziers;     const rt_hip_bezier     *beziers;
  uint32_t n_rotationals; const rt_hip_rotational *rotationals;
  /* textures and bump textures (Object.h:69-74; colour(): :186-196, Norm(): :79-92).  n_textures == 0: none.
   * Texel indices that the reference lets run out of the image (row == rows at a sphere's pole) are clamped
   * into it; the column of a hit on a surface of revolution, whose angle is any real number, wraps around. */
  uint32_t n_textures;    const rt_hip_texture *textures;
  const int32_t *obj_texture;        /* [n_objects] index into textures, -1 = haveTexture false */
  const int32_t *obj_bump_texture;   /* [n_objects] index into textures, -1 = haveBumpTexture false */
  const double  *obj_texture_scale;  /* [n_objects][2] CInftyPlane textureScale, bumpTextureScale (Object.h:704-705); other types ignore it */
  const double  *tri_texture_coord;  /* [n_triangles][3][2] CTriangle::textureCoord (Object.h:342), or NULL */
} rt_hip_scene_desc;

/* CCamera constructor arguments (Camera.h:17-18) */
typedef struct {
  double view_pt[3]; double eye_dir[3]; double left[3];
  double dist_of_plane; double hor_range; double ver_range;
  int32_t resolution; int32_t depth;
} rt_hip_camera;

/* Run-time values of what the reference fixes at compile time. */
typedef struct {
  uint32_t struct_size;
  double   background[3];   /* RayTracer's background argument (RayTracer.h:8) */
  double   weight_threshold;/* 0 -> 0.1 (RayTracer.h:7) */
  uint64_t seed;            /* counter-based generator standing in for rand() (Light.h:11) */
  uint32_t spp;             /* Render_Focusing samples per pixel; 0 -> 100 (Camera.h:171) */
  double   lens_radius;     /* Render_Focusing radius argument */
  uint64_t pixel_first;     /* shard of the v*W+h raster to render; pixel_count 0 = all */
  uint64_t pixel_count;
  uint32_t sample_first;    /* shard of the spp loop; sample_count 0 = all */
  uint32_t sample_count;
  uint32_t collect_stats;   /* non-zero: count node visits / primitive tests (slower kernels) */
  /* Interleaved tile shard of the pixel loop of Camera.h:59-63 (rt_hip_render_whitted only): the image is cut into
   * super-tiles of 128 x 64 pixels, numbered row-major ((W+127)/128 per row); with tile_stride = N > 1 the call renders
   * super-tiles tile_offset, tile_offset + N, ... -- every rank's share is spread over the whole image, so the shares cost
   * the same (a contiguous eighth of BASELINE's C2 frame costs up to 1.8x the average).  The outputs then hold the call's
   * super-tiles one after the other, 128 x 64 entries each, row-major inside the tile; entries that fall outside the
   * image are not written.  rt_hip_tile_shard_info gives the sizes, rt_hip_tiles_to_frame_dev turns the all-gathered
   * shards of all ranks back into the raster frame.  0 or 1 = no tile shard.  Excludes pixel_count. */
  uint32_t tile_offset;
  uint32_t tile_stride;
  uint32_t frame_format;    /* RT_HIP_FRAME_*: what the rgb argument of rt_hip_render_whitted(_dev / _noalloc) receives */
} rt_hip_params;
#define RT_HIP_FRAME_RGB_F64 0u   /* 3 doubles R,G,B per pixel: CColour before the store of Camera.h:69 (default) */
#define RT_HIP_FRAME_BGR8    1u   /* 3 bytes B,G,R per pixel: the Vec3b picture after (uchar) truncation, Camera.h:69 */

typedef struct {
  uint64_t rays;            /* closest-hit + occlusion queries */
  uint64_t shadow_rays;     /* occlusion queries among them */
  uint64_t node_visits;     /* per ray (lane) box tests */
  uint64_t prim_tests;      /* per ray (lane) primitive cull tests */
  uint64_t exact_tests;     /* per ray (lane) reference-arithmetic confirmations */
  uint64_t wave_node_visits;/* per wave node pops */
  uint64_t wave_prim_tests; /* per wave primitive iterations */
  uint64_t reserved;
} rt_hip_stats;

/* ---- context / errors ------------------------------------------------------------ */
int rt_hip_device_count(int *n);
int rt_hip_context_create(int device_id, rt_hip_ctx **out);
int rt_hip_context_destroy(rt_hip_ctx *ctx);
const char *rt_hip_last_error(void);
/* counters accumulated by calls made with collect_stats != 0; reset = 1 clears them */
int rt_hip_get_stats(rt_hip_ctx *ctx, rt_hip_stats *out, int reset);

/* ---- scene: replaces CScene::ConstructBSPTree's product (Scene.h:286-293) ----------- */
/* Copies the description to the device (replicated per GPU) and derives the traversal layout.
 * The description carries the reference's tree (nodes, leaf references, root objects); which tree the device
 * walks is the library's business, because the closest hit does not depend on it: an object is listed in every
 * cell that its box overlaps (the reference's own rule, Object.h:212-222), candidates of a leaf are compared as
 * in Scene.h:349-387, and every hit is confirmed with the reference's arithmetic.  leaf_refine selects:
 *   RT_HIP_LEAF_REFINE_DEFAULT      the library's own hierarchy over the finite objects: an object partition
 *                                   (every object in one leaf, child boxes = bounds of their objects, cuts chosen
 *                                   by a surface-area cost), tuned leaf size (leaves of fewer than 20 objects,
 *                                   fewer than 8 in scenes of at most 4096 objects);
 *   RT_HIP_LEAF_REFINE_NONE         the reference's tree exactly as described (leaves < 100 objects, BSPTree.h:56-60);
 *   n (2 ...)                       the reference's tree, its leaves split further down to fewer than n objects;
 *   RT_HIP_LEAF_REFINE_OWN | n      the library's own hierarchy with leaves of fewer than n objects.
 * A description without nodes (ConstructBSPTree never called) has no tree in any mode (Scene.h:87-88). */
#define RT_HIP_LEAF_REFINE_DEFAULT 0u
#define RT_HIP_LEAF_REFINE_NONE    0xFFFFFFFFu
#define RT_HIP_LEAF_REFINE_OWN     0x40000000u
int rt_hip_scene_create(rt_hip_ctx *ctx, const rt_hip_scene_desc *desc, uint32_t leaf_refine, rt_hip_scene **out);
int rt_hip_scene_destroy(rt_hip_scene *scene);
/* Another handle on the same scene: shares the geometry in HBM (released with the last handle), owns its
 * render workspace.  One handle serves one call at a time; two handles may render concurrently on two
 * streams.  The reference passes CScene BY VALUE into every RayTracer call (RayTracer.h:8) - a copy per ray;
 * this is the one copy per frame in flight that remains of it.
 * A handle also remembers how long every 8x8-pixel tile of its last depth-1 frame took and starts the longest tiles first when it renders
 * the same tiling again (an animation: same size and shard, camera moved a little).  Only the order of work changes, never a result;
 * RT_HIP_COST_ORDER=0 in the environment turns it off (it is used from about 25,000 tiles per launch on: smaller launches have no tail worth filling). */
int rt_hip_scene_share(rt_hip_scene *scene, rt_hip_scene **out);

/* ---- queries --------------------------------------------------------------------- */
/* CScene::FirstInterPt_BSPTree(ray, record) (Scene.h:85-91) for n rays.
 * use_tree = 0 forces CScene::FirstInterPt (brute force, Scene.h:62-83). */
int rt_hip_trace_closest(rt_hip_scene *scene, const rt_hip_ray *rays, size_t n, int use_tree, int record,
                         rt_hip_hit *hits);
int rt_hip_trace_closest_dev(rt_hip_scene *scene, const rt_hip_ray *d_rays, size_t n, int use_tree, int record,
                             rt_hip_hit *d_hits, uint32_t collect_stats, void *stream);

/* The shadow predicate of CAreaLight::luminosity_BSPTree (Scene.h:447-450) /
 * CPointLight::InShadow (Light.h:62-69): occluded[i] = 1 iff some object is hit at a point X with
 * Dot(o - X, ref_pts[i] - X) < -eps; with ref_pts == NULL the CParallelLight rule
 * (any hit at all, Light.h:87-94) applies. */
int rt_hip_trace_occluded(rt_hip_scene *scene, const rt_hip_ray *rays, const double *ref_pts, size_t n,
                          int use_tree, uint8_t *occluded);
int rt_hip_trace_occluded_dev(rt_hip_scene *scene, const rt_hip_ray *d_rays, const double *d_ref_pts, size_t n, int use_tree,
                              uint8_t *d_occluded, uint32_t collect_stats, void *stream);

/* ---- renders --------------------------------------------------------------------- */
/* image size of a camera: int(horRange*resolution) x int(verRange*resolution) (Camera.h:55-56) */
int rt_hip_camera_size(const rt_hip_camera *cam, int *width, int *height);

/* CCamera::Render_RayTracer (Camera.h:53-75) up to, not including, the uchar store:
 * rgb[(v*W+h)*3 + {0,1,2}] = colour.R(),G(),B() as doubles; prim_id (optional) = primary hit id.
 * With a pixel shard in params the outputs hold pixel_count entries. */
int rt_hip_render_whitted(rt_hip_scene *scene, const rt_hip_camera *cam, const rt_hip_params *params,
                          double *rgb, int32_t *prim_id);
int rt_hip_render_whitted_dev(rt_hip_scene *scene, const rt_hip_camera *cam, const rt_hip_params *params,
                              double *d_rgb, int32_t *d_prim_id, void *stream);

/* CCamera::Render_RayTracer (Camera.h:53-75, depth == 1) for n_frames cameras of one picture size in ONE launch: frame k of cams[k] goes to
 * (char *)d_rgb + k * rgb_pitch_bytes in params->frame_format, its ids (optional) to d_prim_id + k * id_pitch; the results are those of n_frames calls of
 * rt_hip_render_whitted_dev.  For the ranks of a multi-GPU run (params->tile_offset / tile_stride): an eighth of a 1080p frame fills the GPU's wave slots
 * once, the shares of a sub-batch of frames fill it like a whole frame, and the sub-batch is what one all-gather moves anyway.  Scenes without
 * Bezier / rotational objects, at least one light, 1..64 frames; RT_HIP_EUNSUPPORTED otherwise (render the frames one by one). */
int rt_hip_render_whitted_batch_dev(rt_hip_scene *scene, const rt_hip_camera *cams, uint32_t n_frames, const rt_hip_params *params,
                                    void *d_rgb, uint64_t rgb_pitch_bytes, int32_t *d_prim_id, uint64_t id_pitch, void *stream);

/* NO counterpart in the reference.  rt_hip_render_whitted_dev for callers that cannot allocate device memory: code
 * under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same kernels and same results as
 * rt_hip_render_whitted_dev for a camera of depth 1 (Camera.h:53-75 with depth == 1: closest hit, luminosities, local
 * illumination; the frame of BASELINE config C2); the workspace (ray counters, heavy-first wave order, per-light
 * luminosities) lives in a fixed 32 MiB array that every GPU the library runs on holds until the process ends, so the
 * call makes no hipMalloc / hipFree and creates no events or streams.  Limits: camera depth 1, a scene with at least
 * one light, and 256 + 10.1 bytes per 8x8-pixel tile + (unless the scene has exactly one light of one shadow ray)
 * 8 bytes per padded pixel and light must fit the array -- a 1920x1080 frame with up to one multi-sample light, or
 * any frame up to ~25 Mpixel with one single-sample light; RT_HIP_EINVAL with the byte counts otherwise.  All calls
 * on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads, two scene handles)
 * race; calls on one stream are ordered and safe.  rt_hip_last_ray_count reports this call's rays until the next
 * allocation-free call on the same GPU overwrites the counters. */
int rt_hip_render_whitted_noalloc(rt_hip_scene *scene, const rt_hip_camera *cam, const rt_hip_params *params,
                                  double *d_rgb, int32_t *d_prim_id, void *stream);

/* Sizes of a tile shard (rt_hip_params.tile_stride): super-tiles of the camera's image, super-tiles per rank
 * (= ceil(total / tile_stride): every rank's output arrays hold that many x 8192 entries, so that the shards all-gather
 * with equal counts; a rank whose last slot has no super-tile leaves it unwritten). */
int rt_hip_tile_shard_info(const rt_hip_camera *cam, uint32_t tile_stride, uint32_t *super_tiles, uint32_t *per_rank);
/* The all-gathered shards (rank r's per_rank x 8192 entries at d_tiles + r * rank_pitch_bytes; 0 = tightly packed,
 * per_rank * 8192 * elem_bytes; a larger pitch when every rank contributed several frames to one all-gather) -> the
 * raster frame d_frame[v*W+h]; elem_bytes = 24 (f64 R,G,B), 3 (B,G,R bytes), 4 (prim ids) or any other entry size.
 * This is the store picture.at<Vec3b>(v,h) of Camera.h:69 for tiles rendered elsewhere. */
int rt_hip_tiles_to_frame_dev(const rt_hip_camera *cam, uint32_t tile_stride, const void *d_tiles, uint64_t rank_pitch_bytes,
                              uint32_t elem_bytes, void *d_frame, void *stream);
/* The same for n_frames frames in one launch, when every rank contributed several frames to one all-gather: frame g of rank r starts at
 * d_tiles + r * rank_pitch_bytes + g * tiles_frame_pitch_bytes and goes to d_frames + g * frame_pitch_bytes (pitches 0 = tightly packed). */
int rt_hip_tiles_to_frames_dev(const rt_hip_camera *cam, uint32_t tile_stride, const void *d_tiles, uint64_t rank_pitch_bytes, uint32_t elem_bytes,
                               uint32_t n_frames, uint64_t tiles_frame_pitch_bytes, void *d_frames, uint64_t frame_pitch_bytes, void *stream);

/* CCamera::Render_Focusing (Camera.h:171-199): spp lens samples per pixel.  The output is the SUM over
 * the sample shard of sample_colour/spp WITHOUT the running clamp, so that shards add up
 * (all-reduce) and one final min(255, .) gives the reference's clamped running sum (every term
 * is >= 0, Colour.h:41-47). */
int rt_hip_render_focusing(rt_hip_scene *scene, const rt_hip_camera *cam, const rt_hip_params *params,
                           double *rgb);
int rt_hip_render_focusing_dev(rt_hip_scene *scene, const rt_hip_camera *cam, const rt_hip_params *params,
                               double *d_rgb, void *stream);

/* ---- adaptive anti-aliasing: CCamera::Render_RayTracer_Antialiasing (Camera.h:83-169) and, with focusing != 0,
 * CCamera::Render_Focusing_Antialiasing (Camera.h:201-280; params->spp lens samples, params->lens_radius).
 * A pinhole pass gives the depth / normal / type (/ colour) maps; pixels where |Laplace| of a map (3x3 kernel of
 * Camera.h:285, cv::filter2D's default reflect-101 border) exceeds its threshold are re-sampled with
 * oversample_times rays jittered by random(r0) on the picture plane.  Zero fields take the reference's constants
 * (Camera.h:77-82).  rgb as in rt_hip_render_whitted (final CColour values); edge_mask (optional) = 1 where the
 * pixel was re-sampled.  Shards by pixels only (the stencil's neighbour rows are rendered as a halo).  A pixel
 * whose primary ray hits nothing (the reference dereferences NULL there, Camera.h:109) counts with
 * depth = Dist(inftyPt, viewPt), a zero normal and type -1. */
typedef struct {
  uint32_t struct_size;
  uint32_t oversample_times;        /* 0 -> 100 */
  double   r0;                      /* 0 -> 0.007 */
  double   depth_grad_threshold;    /* 0 -> 0.8; a negative threshold is always exceeded: every pixel is re-sampled
                                     * (the fixed-spp jittered render of BASELINE config C4) */
  double   norm_grad_threshold;     /* 0 -> 0.8 */
  double   type_grad_threshold;     /* 0 -> 0.5 */
  double   colour_grad_threshold;   /* 0 -> 80; not used when focusing (Camera.h:258-262) */
  uint32_t focusing;                /* 0: Render_RayTracer_Antialiasing, 1: Render_Focusing_Antialiasing */
  uint32_t reserved;
} rt_hip_aa_params;
int rt_hip_render_antialiasing(rt_hip_scene *scene, const rt_hip_camera *cam, const rt_hip_params *params, const rt_hip_aa_params *aa,
                               double *rgb, uint8_t *edge_mask);
int rt_hip_render_antialiasing_dev(rt_hip_scene *scene, const rt_hip_camera *cam, const rt_hip_params *params, const rt_hip_aa_params *aa,
                                   double *d_rgb, uint8_t *d_edge_mask, void *stream);
/* number of pixels the last anti-aliasing call on this scene re-sampled */
int rt_hip_last_aa_edges(rt_hip_scene *scene, uint64_t *n);

/* ---- progressive photon mapping: CPhotonMapper (PhotonMapper.h:133-376) driven as in CCamera::Render_PhotonMapper
 * (Camera.h:291-313).  begin = the eye pass (RayTracingPass per pixel, :159-220: Whitted + a hit point on every
 * diffuse surface of the ray tree); photon_num = photonNum (:155, 0 -> 50000), times = the passes that will be run
 * (N_emitted, :225).  pass = one iteration of PhotonMappingPass (:226-238).  For several GPUs: every rank calls
 * begin (same hit points), traces its own photon range, all-reduces the accumulators (sum), and calls update.
 * resolve = the CV_8UC3 picture as R,G,B doubles 0..255. ---- */
typedef struct rt_hip_ppm rt_hip_ppm;
int rt_hip_ppm_begin(rt_hip_scene *scene, const rt_hip_camera *cam, const rt_hip_params *params, uint32_t photon_num, uint32_t times,
                     rt_hip_ppm **out);
/* CCamera::Render_PhotonMapper_Antialiasing (Camera.h:315-423) up to PhotonMappingPass: the maps and edge mask of
 * rt_hip_render_antialiasing, then the eye pass in which a masked pixel sends oversample_times jittered rays through RayTracingPass,
 * all of them depositing hit points on that pixel (:384-389), and the other pixels one ray (:393-394).  The passes then add into
 * the CV_32SC3 picture (PhotonMapper.h:365-367) and rt_hip_ppm_resolve divides a masked pixel's sums by oversample_times before the
 * saturating add (Camera.h:414-418).  Everything after begin (trace / update / pass / resolve / checkpoint) is the same object.
 * aa == NULL: the constants of Camera.h:77-82. */
int rt_hip_ppm_begin_aa(rt_hip_scene *scene, const rt_hip_camera *cam, const rt_hip_params *params, const rt_hip_aa_params *aa,
                        uint32_t photon_num, uint32_t times, rt_hip_ppm **out);
int rt_hip_ppm_edge_mask(rt_hip_ppm *ppm, uint8_t *mask /* W*H: 1 = re-sampled pixel; all 0 for rt_hip_ppm_begin */);
int rt_hip_ppm_info(rt_hip_ppm *ppm, uint64_t *n_hitpoints, uint64_t *photons_per_pass, uint64_t *photon_segments, uint64_t *photon_hits);
int rt_hip_ppm_trace(rt_hip_ppm *ppm, uint32_t pass_index, uint64_t photon_first, uint64_t photon_count, void *stream);
int rt_hip_ppm_accumulators(rt_hip_ppm *ppm, double **d_acc, uint64_t *n_doubles);   /* 4 per hit point: M, tau R, G, B */
int rt_hip_ppm_update(rt_hip_ppm *ppm, void *stream);                                  /* RadianceEvaluation, :357-376 */
int rt_hip_ppm_pass(rt_hip_ppm *ppm, uint32_t pass_index);                             /* trace(all) + update, synchronous */
int rt_hip_ppm_resolve(rt_hip_ppm *ppm, double *rgb);
int rt_hip_ppm_resolve_dev(rt_hip_ppm *ppm, double *d_rgb, void *stream);
int rt_hip_ppm_hitpoints(rt_hip_ppm *ppm, double *out /* 11 doubles each: pixel, path, pt[3], norm[3], wgt, R, N */);
int rt_hip_ppm_end(rt_hip_ppm *ppm);
/* Checkpoint / resume of a long run (the reference keeps its 64 passes in one process, Camera.h:309): the state the
 * passes change -- R and N per hit point (PhotonMapper.h:369-373), the sums added to the picture (:363-367), the
 * photon census, the index of the next pass.  Resume = rt_hip_ppm_begin with the same scene, camera, params,
 * photon_num and times (the eye pass is deterministic) + load_state, then passes next_pass, next_pass + 1, ...
 * A state from a different run is refused. */
int rt_hip_ppm_state_size(rt_hip_ppm *ppm, uint64_t *bytes);
int rt_hip_ppm_save_state(rt_hip_ppm *ppm, void *buf, uint64_t bytes);
int rt_hip_ppm_load_state(rt_hip_ppm *ppm, const void *buf, uint64_t bytes, uint32_t *next_pass);

/* ---- god ray: CGodRay (GodRay.h:109-245) driven as in CCamera::Render_RayTracer_GodRay (Camera.h:425-448).
 * create = CGodRay(scene, lightColour); add_area_xz = AddAreaLightSource_XZ (:193-200); build = PhotonMappingPass
 * (:136-140: light-shaft photons + their kd-tree); render = RayTracingPass per pixel (:142-191). ---- */
typedef struct rt_hip_godray rt_hip_godray;
int rt_hip_godray_create(rt_hip_scene *scene, const double light_colour[3], uint64_t seed, rt_hip_godray **out);
int rt_hip_godray_add_area_xz(rt_hip_godray *g, double y, const double min_xz[2], const double max_xz[2], double light_sample_interval,
                              const double direction[3]);
int rt_hip_godray_build(rt_hip_godray *g);
int rt_hip_godray_num_photons(rt_hip_godray *g, uint64_t *n);
int rt_hip_godray_photons(rt_hip_godray *g, double *out /* 4 per photon: x, y, z, energy; creation order */);
int rt_hip_render_godray(rt_hip_scene *scene, rt_hip_godray *g, const rt_hip_camera *cam, const rt_hip_params *params, double *rgb,
                         int32_t *prim_id);
int rt_hip_render_godray_dev(rt_hip_scene *scene, rt_hip_godray *g, const rt_hip_camera *cam, const rt_hip_params *params, double *d_rgb,
                             int32_t *d_prim_id, void *stream);
/* CCamera::Render_RayTracer_GodRay_Antialiasing (Camera.h:450-537) after PhotonMappingPass: maps and mask from the plain RayTracer
 * (:483), masked pixels re-sampled through CGodRay::RayTracingPass (:516-523), the others one such ray (:527-528); rgb, edge_mask and
 * the pixel shard as in rt_hip_render_antialiasing */
int rt_hip_render_godray_antialiasing(rt_hip_scene *scene, rt_hip_godray *g, const rt_hip_camera *cam, const rt_hip_params *params,
                                      const rt_hip_aa_params *aa, double *rgb, uint8_t *edge_mask);
int rt_hip_render_godray_antialiasing_dev(rt_hip_scene *scene, rt_hip_godray *g, const rt_hip_camera *cam, const rt_hip_params *params,
                                          const rt_hip_aa_params *aa, double *d_rgb, uint8_t *d_edge_mask, void *stream);
int rt_hip_godray_destroy(rt_hip_godray *g);

/* ---- multi-GPU exchange: one process per GPU, RCCL over xGMI behind the ABI.  The reference has no counterpart: its only
 * parallelism is the OpenMP pixel loop of Camera.h:59-63, which these calls let a host shard over GPUs without Python --
 * tiles of the picture are all-gathered (rt_hip_params.tile_offset / tile_stride), sample sums of Render_Focusing
 * (Camera.h:191) and the photon statistics of a pass (PhotonMapper.h:334-355) are all-reduced.  Rank 0 makes the id
 * (ncclGetUniqueId) and the launcher hands its bytes to the other ranks (a file, a socket, torch.distributed's store);
 * every rank then calls rt_hip_comm_create (ncclCommInitRank: collective, blocks until all ranks have called).  RCCL is
 * loaded at run time when a communicator of more than one rank is first made; world == 1 needs neither RCCL nor an id
 * (all_gather copies, all_reduce does nothing; with RT_HIP_COMM_FORCE_RCCL=1 in the environment -- a test aid -- a
 * one-rank communicator goes through RCCL as well).  Collectives are enqueued on `stream` and return. ---- */
#define RT_HIP_COMM_ID_BYTES 128
typedef struct rt_hip_comm rt_hip_comm;
int rt_hip_comm_unique_id(unsigned char *id /* RT_HIP_COMM_ID_BYTES */);
int rt_hip_comm_create(rt_hip_ctx *ctx, const unsigned char *id, int rank, int world, rt_hip_comm **out);
int rt_hip_comm_info(rt_hip_comm *comm, int *rank, int *world);
int rt_hip_comm_all_gather(rt_hip_comm *comm, const void *d_send, void *d_recv, size_t bytes_per_rank, void *stream);
int rt_hip_comm_all_reduce_sum_f64(rt_hip_comm *comm, double *d_buf, size_t n, void *stream);
int rt_hip_comm_destroy(rt_hip_comm *comm);

/* Device buffers for a host that holds no HIP headers (ray_tracing_amd/host/rt_host.hpp: the sharded Camera::Render_* keep a rank's shard and the
 * gathered frame on the GPU between the render, the collective and rt_hip_tiles_to_frame_dev).  No counterpart in the reference (one process, host
 * memory only): plain hipMalloc / hipFree / hipMemset / hipMemcpy on the context's device.  upload / download are synchronous; download waits for
 * everything enqueued on the device before it copies. */
int rt_hip_device_alloc(rt_hip_ctx *ctx, size_t bytes, void **d_ptr);
int rt_hip_device_free(rt_hip_ctx *ctx, void *d_ptr);
int rt_hip_device_zero(rt_hip_ctx *ctx, void *d_ptr, size_t bytes);
int rt_hip_device_upload(rt_hip_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int rt_hip_device_download(rt_hip_ctx *ctx, void *dst, const void *d_src, size_t bytes);

/* Per-kernel device time, measured with hipEvents recorded on the caller's stream around every
 * kernel launch of this scene while profiling is on (rt_hip_set_profiling).  Index = RT_HIP_K_*. */
#define RT_HIP_K_PRIMARY        0   /* camera/lens ray generation + closest hit (level 0) */
#define RT_HIP_K_TRACE_NODES    1   /* closest hit of queued secondary rays */
#define RT_HIP_K_SHADE          2   /* local illumination + child ray generation */
#define RT_HIP_K_COMBINE        3
#define RT_HIP_K_RESOLVE        4
#define RT_HIP_K_TRACE_CLOSEST  5   /* rt_hip_trace_closest(_dev) */
#define RT_HIP_K_TRACE_OCCLUDED 6
#define RT_HIP_K_SHADOW         7   /* luminosity of every light at every hit: occlusion walks */
#define RT_HIP_K_ORDER          8   /* heavy-first work ordering (classification + stable partition) */
#define RT_HIP_K_AA             9   /* anti-aliasing drivers: G-buffer maps, edge mask, per-pixel sums */
#define RT_HIP_K_FRAME_DIRECT  10   /* a depth-1 frame in one launch: camera rays, closest hit, luminosities, local illumination */
#define RT_HIP_K_COUNT          12
typedef struct { double ms[RT_HIP_K_COUNT]; uint64_t launches[RT_HIP_K_COUNT]; } rt_hip_profile;
int rt_hip_set_profiling(rt_hip_scene *scene, int on);
/* synchronises the device, adds up the recorded events; reset = 1 clears the sums */
int rt_hip_get_profile(rt_hip_scene *scene, rt_hip_profile *out, int reset);

/* number of rays (closest-hit + occlusion queries) issued by the last render call on this scene */
int rt_hip_last_ray_count(rt_hip_scene *scene, uint64_t *rays, uint64_t *shadow_rays);
/* Diagnostic, no counterpart in the reference: of the shadow rays rt_hip_last_ray_count reports for the last render call, those the library did
 * not trace because it proved their points unoccluded from the point and the area light's rectangle alone (area_light_all_clear;
 * CAreaLight::luminosity, Light.h:126-162, makes every one of them).  They are counted as rays: the figure says how many. */
int rt_hip_last_proved_shadow_rays(rt_hip_scene *scene, uint64_t *n);
/* Diagnostic, no counterpart in the reference: shader-clock cycles every wave (8x8-pixel tile, in the library's tile order: 128x64-pixel
 * super-tiles row by row, 16x8 tiles inside each) of this handle's last depth-1 frame was resident - the figures that order the next frame's
 * work (rt_hip_scene_share).  *n_waves = number of entries (0: no such frame yet); out may be NULL to ask for the size.  Synchronises the device. */
int rt_hip_last_wave_costs(rt_hip_scene *scene, uint32_t *out, uint64_t cap, uint64_t *n_waves);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
