// rt_host_capi.cpp -- flat C entry points over the C++ host layer (rt_host.hpp) so that the test and
// benchmark harness (Python, ctypes) can drive the same Scene / Camera code a C++ layout would.
// Nothing here computes: it forwards to rt::host classes, which forward to the rt_hip_* ABI.
#include "rt_host.hpp"

using namespace rt::host;

static thread_local std::string g_msg;
#define GUARD(body)                                                   \
  try { body; return 0; }                                             \
  catch (const Error &e) { g_msg = e.what(); return e.code; }         \
  catch (const std::exception &e) { g_msg = e.what(); return -100; }

extern "C" {

const char *rth_last_error() { return g_msg.c_str(); }

int rth_context_new(int device, void **out) { GUARD(*out = new Context(device)) }
void rth_context_free(void *c) { delete (Context *)c; }
void *rth_context_raw(void *c) { return ((Context *)c)->get(); }

int rth_scene_new(double ambient, void **out) { GUARD(*out = new Scene(ambient)) }
void rth_scene_free(void *s) { delete (Scene *)s; }
static Material mat_from(const double m[8])
{ Material r(m[0], m[1], m[2]); r.m.diff = m[3]; r.m.spec = m[4]; r.m.shiness = m[5]; r.m.emissive = m[6]; r.m.ambient = m[7]; return r; }
void rth_material_preset(int name, double out[8])
{ Material m(name); out[0] = m.m.reflW; out[1] = m.m.refrW; out[2] = m.m.RI; out[3] = m.m.diff; out[4] = m.m.spec; out[5] = m.m.shiness; out[6] = m.m.emissive; out[7] = m.m.ambient; }

int rth_add_sphere(void *s, const double c[3], double r, const double m[8], const double col[3])
{ return ((Scene *)s)->AddSphere(Point3D(c[0], c[1], c[2]), r, mat_from(m), Colour(col[0], col[1], col[2])); }
int rth_add_plane(void *s, const double p0[3], const double n[3], const double m[8], const double col[3])
{ return ((Scene *)s)->AddInftyPlane(Point3D(p0[0], p0[1], p0[2]), Point3D(n[0], n[1], n[2]), mat_from(m), Colour(col[0], col[1], col[2])); }
int rth_add_circle(void *s, const double p0[3], const double n[3], double r, const double m[8], const double col[3])
{ return ((Scene *)s)->AddCircle(Point3D(p0[0], p0[1], p0[2]), Point3D(n[0], n[1], n[2]), r, mat_from(m), Colour(col[0], col[1], col[2])); }
static void add_bezier_impl(Scene *s, const double *ctrl, const double *c, const double *m, const double *col)
{
  Point3D p[4][4];
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) p[i][j] = Point3D(ctrl[(i * 4 + j) * 3], ctrl[(i * 4 + j) * 3 + 1], ctrl[(i * 4 + j) * 3 + 2]);
  s->AddBezierObject(p, Point3D(c[0], c[1], c[2]), mat_from(m), Colour(col[0], col[1], col[2]));
}
static void add_rotational_impl(Scene *s, const double *prof, int n_prof, const double *c, int u_sample_num, double u_range, const double *m, const double *col)
{
  std::vector<std::array<double, 2>> pr;
  for (int k = 0; k < n_prof; k++) pr.push_back(std::array<double, 2>{prof[2 * k], prof[2 * k + 1]});
  s->AddRotationalObject(pr, Point3D(c[0], c[1], c[2]), mat_from(m), Colour(col[0], col[1], col[2]), u_sample_num, u_range);
}
int rth_add_bezier(void *s, const double ctrl[48], const double c[3], const double m[8], const double col[3])
{ GUARD(add_bezier_impl((Scene *)s, ctrl, c, m, col)) }
int rth_add_rotational(void *s, const double *prof, int n_prof, const double c[3], int u_sample_num, double u_range, const double m[8], const double col[3])
{ GUARD(add_rotational_impl((Scene *)s, prof, n_prof, c, u_sample_num, u_range, m, col)) }
int rth_add_triangle(void *s, const double v[9], const double *vn, const double m[8], const double col[3])
{
  Point3D p[3], q[3];
  for (int i = 0; i < 3; i++) { p[i] = Point3D(v[3 * i], v[3 * i + 1], v[3 * i + 2]); if (vn) q[i] = Point3D(vn[3 * i], vn[3 * i + 1], vn[3 * i + 2]); }
  return ((Scene *)s)->AddTriangle(p, vn ? q : nullptr, mat_from(m), Colour(col[0], col[1], col[2]));
}
// bulk variant for large synthetic meshes: n triangles, 9 doubles each, no vertex normals
int rth_add_triangles(void *s, const double *v, size_t n, const double m[8], const double col[3])
{
  Material mm = mat_from(m); Colour c(col[0], col[1], col[2]);
  GUARD(for (size_t k = 0; k < n; k++) { Point3D p[3]; for (int i = 0; i < 3; i++) p[i] = Point3D(v[9 * k + 3 * i], v[9 * k + 3 * i + 1], v[9 * k + 3 * i + 2]); ((Scene *)s)->AddTriangle(p, nullptr, mm, c); })
}
// textures: pixels = rows x cols x 3 bytes, B,G,R (cv::Mat CV_8UC3)
int rth_add_texture_image(void *s, unsigned rows, unsigned cols, const unsigned char *bgr, int *tex)
{ GUARD(*tex = ((Scene *)s)->AddTextureImage(Image(rows, cols, bgr))) }
int rth_read_ppm_texture(void *s, const char *path, int *tex) { GUARD(*tex = ((Scene *)s)->AddTextureImage(Image::ReadPPM(path))) }
int rth_texture_info(void *s, int tex, unsigned *rows, unsigned *cols, const unsigned char **bgr)
{ GUARD(const Image &im = ((Scene *)s)->TextureImage(tex); *rows = im.rows; *cols = im.cols; *bgr = im.bgr.data()) }
int rth_obj_add_texture(void *s, int obj, int tex, double scale) { GUARD(((Scene *)s)->AddTexture(obj, tex, scale)) }
int rth_obj_add_bump_texture(void *s, int obj, int tex, double scale) { GUARD(((Scene *)s)->AddBumpTexture(obj, tex, scale)) }
static int add_triangle_tex_impl(Scene *s, const double *v, const double *vn, const double *tc, const double *m, const double *col)
{
  Point3D p[3], q[3]; double t[3][2];
  for (int i = 0; i < 3; i++) { p[i] = Point3D(v[3 * i], v[3 * i + 1], v[3 * i + 2]); if (vn) q[i] = Point3D(vn[3 * i], vn[3 * i + 1], vn[3 * i + 2]); t[i][0] = tc[2 * i]; t[i][1] = tc[2 * i + 1]; }
  return s->AddTriangle(p, vn ? q : nullptr, t, mat_from(m), Colour(col[0], col[1], col[2]));
}
int rth_add_triangle_tex(void *s, const double v[9], const double *vn, const double tc[6], const double m[8], const double col[3], int *id)
{ GUARD(*id = add_triangle_tex_impl((Scene *)s, v, vn, tc, m, col)) }
static int rot_to_mesh_impl(Scene *s, const double *prof, int n_prof, int u_num, int v_num, int tex, const double *c, const double *m, const double *col)
{
  RotationalSurface surf;
  for (int k = 0; k < n_prof; k++) surf.AddControlPt(prof[2 * k], prof[2 * k + 1]);
  Point3D center(c[0], c[1], c[2]);
  return tex >= 0 ? s->RotationalSurfaceToMesh_Texture(surf, u_num, v_num, tex, center, mat_from(m), Colour(col[0], col[1], col[2]))
                  : s->RotationalSurfaceToMesh(surf, u_num, v_num, center, mat_from(m), Colour(col[0], col[1], col[2]));
}
// CScene::RotationalSurfaceToMesh (tex < 0) / RotationalSurfaceToMesh_Texture (Scene.h:168-266)
int rth_rotational_surface_to_mesh(void *s, const double *prof, int n_prof, int u_num, int v_num, int tex, const double c[3], const double m[8],
                                   const double col[3], int *added)
{ GUARD(*added = rot_to_mesh_impl((Scene *)s, prof, n_prof, u_num, v_num, tex, c, m, col)) }
int rth_read_obj_triangle(void *s, const char *path, const double c[3], double scale, const double m[8], const double col[3], int *added)
{ GUARD(*added = ((Scene *)s)->ReadObjFile_Triangle(path, Point3D(c[0], c[1], c[2]), scale, mat_from(m), Colour(col[0], col[1], col[2]))) }
void rth_add_point_light(void *s, const double p[3], double lum) { ((Scene *)s)->AddPointLight(Point3D(p[0], p[1], p[2]), lum); }
void rth_add_parallel_light(void *s, const double d[3], double lum) { ((Scene *)s)->AddParallelLight(Point3D(d[0], d[1], d[2]), lum); }
void rth_add_area_light(void *s, const double p[3], double lum, double len, int sn) { ((Scene *)s)->AddAreaLight(Point3D(p[0], p[1], p[2]), lum, len, sn); }
int rth_construct_bsp(void *s, int max_obj, int max_depth) { GUARD(((Scene *)s)->ConstructBSPTree(max_obj > 0 ? max_obj : 100, max_depth > 0 ? max_depth : 40)) }
void rth_set_leaf_refine(void *s, unsigned n) { ((Scene *)s)->SetLeafRefine(n); }
size_t rth_num_objects(void *s) { return ((Scene *)s)->NumObjects(); }
// flattened tree access (for the BSP parity tests): stats = nodes, leaves, empty leaves, refs, max depth, max leaf, root objs, has tree
void rth_tree_stats(void *sv, long long st[8], double root_box[6])
{
  Scene *s = (Scene *)sv;
  for (int i = 0; i < 8; i++) st[i] = 0;
  const auto &N = s->Nodes();
  if (N.empty()) return;
  st[7] = 1; st[0] = (long long)N.size();
  std::vector<std::pair<uint32_t, int>> stack{{0u, 0}};
  while (!stack.empty()) {
    auto [i, d] = stack.back(); stack.pop_back();
    if (d > st[4]) st[4] = d;
    if (N[i].split_axis < 0) { st[1]++; if (N[i].b == 0) st[2]++; st[3] += N[i].b; if (N[i].b > st[5]) st[5] = N[i].b; }
    else { stack.push_back({(uint32_t)N[i].a, d + 1}); stack.push_back({(uint32_t)N[i].b, d + 1}); }
  }
  st[6] = (N[0].split_axis >= 0) ? (long long)s->RootObjects().size() : 0;
  for (int a = 0; a < 3; a++) { root_box[a] = N[0].box_min[a]; root_box[3 + a] = N[0].box_max[a]; }
}
size_t rth_tree_arrays(void *sv, const rt_hip_node **nodes, const uint32_t **refs, size_t *n_refs)
{ Scene *s = (Scene *)sv; *nodes = s->Nodes().data(); *refs = s->Refs().data(); *n_refs = s->Refs().size(); return s->Nodes().size(); }
size_t rth_triangles(void *sv, const rt_hip_triangle **t) { Scene *s = (Scene *)sv; *t = s->Triangles().data(); return s->Triangles().size(); }

int rth_scene_device(void *s, void *ctx, void **raw) { GUARD(*raw = ((Scene *)s)->Device(*(Context *)ctx)) }
int rth_first_inter_pt(void *s, void *ctx, const rt_hip_ray *rays, size_t n, int use_tree, int record, rt_hip_hit *hits)
{ GUARD(if (use_tree) ((Scene *)s)->FirstInterPt_BSPTree(*(Context *)ctx, rays, n, record != 0, hits); else ((Scene *)s)->FirstInterPt(*(Context *)ctx, rays, n, record != 0, hits)) }

int rth_camera_new(const double view[3], const double eye[3], const double left[3], double dist, double hor, double ver, int res, int depth, void **out)
{ GUARD(*out = new Camera(Point3D(view[0], view[1], view[2]), Point3D(eye[0], eye[1], eye[2]), Point3D(left[0], left[1], left[2]), dist, hor, ver, res, depth)) }
void rth_camera_free(void *c) { delete (Camera *)c; }
void rth_camera_size(void *c, int *w, int *h) { *w = ((Camera *)c)->width(); *h = ((Camera *)c)->height(); }
void rth_camera_seed(void *c, uint64_t seed) { ((Camera *)c)->seed = seed; }
const rt_hip_camera *rth_camera_raw(void *c) { return &((Camera *)c)->raw(); }
int rth_render_raytracer(void *c, void *ctx, void *s, const double bg[3])
{ GUARD(((Camera *)c)->Render_RayTracer(*(Context *)ctx, *(Scene *)s, Colour(bg[0], bg[1], bg[2]))) }
int rth_render_focusing(void *c, void *ctx, void *s, double radius, const double bg[3], unsigned spp)
{ GUARD(((Camera *)c)->Render_Focusing(*(Context *)ctx, *(Scene *)s, radius, Colour(bg[0], bg[1], bg[2]), spp)) }
int rth_render_raytracer_antialiasing(void *c, void *ctx, void *s, const double bg[3], unsigned os)
{ GUARD(((Camera *)c)->Render_RayTracer_Antialiasing(*(Context *)ctx, *(Scene *)s, Colour(bg[0], bg[1], bg[2]), os)) }
int rth_render_focusing_antialiasing(void *c, void *ctx, void *s, double radius, const double bg[3], unsigned spp, unsigned os)
{ GUARD(((Camera *)c)->Render_Focusing_Antialiasing(*(Context *)ctx, *(Scene *)s, radius, Colour(bg[0], bg[1], bg[2]), spp, os)) }
int rth_render_photon_mapper(void *c, void *ctx, void *s, int times, const double bg[3], unsigned photon_num, const char *checkpoint, const char *progress)
{ GUARD(((Camera *)c)->Render_PhotonMapper(*(Context *)ctx, *(Scene *)s, times, Colour(bg[0], bg[1], bg[2]), photon_num, checkpoint ? checkpoint : "", progress ? progress : "")) }
const uint8_t *rth_camera_edge_mask(void *c) { return ((Camera *)c)->edge_mask().empty() ? nullptr : ((Camera *)c)->edge_mask().data(); }
const double *rth_camera_rgb(void *c) { return ((Camera *)c)->rgb().data(); }
const int32_t *rth_camera_prim_id(void *c) { return ((Camera *)c)->prim_id().empty() ? nullptr : ((Camera *)c)->prim_id().data(); }
const uint8_t *rth_camera_image(void *c) { return ((Camera *)c)->image().data(); }
int rth_camera_write_ppm(void *c, const char *path) { GUARD(((Camera *)c)->WritePPM(path)) }

}  // extern "C"
