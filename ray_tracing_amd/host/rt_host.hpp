// rt_host.hpp -- C++ host layer that keeps the reference renderer's scene-side interface
// (CScene / CCamera / CMaterial of Scene.h, Camera.h, Material.h) and executes the hot path through
// the rt_hip_* C ABI (include/rt_hip.h).  A layout written against the reference's Layout.h reads
// the same here:
//
//     rt::host::Scene scene(0.5);
//     scene.AddInftyPlane({0,0,0}, {0,0,1}, Material(_PLASTIC_), Colour(255,255,255));
//     scene.ReadObjFile_Triangle("Icosahedron_Triangle.obj", {0,-1,0}, 0.0015, Material(_PLASTIC_), Colour(241,46,24));
//     scene.ConstructBSPTree();
//     scene.AddAreaLight({-10,20,20}, 1, 10, 10);
//     rt::host::Camera camera({-10,0,1.5}, {1,0,0}, {0,-1,0}, 10, 6, 6, 200, 1);
//     camera.Render_RayTracer(scene, Colour(200,200,200));      // -> camera.image()
//
// What stays on the host (as in the reference): OBJ parsing (Parser.h), materials, lights, the
// BSP-tree BUILD (BSPTree.h:58-109).  What moves behind the ABI: every ray-scene query and all
// shading.  The OpenMP pixel loops of Camera.h disappear because rt_hip_render_* is a batch call.
//
// Build with -ffp-contract=off so that host-side derived values (unitized normals, boxes) round as
// in the reference.  Paths cited are relative to /root/reference/RayTracing/.
#pragma once
#include "../../include/rt_hip.h"

#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

// material preset names, Material.h:4-10
#define _MIRROR_ 200
#define _GLASS_ 201
#define _ICE_ 202
#define _METAL_ 203
#define _PORCELAIN_ 204
#define _RUBBER_ 205
#define _PLASTIC_ 206

namespace rt { namespace host {

constexpr double eps = 1e-8;     // Geometry.h:12
constexpr double infty = 1e+8;   // Geometry.h:48

struct Point3D {                 // CPoint3D, Geometry.h:16-46
  double x = 0, y = 0, z = 0;
  Point3D() {}
  Point3D(double a, double b, double c) : x(a), y(b), z(c) {}
  Point3D operator+(const Point3D &p) const { return Point3D(x + p.x, y + p.y, z + p.z); }
  Point3D operator-(const Point3D &p) const { return Point3D(x - p.x, y - p.y, z - p.z); }
  Point3D operator*(double l) const { return Point3D(x * l, y * l, z * l); }
  Point3D operator^(const Point3D &p) const { return Point3D(y * p.z - z * p.y, z * p.x - x * p.z, x * p.y - y * p.x); }
  bool operator==(const Point3D &p) const { return std::fabs(x - p.x) < eps && std::fabs(y - p.y) < eps && std::fabs(z - p.z) < eps; }
  double Len() const { return std::sqrt(x * x + y * y + z * z); }
  Point3D Unitize() const                                   // Geometry.h:113-119
  {
    if (*this == Point3D(0, 0, 0) || *this == Point3D(infty, infty, infty)) return Point3D(infty, infty, infty);
    double l = Len();
    return Point3D(x / l, y / l, z / l);
  }
};

struct Colour { double R = 0, G = 0, B = 0; Colour() {} Colour(double r, double g, double b) : R(r), G(g), B(b) {} };  // Colour.h:7-24

struct Material {                // CMaterial, Material.h:13-124
  rt_hip_material m{};
  Material(double reflW, double refrW, double RI) { m.reflW = reflW; m.refrW = refrW; m.RI = RI; }
  explicit Material(int name)
  {
    auto set = [&](double a, double b, double c, double d, double e, double f, double g, double h) {
      m.reflW = a; m.refrW = b; m.RI = c; m.diff = d; m.spec = e; m.shiness = f; m.emissive = g; m.ambient = h; };
    switch (name) {
      case _MIRROR_: set(1, 0, 0, 0, 1, 2, 0, 0); break;
      case _GLASS_: set(0.05, 0.95, 1.52, 0, 0.05, 20, 0, 0); break;
      case _ICE_: set(0.05, 0.95, 1.31, 0, 0.05, 20, 0, 1); break;
      case _METAL_: set(0.7, 0, 0, 0.2, 0.8, 20, 0, 1); break;
      case _PORCELAIN_: set(0.1, 0, 0, 1, 0.1, 2, 0, 1); break;
      case _RUBBER_: set(0, 0, 0, 0.3, 0.4, 1, 0, 1); break;
      case _PLASTIC_: set(0, 0, 0, 0.6, 0.0, 1, 0, 1); break;
      default: throw std::invalid_argument("unknown material preset");
    }
  }
};

struct Error : std::runtime_error { int code; Error(int c, const std::string &w) : std::runtime_error(w), code(c) {} };
inline void check(int rc, const char *what) { if (rc != RT_HIP_OK) throw Error(rc, std::string(what) + ": " + rt_hip_last_error()); }

class Context {                  // one GPU
 public:
  explicit Context(int device = 0) { check(rt_hip_context_create(device, &ctx_), "rt_hip_context_create"); }
  ~Context() { rt_hip_context_destroy(ctx_); }
  Context(const Context &) = delete; Context &operator=(const Context &) = delete;
  rt_hip_ctx *get() const { return ctx_; }
 private:
  rt_hip_ctx *ctx_ = nullptr;
};

// CObjMesh, Parser.h:9-82 (token loop of LoadObjFile restated; only v / vt / vn / f are read)
// One rank of a multi-GPU render (one process per GPU, RCCL over xGMI behind rt_hip_comm_*).  The reference has no counterpart: its only
// parallelism is the OpenMP pixel loop of Camera.h:59-63, which the Camera::Render_*(..., comm) overloads below split over the ranks.
//   rank 0:      auto id = Comm::UniqueId();   ... hand the 128 bytes to the other ranks (file, socket, MPI, launcher) ...
//   every rank:  Comm comm(gpu, id, rank, world);       // collective: returns when all ranks have called
class Comm {
 public:
  typedef std::array<unsigned char, RT_HIP_COMM_ID_BYTES> Id;
  static Id UniqueId() { Id id{}; check(rt_hip_comm_unique_id(id.data()), "rt_hip_comm_unique_id"); return id; }
  Comm(Context &ctx, const Id &id, int rank, int world) : ctx_(ctx.get()) { check(rt_hip_comm_create(ctx.get(), id.data(), rank, world, &c_), "rt_hip_comm_create"); rank_ = rank; world_ = world; }
  explicit Comm(Context &ctx) : ctx_(ctx.get()) { check(rt_hip_comm_create(ctx.get(), nullptr, 0, 1, &c_), "rt_hip_comm_create"); }     // a single rank: no RCCL needed
  ~Comm() { rt_hip_comm_destroy(c_); }
  Comm(const Comm &) = delete; Comm &operator=(const Comm &) = delete;
  int rank() const { return rank_; }
  int world() const { return world_; }
  rt_hip_comm *get() const { return c_; }
  rt_hip_ctx *ctx() const { return ctx_; }
 private:
  rt_hip_ctx *ctx_ = nullptr; rt_hip_comm *c_ = nullptr; int rank_ = 0, world_ = 1;
};
// a device buffer of the C ABI's (the host layer includes no HIP header)
class DeviceBuffer {
 public:
  DeviceBuffer(rt_hip_ctx *ctx, size_t bytes) : ctx_(ctx), bytes_(bytes) { check(rt_hip_device_alloc(ctx, bytes, &p_), "rt_hip_device_alloc"); check(rt_hip_device_zero(ctx, p_, bytes), "rt_hip_device_zero"); }
  ~DeviceBuffer() { rt_hip_device_free(ctx_, p_); }
  DeviceBuffer(const DeviceBuffer &) = delete; DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  void *get() const { return p_; }
  size_t bytes() const { return bytes_; }
  void download(void *dst, size_t n) const { check(rt_hip_device_download(ctx_, dst, p_, n), "rt_hip_device_download"); }
 private:
  rt_hip_ctx *ctx_; void *p_ = nullptr; size_t bytes_;
};

struct ObjMesh {
  struct Tuple { int pointIndex, texCoordsIndex, normIndex; };
  std::vector<Point3D> points, normals;
  std::vector<std::array<double, 2>> texCoords;
  std::vector<std::vector<Tuple>> faces;
  explicit ObjMesh(const std::string &filename)
  {
    std::fstream in; in.open(filename, std::ios_base::in);
    if (!in.is_open()) throw std::runtime_error("cannot open " + filename);
    std::string flag;
    for (; in.good();) {
      in >> flag;                                      // a failed read leaves `flag` as it was (Parser.h:42)
      if (flag == "v") { double x = 0, y = 0, z = 0; in >> x >> y >> z; points.push_back(Point3D(x, y, z)); }
      else if (flag == "vt") { double u = 0, v = 0; in >> u >> v; texCoords.push_back({u, v}); }
      else if (flag == "vn") { double x = 0, y = 0, z = 0; in >> x >> y >> z; normals.push_back(Point3D(x, y, z)); }
      else if (flag == "f") {
        char tuples[2048] = {};
        in.getline(tuples, 2048, '\n');
        std::vector<Tuple> face;
        int a = 0, b = 0, c = 0, read = 0;
        for (int i = 0; std::sscanf(&tuples[i], "%i/%i/%i%n", &a, &b, &c, &read) == 3;) {   // Parser.h:68
          i += read; face.push_back({a - 1, b - 1, c - 1});
        }
        if (!face.empty()) faces.push_back(face);
      }
    }
  }
};

// cv::Mat CV_8UC3 as cv::imread returns it (Object.h:198-210): rows x cols, B,G,R bytes.  Decoding JPEG / PNG is the
// application's business (the reference leaves it to OpenCV); binary PPM is read here for C++ layouts and tests.
struct Image {
  uint32_t rows = 0, cols = 0;
  std::vector<uint8_t> bgr;
  Image() {}
  Image(uint32_t r, uint32_t c, const uint8_t *data) : rows(r), cols(c), bgr(data, data + (size_t)r * c * 3) {}
  static Image ReadPPM(const std::string &path)
  {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    auto token = [&](long &out) {                        // header integers, '#' comments skipped
      int ch = std::fgetc(f);
      for (;;) {
        while (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r') ch = std::fgetc(f);
        if (ch != '#') break;
        while (ch != '\n' && ch != EOF) ch = std::fgetc(f);
      }
      if (ch < '0' || ch > '9') return false;
      out = 0;
      while (ch >= '0' && ch <= '9') { out = out * 10 + (ch - '0'); ch = std::fgetc(f); }
      return true;                                      // the single whitespace after the token is consumed
    };
    Image im; long w = 0, h = 0, mx = 0;
    bool ok = std::fgetc(f) == 'P' && std::fgetc(f) == '6' && token(w) && token(h) && token(mx) && w > 0 && h > 0 && mx == 255 &&
              w <= 32768 && h <= 32768;
    if (ok) {
      im.rows = (uint32_t)h; im.cols = (uint32_t)w; im.bgr.resize((size_t)w * h * 3);
      ok = std::fread(im.bgr.data(), 1, im.bgr.size(), f) == im.bgr.size();
      for (size_t i = 0; ok && i < (size_t)w * h; i++) std::swap(im.bgr[3 * i], im.bgr[3 * i + 2]);    // RGB -> BGR
    }
    std::fclose(f);
    if (!ok) throw std::runtime_error("not a binary 8-bit PPM: " + path);
    return im;
  }
};

// CBezierCurve + CRotationalSurface (Bezier.h:8-87): the profile curve P(t) = (radius, height) and the surface
// S(t, theta) it sweeps about the z axis; used on the host by RotationalSurfaceToMesh
struct RotationalSurface {
  std::vector<std::array<double, 2>> controlPts;         // AddControlPt order (Bezier.h:19)
  void AddControlPt(double u, double v) { controlPts.push_back({u, v}); }
  int degree() const { return (int)controlPts.size() - 1; }
  std::array<double, 2> Point2D(double t, int i, int deg) const                               // Bezier.h:28-34
  {
    if (deg == 0) return controlPts[i];
    auto a = Point2D(t, i, deg - 1), b = Point2D(t, i + 1, deg - 1);
    return {a[0] * (1 - t) + b[0] * t, a[1] * (1 - t) + b[1] * t};
  }
  std::array<double, 2> TanVec(double t) const                                                // Bezier.h:36-39
  { auto a = Point2D(t, 1, degree() - 1), b = Point2D(t, 0, degree() - 1); return {(a[0] - b[0]) * degree(), (a[1] - b[1]) * degree()}; }
  std::array<double, 2> Norm2D(double t) const                                                // Bezier.h:41-45, Geometry.h:219-225
  {
    auto tv = TanVec(t);
    double u = tv[1], v = -(tv[0]);
    if ((std::fabs(u) < eps && std::fabs(v) < eps) || (std::fabs(u - infty) < eps && std::fabs(v - infty) < eps)) return {infty, infty};
    double l = std::sqrt(u * u + v * v);
    return {u / l, v / l};
  }
  Point3D Point3D_(double t, double th) const { auto p = Point2D(t, 0, degree()); return Point3D(p[0] * std::cos(th), p[0] * std::sin(th), p[1]); }   // Bezier.h:65-69
  Point3D Norm(double t, double th) const { auto r = Norm2D(t); return Point3D(r[0] * std::cos(th), r[0] * std::sin(th), r[1]).Unitize(); }         // Bezier.h:71-75
};

class Scene {                    // CScene, Scene.h:12-60
 public:
  explicit Scene(double ambient = 0) : ambient_(ambient) {}
  ~Scene() { rt_hip_scene_destroy(dev_); }
  Scene(const Scene &) = delete; Scene &operator=(const Scene &) = delete;

  // ---- objects (ids = insertion order = index in CScene::objects) ----
  int AddSphere(Point3D c, double r, const Material &m, Colour col)               // Scene.h:124
  { rt_hip_sphere s{{c.x, c.y, c.z}, r}; spheres_.push_back(s); return add(RT_HIP_SPHERE, (uint32_t)spheres_.size() - 1, m, col); }
  int AddInftyPlane(Point3D p0, Point3D n, const Material &m, Colour col)         // Scene.h:130, Object.h:685-689
  { Point3D u = n.Unitize(); rt_hip_plane p{{p0.x, p0.y, p0.z}, {u.x, u.y, u.z}}; planes_.push_back(p); return add(RT_HIP_INFTY_PLANE, (uint32_t)planes_.size() - 1, m, col); }
  int AddCircle(Point3D p0, Point3D n, double r, const Material &m, Colour col)   // Object.h:651-656
  { Point3D u = n.Unitize(); rt_hip_circle c{{p0.x, p0.y, p0.z}, {u.x, u.y, u.z}, r}; circles_.push_back(c); return add(RT_HIP_CIRCLE, (uint32_t)circles_.size() - 1, m, col); }
  int AddTriangle(const Point3D v[3], const Point3D *vn, const Material &m, Colour col)   // Object.h:360-384
  {
    rt_hip_triangle t{};
    for (int i = 0; i < 3; i++) { t.v[i][0] = v[i].x; t.v[i][1] = v[i].y; t.v[i][2] = v[i].z; }
    if (vn) { for (int i = 0; i < 3; i++) { t.vn[i][0] = vn[i].x; t.vn[i][1] = vn[i].y; t.vn[i][2] = vn[i].z; } t.has_vn = 1; }
    triangles_.push_back(t);
    for (int i = 0; i < 6; i++) tri_tex_coord_.push_back(0.0);
    return add(RT_HIP_TRIANGLE, (uint32_t)triangles_.size() - 1, m, col);
  }
  // CTriangle(vertex, vertexNorm, textureCoord, ...), Object.h:386-400: textureCoord in texel units (row, col)
  int AddTriangle(const Point3D v[3], const Point3D *vn, const double textureCoord[3][2], const Material &m, Colour col)
  {
    int id = AddTriangle(v, vn, m, col);
    for (int i = 0; i < 3; i++) { tri_tex_coord_[tri_tex_coord_.size() - 6 + 2 * i] = textureCoord[i][0]; tri_tex_coord_[tri_tex_coord_.size() - 6 + 2 * i + 1] = textureCoord[i][1]; }
    return id;
  }
  // ---- textures: an image is registered once (the reference shares one cv::Mat* between objects, Scene.h:243) ----
  int AddTextureImage(const Image &img)
  {
    if (img.rows == 0 || img.cols == 0 || img.bgr.size() != (size_t)img.rows * img.cols * 3) throw std::runtime_error("empty or inconsistent texture image");
    images_.push_back(img); dirty_ = true;
    return (int)images_.size() - 1;
  }
  const Image &TextureImage(int tex) const { return images_.at((size_t)tex); }
  // CBaseObject::AddTexture (Object.h:198-203) / CInftyPlane::AddTexture(filename, textureScale) (Object.h:773-779)
  void AddTexture(int obj, int tex, double textureScale = 1)
  { check_obj_tex(obj, tex); obj_tex_[(size_t)obj] = tex; obj_tex_scale_[2 * (size_t)obj] = textureScale; dirty_ = true; }
  // CBaseObject::AddBumpTexture (Object.h:205-210) / CInftyPlane::AddBumpTexture(filename, bumpTextureScale) (Object.h:781-787)
  void AddBumpTexture(int obj, int tex, double bumpTextureScale = 1)
  { check_obj_tex(obj, tex); obj_bump_[(size_t)obj] = tex; obj_tex_scale_[2 * (size_t)obj + 1] = bumpTextureScale; dirty_ = true; }
  // Scene.h:168-213: the surface of revolution as (uNum x vNum x 2) triangles with vertex normals.  The loops run on
  // accumulated doubles exactly as the reference's do (u += du, v += dv, snapped to 1 and 2 PI within 100 eps).
  int RotationalSurfaceToMesh(const RotationalSurface &surface, int uNum, int vNum, Point3D center, const Material &m, Colour col)
  { return rot_to_mesh(surface, uNum, vNum, -1, center, m, col); }
  // Scene.h:215-266: the same with textureCoord = (u rows, v cols / 2 PI) per vertex and the texture on every triangle
  // (no snapping of u, v in this variant)
  int RotationalSurfaceToMesh_Texture(const RotationalSurface &surface, int uNum, int vNum, int tex, Point3D center, const Material &m, Colour col)
  { if (tex < 0 || (size_t)tex >= images_.size()) throw std::runtime_error("texture index out of range"); return rot_to_mesh(surface, uNum, vNum, tex, center, m, col); }
  // CBezierObject(controlPts[4][4], center, material, colour), Object.h:1075
  int AddBezierObject(const Point3D ctrl[4][4], Point3D center, const Material &m, Colour col)
  {
    rt_hip_bezier b{};
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { b.ctrl[i][j][0] = ctrl[i][j].x; b.ctrl[i][j][1] = ctrl[i][j].y; b.ctrl[i][j][2] = ctrl[i][j].z; }
    b.center[0] = center.x; b.center[1] = center.y; b.center[2] = center.z;
    beziers_.push_back(b);
    return add(RT_HIP_BEZIER, (uint32_t)beziers_.size() - 1, m, col);
  }
  // CRotationalObject(bezierCurve, center, material, colour, uSampleNum, uRange = 1), Object.h:818; profile = the
  // curve's control points in AddControlPt order (Bezier.h:19), (u, v) = (radius, height)
  int AddRotationalObject(const std::vector<std::array<double, 2>> &profile, Point3D center, const Material &m, Colour col,
                          int uSampleNum, double uRange = 1)
  {
    if (profile.size() < 2 || profile.size() > 8) throw std::runtime_error("rotational object: 2..8 profile control points");
    if (uSampleNum < 1) throw std::runtime_error("rotational object: uSampleNum < 1");
    rt_hip_rotational r{};
    for (size_t k = 0; k < profile.size(); k++) { r.profile[k][0] = profile[k][0]; r.profile[k][1] = profile[k][1]; }
    r.center[0] = center.x; r.center[1] = center.y; r.center[2] = center.z;
    r.u_range = uRange; r.n_profile = (int32_t)profile.size(); r.u_sample_num = uSampleNum;
    rotationals_.push_back(r);
    return add(RT_HIP_ROTATIONAL, (uint32_t)rotationals_.size() - 1, m, col);
  }
  // Scene.h:150-166: every face -> CTriangle of its first three tuples, vertex = point*scale + center
  int ReadObjFile_Triangle(const std::string &filename, Point3D center, double scale, const Material &m, Colour col)
  {
    ObjMesh mesh(filename);
    int added = 0;
    for (const auto &face : mesh.faces) {
      if (face.size() < 3) throw std::runtime_error("OBJ face with fewer than 3 vertices");
      Point3D v[3], vn[3];
      for (int k = 0; k < 3; k++) {
        if (face[k].pointIndex < 0 || (size_t)face[k].pointIndex >= mesh.points.size() ||
            face[k].normIndex < 0 || (size_t)face[k].normIndex >= mesh.normals.size())
          throw std::runtime_error("OBJ index out of range");
        v[k] = mesh.points[face[k].pointIndex] * scale + center;
        vn[k] = mesh.normals[face[k].normIndex];
      }
      AddTriangle(v, vn, m, col); added++;
    }
    return added;
  }
  // ---- lights ----
  void AddParallelLight(Point3D dir, double lum)                                   // Scene.h:268, Light.h:76-77
  { Point3D u = dir.Unitize(); rt_hip_light l{RT_HIP_PARALLEL_LIGHT, 0, {u.x, u.y, u.z}, lum, 0}; lights_.push_back(l); dirty_ = true; }
  void AddPointLight(Point3D pos, double lum)                                      // Scene.h:274
  { rt_hip_light l{RT_HIP_POINT_LIGHT, 0, {pos.x, pos.y, pos.z}, lum, 0}; lights_.push_back(l); dirty_ = true; }
  void AddAreaLight(Point3D pos, double lum, double len, int sampleNum)            // Scene.h:280
  { rt_hip_light l{RT_HIP_AREA_LIGHT, sampleNum, {pos.x, pos.y, pos.z}, lum, len}; lights_.push_back(l); dirty_ = true; }

  // ---- BSP tree: CScene::ConstructBSPTree (Scene.h:286-293) + BSPTree::Subdivide (BSPTree.h:58-109),
  //      emitted directly as the flattened arrays of rt_hip_scene_desc ----
  void ConstructBSPTree(int maxObjNum = 100, int maxDepth = 40)
  {
    nodes_.clear(); refs_.clear(); root_objects_.clear(); dirty_ = true;
    compute_boxes();
    Box big;
    if (!biggest_aabb(big)) return;                       // no finite object: no tree (Scene.h:289-290)
    std::vector<uint32_t> all(type_.size());
    for (uint32_t i = 0; i < all.size(); i++) all[i] = i;
    max_obj_ = maxObjNum; max_depth_ = maxDepth;
    subdivide(all, big, 0, 0, true);
  }
  bool HasTree() const { return !nodes_.empty(); }
  size_t NumObjects() const { return type_.size(); }
  const std::vector<rt_hip_node> &Nodes() const { return nodes_; }
  const std::vector<uint32_t> &Refs() const { return refs_; }
  const std::vector<uint32_t> &RootObjects() const { return root_objects_; }
  const std::vector<rt_hip_triangle> &Triangles() const { return triangles_; }

  // ---- device ----
  // RT_HIP_LEAF_REFINE_DEFAULT / RT_HIP_LEAF_REFINE_NONE / a leaf size (acceleration only, results unchanged)
  void SetLeafRefine(uint32_t n) { leaf_refine_ = n; dirty_ = true; }
  rt_hip_scene *Device(Context &ctx)
  {
    if (dev_ && !dirty_ && dev_ctx_ == ctx.get()) return dev_;
    rt_hip_scene_destroy(dev_); dev_ = nullptr;
    rt_hip_scene_desc d{};
    d.struct_size = sizeof d;
    d.n_objects = (uint32_t)type_.size();
    d.obj_type = type_.data(); d.obj_index = index_.data(); d.obj_material = mat_index_.data(); d.obj_colour = colour_.data();
    d.n_triangles = (uint32_t)triangles_.size(); d.triangles = triangles_.data();
    d.n_planes = (uint32_t)planes_.size(); d.planes = planes_.data();
    d.n_spheres = (uint32_t)spheres_.size(); d.spheres = spheres_.data();
    d.n_circles = (uint32_t)circles_.size(); d.circles = circles_.data();
    d.n_materials = (uint32_t)materials_.size(); d.materials = materials_.data();
    d.n_lights = (uint32_t)lights_.size(); d.lights = lights_.data();
    d.ambient = ambient_;
    d.n_nodes = (uint32_t)nodes_.size(); d.nodes = nodes_.data();
    d.n_refs = (uint32_t)refs_.size(); d.refs = refs_.data();
    d.n_root_objects = (uint32_t)root_objects_.size(); d.root_objects = root_objects_.data();
    d.n_beziers = (uint32_t)beziers_.size(); d.beziers = beziers_.data();
    d.n_rotationals = (uint32_t)rotationals_.size(); d.rotationals = rotationals_.data();
    std::vector<rt_hip_texture> tex(images_.size());
    for (size_t k = 0; k < images_.size(); k++) tex[k] = rt_hip_texture{images_[k].rows, images_[k].cols, images_[k].bgr.data()};
    if (!images_.empty()) {
      d.n_textures = (uint32_t)tex.size(); d.textures = tex.data();
      d.obj_texture = obj_tex_.data(); d.obj_bump_texture = obj_bump_.data(); d.obj_texture_scale = obj_tex_scale_.data();
      d.tri_texture_coord = tri_tex_coord_.data();
    }
    check(rt_hip_scene_create(ctx.get(), &d, leaf_refine_, &dev_), "rt_hip_scene_create");
    dev_ctx_ = ctx.get(); dirty_ = false;
    return dev_;
  }
  // CScene::FirstInterPt_BSPTree (Scene.h:85-91) / FirstInterPt (Scene.h:62-83) for a batch of rays
  void FirstInterPt_BSPTree(Context &ctx, const rt_hip_ray *rays, size_t n, bool record, rt_hip_hit *hits)
  { check(rt_hip_trace_closest(Device(ctx), rays, n, 1, record ? 1 : 0, hits), "rt_hip_trace_closest"); }
  void FirstInterPt(Context &ctx, const rt_hip_ray *rays, size_t n, bool record, rt_hip_hit *hits)
  { check(rt_hip_trace_closest(Device(ctx), rays, n, 0, record ? 1 : 0, hits), "rt_hip_trace_closest"); }

 private:
  struct Box { double mn[3], mx[3]; };
  int add(uint8_t type, uint32_t idx, const Material &m, Colour col)
  {
    type_.push_back(type); index_.push_back(idx);
    // materials are de-duplicated by value
    uint32_t mi = 0;
    for (; mi < materials_.size(); mi++) if (std::memcmp(&materials_[mi], &m.m, sizeof m.m) == 0) break;
    if (mi == materials_.size()) materials_.push_back(m.m);
    mat_index_.push_back(mi);
    colour_.push_back(col.R); colour_.push_back(col.G); colour_.push_back(col.B);
    obj_tex_.push_back(-1); obj_bump_.push_back(-1); obj_tex_scale_.push_back(1.0); obj_tex_scale_.push_back(1.0);
    dirty_ = true;
    return (int)type_.size() - 1;
  }
  void check_obj_tex(int obj, int tex) const
  {
    if (obj < 0 || (size_t)obj >= type_.size()) throw std::runtime_error("object id out of range");
    if (tex < 0 || (size_t)tex >= images_.size()) throw std::runtime_error("texture index out of range");
  }
  int rot_to_mesh(const RotationalSurface &surface, int uNum, int vNum, int tex, Point3D center, const Material &m, Colour col)
  {
    const double PI = 3.1415926535;                                                        // Geometry.h:13
    if (uNum < 1 || vNum < 1 || surface.degree() < 1) throw std::runtime_error("RotationalSurfaceToMesh: bad arguments");
    std::vector<Point3D> points, normals;
    const double du = 1 / double(uNum), dv = 2 * PI / double(vNum);
    const double rows = tex >= 0 ? (double)images_[(size_t)tex].rows : 0, cols = tex >= 0 ? (double)images_[(size_t)tex].cols : 0;
    long ptId = 0; int added = 0;
    for (double u = 0, i = 0; u < 1 + eps; u += du, i++) {
      if (tex < 0 && std::fabs(u - 1) < 100 * eps) u = 1;                                  // Scene.h:177-178 (plain variant only)
      for (double v = 0, j = 0; v < 2 * PI + eps; v += dv, j++) {
        if (tex < 0 && std::fabs(v - 2 * PI) < 100 * eps) v = 2 * PI;                      // Scene.h:181-182
        points.push_back(surface.Point3D_(u, v));
        normals.push_back(surface.Norm(u, v));
        if (i != 0 && j != 0) {
          if (ptId - vNum - 2 < 0) throw std::runtime_error("RotationalSurfaceToMesh: the v loop did not run vNum + 1 times");
          const long a = ptId, b = ptId - 1, c = ptId - vNum - 1, d = ptId - vNum - 2;
          Point3D v1[3] = {points[a] + center, points[b] + center, points[c] + center}, n1[3] = {normals[a], normals[b], normals[c]};
          Point3D v2[3] = {points[b] + center, points[d] + center, points[c] + center}, n2[3] = {normals[b], normals[d], normals[c]};
          if (tex >= 0) {
            const double t1[3][2] = {{u * rows, v * cols / (2 * PI)}, {u * rows, (v - dv) * cols / (2 * PI)}, {(u - du) * rows, v * cols / (2 * PI)}};
            const double t2[3][2] = {{u * rows, (v - dv) * cols / (2 * PI)}, {(u - du) * rows, (v - dv) * cols / (2 * PI)}, {(u - du) * rows, v * cols / (2 * PI)}};
            AddTexture(AddTriangle(v1, n1, t1, m, col), tex);
            AddTexture(AddTriangle(v2, n2, t2, m, col), tex);
          } else { AddTriangle(v1, n1, m, col); AddTriangle(v2, n2, m, col); }
          added += 2;
        }
        ptId++;
      }
    }
    return added;
  }
  static bool is_infty_box(const Box &b)
  { for (int a = 0; a < 3; a++) if (!(std::fabs(b.mn[a] - infty) < eps && std::fabs(b.mx[a] - infty) < eps)) return false; return true; }
  void compute_boxes()
  {
    boxes_.resize(type_.size());
    for (size_t i = 0; i < type_.size(); i++) {
      Box b;
      for (int a = 0; a < 3; a++) { b.mn[a] = infty; b.mx[a] = infty; }                  // Object.h:34
      if (type_[i] == RT_HIP_TRIANGLE) {                                                  // Object.h:459-468
        const rt_hip_triangle &t = triangles_[index_[i]];
        for (int a = 0; a < 3; a++) {
          b.mn[a] = std::min(std::min(t.v[0][a], t.v[1][a]), t.v[2][a]) - eps;
          b.mx[a] = std::max(std::max(t.v[0][a], t.v[1][a]), t.v[2][a]) + eps;
        }
      } else if (type_[i] == RT_HIP_SPHERE) {                                             // Object.h:234
        const rt_hip_sphere &s = spheres_[index_[i]];
        for (int a = 0; a < 3; a++) { b.mn[a] = s.c[a] - s.r; b.mx[a] = s.c[a] + s.r; }
      } else if (type_[i] == RT_HIP_BEZIER) {                                             // Object.h:1204-1227
        const rt_hip_bezier &z = beziers_[index_[i]];
        for (int a = 0; a < 3; a++) {
          double lo = infty, hi = -infty;
          for (int p = 0; p < 4; p++) for (int q = 0; q < 4; q++) { if (z.ctrl[p][q][a] < lo) lo = z.ctrl[p][q][a]; if (z.ctrl[p][q][a] > hi) hi = z.ctrl[p][q][a]; }
          b.mn[a] = lo + z.center[a]; b.mx[a] = hi + z.center[a];
        }
      } else if (type_[i] == RT_HIP_ROTATIONAL) {                                         // Object.h:954-970
        const rt_hip_rotational &z = rotationals_[index_[i]];
        const int n = z.n_profile;
        double r = 0;
        for (int k = 0; k < n; k++) if (z.profile[k][0] > r) r = z.profile[k][0];
        b.mn[0] = z.center[0] - r - eps; b.mx[0] = z.center[0] + r + eps;
        b.mn[1] = z.center[1] - r - eps; b.mx[1] = z.center[1] + r + eps;
        b.mn[2] = z.center[2] + std::min(z.profile[0][1], z.profile[n - 1][1]) - eps;
        b.mx[2] = z.center[2] + std::max(z.profile[0][1], z.profile[n - 1][1]) + eps;
      }
      boxes_[i] = b;
    }
  }
  bool biggest_aabb(Box &out) const                                                       // Scene.h:295-317
  {
    double mn[3] = {infty, infty, infty}, mx[3] = {-infty, -infty, -infty};
    for (const Box &b : boxes_) {
      if (is_infty_box(b)) continue;
      for (int a = 0; a < 3; a++) { if (b.mn[a] < mn[a]) mn[a] = b.mn[a]; if (b.mx[a] > mx[a]) mx[a] = b.mx[a]; }
    }
    auto ne = [](double p, double q) { return !(std::fabs(p - q) < eps); };   // operator!= on doubles is exact in the reference
    (void)ne;
    if (mn[0] != infty && mn[1] != infty && mn[2] != infty && mx[0] != -infty && mx[1] != -infty && mx[2] != infty) {
      for (int a = 0; a < 3; a++) { out.mn[a] = mn[a]; out.mx[a] = mx[a]; }
      return true;
    }
    return false;
  }
  bool point_in_box(uint32_t id, const Box &box) const                                     // Object.h:212-222
  {
    if (is_infty_box(box)) return true;
    const Box &b = boxes_[id];
    for (int a = 0; a < 3; a++) if (b.mn[a] > box.mx[a] + eps || b.mx[a] < box.mn[a] - eps) return false;
    return true;
  }
  // returns the index of the emitted node
  uint32_t subdivide(std::vector<uint32_t> &objs, const Box &box, int depth, int axis, bool is_root)
  {
    const uint32_t me = (uint32_t)nodes_.size();
    nodes_.emplace_back();
    {
      rt_hip_node &n = nodes_[me];
      for (int a = 0; a < 3; a++) { n.box_min[a] = box.mn[a]; n.box_max[a] = box.mx[a]; }
      n.split_axis = -1; n.a = 0; n.b = 0; n.reserved = 0;
    }
    if ((int)objs.size() < max_obj_ || depth == max_depth_) {                              // BSPTree.h:60
      nodes_[me].a = (int32_t)refs_.size(); nodes_[me].b = (int32_t)objs.size();
      refs_.insert(refs_.end(), objs.begin(), objs.end());
      if (is_root) root_objects_ = objs;                 // an unsplit root keeps every object (BSPTree.h:60-61)
      return me;
    }
    Box L = box, R = box;
    const double mid = (box.mn[axis] + box.mx[axis]) / 2;                                  // BSPTree.h:71-72 etc.
    L.mx[axis] = mid; R.mn[axis] = mid;
    std::vector<uint32_t> lo, ro;
    for (uint32_t id : objs) {                                                             // BSPTree.h:94-102
      if (point_in_box(id, L)) lo.push_back(id);
      if (point_in_box(id, R)) ro.push_back(id);
      if (is_root && is_infty_box(boxes_[id])) root_objects_.push_back(id);
    }
    std::vector<uint32_t>().swap(objs);                                                    // BSPTree.h:103
    const int next = (axis + 1) % 3;
    uint32_t l = subdivide(lo, L, depth + 1, next, false);
    uint32_t r = subdivide(ro, R, depth + 1, next, false);
    nodes_[me].split_axis = axis; nodes_[me].a = (int32_t)l; nodes_[me].b = (int32_t)r;
    return me;
  }

  double ambient_;
  std::vector<uint8_t> type_; std::vector<uint32_t> index_, mat_index_; std::vector<double> colour_;
  std::vector<rt_hip_triangle> triangles_; std::vector<rt_hip_plane> planes_; std::vector<rt_hip_sphere> spheres_;
  std::vector<rt_hip_circle> circles_; std::vector<rt_hip_material> materials_; std::vector<rt_hip_light> lights_;
  std::vector<rt_hip_bezier> beziers_; std::vector<rt_hip_rotational> rotationals_;
  std::vector<Image> images_; std::vector<int32_t> obj_tex_, obj_bump_; std::vector<double> obj_tex_scale_, tri_tex_coord_;
  std::vector<Box> boxes_;
  std::vector<rt_hip_node> nodes_; std::vector<uint32_t> refs_, root_objects_;
  int max_obj_ = 100, max_depth_ = 40;
  uint32_t leaf_refine_ = RT_HIP_LEAF_REFINE_DEFAULT;
  rt_hip_scene *dev_ = nullptr; rt_hip_ctx *dev_ctx_ = nullptr; bool dirty_ = true;
};

class GodRay {                   // CGodRay, GodRay.h:109-134
 public:
  GodRay(Context &ctx, Scene &scene, Colour lightColour, uint64_t seed = 0) : scene_(scene.Device(ctx))
  {
    const double c[3] = {lightColour.R, lightColour.G, lightColour.B};
    check(rt_hip_godray_create(scene_, c, seed, &g_), "rt_hip_godray_create");
  }
  ~GodRay() { rt_hip_godray_destroy(g_); }
  GodRay(const GodRay &) = delete; GodRay &operator=(const GodRay &) = delete;
  // GodRay.h:193-200; min / max are (x, z) corners
  void AddAreaLightSource_XZ(double y, const double min_xz[2], const double max_xz[2], double lightSampleInterval, Point3D direction)
  {
    const double d[3] = {direction.x, direction.y, direction.z};
    check(rt_hip_godray_add_area_xz(g_, y, min_xz, max_xz, lightSampleInterval, d), "rt_hip_godray_add_area_xz");
  }
  void PhotonMappingPass() { check(rt_hip_godray_build(g_), "rt_hip_godray_build"); }     // GodRay.h:136-140
  rt_hip_godray *get() const { return g_; }
  rt_hip_scene *scene() const { return scene_; }
 private:
  rt_hip_scene *scene_ = nullptr; rt_hip_godray *g_ = nullptr;
};

class Camera {                   // CCamera, Camera.h:14-51
 public:
  Camera(Point3D viewPt, Point3D eyeDirection, Point3D left, double distOfPlane, double horRange, double verRange,
         int resolution, int depth)
  {
    cam_ = rt_hip_camera{{viewPt.x, viewPt.y, viewPt.z}, {eyeDirection.x, eyeDirection.y, eyeDirection.z}, {left.x, left.y, left.z},
                         distOfPlane, horRange, verRange, resolution, depth};
    check(rt_hip_camera_size(&cam_, &w_, &h_), "rt_hip_camera_size");
  }
  int width() const { return w_; }
  int height() const { return h_; }
  const rt_hip_camera &raw() const { return cam_; }
  uint64_t seed = 0;
  // Render_RayTracer, Camera.h:53-75.  rgb(): doubles before the uchar store; image(): the CV_8UC3 picture
  // (BGR, row-major [v][h][3]) with the reference's double->uchar truncation (Camera.h:69).
  void Render_RayTracer(Context &ctx, Scene &scene, Colour background = Colour(0, 0, 0))
  {
    rt_hip_params p = params(background);
    rgb_.assign((size_t)w_ * h_ * 3, 0.0); prim_.assign((size_t)w_ * h_, -1);
    check(rt_hip_render_whitted(scene.Device(ctx), &cam_, &p, rgb_.data(), prim_.data()), "rt_hip_render_whitted");
    to_picture();
  }
  // Render_Focusing, Camera.h:171-199 (spp = the reference's sampleTimes, default 100)
  void Render_Focusing(Context &ctx, Scene &scene, double radius, Colour background = Colour(0, 0, 0), uint32_t spp = 100)
  {
    rt_hip_params p = params(background);
    p.spp = spp; p.lens_radius = radius;
    rgb_.assign((size_t)w_ * h_ * 3, 0.0); prim_.clear();
    check(rt_hip_render_focusing(scene.Device(ctx), &cam_, &p, rgb_.data()), "rt_hip_render_focusing");
    for (double &c : rgb_) c = std::min(255.0, c);        // the running clamp of Colour.h:41-47, applied once
    to_picture();
  }
  // ---- the same renders over the ranks of a communicator (one process per GPU).  Every rank ends with the whole picture. ----
  // Render_RayTracer: the pixel loop of Camera.h:59-63 split by interleaved 128x64-pixel super-tiles (rank r renders super-tiles r, r + world, ...:
  // every rank gets the same mix of cheap and expensive image regions), the shards all-gathered, put back into raster order on the GPU.
  // Bit-identical to the single-GPU call.
  void Render_RayTracer(Context &ctx, Scene &scene, Comm &comm, Colour background = Colour(0, 0, 0))
  {
    rt_hip_params p = params(background);
    p.tile_offset = (uint32_t)comm.rank(); p.tile_stride = (uint32_t)comm.world();
    if (comm.world() == 1) { p.tile_offset = 0; p.tile_stride = 0; }
    const size_t n = (size_t)w_ * h_;
    rgb_.assign(n * 3, 0.0); prim_.assign(n, -1);
    uint32_t total = 0, per_rank = 0;
    check(rt_hip_tile_shard_info(&cam_, (uint32_t)comm.world(), &total, &per_rank), "rt_hip_tile_shard_info");
    const size_t slots = (size_t)per_rank * 8192, W = (size_t)comm.world();
    rt_hip_scene *dev = scene.Device(ctx);
    if (comm.world() == 1) {          // one rank: the frame is already in raster order; the gather is a copy (through RCCL when the communicator has one)
      DeviceBuffer d_rgb(comm.ctx(), n * 24), d_id(comm.ctx(), n * 4), g_rgb(comm.ctx(), n * 24), g_id(comm.ctx(), n * 4);
      check(rt_hip_render_whitted_dev(dev, &cam_, &p, (double *)d_rgb.get(), (int32_t *)d_id.get(), nullptr), "rt_hip_render_whitted_dev");
      check(rt_hip_comm_all_gather(comm.get(), d_rgb.get(), g_rgb.get(), n * 24, nullptr), "rt_hip_comm_all_gather");
      check(rt_hip_comm_all_gather(comm.get(), d_id.get(), g_id.get(), n * 4, nullptr), "rt_hip_comm_all_gather");
      g_rgb.download(rgb_.data(), n * 24); g_id.download(prim_.data(), n * 4);
    } else {
      DeviceBuffer s_rgb(comm.ctx(), slots * 24), s_id(comm.ctx(), slots * 4), g_rgb(comm.ctx(), W * slots * 24), g_id(comm.ctx(), W * slots * 4);
      DeviceBuffer f_rgb(comm.ctx(), n * 24), f_id(comm.ctx(), n * 4);
      check(rt_hip_render_whitted_dev(dev, &cam_, &p, (double *)s_rgb.get(), (int32_t *)s_id.get(), nullptr), "rt_hip_render_whitted_dev");
      check(rt_hip_comm_all_gather(comm.get(), s_rgb.get(), g_rgb.get(), slots * 24, nullptr), "rt_hip_comm_all_gather");
      check(rt_hip_comm_all_gather(comm.get(), s_id.get(), g_id.get(), slots * 4, nullptr), "rt_hip_comm_all_gather");
      check(rt_hip_tiles_to_frame_dev(&cam_, (uint32_t)comm.world(), g_rgb.get(), slots * 24, 24, f_rgb.get(), nullptr), "rt_hip_tiles_to_frame_dev");
      check(rt_hip_tiles_to_frame_dev(&cam_, (uint32_t)comm.world(), g_id.get(), slots * 4, 4, f_id.get(), nullptr), "rt_hip_tiles_to_frame_dev");
      f_rgb.download(rgb_.data(), n * 24); f_id.download(prim_.data(), n * 4);
    }
    to_picture();
  }
  // Render_Focusing: the sample loop of Camera.h:187-192 split over the ranks (rank r takes a contiguous slice of the spp lens samples; the
  // sampler is keyed by (pixel, sample), so the slices are the single-GPU samples), the un-clamped sums all-reduced, one clamp at the end
  // (= the reference's running clamp: every term is >= 0).  Equal to the single-GPU picture up to the order of the additions (<= 1e-12).
  void Render_Focusing(Context &ctx, Scene &scene, Comm &comm, double radius, Colour background = Colour(0, 0, 0), uint32_t spp = 100)
  {
    rt_hip_params p = params(background);
    p.spp = spp; p.lens_radius = radius;
    const uint32_t W = (uint32_t)comm.world(), r = (uint32_t)comm.rank(), per = spp / W, extra = spp % W;
    p.sample_first = r * per + std::min(r, extra); p.sample_count = per + (r < extra ? 1u : 0u);
    const size_t n = (size_t)w_ * h_;
    rgb_.assign(n * 3, 0.0); prim_.clear();
    DeviceBuffer d_rgb(comm.ctx(), n * 24);
    if (p.sample_count) check(rt_hip_render_focusing_dev(scene.Device(ctx), &cam_, &p, (double *)d_rgb.get(), nullptr), "rt_hip_render_focusing_dev");
    check(rt_hip_comm_all_reduce_sum_f64(comm.get(), (double *)d_rgb.get(), n * 3, nullptr), "rt_hip_comm_all_reduce_sum_f64");
    d_rgb.download(rgb_.data(), n * 24);
    for (double &c : rgb_) c = std::min(255.0, c);
    to_picture();
  }
  // Render_PhotonMapper: every rank runs the (deterministic) eye pass and holds the same hit points; a pass's photons are split by index
  // (rt_hip_ppm_trace(first, count)), the per-hit-point sums (M, tau) all-reduced, and every rank applies the same RadianceEvaluation
  // (PhotonMapper.h:357-376).  Photon counts are exact; tau up to the order of the additions.
  void Render_PhotonMapper(Context &ctx, Scene &scene, Comm &comm, int times, Colour background = Colour(0, 0, 0), uint32_t photonNum = 50000)
  {
    rt_hip_params p = params(background);
    rt_hip_ppm *m = nullptr;
    check(rt_hip_ppm_begin(scene.Device(ctx), &cam_, &p, photonNum, (uint32_t)times, &m), "rt_hip_ppm_begin");
    uint64_t hp = 0, per_pass = 0, seg = 0, hits = 0;
    int rc = rt_hip_ppm_info(m, &hp, &per_pass, &seg, &hits);
    const uint64_t W = (uint64_t)comm.world(), r = (uint64_t)comm.rank(), per = per_pass / W, extra = per_pass % W;
    const uint64_t first = r * per + std::min(r, extra), count = per + (r < extra ? 1 : 0);
    for (int t = 0; t < times && !rc; t++) {
      if (count) rc = rt_hip_ppm_trace(m, (uint32_t)t, first, count, nullptr);
      double *acc = nullptr; uint64_t nacc = 0;
      if (!rc) rc = rt_hip_ppm_accumulators(m, &acc, &nacc);
      if (!rc) rc = rt_hip_comm_all_reduce_sum_f64(comm.get(), acc, (size_t)nacc, nullptr);
      if (!rc) rc = rt_hip_ppm_update(m, nullptr);
    }
    rgb_.assign((size_t)w_ * h_ * 3, 0.0); prim_.clear();
    if (!rc) rc = rt_hip_ppm_resolve(m, rgb_.data());
    rt_hip_ppm_end(m);
    check(rc, "Render_PhotonMapper over a communicator");
    to_picture();
  }
  // Render_RayTracer_Antialiasing, Camera.h:83-169: G-buffer pass, Laplacian edge mask, oversampleTimes jittered rays
  // on the masked pixels.  edge_mask(): 1 where a pixel was re-sampled.
  void Render_RayTracer_Antialiasing(Context &ctx, Scene &scene, Colour background = Colour(0, 0, 0), uint32_t oversampleTimes = 100)
  {
    rt_hip_params p = params(background);
    rt_hip_aa_params a{}; a.struct_size = sizeof a; a.oversample_times = oversampleTimes;
    rgb_.assign((size_t)w_ * h_ * 3, 0.0); prim_.clear(); mask_.assign((size_t)w_ * h_, 0);
    check(rt_hip_render_antialiasing(scene.Device(ctx), &cam_, &p, &a, rgb_.data(), mask_.data()), "rt_hip_render_antialiasing");
    to_picture();
  }
  // Render_Focusing_Antialiasing, Camera.h:201-280: spp lens samples per pixel, x oversampleTimes on the masked pixels
  void Render_Focusing_Antialiasing(Context &ctx, Scene &scene, double radius, Colour background = Colour(0, 0, 0), uint32_t spp = 100,
                                    uint32_t oversampleTimes = 100)
  {
    rt_hip_params p = params(background);
    p.spp = spp; p.lens_radius = radius;
    rt_hip_aa_params a{}; a.struct_size = sizeof a; a.oversample_times = oversampleTimes; a.focusing = 1;
    rgb_.assign((size_t)w_ * h_ * 3, 0.0); prim_.clear(); mask_.assign((size_t)w_ * h_, 0);
    check(rt_hip_render_antialiasing(scene.Device(ctx), &cam_, &p, &a, rgb_.data(), mask_.data()), "rt_hip_render_antialiasing");
    to_picture();
  }
  // Render_PhotonMapper, Camera.h:291-313: eye pass, then `times` photon passes of photonNum photons per light
  // (PhotonMapper.h:155: 50000); the picture is the reference's CV_8UC3 image.
  // checkpoint (optional): file that receives the run's state after every pass (written beside it, then renamed); when
  // it exists at the start and belongs to this run, the passes it covers are skipped.  progress (optional): binary PPM
  // rewritten with the picture after every pass (the reference shows the picture once, at the end, Camera.h:310-312).
  void Render_PhotonMapper(Context &ctx, Scene &scene, int times, Colour background = Colour(0, 0, 0), uint32_t photonNum = 50000,
                           const std::string &checkpoint = "", const std::string &progress = "")
  { photon_mapper(ctx, scene, times, background, photonNum, checkpoint, progress, 0); }
  // Render_PhotonMapper_Antialiasing, Camera.h:315-423 (what Layout5_Wine calls, Layout.h:299): the edge mask of
  // Render_RayTracer_Antialiasing, oversampleTimes jittered eye rays per masked pixel all depositing hit points on it, the photon
  // sums of a masked pixel divided by oversampleTimes.  edge_mask(): 1 where a pixel was re-sampled.
  void Render_PhotonMapper_Antialiasing(Context &ctx, Scene &scene, int times, Colour background = Colour(0, 0, 0), uint32_t photonNum = 50000,
                                        uint32_t oversampleTimes = 100, const std::string &checkpoint = "", const std::string &progress = "")
  { photon_mapper(ctx, scene, times, background, photonNum, checkpoint, progress, oversampleTimes ? oversampleTimes : 100); }
  // Render_RayTracer_GodRay_Antialiasing, Camera.h:450-537 (what Layout6_Room calls, Layout.h:321)
  void Render_RayTracer_GodRay_Antialiasing(GodRay &godRay, Colour background = Colour(0, 0, 0), uint32_t oversampleTimes = 100)
  {
    godRay.PhotonMappingPass();
    rt_hip_params p = params(background);
    rt_hip_aa_params a{}; a.struct_size = sizeof a; a.oversample_times = oversampleTimes;
    rgb_.assign((size_t)w_ * h_ * 3, 0.0); prim_.clear(); mask_.assign((size_t)w_ * h_, 0);
    check(rt_hip_render_godray_antialiasing(godRay.scene(), godRay.get(), &cam_, &p, &a, rgb_.data(), mask_.data()), "rt_hip_render_godray_antialiasing");
    to_picture();
  }
 private:
  void photon_mapper(Context &ctx, Scene &scene, int times, Colour background, uint32_t photonNum, const std::string &checkpoint,
                     const std::string &progress, uint32_t oversampleTimes)
  {
    rt_hip_params p = params(background);
    rt_hip_ppm *m = nullptr;
    if (oversampleTimes) {
      rt_hip_aa_params a{}; a.struct_size = sizeof a; a.oversample_times = oversampleTimes;
      check(rt_hip_ppm_begin_aa(scene.Device(ctx), &cam_, &p, &a, photonNum, (uint32_t)times, &m), "rt_hip_ppm_begin_aa");
      mask_.assign((size_t)w_ * h_, 0);
      check(rt_hip_ppm_edge_mask(m, mask_.data()), "rt_hip_ppm_edge_mask");
    } else
      check(rt_hip_ppm_begin(scene.Device(ctx), &cam_, &p, photonNum, (uint32_t)times, &m), "rt_hip_ppm_begin");
    int rc = 0;
    uint32_t first = 0;
    std::vector<unsigned char> state;
    if (!checkpoint.empty()) {
      uint64_t nb = 0; rc = rt_hip_ppm_state_size(m, &nb); state.resize((size_t)nb);
      if (FILE *f = std::fopen(checkpoint.c_str(), "rb")) {
        std::vector<unsigned char> old(state.size());
        const bool whole = std::fread(old.data(), 1, old.size(), f) == old.size();
        std::fclose(f);
        if (whole && rt_hip_ppm_load_state(m, old.data(), old.size(), &first) != 0) first = 0;     // another run's file: start over
      }
    }
    rgb_.assign((size_t)w_ * h_ * 3, 0.0); prim_.clear();
    for (uint32_t t = first; t < (uint32_t)times && !rc; t++) {
      rc = rt_hip_ppm_pass(m, t);
      if (!rc && !checkpoint.empty()) {
        rc = rt_hip_ppm_save_state(m, state.data(), state.size());
        const std::string tmp = checkpoint + ".part";
        FILE *f = rc ? nullptr : std::fopen(tmp.c_str(), "wb");
        if (f) {
          const bool ok = std::fwrite(state.data(), 1, state.size(), f) == state.size();
          if (std::fclose(f) == 0 && ok) std::rename(tmp.c_str(), checkpoint.c_str());
        }
      }
      if (!rc && !progress.empty()) { rc = rt_hip_ppm_resolve(m, rgb_.data()); if (!rc) { to_picture(); WritePPM(progress); } }
    }
    if (!rc) rc = rt_hip_ppm_resolve(m, rgb_.data());
    rt_hip_ppm_end(m);
    check(rc, "rt_hip_ppm_pass");
    to_picture();
  }
 public:
  // Render_RayTracer_GodRay, Camera.h:425-448 (the caller runs godRay.PhotonMappingPass() first; the reference does it here)
  void Render_RayTracer_GodRay(GodRay &godRay, Colour background = Colour(0, 0, 0))
  {
    godRay.PhotonMappingPass();
    rt_hip_params p = params(background);
    rgb_.assign((size_t)w_ * h_ * 3, 0.0); prim_.assign((size_t)w_ * h_, -1);
    check(rt_hip_render_godray(godRay.scene(), godRay.get(), &cam_, &p, rgb_.data(), prim_.data()), "rt_hip_render_godray");
    to_picture();
  }
  const std::vector<double> &rgb() const { return rgb_; }
  const std::vector<int32_t> &prim_id() const { return prim_; }
  const std::vector<uint8_t> &image() const { return picture_; }
  const std::vector<uint8_t> &edge_mask() const { return mask_; }
  // binary PPM instead of cv::imwrite (Camera.h:73)
  void WritePPM(const std::string &path) const
  {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    std::fprintf(f, "P6\n%d %d\n255\n", w_, h_);
    for (size_t i = 0; i < (size_t)w_ * h_; i++) { uint8_t px[3] = {picture_[3 * i + 2], picture_[3 * i + 1], picture_[3 * i]}; std::fwrite(px, 1, 3, f); }
    std::fclose(f);
  }
 private:
  rt_hip_params params(Colour bg) const
  {
    rt_hip_params p{}; p.struct_size = sizeof p;
    p.background[0] = bg.R; p.background[1] = bg.G; p.background[2] = bg.B; p.seed = seed;
    return p;
  }
  void to_picture()
  {
    picture_.resize((size_t)w_ * h_ * 3);
    for (size_t i = 0; i < (size_t)w_ * h_; i++) {
      picture_[3 * i] = (uint8_t)rgb_[3 * i + 2]; picture_[3 * i + 1] = (uint8_t)rgb_[3 * i + 1]; picture_[3 * i + 2] = (uint8_t)rgb_[3 * i];
    }
  }
  rt_hip_camera cam_{}; int w_ = 0, h_ = 0;
  std::vector<double> rgb_; std::vector<int32_t> prim_; std::vector<uint8_t> picture_, mask_;
};

}}  // namespace rt::host
