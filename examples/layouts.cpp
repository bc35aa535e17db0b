// layouts.cpp -- scene scripts of the reference's Layout.h written against the host mirror (rt_host.hpp), to show
// that a layout ports line by line and to exercise the C++ side end to end (tests/test_gpu_examples.py runs it).
//
//   layouts focusing <Icosahedron_Triangle.obj> <out.ppm> [resolution] [spp]     Layout3_Focusing, Layout.h:134-160
//   layouts raytracer <Icosahedron_Triangle.obj> <out.ppm> [resolution] [depth]  the same scene (metal mesh) through Render_RayTracer
//
//   focusing / raytracer / wine-plain take  --rank R --world W --id-file F  after their arguments: rank R of W processes (one per GPU; rank R
//   uses GPU R % the GPUs present), sharded by Camera::Render_*(..., comm) over rt_hip_comm_* (RCCL).  Rank 0 writes the communicator's 128-byte id
//   to F, the others wait for it; every rank ends with the whole picture, rank 0 writes it.
//   layouts apples   <out.ppm> [resolution]                                     Layout1_Apples-like: textured mesh from
//                                                                               RotationalSurfaceToMesh_Texture, textured +
//                                                                               bump-mapped planes, anti-aliased render
//   layouts wine     <out.ppm> [resolution] [times] [photonNum] [oversampleTimes]   Layout5_Wine, Layout.h:237-300: three cups,
//                                                                               Render_PhotonMapper_Antialiasing (untextured)
//   layouts room     <Icosahedron_Triangle.obj> <out.ppm> [resolution] [oversampleTimes]
//                                                                               Layout6_Room, Layout.h:302-323: Render_RayTracer_GodRay_
//                                                                               Antialiasing; Corridor_Triangle.obj is missing from the
//                                                                               reference checkout, a few walls and a mesh stand in
//
// Textures: the reference reads JPEG through OpenCV; here the images are synthetic (or binary PPM via Image::ReadPPM).
#include "../ray_tracing_amd/host/rt_host.hpp"

#include <chrono>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <thread>

using namespace rt::host;

static Image checker(uint32_t rows, uint32_t cols, uint32_t cell, const uint8_t a[3], const uint8_t b[3])
{
  Image im; im.rows = rows; im.cols = cols; im.bgr.resize((size_t)rows * cols * 3);
  for (uint32_t r = 0; r < rows; r++) for (uint32_t c = 0; c < cols; c++) {
    const uint8_t *p = (((r / cell) + (c / cell)) & 1) ? a : b;
    for (int k = 0; k < 3; k++) im.bgr[((size_t)r * cols + c) * 3 + k] = p[k];
  }
  return im;
}

// --rank R --world W --id-file F anywhere after the positional arguments; removed from argv
struct Shard { int rank = 0, world = 0; std::string id_file; };
static Shard take_shard_options(int &argc, char **argv)
{
  Shard s; int out = 0;
  for (int i = 0; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--rank" && i + 1 < argc) s.rank = std::atoi(argv[++i]);
    else if (a == "--world" && i + 1 < argc) s.world = std::atoi(argv[++i]);
    else if (a == "--id-file" && i + 1 < argc) s.id_file = argv[++i];
    else argv[out++] = argv[i];
  }
  argc = out;
  return s;
}
static int gpu_of_rank(int rank) { int n = 1; if (rt_hip_device_count(&n) != 0 || n < 1) n = 1; return rank % n; }
// the communicator of this rank: rank 0 makes the id and leaves it in the file (written beside it, then renamed), the others pick it up
static std::unique_ptr<Comm> make_comm(Context &gpu, const Shard &s)
{
  if (s.world < 1) return nullptr;
  if (s.id_file.empty()) throw std::runtime_error("--world needs --id-file");
  Comm::Id id{};
  if (s.rank == 0) {
    id = Comm::UniqueId();
    const std::string tmp = s.id_file + ".part";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(id.data(), 1, id.size(), f) != id.size()) throw std::runtime_error("cannot write " + tmp);
    std::fclose(f);
    std::rename(tmp.c_str(), s.id_file.c_str());
  } else {
    for (int tries = 0;; tries++) {
      if (FILE *f = std::fopen(s.id_file.c_str(), "rb")) { const size_t n = std::fread(id.data(), 1, id.size(), f); std::fclose(f); if (n == id.size()) break; }
      if (tries > 600) throw std::runtime_error("no communicator id in " + s.id_file);
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  return std::unique_ptr<Comm>(new Comm(gpu, id, s.rank, s.world));
}

static int focusing(int argc, char **argv)
{
  const Shard sh = take_shard_options(argc, argv);
  if (argc < 4) { std::cerr << "usage: layouts focusing <obj> <out.ppm> [resolution] [spp]\n"; return 2; }
  const int res = argc > 4 ? std::atoi(argv[4]) : 100;
  const uint32_t spp = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 100;
  Context gpu(gpu_of_rank(sh.rank));
  std::unique_ptr<Comm> comm = make_comm(gpu, sh);
  Scene scene(0.5);                                                                                     // Layout.h:138
  scene.AddInftyPlane(Point3D(0, 0, 0), Point3D(0, 0, 1), Material(_PLASTIC_), Colour(255, 255, 255));  // :140-142
  scene.AddInftyPlane(Point3D(10, 0, 0), Point3D(-1, 0, 0), Material(_PLASTIC_), Colour(0, 0, 0));      // :144-146
  scene.ReadObjFile_Triangle(argv[2], Point3D(0, -1, 0), 0.0015, Material(_PLASTIC_), Colour(241, 46, 24));   // :151
  scene.ConstructBSPTree();                                                                             // :153
  scene.AddAreaLight(Point3D(-10, 20, 20), 1, 10, 10);                                                  // :155
  Camera camera(Point3D(-10, 0, 1.5), Point3D(1, 0, 0), Point3D(0, -1, 0), 10, 6.4, 4.8, res, 1);       // :157
  camera.seed = 1;
  if (comm) camera.Render_Focusing(gpu, scene, *comm, 0.1, Colour(200, 200, 200), spp);                 // the sample loop over the ranks
  else camera.Render_Focusing(gpu, scene, 0.1, Colour(200, 200, 200), spp);                             // :159
  if (sh.rank == 0) camera.WritePPM(argv[3]);
  return 0;
}

static int raytracer(int argc, char **argv)
{
  const Shard sh = take_shard_options(argc, argv);
  if (argc < 4) { std::cerr << "usage: layouts raytracer <obj> <out.ppm> [resolution] [depth] [--rank R --world W --id-file F]\n"; return 2; }
  const int res = argc > 4 ? std::atoi(argv[4]) : 100, depth = argc > 5 ? std::atoi(argv[5]) : 3;
  Context gpu(gpu_of_rank(sh.rank));
  std::unique_ptr<Comm> comm = make_comm(gpu, sh);
  Scene scene(0.5);
  scene.AddInftyPlane(Point3D(0, 0, 0), Point3D(0, 0, 1), Material(_PLASTIC_), Colour(255, 255, 255));
  scene.AddInftyPlane(Point3D(10, 0, 0), Point3D(-1, 0, 0), Material(_PLASTIC_), Colour(0, 0, 0));
  scene.ReadObjFile_Triangle(argv[2], Point3D(0, -1, 0), 0.0015, Material(_METAL_), Colour(241, 46, 24));
  scene.ConstructBSPTree();
  scene.AddAreaLight(Point3D(-10, 20, 20), 1, 0, 1);
  Camera camera(Point3D(-10, 0, 1.5), Point3D(1, 0, 0), Point3D(0, -1, 0), 10, 6.4, 4.8, res, depth);
  if (comm) camera.Render_RayTracer(gpu, scene, *comm, Colour(200, 200, 200));                          // the pixel loop of Camera.h:59-63 over the ranks
  else camera.Render_RayTracer(gpu, scene, Colour(200, 200, 200));
  if (sh.rank == 0) camera.WritePPM(argv[3]);
  return 0;
}

static int apples(int argc, char **argv)
{
  if (argc < 3) { std::cerr << "usage: layouts apples <out.ppm> [resolution]\n"; return 2; }
  const int res = argc > 3 ? std::atoi(argv[3]) : 60;
  Context gpu(0);
  Scene scene(0.3);
  const uint8_t red[3] = {30, 40, 200}, yellow[3] = {60, 200, 230}, grey[3] = {120, 120, 120}, white[3] = {240, 240, 240};
  const int skin = scene.AddTextureImage(checker(40, 60, 5, red, yellow));
  const int tiles = scene.AddTextureImage(checker(64, 64, 8, grey, white));
  RotationalSurface apple;                                                                              // Layout.h:56-59
  apple.AddControlPt(0, 0.3); apple.AddControlPt(2.2, -0.9); apple.AddControlPt(2.6, 2.4); apple.AddControlPt(0.05, 2.0);
  scene.RotationalSurfaceToMesh_Texture(apple, 5, 15, skin, Point3D(1, -3, 0.9), Material(_PLASTIC_), Colour(255, 255, 255));   // :60
  scene.RotationalSurfaceToMesh(apple, 4, 9, Point3D(4, 3, 0.9), Material(_METAL_), Colour(80, 160, 90));
  const int floor = scene.AddInftyPlane(Point3D(0, 0, 0), Point3D(0, 0, 1), Material(_PORCELAIN_), Colour(255, 255, 255));     // :74-77
  scene.AddTexture(floor, tiles, 5); scene.AddBumpTexture(floor, tiles, 3);
  const int wall = scene.AddInftyPlane(Point3D(12, 0, 0), Point3D(-1, 0, 0), Material(_PLASTIC_), Colour(255, 255, 255));      // :79-82
  scene.AddTexture(wall, tiles, 2);
  scene.ConstructBSPTree();
  scene.AddAreaLight(Point3D(-10, 12, 20), 1.2, 4, 2);
  Camera camera(Point3D(-14, 3, 7), Point3D(1, -0.1, -0.3), Point3D(0, -1, 0), 10, 9.6, 7.2, res, 3);
  camera.seed = 2;
  camera.Render_RayTracer_Antialiasing(gpu, scene, Colour(20, 30, 40), 16);                              // :95
  camera.WritePPM(argv[2]);
  size_t masked = 0; for (uint8_t m : camera.edge_mask()) masked += m;
  std::cout << camera.width() << "x" << camera.height() << ", " << masked << " pixels re-sampled\n";
  return 0;
}

// Object_Cup2, Layout.h:190-235
static void Object_Cup2(Scene &scene, Point3D center, const Material &material, Colour colour)
{
  scene.AddRotationalObject({{2, 0.2}, {0.2, 0.2}, {0.2, 2.2}, {0.2, 4.2}}, center, material, colour, 50);
  scene.AddRotationalObject({{0.2, 4.2}, {0.2, 6.2}, {2, 6.2}, {1.7, 10.2}}, center, material, colour, 50);
  scene.AddRotationalObject({{1.6, 10.2}, {1.65, 8.2}, {1.3, 6.2}, {0, 6.2}}, center, material, colour, 50);
  scene.AddRotationalObject({{2, 0}, {2.1, 0.1}, {2, 0.2}}, center, material, colour, 10);
  scene.AddRotationalObject({{1.7, 10.2}, {1.65, 10.3}, {1.6, 10.2}}, center, material, colour, 10);
}

static int wine(int argc, char **argv)
{
  const Shard sh = take_shard_options(argc, argv);
  if (argc < 3) { std::cerr << "usage: layouts wine <out.ppm> [resolution] [times] [photonNum] [oversampleTimes]\n"; return 2; }
  const int res = argc > 3 ? std::atoi(argv[3]) : 100, times = argc > 4 ? std::atoi(argv[4]) : 30;
  const uint32_t photonNum = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 50000, os = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 100;
  const double eps = 1e-8;
  Context gpu(gpu_of_rank(sh.rank));
  std::unique_ptr<Comm> comm = make_comm(gpu, sh);
  Scene scene(0.5);                                                                                     // Layout.h:239
  scene.AddInftyPlane({0, 0, 0}, {0, 0, 1}, Material(_RUBBER_), Colour(140, 100, 60));                  // :241-251 (texture_wood / texture6 left out)
  scene.AddInftyPlane({10, 0, 0}, {-1, 0, 0}, Material(_PLASTIC_), Colour(90, 90, 110));
  scene.AddInftyPlane({0, -10, 0}, {0, 1, 0}, Material(_PLASTIC_), Colour(90, 110, 90));
  const std::vector<std::array<double, 2>> wineCurve = {{0, 6.2 - 10 * eps}, {1.3 + 10 * eps, 6.2 - 10 * eps}, {1.65 + 10 * eps, 8.2 - 10 * eps},
                                                        {1.6 + 10 * eps, 10.2 - 10 * eps}};             // bezierCurve1, :254-258
  RotationalSurface bezierCurve1; for (auto &c : wineCurve) bezierCurve1.AddControlPt(c[0], c[1]);
  struct Cup { Point3D center; double uRange; Colour wine; };
  const Cup cups[3] = {{{-2, -2, 0}, 0.7, Colour(151, 57, 29)}, {{2.2, 2.2, 0}, 0.5, Colour(100, 25, 30)}, {{5, 5, 0}, 0.8, Colour(100, 25, 30)}};
  for (const Cup &c : cups) {                                                                           // :253-286 (rObj2 and rObj3 rotate bezierCurve1 too)
    Object_Cup2(scene, c.center, Material(_GLASS_), Colour(155, 155, 155));
    scene.AddRotationalObject(wineCurve, c.center, Material(_ICE_), c.wine, 50, c.uRange);
    const auto pt = bezierCurve1.Point2D(c.uRange, 0, bezierCurve1.degree());                           // CBezierCurve::Point2D(t), Bezier.h:27
    scene.AddCircle({c.center.x, c.center.y, pt[1]}, {0, 0, 1}, pt[0] + 10 * eps, Material(_ICE_), c.wine);
  }
  scene.AddAreaLight({-10, 25, 25}, 2, 2, 5);                                                           // :289
  Camera camera({-23.5, 27.5, 24}, {2.9, -3, -2}, {-1, -1, 0}, 15, 8, 6, res, 8);                       // :291
  camera.seed = 1;
  if (comm) camera.Render_PhotonMapper(gpu, scene, *comm, times, Colour(0, 0, 0), photonNum);          // Render_PhotonMapper, the photons of a pass over the ranks
  else if (os == 0) camera.Render_PhotonMapper(gpu, scene, times, Colour(0, 0, 0), photonNum);          // (oversampleTimes 0: the plain driver, Camera.h:291-313)
  else camera.Render_PhotonMapper_Antialiasing(gpu, scene, times, Colour(0, 0, 0), photonNum, os);     // :299
  if (sh.rank == 0) camera.WritePPM(argv[2]);
  size_t masked = 0; for (uint8_t m : camera.edge_mask()) masked += m;
  std::cout << camera.width() << "x" << camera.height() << ", " << scene.NumObjects() << " objects, " << masked << " pixels re-sampled\n";
  return 0;
}

static int room(int argc, char **argv)
{
  if (argc < 4) { std::cerr << "usage: layouts room <obj> <out.ppm> [resolution] [oversampleTimes]\n"; return 2; }
  const int res = argc > 4 ? std::atoi(argv[4]) : 100;
  const uint32_t os = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 100;
  Context gpu(0);
  Scene scene(0.1);                                                                                     // Layout.h:304
  scene.AddInftyPlane({50, 0, 0}, {-1, 0, 0}, Material(_PLASTIC_), Colour(90, 90, 120));                // :306-308
  auto quad = [&](Point3D a, Point3D b, Point3D c, Point3D d, Colour col) {
    const Point3D t1[3] = {a, b, c}, t2[3] = {a, c, d};
    scene.AddTriangle(t1, nullptr, Material(_PLASTIC_), col); scene.AddTriangle(t2, nullptr, Material(_PLASTIC_), col);
  };
  // stand-in for ./ObjFiles/Corridor_Triangle.obj (:310): a floor, a wall with a window gap under the light source, a mesh
  quad({-12, -8, 0}, {30, -8, 0}, {30, 8, 0}, {-12, 8, 0}, Colour(241, 46, 24));
  quad({-12, 5, 0}, {2, 5, 0}, {2, 5, 8}, {-12, 5, 8}, Colour(200, 200, 200));
  quad({6, 5, 0}, {30, 5, 0}, {30, 5, 8}, {6, 5, 8}, Colour(200, 200, 200));
  quad({2, 5, 0}, {6, 5, 0}, {6, 5, 1}, {2, 5, 1}, Colour(200, 200, 200));
  scene.ReadObjFile_Triangle(argv[2], {9, -2, 0}, 0.003, Material(_METAL_), Colour(241, 46, 24));
  scene.ConstructBSPTree();                                                                             // :313
  scene.AddAreaLight({-10, 20, 20}, 2, 10, 2);                                                          // :315 (10 x 10 samples there)
  GodRay godRay(gpu, scene, Colour(155, 155, 50), 9);                                                   // :317
  const double mn[2] = {-3, 0}, mx[2] = {9, 10};
  godRay.AddAreaLightSource_XZ(6, mn, mx, 0.4, {1, -2, -1});                                            // :318 (interval 0.1 there)
  Camera camera({-10, 0, 2.5}, {1, 0, 0}, {0, -1, 0}, 10, 8, 6, res, 1);                                // :320
  camera.seed = 9;
  camera.Render_RayTracer_GodRay_Antialiasing(godRay, Colour(20, 20, 30), os);                         // :321
  camera.WritePPM(argv[3]);
  size_t masked = 0; for (uint8_t m : camera.edge_mask()) masked += m;
  std::cout << camera.width() << "x" << camera.height() << ", " << masked << " pixels re-sampled\n";
  return 0;
}

int main(int argc, char **argv)
{
  try {
    const std::string what = argc > 1 ? argv[1] : "";
    if (what == "focusing") return focusing(argc, argv);
    if (what == "raytracer") return raytracer(argc, argv);
    if (what == "apples") return apples(argc, argv);
    if (what == "wine") return wine(argc, argv);
    if (what == "room") return room(argc, argv);
    std::cerr << "usage: layouts focusing|raytracer|apples|wine|room ...\n";
    return 2;
  } catch (const std::exception &e) { std::cerr << "error: " << e.what() << "\n"; return 1; }
}
