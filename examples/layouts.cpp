// layouts.cpp -- scene scripts of the reference's Layout.h written against the host mirror (rt_host.hpp), to show
// that a layout ports line by line and to exercise the C++ side end to end (tests/test_gpu_examples.py runs it).
//
//   layouts focusing <Icosahedron_Triangle.obj> <out.ppm> [resolution] [spp]     Layout3_Focusing, Layout.h:134-160
//   layouts apples   <out.ppm> [resolution]                                     Layout1_Apples-like: textured mesh from
//                                                                               RotationalSurfaceToMesh_Texture, textured +
//                                                                               bump-mapped planes, anti-aliased render
//
// Textures: the reference reads JPEG through OpenCV; here the images are synthetic (or binary PPM via Image::ReadPPM).
#include "../ray_tracing_amd/host/rt_host.hpp"

#include <cstdlib>
#include <iostream>

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

static int focusing(int argc, char **argv)
{
  if (argc < 4) { std::cerr << "usage: layouts focusing <obj> <out.ppm> [resolution] [spp]\n"; return 2; }
  const int res = argc > 4 ? std::atoi(argv[4]) : 100;
  const uint32_t spp = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 100;
  Context gpu(0);
  Scene scene(0.5);                                                                                     // Layout.h:138
  scene.AddInftyPlane(Point3D(0, 0, 0), Point3D(0, 0, 1), Material(_PLASTIC_), Colour(255, 255, 255));  // :140-142
  scene.AddInftyPlane(Point3D(10, 0, 0), Point3D(-1, 0, 0), Material(_PLASTIC_), Colour(0, 0, 0));      // :144-146
  scene.ReadObjFile_Triangle(argv[2], Point3D(0, -1, 0), 0.0015, Material(_PLASTIC_), Colour(241, 46, 24));   // :151
  scene.ConstructBSPTree();                                                                             // :153
  scene.AddAreaLight(Point3D(-10, 20, 20), 1, 10, 10);                                                  // :155
  Camera camera(Point3D(-10, 0, 1.5), Point3D(1, 0, 0), Point3D(0, -1, 0), 10, 6.4, 4.8, res, 1);       // :157
  camera.seed = 1;
  camera.Render_Focusing(gpu, scene, 0.1, Colour(200, 200, 200), spp);                                  // :159
  camera.WritePPM(argv[3]);
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

int main(int argc, char **argv)
{
  try {
    const std::string what = argc > 1 ? argv[1] : "";
    if (what == "focusing") return focusing(argc, argv);
    if (what == "apples") return apples(argc, argv);
    std::cerr << "usage: layouts focusing|apples ...\n";
    return 2;
  } catch (const std::exception &e) { std::cerr << "error: " << e.what() << "\n"; return 1; }
}
