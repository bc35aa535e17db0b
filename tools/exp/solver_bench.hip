// experiment: the patch solver alone, on C4's camera rays that pass the patch's prune box.  MODE 0: one lane per pair (reference loops in lock
// step); MODE 1: pairs_solve (the kernels' solver); MODE 2: the same for an occlusion query without a reference point.  Prints ms per launch and checks the two against each other and against the host's bezier_hit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#include <cstdlib>
#include "../../ray_tracing_amd/csrc/rt_device.hpp"
using namespace rt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#ifndef MINW
#define MINW 2
#endif
template <int MODE>
__global__ void __launch_bounds__(64, MINW) k_solve(DevScene sc, const double *rays, int n, double *out)
{
  __shared__ WaveLds lds;
  const int lane = threadIdx.x & 63, idx = blockIdx.x * 64 + lane;
  const bool active = idx < n;
  RayLds &me = lds.rays[lane];
  for (int a = 0; a < 3; a++) { me.o[a] = active ? rays[idx * 6 + a] : 0.0; me.d[a] = active ? rays[idx * 6 + 3 + a] : 1.0; }
  PairQueue &Q = wave_pair_queue();
  __builtin_amdgcn_wave_barrier();
  const int nq = pairs_push(Q, 0, active, 0);
  if (MODE == 0) {
    if (lane < nq) {
      const RayLds &gr = lds.rays[Q.owner[lane]];
      const DBezier B = sc.beziers[0];
      ParamHit h = bezier_hit(B, mk(gr.o[0], gr.o[1], gr.o[2]), mk(gr.d[0], gr.d[1], gr.d[2]));
      Q.t[lane] = h.t; Q.u[lane] = h.u; Q.v[lane] = h.v;
    }
    __builtin_amdgcn_wave_barrier();
  } else pairs_solve(sc, &lds, Q, nq, MODE == 1, MODE == 2);
  for (int q = 0; q < nq; q++) if ((int)Q.owner[q] == lane) { out[idx * 3] = Q.t[q]; out[idx * 3 + 1] = Q.u[q]; out[idx * 3 + 2] = Q.v[q]; }
}
int main(int argc, char **argv)
{
  const double ctrl[4][4][3] = {{{-2, 6, 9}, {-1, 2, 9}, {-1, -2, 9}, {-2, -6, 9}}, {{-1, 6, 3}, {0, 2, 3}, {0, -2, 3}, {-1, -6, 3}},
                                {{-1, 6, -3}, {0, 2, -2}, {0, -2, -3}, {-1, -6, -3}}, {{-2, 6, -9}, {-1, 2, -9}, {-1, -2, -9}, {-2, -6, -9}}};
  DBezier B; memset(&B, 0, sizeof B);
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int a = 0; a < 3; a++) B.c[i][j][a] = ctrl[i][j][a];
  B.center[0] = 20; B.center[1] = 25; B.center[2] = 17;
  param_prune_box(B);
  V3 o = mk(-10, 20, 20), eye = mk(1, 0, 0), left = unitize(mk(-0.8, -1, 0)); V3 up = cross(eye, left);
  const int W = 1920, H = 1080;
  std::vector<double> rays;
  for (int ty = 0; ty < H / 8; ty++) for (int tx = 0; tx < W / 8; tx++) for (int p = 0; p < 64; p++) {       // 8x8 tiles, as the renderer's waves
    const int x = tx * 8 + (p & 7), y = ty * 8 + (p >> 3);
    double fx = (x + 0.5) / W * 6 - 3, fy = (y + 0.5) / H * 3.375 - 1.6875;
    V3 d = unitize(eye * 5 + left * fx + up * fy);
    double tmin = -1e300, tmax = 1e300; const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    for (int a = 0; a < 3; a++) { double t1 = (B.pmin[a] - oo[a]) / dd[a], t2 = (B.pmax[a] - oo[a]) / dd[a]; tmin = fmax(tmin, fmin(t1, t2)); tmax = fmin(tmax, fmax(t1, t2)); }
    if (!(tmin <= tmax && tmax > 0)) continue;
    rays.insert(rays.end(), {o.x, o.y, o.z, d.x, d.y, d.z});
  }
  const int n = (int)(rays.size() / 6);
  uint8_t type = 6; uint32_t index = 0;
  DBezier *dB; uint8_t *dT; uint32_t *dI; double *dR, *dO0, *dO1;
  CK(hipMalloc(&dB, sizeof B)); CK(hipMemcpy(dB, &B, sizeof B, hipMemcpyHostToDevice));
  CK(hipMalloc(&dT, 16)); CK(hipMemcpy(dT, &type, 1, hipMemcpyHostToDevice));
  CK(hipMalloc(&dI, 16)); CK(hipMemcpy(dI, &index, 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dR, rays.size() * 8)); CK(hipMemcpy(dR, rays.data(), rays.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dO0, n * 24ull)); CK(hipMalloc(&dO1, n * 24ull));
  CK(hipMemset(dO0, 0, n * 24ull)); CK(hipMemset(dO1, 0, n * 24ull));
  DevScene sc; memset(&sc, 0, sizeof sc);
  sc.beziers = dB; sc.obj_type = dT; sc.obj_index = dI; sc.n_objects = 1; sc.has_param = 1;
  const int blocks = (n + 63) / 64;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms[2] = {0, 0};
  for (int mode = 0; mode < 2; mode++) for (int rep = 0; rep < (argc > 1 ? atoi(argv[1]) : 60); rep++) {
    CK(hipEventRecord(e0));
    if (mode == 0) k_solve<0><<<blocks, 64>>>(sc, dR, n, dO0); else k_solve<1><<<blocks, 64>>>(sc, dR, n, dO1);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float t; CK(hipEventElapsedTime(&t, e0, e1)); if (rep == 0 || t < ms[mode]) ms[mode] = t;
  }
  std::vector<double> h0(n * 3), h1(n * 3);
  CK(hipMemcpy(h0.data(), dO0, n * 24ull, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), dO1, n * 24ull, hipMemcpyDeviceToHost));
  long diff = 0, diff_host = 0, hits = 0;
  for (int i = 0; i < n; i++) {
    if (memcmp(&h0[i * 3], &h1[i * 3], 24)) diff++;
    if (h0[i * 3] < kMiss) hits++;
    if (i % 97 == 0) { ParamHit h = bezier_hit(B, mk(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), mk(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5])); if (h.t != h1[i * 3] || h.u != h1[i * 3 + 1] || h.v != h1[i * 3 + 2]) diff_host++; }
  }
  printf("pairs %d hits %ld | lane-per-pair %.3f ms (%.1f M pairs/s) | pairs_solve %.3f ms (%.1f M pairs/s) | differing %ld, vs host (1 in 97) %ld\n", n, hits, ms[0], n / ms[0] / 1e3, ms[1], n / ms[1] / 1e3, diff, diff_host);
  // ---- shadow rays towards the parallel light (1,-1,-1) of C4 from the primary hit points (patch or the wall x = 20) ----
  {
    std::vector<double> sh;
    const V3 tl = unitize(mk(-1, 1, 1));
    for (int i = 0; i < n; i++) {
      const V3 ro = mk(rays[i * 6], rays[i * 6 + 1], rays[i * 6 + 2]), rd = mk(rays[i * 6 + 3], rays[i * 6 + 4], rays[i * 6 + 5]);
      const double t = h0[i * 3] < kMiss ? h0[i * 3] : (20 - ro.x) / rd.x;
      const V3 x = ro + rd * t;
      double tmin = -1e300, tmax = 1e300; const double oo[3] = {x.x, x.y, x.z}, dd[3] = {tl.x, tl.y, tl.z};
      for (int a = 0; a < 3; a++) { double t1 = (B.pmin[a] - oo[a]) / dd[a], t2 = (B.pmax[a] - oo[a]) / dd[a]; tmin = fmax(tmin, fmin(t1, t2)); tmax = fmin(tmax, fmax(t1, t2)); }
      if (!(tmin <= tmax && tmax > 0)) continue;
      sh.insert(sh.end(), {x.x, x.y, x.z, tl.x, tl.y, tl.z});
    }
    const int ns = (int)(sh.size() / 6);
    double *dS; CK(hipMalloc(&dS, sh.size() * 8)); CK(hipMemcpy(dS, sh.data(), sh.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dO0, 0, n * 24ull)); CK(hipMemset(dO1, 0, n * 24ull));
    const int bl = (ns + 63) / 64;
    float m2[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; mode++) for (int rep = 0; rep < (argc > 1 ? atoi(argv[1]) : 60); rep++) {
      CK(hipEventRecord(e0));
      if (mode == 0) k_solve<0><<<bl, 64>>>(sc, dS, ns, dO0); else if (mode == 1) k_solve<1><<<bl, 64>>>(sc, dS, ns, dO1); else k_solve<2><<<bl, 64>>>(sc, dS, ns, dO1);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
      float t; CK(hipEventElapsedTime(&t, e0, e1)); if (rep == 0 || t < m2[mode]) m2[mode] = t;
    }
    CK(hipMemcpy(h0.data(), dO0, ns * 24ull, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), dO1, ns * 24ull, hipMemcpyDeviceToHost));
    long occ = 0, dif = 0;
    for (int i = 0; i < ns; i++) { const bool a = h0[i * 3] < kMiss, b = h1[i * 3] < kMiss; occ += a; dif += a != b; }
    printf("shadow pairs %d occluded %ld | lane-per-pair %.3f ms | pairs_solve %.3f ms | pairs_solve, first accepted iterate settles %.3f ms | differing verdicts %ld\n", ns, occ, m2[0], m2[1], m2[2], dif);
  }
  return 0;
}
