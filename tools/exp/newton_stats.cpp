// experiment: how many Newton steps does the patch solver run on C4-like rays, and how many would an exact "iterate repeats" break save?
#include <cstdio>
#include <cmath>
#include <cstdint>
#include "../../ray_tracing_amd/csrc/rt_param.hpp"
using namespace rt;
static int seed_steps[25]; static long hist[12]; static long lock_sum = 0, lock_n = 0; static int lock_max[25]; static int lock_cnt = 0; static long flat_sum; static int flat_max;
static long steps_ref = 0, steps_p1 = 0, steps_p2 = 0, solves = 0, hits = 0, diff = 0;
static ParamHit hit_mode(const DBezier &B, V3 o, V3 d, int mode, long &steps)
{
  const V3 center = mk(B.center[0], B.center[1], B.center[2]);
  double cur_u = kInfty, cur_v = kInfty, cur_t = kInfty; bool have = false;
  for (int i = 0; i <= 4; i++) for (int j = 0; j <= 4; j++) {
    double u = i * 0.25, v = j * 0.25;
    double t = dist(bez_point(B, u, v) + center, o);
    double pt = NAN, pu = NAN, pv = NAN;     // iterate before (t,u,v)
    int ns = 0;
    for (int step = 0; step < kMaxStepNum; step++) {
      steps++; ns++;
      V3 S, dPu, dPv; bez_eval(B, u, v, S, dPu, dPv);
      double tp, up, vp; newton_step(d, o, center, S, dPu, dPv, t, u, v, tp, up, vp);
      if (up < -kMaxDeviation || up > 1 + kMaxDeviation || vp < -kMaxDeviation || vp > 1 + kMaxDeviation || tp < -kMaxDeviation) break;
      if (len2(mk(tp - t, up - u, vp - v)) < kMinStepLen && tp > kEps && 0 < up && up < 1 && 0 < vp && vp < 1) {
        have = true;
        if (tp < cur_t - kEps) { cur_u = up; cur_v = vp; cur_t = tp; break; }
      }
      if (mode >= 1 && tp == t && up == u && vp == v) break;
      if (mode >= 2 && tp == pt && up == pu && vp == pv) break;
      pt = t; pu = u; pv = v;
      u = up; v = vp; t = tp;
    }
    if (mode == 2) { seed_steps[i * 5 + j] = ns; hist[ns]++; static int shown = 0; if (ns == 10 && shown < 6 && (solves % 977) == 0) { shown++; printf("straggler seed %d,%d: end t %.17g u %.17g v %.17g  prev t %.17g u %.17g v %.17g\n", i, j, t, u, v, pt, pu, pv); } }
  }
  ParamHit h; h.t = kMiss; h.u = kInfty; h.v = kInfty;
  if (have) { h.t = dist(center + bez_point(B, cur_u, cur_v), o); h.u = cur_u; h.v = cur_v; }
  return h;
}
int main()
{
  const double ctrl[4][4][3] = {{{-2, 6, 9}, {-1, 2, 9}, {-1, -2, 9}, {-2, -6, 9}}, {{-1, 6, 3}, {0, 2, 3}, {0, -2, 3}, {-1, -6, 3}},
                                {{-1, 6, -3}, {0, 2, -2}, {0, -2, -3}, {-1, -6, -3}}, {{-2, 6, -9}, {-1, 2, -9}, {-1, -2, -9}, {-2, -6, -9}}};
  DBezier B; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int a = 0; a < 3; a++) B.c[i][j][a] = ctrl[i][j][a];
  B.center[0] = 20; B.center[1] = 25; B.center[2] = 17;
  param_prune_box(B);
  V3 o = mk(-10, 20, 20), eye = mk(1, 0, 0), left = unitize(mk(-0.8, -1, 0)); V3 up = cross(eye, left);
  const int W = 480, H = 270;
  for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
    double fx = (x + 0.5) / W * 6 - 3, fy = (y + 0.5) / H * 3.375 - 1.6875;
    V3 d = unitize(eye * 5 + left * fx + up * fy);
    // prune box slab
    double tmin = -1e300, tmax = 1e300; const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
    for (int a = 0; a < 3; a++) { double t1 = (B.pmin[a] - oo[a]) / dd[a], t2 = (B.pmax[a] - oo[a]) / dd[a]; tmin = fmax(tmin, fmin(t1, t2)); tmax = fmin(tmax, fmax(t1, t2)); }
    if (!(tmin <= tmax && tmax > 0)) continue;
    solves++;
    ParamHit a = hit_mode(B, o, d, 0, steps_ref), b = hit_mode(B, o, d, 1, steps_p1), c = hit_mode(B, o, d, 2, steps_p2);
    if (a.t < kMiss) hits++;
    for (int k = 0; k < 25; k++) lock_max[k] = lock_max[k] > seed_steps[k] ? lock_max[k] : seed_steps[k];
    { int tot = 0; for (int k = 0; k < 25; k++) tot += seed_steps[k]; if (tot > flat_max) flat_max = tot; }
    if (++lock_cnt == 64) { flat_sum += flat_max; flat_max = 0; for (int k = 0; k < 25; k++) { lock_sum += lock_max[k]; lock_max[k] = 0; } lock_n++; lock_cnt = 0; }
    if (a.t != b.t || a.u != b.u || a.v != b.v || a.t != c.t || a.u != c.u || a.v != c.v) diff++;
  }
  printf("flattened: max lane steps per batch %.1f\n", double(flat_sum) / lock_n); printf("lockstep(64 consecutive solves) steps per solve-batch %.1f; per-seed step histogram:", double(lock_sum) / lock_n); for (int k = 0; k < 12; k++) printf(" %d:%ld", k, hist[k]); printf("\n");
  printf("pixels %d solves %ld hits %ld  steps/solve ref %.1f  stationary-break %.1f  +2-cycle %.1f  differing results %ld\n", W * H, solves, hits,
         double(steps_ref) / solves, double(steps_p1) / solves, double(steps_p2) / solves, diff);
}
