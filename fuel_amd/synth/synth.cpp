// synth.cpp -- deterministic synthetic worlds, knowledge states and depth frames.
//
// Host-side input generator for bench.py and the tests (there is no network for datasets and
// office.pcd is a missing blob in the reference snapshot).  It is independent of both the oracle
// and libfuelmi: plain grid parameters in, plain arrays out.  Recipe: SURVEY.md section 8(d) --
// splitmix64-seeded pillars/walls on a floor slab (the reference gets worlds from .pcd files,
// uav_simulator/map_generator/src/map_publisher.cpp:20-52), pinhole frames (intrinsics
// exploration_manager/launch/exploration.launch:38-41) rendered by exact voxel traversal of the
// ground truth, projected to world points the way MapROS::proessDepthImage does
// (plan_env/src/map_ros.cpp:176-212: depth beyond depth_filter_maxdist -> maxdist,
// pt = R*[(u-cx)d/fx,(v-cy)d/fy,d]+t, stored as float).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

extern "C" {
typedef struct {
  int nv[3];
  double origin[3];
  double res;
  double logodds[5]; /* prob_hit_log, prob_miss_log, clamp_min_log, clamp_max_log, min_occupancy_log */
} synth_grid;
}

namespace {
struct Rng {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double uni(double a, double b) { return a + (b - a) * uni(); }
};
}  // namespace

extern "C" {

// Ground-truth occupancy (1 = solid) on the map's voxel grid.  Returns number of solid voxels.
long synth_world(const synth_grid* G, uint64_t seed, int n_obstacles, unsigned char* truth) {
  const int* nv = G->nv;
  const double* org = G->origin;
  const long N = (long)nv[0] * nv[1] * nv[2];
  std::memset(truth, 0, (size_t)N);
  auto adr = [&](int x, int y, int z) { return ((long)x * nv[1] + y) * nv[2] + z; };
  // floor slab: the voxel layer whose top face is z = 0  (z in [-res, 0))
  const double vres = G->res;
  int zf = (int)std::floor((-0.5 * vres - org[2]) / vres);
  if (zf < 0) zf = 0;
  for (int x = 0; x < nv[0]; ++x)
    for (int y = 0; y < nv[1]; ++y) truth[adr(x, y, zf)] = 1;
  Rng r{seed};
  const int margin = 3;  // keep obstacles >= 0.3 m from the map faces (avoids the inflate wrap quirk)
  for (int k = 0; k < n_obstacles; ++k) {
    bool wall = (r.next() % 5) == 0;
    int sx = (int)std::lround(r.uni(0.3, 2.0) / vres), sy = (int)std::lround(r.uni(0.3, 2.0) / vres);
    if (wall) {
      if (r.next() & 1)
        sx = (int)std::lround(r.uni(3.0, 8.0) / vres), sy = (int)std::lround(0.3 / vres);
      else
        sy = (int)std::lround(r.uni(3.0, 8.0) / vres), sx = (int)std::lround(0.3 / vres);
    }
    double hmax = 0.8 * nv[2] * vres;
    int sz = (int)std::lround(r.uni(1.0, hmax) / vres);
    int x0 = margin + (int)(r.next() % (uint64_t)std::max(1, nv[0] - 2 * margin - sx));
    int y0 = margin + (int)(r.next() % (uint64_t)std::max(1, nv[1] - 2 * margin - sy));
    int z0 = zf + 1;
    for (int x = x0; x < x0 + sx && x < nv[0] - margin; ++x)
      for (int y = y0; y < y0 + sy && y < nv[1] - margin; ++y)
        for (int z = z0; z < z0 + sz && z < nv[2] - margin; ++z) truth[adr(x, y, z)] = 1;
  }
  long cnt = 0;
  for (long i = 0; i < N; ++i) cnt += truth[i];
  return cnt;
}

// "As-if-explored" knowledge state for the large benchmark maps (ray-fusing enough frames to know
// ~45 % of a 16 M-voxel map would take minutes on the CPU).  Known region K = union of seeded
// spheres; inside K free-truth voxels become FREE, solid-truth voxels with a free 6-neighbour
// inside K become OCCUPIED (visible surfaces), everything else stays UNKNOWN.  Log-odds values
// are ones the fusion rule reaches: clamp_min / one-miss for free, clamp_max / one-hit for
// occupied (sdf_map.cpp:332-344).  Writes occ[N]; returns the number of known voxels.
long synth_known_state(const synth_grid* G, const unsigned char* truth, uint64_t seed, int n_spheres,
                       double rmin, double rmax, double* occ) {
  const int* nv = G->nv;
  const double* org = G->origin;
  const double* lo = G->logodds;
  const double l_hit = lo[0], l_miss = lo[1], l_min = lo[2], l_max = lo[3], l_occ = lo[4];
  const double vres = G->res;
  const long N = (long)nv[0] * nv[1] * nv[2];
  std::vector<unsigned char> known((size_t)N, 0);
  Rng r{seed ^ 0x5EEDull};
  for (int s = 0; s < n_spheres; ++s) {
    double cx = r.uni(org[0] + 1.0, -org[0] - 1.0), cy = r.uni(org[1] + 1.0, -org[1] - 1.0);
    double cz = r.uni(0.8, 2.5), rad = r.uni(rmin, rmax);
    int x0 = std::max(0, (int)std::floor((cx - rad - org[0]) / vres)), x1 = std::min(nv[0] - 1, (int)std::floor((cx + rad - org[0]) / vres));
    int y0 = std::max(0, (int)std::floor((cy - rad - org[1]) / vres)), y1 = std::min(nv[1] - 1, (int)std::floor((cy + rad - org[1]) / vres));
    int z0 = std::max(0, (int)std::floor((cz - rad - org[2]) / vres)), z1 = std::min(nv[2] - 1, (int)std::floor((cz + rad - org[2]) / vres));
    for (int x = x0; x <= x1; ++x)
      for (int y = y0; y <= y1; ++y)
        for (int z = z0; z <= z1; ++z) {
          double dx = (x + 0.5) * vres + org[0] - cx, dy = (y + 0.5) * vres + org[1] - cy, dz = (z + 0.5) * vres + org[2] - cz;
          if (dx * dx + dy * dy + dz * dz <= rad * rad) known[((long)x * nv[1] + y) * nv[2] + z] = 1;
        }
  }
  long cnt = 0;
  const double unknown = l_min - 0.01;
  for (int x = 0; x < nv[0]; ++x)
    for (int y = 0; y < nv[1]; ++y)
      for (int z = 0; z < nv[2]; ++z) {
        long a = ((long)x * nv[1] + y) * nv[2] + z;
        double v = unknown;
        if (known[a]) {
          uint64_t h = (uint64_t)a * 0x9E3779B97F4A7C15ull;
          h ^= h >> 29;
          if (!truth[a]) {
            v = (h & 3) ? l_min : std::min(std::max(l_occ + l_miss, l_min), l_max);
            ++cnt;
          } else {
            bool surf = false;
            const int d[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
            for (auto& o : d) {
              int xx = x + o[0], yy = y + o[1], zz = z + o[2];
              if (xx < 0 || yy < 0 || zz < 0 || xx >= nv[0] || yy >= nv[1] || zz >= nv[2]) continue;
              long b = ((long)xx * nv[1] + yy) * nv[2] + zz;
              if (known[b] && !truth[b]) surf = true;
            }
            if (surf) {
              v = (h & 3) ? l_max : std::min(std::max(l_occ + l_hit, l_min), l_max);
              ++cnt;
            }
          }
        }
        occ[a] = v;
      }
  return cnt;
}

// k-th camera pose of the seeded tour: a Lissajous path at flight height with sweeping yaw,
// nudged out of solid voxels.  pose = {x,y,z,yaw,pitch}
void synth_camera(const synth_grid* G, const unsigned char* truth, uint64_t seed, int k, int n_total,
                  double extent_frac, double pose[5]) {
  const int* nv = G->nv;
  const double* org = G->origin;
  const double vres = G->res;
  Rng r{seed ^ 0xC0FFEEull};
  double ph1 = r.uni(0, 6.28318), ph2 = r.uni(0, 6.28318);
  double t = (n_total > 1) ? (double)k / (double)n_total : 0.0;
  double ax = extent_frac * (-org[0] - 1.5), ay = extent_frac * (-org[1] - 1.5);
  double x = ax * std::sin(2 * 3.14159265358979 * 3 * t + ph1);
  double y = ay * std::sin(2 * 3.14159265358979 * 2 * t + ph2);
  double z = 1.0 + 0.5 * std::sin(2 * 3.14159265358979 * 5 * t);
  auto solid = [&](double px, double py, double pz) {
    int ix = (int)std::floor((px - org[0]) / vres), iy = (int)std::floor((py - org[1]) / vres),
        iz = (int)std::floor((pz - org[2]) / vres);
    if (ix < 0 || iy < 0 || iz < 0 || ix >= nv[0] || iy >= nv[1] || iz >= nv[2]) return true;
    return truth[((long)ix * nv[1] + iy) * nv[2] + iz] != 0;
  };
  // nudge along +x/+y in 0.25 m steps until free (deterministic)
  for (int it = 0; it < 40 && solid(x, y, z); ++it) {
    x += 0.25 * ((it & 1) ? 1 : 0);
    y += 0.25 * ((it & 1) ? 0 : 1);
  }
  pose[0] = x;
  pose[1] = y;
  pose[2] = z;
  pose[3] = 2 * 3.14159265358979 * 7 * t + ph1;  // yaw sweeps
  pose[4] = 0.25 * std::sin(2 * 3.14159265358979 * 11 * t);  // small pitch wobble
}

// Render one depth frame against the truth grid and project to world points (float xyz, 12-byte
// stride).  Returns number of points written (<= cap).
int synth_render(const synth_grid* G, const unsigned char* truth, const double pose[5], int width, int height,
                 int skip, int margin, double fx, double fy, double cx, double cy, double maxdist,
                 double mindist, float* out, int cap) {
  const int* nv = G->nv;
  const double* org = G->origin;
  const double vres = G->res;
  const double yaw = pose[3], pitch = pose[4];
  // camera axes in world: z_c forward, x_c right, y_c down
  double cyw = std::cos(yaw), syw = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch);
  double fwd[3] = {cyw * cp, syw * cp, sp};
  double right[3] = {syw, -cyw, 0};
  double down[3] = {cyw * sp, syw * sp, -cp};  // = fwd x right (points down when pitch = 0)
  int n = 0;
  for (int v = margin; v < height - margin; v += skip)
    for (int u = margin; u < width - margin; u += skip) {
      double dc[3] = {(u - cx) / fx, (v - cy) / fy, 1.0};
      double d[3];
      for (int i = 0; i < 3; ++i) d[i] = right[i] * dc[0] + down[i] * dc[1] + fwd[i] * dc[2];
      // exact voxel traversal from the camera along d, parameter = camera-frame depth
      double p[3] = {(pose[0] - org[0]) / vres, (pose[1] - org[1]) / vres, (pose[2] - org[2]) / vres};
      int c[3], st[3];
      double tmax[3], tdel[3];
      for (int i = 0; i < 3; ++i) {
        c[i] = (int)std::floor(p[i]);
        double di = d[i] / vres;
        st[i] = di > 0 ? 1 : (di < 0 ? -1 : 0);
        if (st[i] == 0) {
          tmax[i] = 1e300;
          tdel[i] = 1e300;
        } else {
          double nb = st[i] > 0 ? (c[i] + 1 - p[i]) : (p[i] - c[i]);
          tmax[i] = nb / std::fabs(di);
          tdel[i] = 1.0 / std::fabs(di);
        }
      }
      double depth = -1.0, t = 0.0;
      while (t <= maxdist) {
        if (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= nv[0] || c[1] >= nv[1] || c[2] >= nv[2]) break;
        if (truth[((long)c[0] * nv[1] + c[1]) * nv[2] + c[2]]) {
          depth = t;
          break;
        }
        int a = (tmax[0] < tmax[1]) ? (tmax[0] < tmax[2] ? 0 : 2) : (tmax[1] < tmax[2] ? 1 : 2);
        t = tmax[a];
        tmax[a] += tdel[a];
        c[a] += st[a];
      }
      // sensor quantisation to millimetres (16UC1 depth image), then the reference's filter
      if (depth >= 0) depth = std::floor(depth * 1000.0 + 0.5) / 1000.0 + 0.05;  // hit slightly inside
      if (depth < 0 || depth > maxdist)
        depth = maxdist;
      else if (depth < mindist)
        continue;
      if (n >= cap) return n;
      double pc[3] = {(u - cx) * depth / fx, (v - cy) * depth / fy, depth};
      for (int i = 0; i < 3; ++i)
        out[3 * n + i] = (float)(right[i] * pc[0] + down[i] * pc[1] + fwd[i] * pc[2] + pose[i]);
      ++n;
    }
  return n;
}

// Full-resolution 16UC1 depth image (millimetres, 0 = no return within max_range) of the truth grid
// seen from `pose` -- what the simulator's depth_render_node publishes
// (uav_simulator/local_sensing/src/depth_render_node.cpp:112-168) and MapROS consumes.
void synth_depth_image(const synth_grid* G, const unsigned char* truth, const double pose[5], int width,
                       int height, double fx, double fy, double cx, double cy, double max_range,
                       unsigned short* img) {
  const int* nv = G->nv;
  const double* org = G->origin;
  const double vres = G->res;
  const double yaw = pose[3], pitch = pose[4];
  double cyw = std::cos(yaw), syw = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch);
  double fwd[3] = {cyw * cp, syw * cp, sp};
  double right[3] = {syw, -cyw, 0};
  double down[3] = {cyw * sp, syw * sp, -cp};
  for (int v = 0; v < height; ++v)
    for (int u = 0; u < width; ++u) {
      double dc[3] = {(u - cx) / fx, (v - cy) / fy, 1.0};
      double d[3];
      for (int i = 0; i < 3; ++i) d[i] = right[i] * dc[0] + down[i] * dc[1] + fwd[i] * dc[2];
      double p[3] = {(pose[0] - org[0]) / vres, (pose[1] - org[1]) / vres, (pose[2] - org[2]) / vres};
      int c[3], st[3];
      double tmax[3], tdel[3];
      for (int i = 0; i < 3; ++i) {
        c[i] = (int)std::floor(p[i]);
        double di = d[i] / vres;
        st[i] = di > 0 ? 1 : (di < 0 ? -1 : 0);
        if (st[i] == 0) {
          tmax[i] = 1e300;
          tdel[i] = 1e300;
        } else {
          double nb = st[i] > 0 ? (c[i] + 1 - p[i]) : (p[i] - c[i]);
          tmax[i] = nb / std::fabs(di);
          tdel[i] = 1.0 / std::fabs(di);
        }
      }
      double depth = -1.0, t = 0.0;
      while (t <= max_range) {
        if (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= nv[0] || c[1] >= nv[1] || c[2] >= nv[2]) break;
        if (truth[((long)c[0] * nv[1] + c[1]) * nv[2] + c[2]]) {
          depth = t;
          break;
        }
        int a = (tmax[0] < tmax[1]) ? (tmax[0] < tmax[2] ? 0 : 2) : (tmax[1] < tmax[2] ? 1 : 2);
        t = tmax[a];
        tmax[a] += tdel[a];
        c[a] += st[a];
      }
      long mm = depth < 0 ? 0 : std::lround((depth + 0.05) * 1000.0);  // hit slightly inside the surface
      img[(long)v * width + u] = (unsigned short)(mm > 65535 ? 65535 : mm);
    }
}

}  // extern "C"
