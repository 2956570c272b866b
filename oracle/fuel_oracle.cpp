// fuel_oracle.cpp -- CPU ORACLE (test infrastructure only; see fuel_oracle.h).
// Dependency-free restatement of the FUEL hot path in double precision, single thread.
// Every function cites the reference lines it follows (paths relative to
// /root/reference/fuel_planner/).  Quirks of the reference are reproduced on purpose.
#include "fuel_oracle.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <limits>
#include <list>
#include <vector>

namespace {

struct V3d {
  double v[3];
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
};
struct V3i {
  int v[3];
  int& operator[](int i) { return v[i]; }
  const int& operator[](int i) const { return v[i]; }
};
inline V3d mk(double a, double b, double c) { return V3d{{a, b, c}}; }
inline V3d operator+(const V3d& a, const V3d& b) { return mk(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline V3d operator-(const V3d& a, const V3d& b) { return mk(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline V3d operator*(const V3d& a, double s) { return mk(a[0] * s, a[1] * s, a[2] * s); }
inline V3d operator*(double s, const V3d& a) { return mk(a[0] * s, a[1] * s, a[2] * s); }
inline V3d operator/(const V3d& a, double s) { return mk(a[0] / s, a[1] / s, a[2] / s); }
inline double dot(const V3d& a, const V3d& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double sqnorm(const V3d& a) { return dot(a, a); }
inline double norm(const V3d& a) { return std::sqrt(sqnorm(a)); }

// ---------------------------------------------------------------------------------------------
// RayCaster -- plan_env/src/raycast.cpp:6-23 (helpers), :323-327 setParams, :329-372 input,
// :374-407 nextId.  Works in coordinates scaled by 1/resolution (NOT origin shifted).
// ---------------------------------------------------------------------------------------------
inline int signum_i(int x) { return x == 0 ? 0 : (x < 0 ? -1 : 1); }
inline double mod_pos(double value, double modulus) {
  return std::fmod(std::fmod(value, modulus) + modulus, modulus);
}
double intbound(double s, double ds) {
  // smallest positive t with s + t*ds integer (raycast.cpp:14-23)
  if (ds < 0) return intbound(-s, -ds);
  s = mod_pos(s, 1);
  return (1 - s) / ds;
}

struct RayWalk {
  double res;
  V3d offset;  // 0.5 - origin/res
  int x, y, z, ex, ey, ez;
  int sx, sy, sz;
  double tmx, tmy, tmz, tdx, tdy, tdz;

  void setParams(double resolution, const V3d& origin) {
    res = resolution;
    offset = mk(0.5, 0.5, 0.5) - origin / resolution;
  }
  bool input(const V3d& start, const V3d& end) {
    V3d s = start / res, e = end / res;
    x = (int)std::floor(s[0]);
    y = (int)std::floor(s[1]);
    z = (int)std::floor(s[2]);
    ex = (int)std::floor(e[0]);
    ey = (int)std::floor(e[1]);
    ez = (int)std::floor(e[2]);
    // the reference stores the INTEGER cell differences in doubles and uses them as direction
    double dx = ex - x, dy = ey - y, dz = ez - z;
    sx = signum_i((int)dx);
    sy = signum_i((int)dy);
    sz = signum_i((int)dz);
    tmx = intbound(s[0], dx);
    tmy = intbound(s[1], dy);
    tmz = intbound(s[2], dz);
    tdx = ((double)sx) / dx;
    tdy = ((double)sy) / dy;
    tdz = ((double)sz) / dz;
    return !(sx == 0 && sy == 0 && sz == 0);
  }
  // writes the CURRENT cell (map index), then steps; false when current == end cell
  bool nextId(V3i& idx) {
    idx[0] = (int)((double)x + offset[0]);  // Eigen cast<int>: truncation
    idx[1] = (int)((double)y + offset[1]);
    idx[2] = (int)((double)z + offset[2]);
    if (x == ex && y == ey && z == ez) return false;
    if (tmx < tmy) {
      if (tmx < tmz) {
        x += sx;
        tmx += tdx;
      } else {
        z += sz;
        tmz += tdz;
      }
    } else {
      if (tmy < tmz) {
        y += sy;
        tmy += tdy;
      } else {
        z += sz;
        tmz += tdz;
      }
    }
    return true;
  }
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// SDFMap state -- plan_env/include/plan_env/sdf_map.h:86-125
// ---------------------------------------------------------------------------------------------
struct fo_map {
  // MapParam
  V3d origin, size, min_bound, max_bound;
  V3i nvox;
  double res, res_inv, obstacles_inflation, virtual_ceil_height, ground_height;
  V3i box_min, box_max;
  V3d box_mind, box_maxd;
  double default_dist;
  bool optimistic, signed_dist;
  double p_hit, p_miss, p_min, p_max, p_occ;
  double l_hit, l_miss, l_min, l_max, l_occ;
  double max_ray_length, local_bound_inflate, unknown_flag;
  // MapData
  std::vector<double> occ, dist_neg, dist, tmp1, tmp2;
  std::vector<char> infl;
  std::vector<short> cnt_hit, cnt_miss;
  std::vector<char> flag_rayend;
  char raycast_num;
  std::deque<int> cache;
  V3i lb_min, lb_max;
  V3d upd_min, upd_max;
  bool reset_updated_box;
  bool local_updated;
  RayWalk caster;

  // sdf_map.h:127-147
  void posToIndex(const V3d& p, V3i& id) const {
    for (int i = 0; i < 3; ++i) id[i] = (int)std::floor((p[i] - origin[i]) * res_inv);
  }
  void indexToPos(const V3i& id, V3d& p) const {
    for (int i = 0; i < 3; ++i) p[i] = (id[i] + 0.5) * res + origin[i];
  }
  void boundIndex(V3i& id) const {
    for (int i = 0; i < 3; ++i) id[i] = std::max(std::min(id[i], nvox[i] - 1), 0);
  }
  int adr(int x, int y, int z) const { return x * nvox[1] * nvox[2] + y * nvox[2] + z; }
  int adr(const V3i& id) const { return adr(id[0], id[1], id[2]); }
  int total() const { return nvox[0] * nvox[1] * nvox[2]; }
  // sdf_map.h:149-166
  bool isInMap(const V3d& p) const {
    for (int i = 0; i < 3; ++i)
      if (p[i] < min_bound[i] + 1e-4) return false;
    for (int i = 0; i < 3; ++i)
      if (p[i] > max_bound[i] - 1e-4) return false;
    return true;
  }
  bool isInMap(const V3i& id) const {
    for (int i = 0; i < 3; ++i)
      if (id[i] < 0 || id[i] > nvox[i] - 1) return false;
    return true;
  }
  // sdf_map.h:168-175 (index version: min <= id < max)
  bool isInBox(const V3i& id) const {
    for (int i = 0; i < 3; ++i)
      if (id[i] < box_min[i] || id[i] >= box_max[i]) return false;
    return true;
  }
  // sdf_map.h:180-187 (position version: strictly inside the box)
  bool isInBoxPos(const V3d& p) const {
    for (int i = 0; i < 3; ++i)
      if (p[i] <= box_mind[i] || p[i] >= box_maxd[i]) return false;
    return true;
  }
  // sdf_map.h:196-203
  int getOccupancy(const V3i& id) const {
    if (!isInMap(id)) return -1;
    double o = occ[adr(id)];
    if (o < l_min - 1e-3) return 0;  // UNKNOWN
    if (o > l_occ) return 2;         // OCCUPIED
    return 1;                        // FREE
  }
  double getDistance(const V3i& id) const {
    if (!isInMap(id)) return -1;
    return dist[adr(id)];
  }
};

extern "C" {

// plan_env/src/sdf_map.cpp:12-93 (initMap) minus ROS
fo_map* fo_map_create(const fo_map_cfg* c) {
  fo_map* m = new fo_map;
  m->res = c->resolution;
  m->obstacles_inflation = c->obstacles_inflation;
  m->local_bound_inflate = std::max(m->res, c->local_bound_inflate);
  m->ground_height = c->ground_height;
  m->default_dist = c->default_dist;
  m->optimistic = c->optimistic != 0;
  m->signed_dist = c->signed_dist != 0;
  m->res_inv = 1 / m->res;
  m->origin = mk(-c->map_size[0] / 2.0, -c->map_size[1] / 2.0, c->ground_height);
  m->size = mk(c->map_size[0], c->map_size[1], c->map_size[2]);
  for (int i = 0; i < 3; ++i) m->nvox[i] = (int)std::ceil(m->size[i] / m->res);
  m->min_bound = m->origin;
  m->max_bound = m->origin + m->size;
  m->p_hit = c->p_hit;
  m->p_miss = c->p_miss;
  m->p_min = c->p_min;
  m->p_max = c->p_max;
  m->p_occ = c->p_occ;
  m->max_ray_length = c->max_ray_length;
  m->virtual_ceil_height = c->virtual_ceil_height;
  auto logit = [](double x) { return std::log(x / (1 - x)); };
  m->l_hit = logit(m->p_hit);
  m->l_miss = logit(m->p_miss);
  m->l_min = logit(m->p_min);
  m->l_max = logit(m->p_max);
  m->l_occ = logit(m->p_occ);
  m->unknown_flag = 0.01;
  size_t n = (size_t)m->total();
  m->occ.assign(n, m->l_min - m->unknown_flag);
  m->infl.assign(n, 0);
  m->dist_neg.assign(n, m->default_dist);
  m->dist.assign(n, m->default_dist);
  m->cnt_hit.assign(n, 0);
  m->cnt_miss.assign(n, 0);
  m->flag_rayend.assign(n, -1);
  m->tmp1.assign(n, 0);
  m->tmp2.assign(n, 0);
  m->raycast_num = 0;
  m->reset_updated_box = true;
  m->upd_min = m->upd_max = mk(0, 0, 0);
  for (int i = 0; i < 3; ++i) {
    m->box_mind[i] = c->box_min[i];
    m->box_maxd[i] = c->box_max[i];
  }
  m->posToIndex(m->box_mind, m->box_min);
  m->posToIndex(m->box_maxd, m->box_max);
  m->lb_min = V3i{{0, 0, 0}};
  m->lb_max = V3i{{0, 0, 0}};
  m->local_updated = false;
  m->caster.setParams(m->res, m->origin);
  return m;
}
void fo_map_destroy(fo_map* m) { delete m; }

void fo_map_voxel_num(const fo_map* m, int out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = m->nvox[i];
}
void fo_map_origin(const fo_map* m, double out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = m->origin[i];
}
void fo_map_box_index(const fo_map* m, int bmin[3], int bmax[3]) {
  for (int i = 0; i < 3; ++i) bmin[i] = m->box_min[i], bmax[i] = m->box_max[i];
}
void fo_map_logodds(const fo_map* m, double out[5]) {
  out[0] = m->l_hit, out[1] = m->l_miss, out[2] = m->l_min, out[3] = m->l_max, out[4] = m->l_occ;
}
double* fo_map_occupancy(fo_map* m) { return m->occ.data(); }
char* fo_map_inflate(fo_map* m) { return m->infl.data(); }
double* fo_map_distance(fo_map* m) { return m->dist.data(); }
double* fo_map_distance_neg(fo_map* m) { return m->dist_neg.data(); }
char* fo_map_flag_rayend(fo_map* m) { return m->flag_rayend.data(); }

// plan_env/src/sdf_map.cpp:95-114
void fo_map_reset_buffer(fo_map* m, const double min_pos[3], const double max_pos[3]) {
  V3i a, b;
  m->posToIndex(mk(min_pos[0], min_pos[1], min_pos[2]), a);
  m->posToIndex(mk(max_pos[0], max_pos[1], max_pos[2]), b);
  m->boundIndex(a);
  m->boundIndex(b);
  for (int x = a[0]; x <= b[0]; ++x)
    for (int y = a[1]; y <= b[1]; ++y)
      for (int z = a[2]; z <= b[2]; ++z) {
        m->infl[m->adr(x, y, z)] = 0;
        m->dist[m->adr(x, y, z)] = m->default_dist;
      }
}
void fo_map_reset_buffer_all(fo_map* m) {
  fo_map_reset_buffer(m, m->min_bound.v, m->max_bound.v);
  m->lb_min = V3i{{0, 0, 0}};
  m->lb_max = V3i{{m->nvox[0] - 1, m->nvox[1] - 1, m->nvox[2] - 1}};
}

// sdf_map.h:210-215
void fo_map_set_occupied(fo_map* m, const double pos[3], int occ) {
  V3d p = mk(pos[0], pos[1], pos[2]);
  if (!m->isInMap(p)) return;
  V3i id;
  m->posToIndex(p, id);
  m->infl[m->adr(id)] = (char)occ;
}

void fo_map_get_local_bound(const fo_map* m, int bmin[3], int bmax[3]) {
  for (int i = 0; i < 3; ++i) bmin[i] = m->lb_min[i], bmax[i] = m->lb_max[i];
}
void fo_map_set_local_bound(fo_map* m, const int bmin[3], const int bmax[3]) {
  for (int i = 0; i < 3; ++i) m->lb_min[i] = bmin[i], m->lb_max[i] = bmax[i];
}
// plan_env/src/sdf_map.cpp:491-495
void fo_map_get_updated_box(fo_map* m, double bmin[3], double bmax[3], int reset) {
  for (int i = 0; i < 3; ++i) bmin[i] = m->upd_min[i], bmax[i] = m->upd_max[i];
  if (reset) m->reset_updated_box = true;
}
void fo_map_set_updated_box(fo_map* m, const double bmin[3], const double bmax[3]) {
  for (int i = 0; i < 3; ++i) m->upd_min[i] = bmin[i], m->upd_max[i] = bmax[i];
  m->reset_updated_box = false;
}

int fo_map_get_occupancy_idx(const fo_map* m, const int id[3]) {
  return m->getOccupancy(V3i{{id[0], id[1], id[2]}});
}
int fo_map_get_occupancy_pos(const fo_map* m, const double pos[3]) {
  V3i id;
  m->posToIndex(mk(pos[0], pos[1], pos[2]), id);
  return m->getOccupancy(id);
}
int fo_map_get_inflate_idx(const fo_map* m, const int id[3]) {
  V3i i3{{id[0], id[1], id[2]}};
  if (!m->isInMap(i3)) return -1;
  return (int)m->infl[m->adr(i3)];
}
double fo_map_get_distance_idx(const fo_map* m, const int id[3]) {
  return m->getDistance(V3i{{id[0], id[1], id[2]}});
}

// ---------------------------------------------------------------------------------------------
// Fusion -- plan_env/src/sdf_map.cpp:243-257 (setCacheOccupancy), :259-345 (inputPointCloud),
// :347-362 (closetPointInMap)
// ---------------------------------------------------------------------------------------------
static void set_cache_occupancy(fo_map* m, int a, int occ) {
  if (m->cnt_hit[a] == 0 && m->cnt_miss[a] == 0) m->cache.push_back(a);
  if (occ == 0)
    m->cnt_miss[a] = 1;  // NOT incremented in the reference
  else if (occ == 1)
    m->cnt_hit[a] += 1;
}

static V3d closest_point_in_map(const fo_map* m, const V3d& pt, const V3d& cam) {
  V3d diff = pt - cam;
  V3d max_tc = m->max_bound - cam;
  V3d min_tc = m->min_bound - cam;
  double min_t = 1000000;
  for (int i = 0; i < 3; ++i) {
    if (std::fabs(diff[i]) > 0) {
      double t1 = max_tc[i] / diff[i];
      if (t1 > 0 && t1 < min_t) min_t = t1;
      double t2 = min_tc[i] / diff[i];
      if (t2 > 0 && t2 < min_t) min_t = t2;
    }
  }
  return cam + (min_t - 1e-3) * diff;
}

void fo_map_input_points(fo_map* m, const float* xyz, int stride_bytes, int n, const double camv[3]) {
  if (n == 0) return;
  m->raycast_num += 1;  // char: wraps like the reference (sdf_map.h:118)
  V3d cam = mk(camv[0], camv[1], camv[2]);
  V3d umin = cam, umax = cam;
  if (m->reset_updated_box) {
    m->upd_min = cam;
    m->upd_max = cam;
    m->reset_updated_box = false;
  }
  V3i idx;
  for (int i = 0; i < n; ++i) {
    const float* p = (const float*)((const char*)xyz + (size_t)i * stride_bytes);
    V3d pt = mk(p[0], p[1], p[2]);
    int flag;
    double length;
    if (!m->isInMap(pt)) {
      pt = closest_point_in_map(m, pt, cam);
      length = norm(pt - cam);
      if (length > m->max_ray_length) pt = (pt - cam) / length * m->max_ray_length + cam;
      if (pt[2] < 0.2) continue;
      flag = 0;
    } else {
      length = norm(pt - cam);
      if (length > m->max_ray_length) {
        pt = (pt - cam) / length * m->max_ray_length + cam;
        if (pt[2] < 0.2) continue;
        flag = 0;
      } else
        flag = 1;
    }
    m->posToIndex(pt, idx);
    int a = m->adr(idx);
    set_cache_occupancy(m, a, flag);
    for (int k = 0; k < 3; ++k) {
      umin[k] = std::min(umin[k], pt[k]);
      umax[k] = std::max(umax[k], pt[k]);
    }
    // only the FIRST point landing in an end voxel this frame casts a ray
    if (m->flag_rayend[a] == m->raycast_num) continue;
    m->flag_rayend[a] = m->raycast_num;

    m->caster.input(pt, cam);
    m->caster.nextId(idx);  // end voxel itself discarded
    // safety cap (the reference has none): a correct walk takes exactly |dx|+|dy|+|dz| steps
    int cap = std::abs(m->caster.ex - m->caster.x) + std::abs(m->caster.ey - m->caster.y) +
        std::abs(m->caster.ez - m->caster.z) + 4;
    while (m->caster.nextId(idx)) {
      set_cache_occupancy(m, m->adr(idx), 0);
      if (--cap < 0) break;
    }
  }
  V3d inf = mk(m->local_bound_inflate, m->local_bound_inflate, 0);
  m->posToIndex(umax + inf, m->lb_max);
  m->posToIndex(umin - inf, m->lb_min);
  m->boundIndex(m->lb_min);
  m->boundIndex(m->lb_max);
  m->local_updated = true;
  for (int k = 0; k < 3; ++k) {
    m->upd_min[k] = std::min(umin[k], m->upd_min[k]);
    m->upd_max[k] = std::max(umax[k], m->upd_max[k]);
  }
  while (!m->cache.empty()) {
    int a = m->cache.front();
    m->cache.pop_front();
    double upd = m->cnt_hit[a] >= m->cnt_miss[a] ? m->l_hit : m->l_miss;
    m->cnt_hit[a] = m->cnt_miss[a] = 0;
    if (m->occ[a] < m->l_min - 1e-3) m->occ[a] = m->l_occ;
    m->occ[a] = std::min(std::max(m->occ[a] + upd, m->l_min), m->l_max);
  }
}

int fo_raycast_cells(const fo_map* m, const double start[3], const double end[3], int* out, int cap) {
  RayWalk w;
  w.setParams(m->res, m->origin);
  w.input(mk(start[0], start[1], start[2]), mk(end[0], end[1], end[2]));
  V3i idx;
  w.nextId(idx);
  int n = 0;
  int guard = std::abs(w.ex - w.x) + std::abs(w.ey - w.y) + std::abs(w.ez - w.z) + 4;
  while (w.nextId(idx)) {
    if (n < cap) {
      out[3 * n + 0] = idx[0];
      out[3 * n + 1] = idx[1];
      out[3 * n + 2] = idx[2];
    }
    ++n;
    if (--guard < 0) break;
  }
  return n;
}

// ---------------------------------------------------------------------------------------------
// Inflation -- plan_env/src/sdf_map.cpp:434-471, sdf_map.h:239-266 (all-inflate cube).
// Quirk kept: the only bounds test is 0 <= linear address < N, so stamps wrap across rows.
// ---------------------------------------------------------------------------------------------
void fo_map_inflate_local(fo_map* m) {
  int step = (int)std::ceil(m->obstacles_inflation / m->res);
  const int N = m->total();
  for (int x = m->lb_min[0]; x <= m->lb_max[0]; ++x)
    for (int y = m->lb_min[1]; y <= m->lb_max[1]; ++y)
      for (int z = m->lb_min[2]; z <= m->lb_max[2]; ++z) m->infl[m->adr(x, y, z)] = 0;
  for (int x = m->lb_min[0]; x <= m->lb_max[0]; ++x)
    for (int y = m->lb_min[1]; y <= m->lb_max[1]; ++y)
      for (int z = m->lb_min[2]; z <= m->lb_max[2]; ++z) {
        if (m->occ[m->adr(x, y, z)] > m->l_occ) {
          for (int dx = -step; dx <= step; ++dx)
            for (int dy = -step; dy <= step; ++dy)
              for (int dz = -step; dz <= step; ++dz) {
                int a = m->adr(x + dx, y + dy, z + dz);
                if (a >= 0 && a < N) m->infl[a] = 1;
              }
        }
      }
  if (m->virtual_ceil_height > -0.5) {
    int ceil_id = (int)std::floor((m->virtual_ceil_height - m->origin[2]) * m->res_inv);
    for (int x = m->lb_min[0]; x <= m->lb_max[0]; ++x)
      for (int y = m->lb_min[1]; y <= m->lb_max[1]; ++y) m->occ[m->adr(x, y, ceil_id)] = m->l_max;
  }
}

// ---------------------------------------------------------------------------------------------
// ESDF -- plan_env/src/sdf_map.cpp:116-150 (fillESDF: 1-D lower envelope of parabolas),
// :152-241 (updateESDF3d: z, y, x passes restricted to the local box; optional signed pass).
// "Infinity" is DBL_MAX exactly as in the reference.
// ---------------------------------------------------------------------------------------------
}  // extern "C"
template <typename FG, typename FS>
static void fill_esdf(FG get, FS set, int start, int end, int dimlen) {
  std::vector<int> v(dimlen);
  std::vector<double> z(dimlen + 1);
  const double DMAX = std::numeric_limits<double>::max();
  int k = start;
  v[start] = start;
  z[start] = -DMAX;
  z[start + 1] = DMAX;
  for (int q = start + 1; q <= end; q++) {
    k++;
    double s;
    do {
      k--;
      s = ((get(q) + q * q) - (get(v[k]) + v[k] * v[k])) / (2 * q - 2 * v[k]);
    } while (s <= z[k]);
    k++;
    v[k] = q;
    z[k] = s;
    z[k + 1] = DMAX;
  }
  k = start;
  for (int q = start; q <= end; q++) {
    while (z[k + 1] < q) k++;
    double val = (q - v[k]) * (q - v[k]) + get(v[k]);
    set(q, val);
  }
}

extern "C" {
void fo_map_update_esdf(fo_map* m) {
  const V3i lo = m->lb_min, hi = m->lb_max;
  const double DMAX = std::numeric_limits<double>::max();
  auto three_pass = [&](auto is_source, std::vector<double>& out) {
    for (int x = lo[0]; x <= hi[0]; x++)
      for (int y = lo[1]; y <= hi[1]; y++)
        fill_esdf([&](int z) { return is_source(m->adr(x, y, z)) ? 0.0 : DMAX; },
                  [&](int z, double val) { m->tmp1[m->adr(x, y, z)] = val; }, lo[2], hi[2], m->nvox[2]);
    for (int x = lo[0]; x <= hi[0]; x++)
      for (int z = lo[2]; z <= hi[2]; z++)
        fill_esdf([&](int y) { return m->tmp1[m->adr(x, y, z)]; },
                  [&](int y, double val) { m->tmp2[m->adr(x, y, z)] = val; }, lo[1], hi[1], m->nvox[1]);
    for (int y = lo[1]; y <= hi[1]; y++)
      for (int z = lo[2]; z <= hi[2]; z++)
        fill_esdf([&](int x) { return m->tmp2[m->adr(x, y, z)]; },
                  [&](int x, double val) { out[m->adr(x, y, z)] = m->res * std::sqrt(val); }, lo[0], hi[0],
                  m->nvox[0]);
  };
  if (m->optimistic)
    three_pass([&](int a) { return m->infl[a] == 1; }, m->dist);
  else
    three_pass([&](int a) { return m->infl[a] == 1 || m->occ[a] < m->l_min - 1e-3; }, m->dist);
  if (m->signed_dist) {
    three_pass([&](int a) { return m->infl[a] == 0; }, m->dist_neg);
    for (int x = lo[0]; x <= hi[0]; ++x)
      for (int y = lo[1]; y <= hi[1]; ++y)
        for (int z = lo[2]; z <= hi[2]; ++z) {
          int a = m->adr(x, y, z);
          if (m->dist_neg[a] > 0.0) m->dist[a] += (-m->dist_neg[a] + m->res);
        }
  }
}

// ---------------------------------------------------------------------------------------------
// Trilinear distance + gradient -- plan_env/src/sdf_map.cpp:497-536
// (EDTEnvironment::evaluateEDTWithGrad forwards here: edt_environment.cpp:78-87)
// ---------------------------------------------------------------------------------------------
static double dist_with_grad(const fo_map* m, const V3d& pos, V3d& grad) {
  if (!m->isInMap(pos)) {
    grad = mk(0, 0, 0);
    return 0;
  }
  V3d pos_m = pos - 0.5 * m->res * mk(1, 1, 1);
  V3i idx;
  m->posToIndex(pos_m, idx);
  V3d idx_pos, diff;
  m->indexToPos(idx, idx_pos);
  diff = (pos - idx_pos) * m->res_inv;
  double val[2][2][2];
  for (int x = 0; x < 2; x++)
    for (int y = 0; y < 2; y++)
      for (int z = 0; z < 2; z++) val[x][y][z] = m->getDistance(V3i{{idx[0] + x, idx[1] + y, idx[2] + z}});
  double v00 = (1 - diff[0]) * val[0][0][0] + diff[0] * val[1][0][0];
  double v01 = (1 - diff[0]) * val[0][0][1] + diff[0] * val[1][0][1];
  double v10 = (1 - diff[0]) * val[0][1][0] + diff[0] * val[1][1][0];
  double v11 = (1 - diff[0]) * val[0][1][1] + diff[0] * val[1][1][1];
  double v0 = (1 - diff[1]) * v00 + diff[1] * v10;
  double v1 = (1 - diff[1]) * v01 + diff[1] * v11;
  double d = (1 - diff[2]) * v0 + diff[2] * v1;
  grad[2] = (v1 - v0) * m->res_inv;
  grad[1] = ((1 - diff[2]) * (v10 - v00) + diff[2] * (v11 - v01)) * m->res_inv;
  grad[0] = (1 - diff[2]) * (1 - diff[1]) * (val[1][0][0] - val[0][0][0]);
  grad[0] += (1 - diff[2]) * diff[1] * (val[1][1][0] - val[0][1][0]);
  grad[0] += diff[2] * (1 - diff[1]) * (val[1][0][1] - val[0][0][1]);
  grad[0] += diff[2] * diff[1] * (val[1][1][1] - val[0][1][1]);
  grad[0] *= m->res_inv;
  return d;
}

void fo_map_dist_grad(const fo_map* m, const double* pos, int n, double* dist, double* grad) {
  for (int i = 0; i < n; ++i) {
    V3d g;
    dist[i] = dist_with_grad(m, mk(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), g);
    grad[3 * i] = g[0], grad[3 * i + 1] = g[1], grad[3 * i + 2] = g[2];
  }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// FrontierFinder -- active_perception/src/frontier_finder.cpp
//   :54-121 searchFrontiers, :123-164 expandFrontier, :353-363 haveOverlap,
//   :365-372 isFrontierChanged, :374-390 computeFrontierInfo (minus PCL downsample),
//   :811-829 sixNeighbors, :848-860 allNeighbors, :862-877 isNeighborUnknown/knownfree
// ---------------------------------------------------------------------------------------------
namespace {
struct Viewpoint {
  V3d pos;
  double yaw;
  int visib_num;
};
struct Cluster {
  std::vector<Viewpoint> viewpoints;
  std::vector<V3d> cells;     // voxel centres, BFS order
  std::vector<V3d> filtered;  // filtered_cells_: VoxelGrid centroids (float precision, like pcl::PointXYZ)
  V3d average, bmin, bmax;
};

// pcl::VoxelGrid<pcl::PointXYZ>::applyFilter restated (PCL is a third-party dependency that is not
// under /root/reference; PCL 1.8-1.12 filters/include/pcl/filters/impl/voxel_grid.hpp): float
// arithmetic throughout, leaves aligned to global multiples of the leaf size, one centroid per
// occupied leaf, output sorted by leaf index (x fastest).  PCL sorts with std::sort (order of equal
// keys unspecified, which only permutes the float summation inside a leaf); this restatement sums in
// input order.  PARITY UNPINNED against PCL itself.
void voxel_grid_downsample(const std::vector<V3d>& in, double leaf_d, std::vector<V3d>& out) {
  out.clear();
  if (in.empty()) return;
  struct P { float x, y, z; };
  std::vector<P> pts;
  pts.reserve(in.size());
  for (auto& c : in) pts.push_back(P{(float)c[0], (float)c[1], (float)c[2]});  // emplace_back(cell[0..2]) -> float
  const float leaf = (float)leaf_d;  // setLeafSize(float, float, float)
  const float inv = 1.0f / leaf;
  float mn[3] = {pts[0].x, pts[0].y, pts[0].z}, mx[3] = {pts[0].x, pts[0].y, pts[0].z};
  for (auto& p : pts) {
    const float v[3] = {p.x, p.y, p.z};
    for (int i = 0; i < 3; ++i) {
      mn[i] = std::min(mn[i], v[i]);
      mx[i] = std::max(mx[i], v[i]);
    }
  }
  int min_b[3], max_b[3], div_b[3];
  for (int i = 0; i < 3; ++i) {
    min_b[i] = (int)std::floor(mn[i] * inv);
    max_b[i] = (int)std::floor(mx[i] * inv);
    div_b[i] = max_b[i] - min_b[i] + 1;
  }
  const int mul[3] = {1, div_b[0], div_b[0] * div_b[1]};
  std::vector<std::pair<unsigned, unsigned>> iv;  // (leaf idx, point index)
  iv.reserve(pts.size());
  for (unsigned k = 0; k < pts.size(); ++k) {
    const int i0 = (int)(std::floor(pts[k].x * inv) - (float)min_b[0]);
    const int i1 = (int)(std::floor(pts[k].y * inv) - (float)min_b[1]);
    const int i2 = (int)(std::floor(pts[k].z * inv) - (float)min_b[2]);
    iv.emplace_back((unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]), k);
  }
  std::stable_sort(iv.begin(), iv.end(),
                   [](const std::pair<unsigned, unsigned>& a, const std::pair<unsigned, unsigned>& b) {
                     return a.first < b.first;
                   });
  size_t i = 0;
  while (i < iv.size()) {
    size_t j = i;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    while (j < iv.size() && iv[j].first == iv[i].first) {
      sx += pts[iv[j].second].x;
      sy += pts[iv[j].second].y;
      sz += pts[iv[j].second].z;
      ++j;
    }
    const float n = (float)(j - i);
    out.push_back(mk((double)(sx / n), (double)(sy / n), (double)(sz / n)));
    i = j;
  }
}
}  // namespace

struct fo_frontier {
  fo_map* map;
  int cluster_min;
  double min_z;
  double cluster_size_xy = 2.0;
  int down_sample = 3;
  int split = 0;
  int canonical_order = 0;
  int flip_pc = 0;
  fo_viewpoint_cfg vp{};
  // PerceptionUtils state (perception_utils.cpp:6-19 constructor, :49-69 setPose)
  V3d pu_pos;
  V3d pu_normals[4];

  void setPose(const V3d& pos, double yaw) {
    pu_pos = pos;
    const double hp = M_PI_2;
    const V3d n_cam[4] = {mk(0.0, std::sin(hp - vp.top_angle), std::cos(hp - vp.top_angle)),
                          mk(0.0, -std::sin(hp - vp.top_angle), std::cos(hp - vp.top_angle)),
                          mk(std::sin(hp - vp.left_angle), 0.0, std::cos(hp - vp.left_angle)),
                          mk(-std::sin(hp - vp.right_angle), 0.0, std::cos(hp - vp.right_angle))};
    // R_wc = R_wb(yaw) * R_bc with T_bc = inverse(T_cb), T_cb = [0 -1 0; 0 0 1; 1 0 0]: the products with
    // the 0 / +-1 entries are exact, leaving R_wc = [s 0 c; -c 0 s; 0 1 0]
    const double c = std::cos(yaw), s = std::sin(yaw);
    const double R[3][3] = {{s, 0.0, c}, {-c, 0.0, s}, {0.0, 1.0, 0.0}};
    for (int k = 0; k < 4; ++k)
      for (int i = 0; i < 3; ++i)
        pu_normals[k][i] = R[i][0] * n_cam[k][0] + R[i][1] * n_cam[k][1] + R[i][2] * n_cam[k][2];
  }
  bool insideFOV(const V3d& point) const {  // :84-93
    V3d dir = point - pu_pos;
    const double nrm = std::sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    if (nrm > vp.max_dist) return false;
    dir = dir / nrm;
    for (auto& n : pu_normals)
      if (dir[0] * n[0] + dir[1] * n[1] + dir[2] * n[2] < 0.0) return false;
    return true;
  }
  bool isNearUnknown(const V3d& pos) const {  // :721-731
    const int vox_num = (int)std::floor(vp.min_candidate_clearance / map->res);
    for (int x = -vox_num; x <= vox_num; ++x)
      for (int y = -vox_num; y <= vox_num; ++y)
        for (int z = -1; z <= 1; ++z) {
          const V3d vox = mk(pos[0] + x * map->res, pos[1] + y * map->res, pos[2] + z * map->res);
          V3i id;
          map->posToIndex(vox, id);
          if (map->getOccupancy(id) == 0) return true;
        }
    return false;
  }
  int countVisibleCells(const V3d& pos, double yaw, const std::vector<V3d>& cluster) {  // :733-755
    setPose(pos, yaw);
    int visib_num = 0;
    RayWalk rc;
    rc.setParams(map->res, map->origin);
    for (auto& cell : cluster) {
      if (!insideFOV(cell)) continue;
      rc.input(cell, pos);
      bool visib = true;
      V3i idx;
      while (rc.nextId(idx)) {
        if (fo_map_get_inflate_idx(map, idx.v) == 1 || map->getOccupancy(idx) == 0) {
          visib = false;
          break;
        }
      }
      if (visib) visib_num += 1;
    }
    return visib_num;
  }
  static void wrapYaw(double& yaw) {
    while (yaw < -M_PI) yaw += 2 * M_PI;
    while (yaw > M_PI) yaw -= 2 * M_PI;
  }
  void sampleViewpoints(Cluster& ftr) {  // :662-695
    for (double rc = vp.candidate_rmin, dr = (vp.candidate_rmax - vp.candidate_rmin) / vp.candidate_rnum;
         rc <= vp.candidate_rmax + 1e-3; rc += dr)
      for (double phi = -M_PI; phi < M_PI; phi += vp.candidate_dphi) {
        const V3d sample_pos = ftr.average + rc * mk(std::cos(phi), std::sin(phi), 0);
        V3i sid;
        map->posToIndex(sample_pos, sid);
        if (!map->isInBoxPos(sample_pos) || fo_map_get_inflate_idx(map, sid.v) == 1 || isNearUnknown(sample_pos))
          continue;
        auto& cells = ftr.filtered;
        auto normalized = [](const V3d& v) {
          return v / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        };
        const V3d ref_dir = normalized(cells.front() - sample_pos);
        double avg_yaw = 0.0;
        for (size_t i = 1; i < cells.size(); ++i) {
          const V3d dir = normalized(cells[i] - sample_pos);
          double yaw = std::acos(dir[0] * ref_dir[0] + dir[1] * ref_dir[1] + dir[2] * ref_dir[2]);
          if (ref_dir[0] * dir[1] - ref_dir[1] * dir[0] < 0) yaw = -yaw;  // ref_dir.cross(dir)[2]
          avg_yaw += yaw;
        }
        avg_yaw = avg_yaw / cells.size() + std::atan2(ref_dir[1], ref_dir[0]);
        wrapYaw(avg_yaw);
        const int visib_num = countVisibleCells(sample_pos, avg_yaw, cells);
        if (visib_num > vp.min_visib_num) ftr.viewpoints.push_back(Viewpoint{sample_pos, avg_yaw, visib_num});
      }
  }
  void computeFrontiersToVisit() {  // :392-423
    for (auto& t : tmp) {
      sampleViewpoints(t);
      if (!t.viewpoints.empty()) {
        frontiers.push_back(t);
        auto& v = frontiers.back().viewpoints;
        std::sort(v.begin(), v.end(), [](const Viewpoint& a, const Viewpoint& b) { return a.visib_num > b.visib_num; });
      } else
        dormant.push_back(t);
    }
    // (tmp_frontiers_ is not cleared by the reference; the next searchFrontiers() clears it)
  }
  bool isFrontierCovered() {  // :697-719
    V3d umin, umax;
    fo_map_get_updated_box(map, umin.v, umax.v, 0);
    auto check = [&](const std::list<Cluster>& L) {
      for (auto& ftr : L) {
        if (!haveOverlap(ftr.bmin, ftr.bmax, umin, umax)) continue;
        const int change_thresh = vp.min_view_finish_fraction * ftr.cells.size();
        int change_num = 0;
        for (auto& cell : ftr.cells) {
          V3i idx;
          map->posToIndex(cell, idx);
          if (!isFrontier(idx) && ++change_num >= change_thresh) return true;
        }
      }
      return false;
    };
    return check(frontiers) || check(dormant);
  }

  std::vector<char> flag;
  std::list<Cluster> frontiers, dormant, tmp;
  std::vector<int> removed_ids;

  bool knownfree(const V3i& id) const { return map->getOccupancy(id) == 1; }
  bool isNeighborUnknown(const V3i& v) const {
    static const int d[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
    for (auto& o : d)
      if (map->getOccupancy(V3i{{v[0] + o[0], v[1] + o[1], v[2] + o[2]}}) == 0) return true;
    return false;
  }
  bool isFrontier(const V3i& id) const { return knownfree(id) && isNeighborUnknown(id); }

  static bool haveOverlap(const V3d& min1, const V3d& max1, const V3d& min2, const V3d& max2) {
    for (int i = 0; i < 3; ++i) {
      double bmin = std::max(min1[i], min2[i]);
      double bmax = std::min(max1[i], max2[i]);
      if (bmin > bmax + 1e-3) return false;
    }
    return true;
  }
  bool isFrontierChanged(const Cluster& c) const {
    for (auto& cell : c.cells) {
      V3i idx;
      map->posToIndex(cell, idx);
      if (!isFrontier(idx)) return true;
    }
    return false;
  }
  // downsample (:757-774): leaf = resolution * down_sample_
  void downsample(const std::vector<V3d>& in, std::vector<V3d>& out) const {
    if (!canonical_order) {
      voxel_grid_downsample(in, map->res * down_sample, out);
      return;
    }
    std::vector<std::pair<int, size_t>> order;  // (voxel address, position in `in`)
    order.reserve(in.size());
    for (size_t i = 0; i < in.size(); ++i) {
      V3i id;
      map->posToIndex(in[i], id);
      order.emplace_back(map->adr(id), i);
    }
    std::sort(order.begin(), order.end());
    std::vector<V3d> sorted;
    sorted.reserve(in.size());
    for (auto& o : order) sorted.push_back(in[o.second]);
    voxel_grid_downsample(sorted, map->res * down_sample, out);
  }

  // splitHorizontally (:179-242).  The principal direction comes from Eigen::EigenSolver<Matrix2d>
  // in the reference (Eigen: third-party, absent); restated as the closed-form symmetric 2x2
  // decomposition with the convention v = normalise(b, lambda_max - a) (fallbacks for b == 0) -- the
  // SIGN of the eigenvector decides which half is emitted first and is UNPINNED against Eigen.
  static void principal_dir(double c00, double c01, double c10, double c11, double pc[2]) {
    const double a = c00, b = 0.5 * (c01 + c10), d = c11;
    const double tr = a + d, det = a * d - b * b, disc = std::sqrt(std::fmax(tr * tr / 4 - det, 0.0));
    const double l = tr / 2 + disc;
    double vx = b, vy = l - a;
    if (std::fabs(vx) + std::fabs(vy) < 1e-300) {
      vx = l - d;
      vy = b;
    }
    if (std::fabs(vx) + std::fabs(vy) < 1e-300) {
      vx = 1;
      vy = 0;
    }
    const double n = std::sqrt(vx * vx + vy * vy);
    pc[0] = vx / n;
    pc[1] = vy / n;
  }
  bool splitHorizontally(const Cluster& ftr, std::list<Cluster>& splits) const {
    const double mean[2] = {ftr.average[0], ftr.average[1]};
    bool need_split = false;
    for (auto& cell : ftr.filtered) {
      const double dx = cell[0] - mean[0], dy = cell[1] - mean[1];
      if (std::sqrt(dx * dx + dy * dy) > cluster_size_xy) {
        need_split = true;
        break;
      }
    }
    if (!need_split) return false;
    double c00 = 0, c01 = 0, c10 = 0, c11 = 0;
    for (auto& cell : ftr.filtered) {
      const double dx = cell[0] - mean[0], dy = cell[1] - mean[1];
      c00 += dx * dx;
      c01 += dx * dy;
      c10 += dy * dx;
      c11 += dy * dy;
    }
    const double nf = double(ftr.filtered.size());
    c00 /= nf, c01 /= nf, c10 /= nf, c11 /= nf;
    double pc[2];
    principal_dir(c00, c01, c10, c11, pc);
    if (flip_pc) pc[0] = -pc[0], pc[1] = -pc[1];
    Cluster f1, f2;
    for (auto& cell : ftr.cells) {
      if ((cell[0] - mean[0]) * pc[0] + (cell[1] - mean[1]) * pc[1] >= 0)
        f1.cells.push_back(cell);
      else
        f2.cells.push_back(cell);
    }
    // (the reference would dereference cells_.front() of an empty half: cannot happen for a cluster
    // whose cells straddle their own mean along pc, which need_split implies)
    Cluster* halves[2] = {&f1, &f2};
    for (Cluster* h : halves) {
      if (h->cells.empty()) continue;
      computeInfo(*h);
      std::list<Cluster> sub;
      if (splitHorizontally(*h, sub))
        splits.insert(splits.end(), sub.begin(), sub.end());
      else
        splits.push_back(*h);
    }
    return true;
  }
  void splitLarge(std::list<Cluster>& L) const {  // splitLargeFrontiers (:166-177)
    std::list<Cluster> tmps, splits;
    for (auto& c : L) {
      if (splitHorizontally(c, splits)) {
        tmps.insert(tmps.end(), splits.begin(), splits.end());
        splits.clear();
      } else
        tmps.push_back(c);
    }
    L = tmps;
  }

  void computeInfo(Cluster& c) const {
    c.average = mk(0, 0, 0);
    c.bmax = c.cells.front();
    c.bmin = c.cells.front();
    for (auto& cell : c.cells) {
      c.average = c.average + cell;
      for (int i = 0; i < 3; ++i) {
        c.bmin[i] = std::min(c.bmin[i], cell[i]);
        c.bmax[i] = std::max(c.bmax[i], cell[i]);
      }
    }
    c.average = c.average / double(c.cells.size());
    if (canonical_order) {
      // order-free evaluation of the same mean: exact integer sum of the voxel indices, one rounding
      // chain at the end (the sequential f64 sum above carries ~1e-13 of order-dependent noise, enough
      // to move a mean that sits exactly on a voxel face -- e.g. the z of a full-height wall -- across it)
      long long sum[3] = {0, 0, 0};
      for (auto& cell : c.cells) {
        V3i id;
        map->posToIndex(cell, id);
        for (int i = 0; i < 3; ++i) sum[i] += id[i];
      }
      for (int i = 0; i < 3; ++i)
        c.average[i] = ((double)sum[i] / double(c.cells.size()) + 0.5) * map->res + map->origin[i];
    }
    if (down_sample > 0) downsample(c.cells, c.filtered);
  }

  void expand(const V3i& first) {
    std::deque<V3i> queue;
    std::vector<V3d> expanded;
    V3d pos;
    map->indexToPos(first, pos);
    expanded.push_back(pos);
    queue.push_back(first);
    flag[map->adr(first)] = 1;
    while (!queue.empty()) {
      V3i cur = queue.front();
      queue.pop_front();
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dz = -1; dz <= 1; ++dz) {
            if (dx == 0 && dy == 0 && dz == 0) continue;
            V3i nbr{{cur[0] + dx, cur[1] + dy, cur[2] + dz}};
            // The reference reads frontier_flag_[toadr(nbr)] BEFORE the box test (UB when nbr is
            // outside the map).  Outside the map isInBox() is false (box is inside the map), so
            // the outcome is "continue" either way; test the box first to stay defined.
            if (!map->isInMap(nbr) || !map->isInBox(nbr)) continue;
            int a = map->adr(nbr);
            if (flag[a] == 1 || !isFrontier(nbr)) continue;
            map->indexToPos(nbr, pos);
            if (pos[2] < min_z) continue;
            expanded.push_back(pos);
            queue.push_back(nbr);
            flag[a] = 1;
          }
    }
    if ((int)expanded.size() > cluster_min) {
      Cluster c;
      c.cells = expanded;
      computeInfo(c);
      tmp.push_back(c);
    }
    // rejected small clusters keep flag == 1 forever (reference quirk)
  }

  int search() {
    tmp.clear();
    V3d umin, umax;
    fo_map_get_updated_box(map, umin.v, umax.v, 1);
    auto resetFlag = [&](std::list<Cluster>::iterator& it, std::list<Cluster>& L) {
      V3i idx;
      for (auto& cell : it->cells) {
        map->posToIndex(cell, idx);
        flag[map->adr(idx)] = 0;
      }
      it = L.erase(it);
    };
    removed_ids.clear();
    int rmv_idx = 0;
    for (auto it = frontiers.begin(); it != frontiers.end();) {
      if (haveOverlap(it->bmin, it->bmax, umin, umax) && isFrontierChanged(*it)) {
        resetFlag(it, frontiers);
        removed_ids.push_back(rmv_idx);
      } else {
        ++rmv_idx;
        ++it;
      }
    }
    for (auto it = dormant.begin(); it != dormant.end();) {
      if (haveOverlap(it->bmin, it->bmax, umin, umax) && isFrontierChanged(*it))
        resetFlag(it, dormant);
      else
        ++it;
    }
    V3d smin = umin - mk(1, 1, 0.5), smax = umax + mk(1, 1, 0.5);
    for (int k = 0; k < 3; ++k) {
      smin[k] = std::max(smin[k], map->box_mind[k]);
      smax[k] = std::min(smax[k], map->box_maxd[k]);
    }
    V3i lo, hi;
    map->posToIndex(smin, lo);
    map->posToIndex(smax, hi);
    // The reference loops lo..hi inclusive without a map test; hi can equal nvox when the
    // exploration box touches the map face (UB there).  Clamp to the map to stay defined.
    for (int k = 0; k < 3; ++k) {
      lo[k] = std::max(lo[k], 0);
      hi[k] = std::min(hi[k], map->nvox[k] - 1);
    }
    for (int x = lo[0]; x <= hi[0]; ++x)
      for (int y = lo[1]; y <= hi[1]; ++y)
        for (int z = lo[2]; z <= hi[2]; ++z) {
          V3i cur{{x, y, z}};
          if (flag[map->adr(cur)] == 0 && isFrontier(cur)) expand(cur);
        }
    if (split) splitLarge(tmp);  // :120
    return (int)tmp.size();
  }
};

extern "C" {

fo_frontier* fo_frontier_create(fo_map* m, const fo_frontier_cfg* cfg) {
  fo_frontier* f = new fo_frontier;
  f->map = m;
  f->cluster_min = cfg->cluster_min;
  f->min_z = cfg->min_z;
  f->cluster_size_xy = cfg->cluster_size_xy;
  f->down_sample = cfg->down_sample;  // <= 0: filtered_cells_ not computed (F1-F4 contract only)
  f->split = cfg->split;
  f->canonical_order = cfg->canonical_order;
  f->flip_pc = cfg->flip_principal_dir;
  f->flag.assign((size_t)m->total(), 0);
  return f;
}
void fo_frontier_destroy(fo_frontier* f) { delete f; }
char* fo_frontier_flags(fo_frontier* f) { return f->flag.data(); }
int fo_frontier_search(fo_frontier* f) { return f->search(); }
void fo_frontier_commit(fo_frontier* f, int dormant) {
  auto& dst = dormant ? f->dormant : f->frontiers;
  dst.insert(dst.end(), f->tmp.begin(), f->tmp.end());
  f->tmp.clear();
}
static const std::list<Cluster>& pick(const fo_frontier* f, int which) {
  return which == 0 ? f->tmp : (which == 1 ? f->frontiers : f->dormant);
}
static const Cluster& nth(const std::list<Cluster>& L, int k) {
  auto it = L.begin();
  std::advance(it, k);
  return *it;
}
int fo_frontier_count(const fo_frontier* f, int which) { return (int)pick(f, which).size(); }
int fo_frontier_cluster_size(const fo_frontier* f, int which, int k) {
  return (int)nth(pick(f, which), k).cells.size();
}
void fo_frontier_cluster_cells(const fo_frontier* f, int which, int k, int* adr) {
  const Cluster& c = nth(pick(f, which), k);
  V3i idx;
  for (size_t i = 0; i < c.cells.size(); ++i) {
    f->map->posToIndex(c.cells[i], idx);
    adr[i] = f->map->adr(idx);
  }
}
void fo_frontier_cluster_info(const fo_frontier* f, int which, int k, double* out9) {
  const Cluster& c = nth(pick(f, which), k);
  for (int i = 0; i < 3; ++i) out9[i] = c.average[i], out9[3 + i] = c.bmin[i], out9[6 + i] = c.bmax[i];
}
void fo_frontier_set_viewpoint_cfg(fo_frontier* f, const fo_viewpoint_cfg* c) { f->vp = *c; }
void fo_frontier_compute_to_visit(fo_frontier* f) { f->computeFrontiersToVisit(); }
int fo_frontier_is_covered(fo_frontier* f) { return f->isFrontierCovered() ? 1 : 0; }
int fo_frontier_viewpoint_count(const fo_frontier* f, int which, int k) {
  return (int)nth(pick(f, which), k).viewpoints.size();
}
void fo_frontier_viewpoints(const fo_frontier* f, int which, int k, double* pos_yaw4, int* visib) {
  const Cluster& c = nth(pick(f, which), k);
  for (size_t i = 0; i < c.viewpoints.size(); ++i) {
    for (int q = 0; q < 3; ++q) pos_yaw4[4 * i + q] = c.viewpoints[i].pos[q];
    pos_yaw4[4 * i + 3] = c.viewpoints[i].yaw;
    visib[i] = c.viewpoints[i].visib_num;
  }
}
int fo_frontier_cluster_filtered_size(const fo_frontier* f, int which, int k) {
  return (int)nth(pick(f, which), k).filtered.size();
}
void fo_frontier_cluster_filtered(const fo_frontier* f, int which, int k, double* xyz) {
  const Cluster& c = nth(pick(f, which), k);
  for (size_t i = 0; i < c.filtered.size(); ++i)
    for (int q = 0; q < 3; ++q) xyz[3 * i + q] = c.filtered[i][q];
}
int fo_frontier_removed_count(const fo_frontier* f) { return (int)f->removed_ids.size(); }
void fo_frontier_removed_ids(const fo_frontier* f, int* ids) {
  for (size_t i = 0; i < f->removed_ids.size(); ++i) ids[i] = f->removed_ids[i];
}

// ---------------------------------------------------------------------------------------------
// BsplineOptimizer cost terms -- bspline_opt/src/bspline_optimizer.cpp
//   :136-140 pt_dist_, :255-282 smoothness, :284-306 distance, :308-353 feasibility,
//   :355-391 start, :393-431 end, :433-457 waypoints, :462-475 guide, :477-502 view,
//   :504-516 time, :518-691 combineCost
// ---------------------------------------------------------------------------------------------
double fo_bspline_pt_dist(const double* ctrl, int n, int dim) {
  double d = 0.0;
  for (int i = 0; i < n - 1; ++i) {
    double s = 0;
    for (int j = 0; j < dim; ++j) {
      double e = ctrl[dim * (i + 1) + j] - ctrl[dim * i + j];
      s += e * e;
    }
    d += std::sqrt(s);
  }
  return d / double(n);
}

void fo_bspline_cost_grad(const fo_map* m, const fo_bspline_cfg* cfg, const fo_bspline_problem* pb,
                          const double* x, double* cost_out, double* grad) {
  const int N = pb->point_num, dim = pb->dim;
  const int MINTIME = 1 << 8;
  const bool opt_time = (pb->cost_function & MINTIME) != 0;
  const int nvar = opt_time ? dim * N + 1 : dim * N;
  const int order = (dim == 1) ? 3 : cfg->bspline_degree;
  std::vector<V3d> q(N), g(N);
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < dim; ++j) q[i][j] = x[dim * i + j];
    for (int j = dim; j < 3; ++j) q[i][j] = 0.0;
  }
  const double dt = opt_time ? x[nvar - 1] : pb->knot_span;
  double f = 0.0;
  for (int i = 0; i < nvar; ++i) grad[i] = 0.0;
  auto zero_g = [&]() {
    for (auto& e : g) e = mk(0, 0, 0);
  };
  auto add_all = [&](double ld) {
    for (int i = 0; i < N; i++)
      for (int j = 0; j < dim; j++) grad[dim * i + j] += ld * g[i][j];
  };

  if (pb->cost_function & (1 << 0)) {  // SMOOTHNESS
    double c = 0.0;
    zero_g();
    for (int i = 0; i < N - 3; i++) {
      V3d ji = (q[i + 3] - 3 * q[i + 2] + 3 * q[i + 1] - q[i]) / pb->pt_dist;
      c += sqnorm(ji);
      V3d tj = 2 * ji / pb->pt_dist;
      g[i + 0] = g[i + 0] + (-1.0) * tj;
      g[i + 1] = g[i + 1] + 3.0 * tj;
      g[i + 2] = g[i + 2] + (-3.0) * tj;
      g[i + 3] = g[i + 3] + tj;
    }
    f += cfg->ld_smooth * c;
    add_all(cfg->ld_smooth);
    // gt_smoothness stays 0 in the reference
  }
  if (pb->cost_function & (1 << 1)) {  // DISTANCE (static environment branch)
    double c = 0.0;
    zero_g();
    for (int i = 0; i < N; i++) {
      V3d dg;
      double d = dist_with_grad(m, q[i], dg);
      double nrm = norm(dg);
      if (nrm > 1e-4) dg = dg / nrm;  // Eigen normalize(): v /= v.norm()
      if (d < cfg->dist0) {
        c += std::pow(d - cfg->dist0, 2);
        g[i] = g[i] + 2.0 * (d - cfg->dist0) * dg;
      }
    }
    f += cfg->ld_dist * c;
    add_all(cfg->ld_dist);
  }
  if (pb->cost_function & (1 << 2)) {  // FEASIBILITY
    double c = 0.0, gt = 0.0;
    zero_g();
    const double dt_inv = 1 / dt, dt_inv2 = dt_inv * dt_inv;
    for (int i = 0; i < N - 1; ++i) {
      V3d vi = (q[i + 1] - q[i]) * dt_inv;
      for (int k = 0; k < 3; ++k) {
        double vd = std::fabs(vi[k]) - cfg->max_vel;
        if (vd > 0.0) {
          c += std::pow(vd, 2);
          double sign = vi[k] > 0 ? 1.0 : -1.0;
          double tmp = 2 * vd * sign * dt_inv;
          g[i][k] += -tmp;
          g[i + 1][k] += tmp;
          if (opt_time) gt += tmp * (-vi[k]);
        }
      }
    }
    for (int i = 0; i < N - 2; ++i) {
      V3d ai = (q[i + 2] - 2 * q[i + 1] + q[i]) * dt_inv2;
      for (int k = 0; k < 3; ++k) {
        double ad = std::fabs(ai[k]) - cfg->max_acc;
        if (ad > 0.0) {
          c += std::pow(ad, 2);
          double sign = ai[k] > 0 ? 1.0 : -1.0;
          double tmp = 2 * ad * sign * dt_inv2;
          g[i][k] += tmp;
          g[i + 1][k] += -2 * tmp;
          g[i + 2][k] += tmp;
          if (opt_time) gt += tmp * ai[k] * (-2) * dt;
        }
      }
    }
    f += cfg->ld_feasi * c;
    add_all(cfg->ld_feasi);
    if (opt_time) grad[nvar - 1] += cfg->ld_feasi * gt;
  }
  if (pb->cost_function & (1 << 3)) {  // START
    double c = 0.0, gt = 0.0;
    V3d gs[3] = {mk(0, 0, 0), mk(0, 0, 0), mk(0, 0, 0)};
    const double* ss = pb->start_state;
    V3d s0 = mk(ss[0], ss[1], ss[2]), s1 = mk(ss[3], ss[4], ss[5]), s2 = mk(ss[6], ss[7], ss[8]);
    V3d q1 = q[0], q2 = q[1], q3 = q[2], dq;
    static const double w_pos = 10.0;
    dq = 1 / 6.0 * (q1 + 4 * q2 + q3) - s0;
    c += w_pos * sqnorm(dq);
    gs[0] = gs[0] + w_pos * 2 * dq * (1 / 6.0);
    gs[1] = gs[1] + w_pos * 2 * dq * (4 / 6.0);
    gs[2] = gs[2] + w_pos * 2 * dq * (1 / 6.0);
    dq = 1 / (2 * dt) * (q3 - q1) - s1;
    c += sqnorm(dq);
    gs[0] = gs[0] + 2 * dq * (-1.0) / (2 * dt);
    gs[2] = gs[2] + 2 * dq * 1.0 / (2 * dt);
    if (opt_time) gt += dot(dq, q3 - q1) / (-dt * dt);
    dq = 1 / (dt * dt) * (q1 - 2 * q2 + q3) - s2;
    c += sqnorm(dq);
    gs[0] = gs[0] + 2 * dq * 1.0 / (dt * dt);
    gs[1] = gs[1] + 2 * dq * (-2.0) / (dt * dt);
    gs[2] = gs[2] + 2 * dq * 1.0 / (dt * dt);
    if (opt_time) gt += dot(dq, q1 - 2 * q2 + q3) / (-dt * dt * dt);
    f += cfg->ld_start * c;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < dim; j++) grad[dim * i + j] += cfg->ld_start * gs[i][j];
    if (opt_time) grad[nvar - 1] += cfg->ld_start * gt;
  }
  if (pb->cost_function & (1 << 4)) {  // END
    double c = 0.0, gt = 0.0;
    V3d ge[3] = {mk(0, 0, 0), mk(0, 0, 0), mk(0, 0, 0)};  // for points N-3, N-2, N-1
    const double* es = pb->end_state;
    V3d q_3 = q[N - 3], q_2 = q[N - 2], q_1 = q[N - 1], dq;
    dq = 1 / 6.0 * (q_1 + 4 * q_2 + q_3) - mk(es[0], es[1], es[2]);
    c += sqnorm(dq);
    ge[2] = ge[2] + 2 * dq * (1 / 6.0);
    ge[1] = ge[1] + 2 * dq * (4 / 6.0);
    ge[0] = ge[0] + 2 * dq * (1 / 6.0);
    if (pb->end_n >= 2) {
      dq = 1 / (2 * dt) * (q_1 - q_3) - mk(es[3], es[4], es[5]);
      c += sqnorm(dq);
      ge[2] = ge[2] + 2 * dq * 1.0 / (2 * dt);
      ge[0] = ge[0] + 2 * dq * (-1.0) / (2 * dt);
      if (opt_time) gt += dot(dq, q_1 - q_3) / (-dt * dt);
    }
    if (pb->end_n == 3) {
      dq = 1 / (dt * dt) * (q_1 - 2 * q_2 + q_3) - mk(es[6], es[7], es[8]);
      c += sqnorm(dq);
      ge[2] = ge[2] + 2 * dq * 1.0 / (dt * dt);
      ge[1] = ge[1] + 2 * dq * (-2.0) / (dt * dt);
      ge[0] = ge[0] + 2 * dq * 1.0 / (dt * dt);
      if (opt_time) gt += dot(dq, q_1 - 2 * q_2 + q_3) / (-dt * dt * dt);
    }
    f += cfg->ld_end * c;
    for (int i = N - 3; i < N; i++)
      for (int j = 0; j < dim; j++) grad[dim * i + j] += cfg->ld_end * ge[i - (N - 3)][j];
    if (opt_time) grad[nvar - 1] += cfg->ld_end * gt;
  }
  if (pb->cost_function & (1 << 5)) {  // GUIDE
    double c = 0.0;
    zero_g();
    int end_idx = N - order;
    for (int i = order; i < end_idx; i++) {
      const double* gp = pb->guide_pts + 3 * (i - order);
      V3d d = q[i] - mk(gp[0], gp[1], gp[2]);
      c += sqnorm(d);
      g[i] = g[i] + 2 * d;
    }
    f += cfg->ld_guide * c;
    add_all(cfg->ld_guide);
  }
  if (pb->cost_function & (1 << 6)) {  // WAYPOINTS
    double c = 0.0;
    zero_g();
    for (int i = 0; i < pb->n_waypt; ++i) {
      const double* wp = pb->waypoints + 3 * i;
      int idx = pb->waypt_idx[i];
      V3d dq = 1 / 6.0 * (q[idx] + 4 * q[idx + 1] + q[idx + 2]) - mk(wp[0], wp[1], wp[2]);
      c += sqnorm(dq);
      g[idx] = g[idx] + dq * (2.0 / 6.0);
      g[idx + 1] = g[idx + 1] + dq * (8.0 / 6.0);
      g[idx + 2] = g[idx + 2] + dq * (2.0 / 6.0);
    }
    f += cfg->ld_waypt * c;
    add_all(cfg->ld_waypt);
  }
  if (pb->cost_function & (1 << 7)) {  // VIEWCONS
    double c = 0.0;
    zero_g();
    V3d p = mk(pb->view_pt[0], pb->view_pt[1], pb->view_pt[2]);
    V3d dir = mk(pb->view_dir[0], pb->view_dir[1], pb->view_dir[2]);
    V3d v = dir / norm(dir);
    int i = pb->view_idx;
    V3d qp = q[i] - p;
    V3d dn = qp - dot(qp, v) * v;
    c += sqnorm(dn);
    // (I - v v^T) dn
    V3d t = dn - dot(v, dn) * v;
    g[i] = g[i] + 2 * t;
    V3d dl = dot(qp, v) * v;
    double norm_dl = norm(dl);
    double safe = norm(dir);
    if (norm_dl < safe) {
      c += cfg->wnl * std::pow(norm_dl - safe, 2);
      V3d vvT_dl = dot(v, dl) * v;
      g[i] = g[i] + cfg->wnl * 2 * (norm_dl - safe) * vvT_dl / norm_dl;
    }
    f += cfg->ld_view * c;
    add_all(cfg->ld_view);
  }
  if (pb->cost_function & MINTIME) {
    double duration = (N - order) * dt;
    double c = duration;
    double gt = double(N - order);
    if (pb->time_lb > 0 && duration < pb->time_lb) {
      static const double w_lb = 10;
      c += w_lb * std::pow(duration - pb->time_lb, 2);
      gt += w_lb * 2 * (duration - pb->time_lb) * (N - order);
    }
    f += cfg->ld_time * c;
    grad[nvar - 1] += cfg->ld_time * gt;
  }
  *cost_out = f;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// MapROS::proessDepthImage (plan_env/src/map_ros.cpp:176-215): 16-bit depth image -> world points.
// Restated literally, including the pointer quirk: `depth` is read from pixel u, then the row
// pointer advances by skip_pixel and the zero test dereferences THAT pixel (:190-198).  A read past
// the end of the image (undefined in the reference) is defined here as 0.
// Rotation: Eigen's Quaterniond::toRotationMatrix(); pt_world = R * pt_cur + t evaluated row by
// row, left to right; stored as float (pcl::PointXYZ).
// ---------------------------------------------------------------------------------------------
extern "C" int fo_project_depth(const unsigned short* img, int rows, int cols, const fo_depth_cfg* c,
                                const double pos[3], const double q[4], float* xyz, int cap) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  const double R[3][3] = {{1.0 - (tyy + tzz), txy - twz, txz + twy},
                          {txy + twz, 1.0 - (txx + tzz), tyz - twx},
                          {txz - twy, tyz + twx, 1.0 - (txx + tyy)}};
  const double inv_factor = 1.0 / c->k_depth_scaling_factor;
  const long total = (long)rows * cols;
  int cnt = 0;
  for (int v = c->depth_filter_margin; v < rows - c->depth_filter_margin; v += c->skip_pixel) {
    long at = (long)v * cols + c->depth_filter_margin;  // row_ptr
    for (int u = c->depth_filter_margin; u < cols - c->depth_filter_margin; u += c->skip_pixel) {
      double depth = img[at] * inv_factor;
      at += c->skip_pixel;
      const unsigned short nxt = at < total ? img[at] : (unsigned short)0;
      if (nxt == 0 || depth > c->depth_filter_maxdist)
        depth = c->depth_filter_maxdist;
      else if (depth < c->depth_filter_mindist)
        continue;
      const double pc[3] = {(u - c->cx) * depth / c->fx, (v - c->cy) * depth / c->fy, depth};
      if (cnt < cap)
        for (int i = 0; i < 3; ++i)
          xyz[3 * cnt + i] = (float)(R[i][0] * pc[0] + R[i][1] * pc[1] + R[i][2] * pc[2] + pos[i]);
      ++cnt;
    }
  }
  return cnt;
}

// ---------------------------------------------------------------------------------------------
// BsplineOptimizer::optimize() around a box-projected L-BFGS (see fuel_oracle.h).  Sequential f64.
// ---------------------------------------------------------------------------------------------
extern "C" double fo_bspline_optimize(const fo_map* m, const fo_bspline_cfg* cfg, const fo_bspline_problem* pb,
                                      double* x_io, int max_eval, int* evals_out) {
  const int N = pb->point_num, dim = pb->dim;
  const bool opt_time = (pb->cost_function & (1 << 8)) != 0;
  const int n = opt_time ? dim * N + 1 : dim * N, npt = dim * N;
  const int MEM = 8;
  double blo[3], bhi[3];
  for (int k = 0; k < 3; ++k) blo[k] = m->box_mind[k] + 0.1, bhi[k] = m->box_maxd[k] - 0.1;  // :174-178
  std::vector<double> q(n), lb(n, -1e300), ub(n, 1e300);
  for (int i = 0; i < n; ++i) {
    double v = x_io[i];
    if (dim != 1 && i < npt) v = std::max(std::min(v, bhi[i % 3]), blo[i % 3]);  // :194-199
    q[i] = v;
  }
  if (dim != 1) {
    for (int i = 0; i < npt; ++i) lb[i] = std::max(q[i] - 10.0, blo[i % 3]), ub[i] = std::min(q[i] + 10.0, bhi[i % 3]);
    if (opt_time) lb[n - 1] = 0.0, ub[n - 1] = 5.0;
  }
  int evals = 0;
  std::vector<double> best = q, g(n), xn(n), gn(n), d(n);
  double f, fbest;
  fo_bspline_cost_grad(m, cfg, pb, q.data(), &f, g.data());
  ++evals;
  fbest = f;
  std::vector<std::vector<double>> S(MEM, std::vector<double>(n)), Y(MEM, std::vector<double>(n));
  double rho[8], al[8];
  int hist = 0, head = 0;
  auto dot = [&](const double* a, const double* b) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
  };
  while (evals < max_eval) {
    for (int i = 0; i < n; ++i) {
      const bool at_lb = q[i] <= lb[i] && g[i] > 0, at_ub = q[i] >= ub[i] && g[i] < 0;
      d[i] = (at_lb || at_ub) ? 0.0 : -g[i];
    }
    for (int k = hist - 1; k >= 0; --k) {
      const int sl = (head + k) % MEM;
      const double a = dot(S[sl].data(), d.data()) * rho[sl];
      al[sl] = a;
      for (int i = 0; i < n; ++i) d[i] -= a * Y[sl][i];
    }
    if (hist > 0) {
      const int sl = (head + hist - 1) % MEM;
      const double yy = dot(Y[sl].data(), Y[sl].data()), sy = 1.0 / rho[sl];
      const double gamma = yy > 0 ? sy / yy : 1.0;
      for (int i = 0; i < n; ++i) d[i] *= gamma;
    }
    for (int k = 0; k < hist; ++k) {
      const int sl = (head + k) % MEM;
      const double b = dot(Y[sl].data(), d.data()) * rho[sl];
      for (int i = 0; i < n; ++i) d[i] += S[sl][i] * (al[sl] - b);
    }
    double gd = dot(g.data(), d.data());
    if (!(gd < 0)) {
      hist = 0, head = 0;
      for (int i = 0; i < n; ++i) d[i] = -g[i];
      gd = -dot(g.data(), g.data());
      if (gd == 0) break;
    }
    double step = hist == 0 ? 1.0 / std::max(1.0, std::sqrt(-gd)) : 1.0, fn = f;
    bool ok = false;
    for (int ls = 0; ls < 20 && evals < max_eval; ++ls) {
      for (int i = 0; i < n; ++i) xn[i] = std::min(std::max(q[i] + step * d[i], lb[i]), ub[i]);
      fo_bspline_cost_grad(m, cfg, pb, xn.data(), &fn, gn.data());
      ++evals;
      if (fn < fbest) fbest = fn, best = xn;
      double dec = 0;
      for (int i = 0; i < n; ++i) dec += g[i] * (xn[i] - q[i]);
      if (fn <= f + 1e-4 * dec) {
        ok = true;
        break;
      }
      step *= 0.5;
    }
    if (!ok) break;
    double sy = 0, ss = 0, xx = 0;
    const int slot = hist < MEM ? (head + hist) % MEM : head;
    for (int i = 0; i < n; ++i) {
      const double si = xn[i] - q[i], yi = gn[i] - g[i];
      sy += si * yi, ss += si * si, xx += xn[i] * xn[i];
    }
    if (sy > 1e-12) {
      for (int i = 0; i < n; ++i) S[slot][i] = xn[i] - q[i], Y[slot][i] = gn[i] - g[i];
      rho[slot] = 1.0 / sy;
      if (hist < MEM)
        ++hist;
      else
        head = (head + 1) % MEM;
    }
    q = xn, g = gn, f = fn;
    if (std::sqrt(ss) <= 1e-5 * std::sqrt(xx)) break;
  }
  for (int i = 0; i < n; ++i) x_io[i] = best[i];
  if (evals_out) *evals_out = evals;
  return fbest;
}

// ---------------------------------------------------------------------------------------------
// NonUniformBspline glue (fuel_planner/bspline/src/non_uniform_bspline.cpp)
// ---------------------------------------------------------------------------------------------
namespace {
// Least squares by Householder QR with column pivoting -- the published algorithm behind Eigen's
// ColPivHouseholderQR::solve (Eigen is third party and absent; Golub & Van Loan, Alg. 5.4.1): at every
// step the remaining column of largest norm is moved to the front, a reflector zeroes it below the
// diagonal and is applied to the rest and to b; then R x = Q^T b by back substitution, un-permuted.
std::vector<double> lstsq_colpiv_qr(std::vector<double> A, int rows, int cols, std::vector<double> b) {
  std::vector<int> perm(cols);
  for (int j = 0; j < cols; ++j) perm[j] = j;
  auto at = [&](int i, int j) -> double& { return A[(size_t)i * cols + j]; };
  for (int k = 0; k < cols; ++k) {
    int best = k;
    double bestn = -1.0;
    for (int j = k; j < cols; ++j) {
      double s = 0.0;
      for (int i = k; i < rows; ++i) s += at(i, j) * at(i, j);
      if (s > bestn) bestn = s, best = j;
    }
    if (best != k) {
      for (int i = 0; i < rows; ++i) std::swap(at(i, k), at(i, best));
      std::swap(perm[k], perm[best]);
    }
    double norm = std::sqrt(bestn);
    if (norm == 0.0) continue;
    const double alpha = at(k, k) > 0 ? -norm : norm;
    std::vector<double> v(rows - k);
    for (int i = k; i < rows; ++i) v[i - k] = at(i, k);
    v[0] -= alpha;
    double vv = 0.0;
    for (double e : v) vv += e * e;
    if (vv == 0.0) continue;
    for (int j = k; j < cols; ++j) {
      double dot = 0.0;
      for (int i = k; i < rows; ++i) dot += v[i - k] * at(i, j);
      const double f = 2.0 * dot / vv;
      for (int i = k; i < rows; ++i) at(i, j) -= f * v[i - k];
    }
    double dot = 0.0;
    for (int i = k; i < rows; ++i) dot += v[i - k] * b[i];
    const double f = 2.0 * dot / vv;
    for (int i = k; i < rows; ++i) b[i] -= f * v[i - k];
  }
  std::vector<double> y(cols), x(cols);
  for (int j = cols - 1; j >= 0; --j) {
    double s = b[j];
    for (int k = j + 1; k < cols; ++k) s -= at(j, k) * y[k];
    y[j] = s / at(j, j);
  }
  for (int j = 0; j < cols; ++j) x[perm[j]] = y[j];
  return x;
}

// a spline as NonUniformBspline holds it: control points (rows), degree p_, knots u_
struct OSpline {
  std::vector<std::array<double, 3>> cp;
  int p = 0;
  std::vector<double> u;
  int n_() const { return (int)cp.size() - 1; }
  int m_() const { return n_() + p + 1; }
};
// setUniformBspline (:15-32)
OSpline spline_uniform(const double* ctrl, int n, int degree, double interval) {
  OSpline s;
  s.cp.resize(n);
  for (int i = 0; i < n; ++i) s.cp[i] = {ctrl[3 * i], ctrl[3 * i + 1], ctrl[3 * i + 2]};
  s.p = degree;
  const int m = s.m_();
  s.u.assign(m + 1, 0.0);
  for (int i = 0; i <= m; ++i) s.u[i] = i <= degree ? double(-degree + i) * interval : s.u[i - 1] + interval;
  return s;
}
// evaluateDeBoor (:51-71)
std::array<double, 3> spline_deboor(const OSpline& s, double uu) {
  const int p = s.p;
  const double ub = std::min(std::max(s.u[p], uu), s.u[s.m_() - p]);
  int k = p;
  while (s.u[k + 1] < ub) ++k;
  std::vector<std::array<double, 3>> d;
  for (int i = 0; i <= p; ++i) d.push_back(s.cp[k - p + i]);
  for (int r = 1; r <= p; ++r)
    for (int i = p; i >= r; --i) {
      const double alpha = (ub - s.u[i + k - p]) / (s.u[i + 1 + k - r] - s.u[i + k - p]);
      for (int a = 0; a < 3; ++a) d[i][a] = (1 - alpha) * d[i - 1][a] + alpha * d[i][a];
    }
  return d[p];
}
// getDerivative (:89-104) with getDerivativeControlPoints (:77-87)
OSpline spline_derivative(const OSpline& s) {
  OSpline d;
  d.p = s.p - 1;
  d.cp.resize(s.cp.size() - 1);
  for (size_t i = 0; i < d.cp.size(); ++i)
    for (int a = 0; a < 3; ++a)
      d.cp[i][a] = s.p * (s.cp[i + 1][a] - s.cp[i][a]) / (s.u[i + s.p + 1] - s.u[i + 1]);
  d.u.assign(s.u.begin() + 1, s.u.end() - 1);
  return d;
}
}  // namespace

// parameterizeToBspline (:178-265); ctrl: (K + degree - 1) x 3.  Returns 0, or -1 on the inputs the
// reference refuses (ts <= 0, fewer than 2 points).
extern "C" int fo_spline_parameterize(double ts, const double* pts, int K, const double* derivs4, int degree,
                                      double* ctrl) {
  if (ts <= 0 || K < 2 || degree < 3 || degree > 5) return -1;
  const int rows = K + 4, cols = K + degree - 1;
  std::vector<double> A((size_t)rows * cols, 0.0);
  double pp[5], pv[5], pa[5];
  if (degree == 3) {
    const double c3[3] = {1, 4, 1}, v3[3] = {-1, 0, 1}, a3[3] = {1, -2, 1};
    for (int i = 0; i < 3; ++i) pp[i] = 1 / 6.0 * c3[i], pv[i] = 1 / (2 * ts) * v3[i], pa[i] = 1 / (ts * ts) * a3[i];
  } else if (degree == 4) {
    const double c4[4] = {1, 11, 11, 1}, v4[4] = {-1, -3, 3, 1}, a4[4] = {1, -1, -1, 1};
    for (int i = 0; i < 4; ++i)
      pp[i] = 1 / 24.0 * c4[i], pv[i] = 1 / (6 * ts) * v4[i], pa[i] = 1 / (2 * ts * ts) * a4[i];
  } else {
    const double c5[5] = {1, 26, 66, 26, 1}, v5[5] = {-1, -10, 0, 10, 1}, a5[5] = {1, 2, -6, 2, 1};
    for (int i = 0; i < 5; ++i) pp[i] = c5[i] / 120.0, pv[i] = v5[i] / (24 * ts), pa[i] = a5[i] / (6 * ts * ts);
  }
  for (int i = 0; i < K; ++i)
    for (int k = 0; k < degree; ++k) A[(size_t)i * cols + i + k] = pp[k];
  for (int k = 0; k < degree; ++k) {
    A[(size_t)K * cols + k] = pv[k];
    A[(size_t)(K + 1) * cols + K - 1 + k] = pv[k];
    A[(size_t)(K + 2) * cols + k] = pa[k];
    A[(size_t)(K + 3) * cols + K - 1 + k] = pa[k];
  }
  for (int a = 0; a < 3; ++a) {
    std::vector<double> b(rows);
    for (int i = 0; i < K; ++i) b[i] = pts[3 * i + a];
    for (int i = 0; i < 4; ++i) b[K + i] = derivs4[3 * i + a];
    const std::vector<double> x = lstsq_colpiv_qr(A, rows, cols, b);
    for (int j = 0; j < cols; ++j) ctrl[3 * j + a] = x[j];
  }
  return 0;
}

// getBoundaryStates(ks, ke) (:107-122) of the uniform spline (ctrl, degree, ts);
// start: (ks + 1) x 3, end: (ke + 1) x 3
extern "C" void fo_spline_boundary_states(const double* ctrl, int n, int degree, double ts, int ks, int ke,
                                          double* start, double* end) {
  const OSpline s = spline_uniform(ctrl, n, degree, ts);
  std::vector<OSpline> ders;  // computeDerivatives(max(ks, ke)) (:89-97)
  const int kd = std::max(ks, ke);
  if (kd >= 1) ders.push_back(spline_derivative(s));
  for (int i = 2; i <= kd; ++i) ders.push_back(spline_derivative(ders.back()));
  const double duration = s.u[s.m_() - s.p] - s.u[s.p];  // getTimeSum (:267-269)
  auto at = [](const OSpline& sp, double t) { return spline_deboor(sp, t + sp.u[sp.p]); };  // evaluateDeBoorT
  auto put = [](double* dst, const std::array<double, 3>& v) { dst[0] = v[0], dst[1] = v[1], dst[2] = v[2]; };
  put(start, at(s, 0.0));
  for (int i = 0; i < ks; ++i) put(start + 3 * (i + 1), at(ders[i], 0.0));
  put(end, at(s, duration));
  for (int i = 0; i < ke; ++i) put(end + 3 * (i + 1), at(ders[i], duration));
}
