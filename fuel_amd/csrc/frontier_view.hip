// frontier_view.hip -- viewpoint sampling and coverage of frontier clusters on the device:
// FrontierFinder::computeFrontiersToVisit / sampleViewpoints / countVisibleCells / isNearUnknown /
// isFrontierCovered (active_perception/src/frontier_finder.cpp:392-423, 662-755, 697-719) and
// PerceptionUtils::setPose / insideFOV (active_perception/src/perception_utils.cpp:49-69, 84-93).
//
// The reference visits, per new cluster, candidate_rnum+1 circles x 2*pi/candidate_dphi angles around
// the cluster mean; a candidate must lie strictly inside the exploration box, in a non-inflated
// voxel, with no UNKNOWN voxel in a (2v+1)x(2v+1)x3 block around it; its yaw is the mean bearing of
// the down-sampled cells and its score the number of those cells inside the camera frustum whose
// ray back to the candidate crosses no inflated / unknown voxel.  That is (clusters x ~100
// candidates x ~100 cells) short ray walks over the same bit-planes the map already keeps on the
// device: one wavefront per (cluster, candidate), lanes over the cells.
//
// Arithmetic: f64 with the reference's expressions (sqrt / division are IEEE on gfx950; compiled with
// -ffp-contract=off).  The candidate offsets rc*cos(phi), rc*sin(phi) are tabulated on the host with
// the reference's own loops; acos / atan2 / sin / cos on the device may differ from glibc in the
// last ulp and the bearing mean is a tree sum, so yaws agree to ~1e-15 rad, not bit for bit.
#include <cmath>
#include <cstdlib>

#include "frontier_internal.h"

namespace {

struct VArgs {
  const u64* occ;   // occupied plane
  const u64* unk;   // unknown plane
  const u64* infl;  // inflated plane
  const double* avg;     // [ncl][3]
  const u32* foff;       // [ncl + 1] filtered-cell offsets
  const float* fxyz;     // filtered cells
  const double* soff;    // [ns][2] candidate offsets (rc*cos(phi), rc*sin(phi))
  int ncl, ns;
  double box_mind[3], box_maxd[3];
  int vox_num;  // floor(min_candidate_clearance / resolution)
  double max_dist;
  double ncam[4][3];  // frustum normals in the camera frame (top, bottom, left, right)
  double* out;        // [ncl*ns][5]: valid, x, y, yaw, visib
};

__device__ __forceinline__ bool idx_in_map(const Geo& g, const int id[3]) {
  return !(id[0] < 0 || id[1] < 0 || id[2] < 0 || id[0] > g.nx - 1 || id[1] > g.ny - 1 || id[2] > g.nz - 1);
}
__device__ __forceinline__ void pos_to_idx(const Geo& g, const double p[3], int id[3]) {
  for (int k = 0; k < 3; ++k) id[k] = (int)floor((p[k] - g.org[k]) * g.res_inv);
}
__device__ __forceinline__ bool bit_at(const u64* pl, long a) { return (pl[a >> 6] >> (a & 63)) & 1ull; }

// RayCaster::input + nextId loop of countVisibleCells (:741-751): true iff no visited voxel is
// inflated or unknown; the walk starts in the cell's voxel and stops before the candidate's voxel
__device__ bool ray_clear(const Geo& g, const VArgs& V, const double start[3], const double end[3]) {
  int c[3], ec[3], st[3];
  double tmax[3], tdel[3];
  for (int k = 0; k < 3; ++k) {
    const double s = start[k] / g.res, e = end[k] / g.res;
    c[k] = (int)floor(s);
    ec[k] = (int)floor(e);
    const double d = ec[k] - c[k];
    const int di = (int)d;
    st[k] = di == 0 ? 0 : (di < 0 ? -1 : 1);
    // intbound(s, d) (raycast.cpp:14-23)
    double ss = s, ds = d;
    if (ds < 0) {
      ss = -ss;
      ds = -ds;
    }
    ss = fmod(fmod(ss, 1.0) + 1.0, 1.0);
    tmax[k] = (1 - ss) / ds;
    tdel[k] = ((double)st[k]) / d;
  }
  const double off[3] = {0.5 - g.org[0] / g.res, 0.5 - g.org[1] / g.res, 0.5 - g.org[2] / g.res};
  int guard = abs(ec[0] - c[0]) + abs(ec[1] - c[1]) + abs(ec[2] - c[2]) + 4;
  while (true) {
    const int id[3] = {(int)((double)c[0] + off[0]), (int)((double)c[1] + off[1]), (int)((double)c[2] + off[2])};
    if (c[0] == ec[0] && c[1] == ec[1] && c[2] == ec[2]) return true;
    if (idx_in_map(g, id)) {
      const long a = (long)id[0] * g.nyz + (long)id[1] * g.nz + id[2];
      if (bit_at(V.infl, a) || bit_at(V.unk, a)) return false;
    }
    if (tmax[0] < tmax[1]) {
      if (tmax[0] < tmax[2]) {
        c[0] += st[0];
        tmax[0] += tdel[0];
      } else {
        c[2] += st[2];
        tmax[2] += tdel[2];
      }
    } else {
      if (tmax[1] < tmax[2]) {
        c[1] += st[1];
        tmax[1] += tdel[1];
      } else {
        c[2] += st[2];
        tmax[2] += tdel[2];
      }
    }
    if (--guard < 0) return true;  // (the reference has no guard; unreachable for finite inputs)
  }
}

__global__ void __launch_bounds__(64) k_vp_sample(Geo g, VArgs V) {
  const int job = blockIdx.x;
  const int c = job / V.ns, s = job - c * V.ns;
  const int lane = threadIdx.x;
  double* o = V.out + (size_t)job * 5;
  const double pos[3] = {V.avg[3 * c] + V.soff[2 * s], V.avg[3 * c + 1] + V.soff[2 * s + 1], V.avg[3 * c + 2] + 0.0};
  // ---- qualification (:670-673) ----
  bool ok = true;
  for (int k = 0; k < 3; ++k)
    if (pos[k] <= V.box_mind[k] || pos[k] >= V.box_maxd[k]) ok = false;
  if (ok) {
    int id[3];
    pos_to_idx(g, pos, id);
    if (idx_in_map(g, id) && bit_at(V.infl, (long)id[0] * g.nyz + (long)id[1] * g.nz + id[2])) ok = false;
  }
  if (ok) {  // isNearUnknown (:721-731)
    const int w = 2 * V.vox_num + 1, total = w * w * 3;
    bool near = false;
    for (int t = lane; t < total; t += 64) {
      const int z = t % 3 - 1, r = t / 3, y = r % w - V.vox_num, x = r / w - V.vox_num;
      const double vox[3] = {pos[0] + x * g.res, pos[1] + y * g.res, pos[2] + z * g.res};
      int id[3];
      pos_to_idx(g, vox, id);
      if (idx_in_map(g, id) && bit_at(V.unk, (long)id[0] * g.nyz + (long)id[1] * g.nz + id[2])) near = true;
    }
    if (__ballot(near)) ok = false;
  }
  if (!ok) {
    if (lane == 0) o[0] = 0.0;
    return;
  }
  // ---- mean bearing of the down-sampled cells (:676-686) ----
  const u32 f0 = V.foff[c], nf = V.foff[c + 1] - f0;
  auto cell = [&](u32 i, double p[3]) {
    p[0] = (double)V.fxyz[3 * (size_t)(f0 + i)];
    p[1] = (double)V.fxyz[3 * (size_t)(f0 + i) + 1];
    p[2] = (double)V.fxyz[3 * (size_t)(f0 + i) + 2];
  };
  double ref[3];
  {
    double p[3];
    cell(0, p);
    const double d[3] = {p[0] - pos[0], p[1] - pos[1], p[2] - pos[2]};
    const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    ref[0] = d[0] / n, ref[1] = d[1] / n, ref[2] = d[2] / n;
  }
  double acc = 0.0;
  for (u32 i = 1 + lane; i < nf; i += 64) {
    double p[3];
    cell(i, p);
    const double d[3] = {p[0] - pos[0], p[1] - pos[1], p[2] - pos[2]};
    const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double dir[3] = {d[0] / n, d[1] / n, d[2] / n};
    double yaw = acos(dir[0] * ref[0] + dir[1] * ref[1] + dir[2] * ref[2]);
    if (ref[0] * dir[1] - ref[1] * dir[0] < 0) yaw = -yaw;
    acc += yaw;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  double yaw = acc / (double)nf + atan2(ref[1], ref[0]);
  while (yaw < -M_PI) yaw += 2 * M_PI;
  while (yaw > M_PI) yaw -= 2 * M_PI;
  // ---- frustum normals in the world (setPose, :49-69): R_wc = [s 0 c; -c 0 s; 0 1 0] ----
  const double cy = cos(yaw), sy = sin(yaw);
  const double R[3][3] = {{sy, 0.0, cy}, {-cy, 0.0, sy}, {0.0, 1.0, 0.0}};
  double nw[4][3];
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 3; ++i) nw[k][i] = R[i][0] * V.ncam[k][0] + R[i][1] * V.ncam[k][1] + R[i][2] * V.ncam[k][2];
  // ---- countVisibleCells (:733-755) ----
  int cnt = 0;
  for (u32 i0 = 0; i0 < nf; i0 += 64) {
    const u32 i = i0 + lane;
    bool vis = false;
    if (i < nf) {
      double p[3];
      cell(i, p);
      double d[3] = {p[0] - pos[0], p[1] - pos[1], p[2] - pos[2]};
      const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      bool in = !(n > V.max_dist);
      if (in) {
        d[0] /= n, d[1] /= n, d[2] /= n;
        for (int k = 0; k < 4; ++k)
          if (d[0] * nw[k][0] + d[1] * nw[k][1] + d[2] * nw[k][2] < 0.0) in = false;
      }
      if (in) vis = ray_clear(g, V, p, pos);
    }
    cnt += __popcll(__ballot(vis));
  }
  if (lane == 0) {
    o[0] = 1.0;
    o[1] = pos[0];
    o[2] = pos[1];
    o[3] = yaw;
    o[4] = (double)cnt;
  }
}

// isFrontierCovered's per-cluster count of cells that stopped being frontier cells; the cells are read
// from the finder's device pool (cand_off[k] = pool offset of candidate k, cand_start[k] = its first flat
// index)
__global__ void k_vp_changed(Geo g, const u64* __restrict__ occ, const u64* __restrict__ unk,
                             const u32* __restrict__ pool, const u64* __restrict__ cand_off,
                             const u32* __restrict__ cand_start, int ncand, u32 total, u32* __restrict__ changed) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = ncand - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cand_start[mid] <= i)
      lo = mid;
    else
      hi = mid - 1;
  }
  const long a = pool[cand_off[lo] + (i - cand_start[lo])];
  auto bit = [&](const u64* p, long q) { return (p[q >> 6] >> (q & 63)) & 1ull; };
  bool f1 = !(bit(occ, a) || bit(unk, a));
  if (f1) {
    const int x = (int)(a / g.nyz), r = (int)(a - (long)x * g.nyz), y = r / g.nz, z = r - y * g.nz;
    f1 = (x > 0 && bit(unk, a - g.nyz)) || (x < g.nx - 1 && bit(unk, a + g.nyz)) || (y > 0 && bit(unk, a - g.nz)) ||
         (y < g.ny - 1 && bit(unk, a + g.nz)) || (z > 0 && bit(unk, a - 1)) || (z < g.nz - 1 && bit(unk, a + 1));
  }
  if (!f1) atomicAdd(&changed[lo], 1u);
}

}  // namespace

static int view_stage(fuelmi_frontier* f, size_t bytes, unsigned char** p) {
  if (bytes > f->d_stage_bytes) {
    if (f->d_stage) HIPCHK(hipFree(f->d_stage));
    f->d_stage = nullptr;
    f->d_stage_bytes = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    HIPCHK(hipMalloc(&f->d_stage, want));
    f->d_stage_bytes = want;
  }
  *p = reinterpret_cast<unsigned char*>(f->d_stage);
  return FUELMI_OK;
}

extern "C" int fuelmi_frontier_set_viewpoint_cfg(fuelmi_frontier* f, const fuelmi_viewpoint_cfg* c) {
  ARGCHK(f && c);
  // rmax == rmin would make the reference's radius loop (:664-665, `rc += dr` with dr = 0) spin forever
  ARGCHK(c->candidate_rnum > 0 && c->candidate_dphi > 0.0 && c->candidate_rmax > c->candidate_rmin);
  f->vcfg = *c;
  f->have_vcfg = true;
  return FUELMI_OK;
}

// sampleViewpoints for every cluster of `tmp`; fills HCluster::viewpoints in the reference's candidate
// order (radius outer loop, angle inner loop)
static int sample_viewpoints(fuelmi_frontier* f, std::list<HCluster>& L) {
  fuelmi_map* m = f->map;
  const Geo& g = m->g;
  const fuelmi_viewpoint_cfg& c = f->vcfg;
  // candidate offsets with the reference's own loops (:664-667)
  std::vector<double> soff;
  for (double rc = c.candidate_rmin, dr = (c.candidate_rmax - c.candidate_rmin) / c.candidate_rnum;
       rc <= c.candidate_rmax + 1e-3; rc += dr)
    for (double phi = -M_PI; phi < M_PI; phi += c.candidate_dphi) {
      soff.push_back(rc * std::cos(phi));
      soff.push_back(rc * std::sin(phi));
    }
  const int ns = (int)(soff.size() / 2), ncl = (int)L.size();
  if (ns == 0 || ncl == 0) return FUELMI_OK;
  std::vector<double> avg(3 * (size_t)ncl);
  std::vector<u32> foff(ncl + 1, 0u);
  size_t nf = 0;
  int k = 0;
  for (HCluster& cl : L) {
    if (cl.filtered.empty()) {
      fuelmi_set_error("viewpoint sampling needs filtered cells: create the frontier finder with cfg.split != 0");
      return FUELMI_EINVAL;
    }
    for (int q = 0; q < 3; ++q) avg[3 * (size_t)k + q] = cl.avg[q];
    foff[k] = (u32)nf;
    nf += cl.filtered.size() / 3;
    ++k;
  }
  foff[ncl] = (u32)nf;
  std::vector<float> fx(3 * nf);
  k = 0;
  for (HCluster& cl : L) memcpy(fx.data() + 3 * (size_t)foff[k++], cl.filtered.data(), cl.filtered.size() * sizeof(float));

  const size_t b_avg = avg.size() * 8, b_soff = soff.size() * 8, b_foff = ((foff.size() * 4 + 7) / 8) * 8,
               b_fx = ((fx.size() * 4 + 7) / 8) * 8, b_out = (size_t)ncl * ns * 5 * 8;
  unsigned char* d;
  int rc = view_stage(f, b_avg + b_soff + b_foff + b_fx + b_out, &d);
  if (rc) return rc;
  hipStream_t st = f->stream;
  // the inflated plane is written on the map's stream: order after everything queued there
  HIPCHK(hipEventRecord(f->ev_dep, m->stream));
  HIPCHK(hipStreamWaitEvent(st, f->ev_dep, 0));
  VArgs V;
  V.occ = m->occ_bits.p, V.unk = m->unk_bits.p, V.infl = m->infl_bits.p;
  unsigned char* p = d;
  V.avg = reinterpret_cast<const double*>(p);
  HIPCHK(hipMemcpyAsync(p, avg.data(), b_avg, hipMemcpyHostToDevice, st));
  p += b_avg;
  V.soff = reinterpret_cast<const double*>(p);
  HIPCHK(hipMemcpyAsync(p, soff.data(), b_soff, hipMemcpyHostToDevice, st));
  p += b_soff;
  V.foff = reinterpret_cast<const u32*>(p);
  HIPCHK(hipMemcpyAsync(p, foff.data(), foff.size() * 4, hipMemcpyHostToDevice, st));
  p += b_foff;
  V.fxyz = reinterpret_cast<const float*>(p);
  HIPCHK(hipMemcpyAsync(p, fx.data(), fx.size() * 4, hipMemcpyHostToDevice, st));
  p += b_fx;
  V.out = reinterpret_cast<double*>(p);
  V.ncl = ncl, V.ns = ns;
  for (int q = 0; q < 3; ++q) V.box_mind[q] = m->cfg.box_min[q], V.box_maxd[q] = m->cfg.box_max[q];
  V.vox_num = (int)std::floor(c.min_candidate_clearance / m->cfg.resolution);
  V.max_dist = c.max_dist;
  const double hp = M_PI_2;  // perception_utils.cpp:13-17
  const double nc[4][3] = {{0.0, std::sin(hp - c.top_angle), std::cos(hp - c.top_angle)},
                           {0.0, -std::sin(hp - c.top_angle), std::cos(hp - c.top_angle)},
                           {std::sin(hp - c.left_angle), 0.0, std::cos(hp - c.left_angle)},
                           {-std::sin(hp - c.right_angle), 0.0, std::cos(hp - c.right_angle)}};
  memcpy(V.ncam, nc, sizeof(nc));
  k_vp_sample<<<ncl * ns, 64, 0, st>>>(g, V);
  HIPCHK(hipGetLastError());
  std::vector<double> out((size_t)ncl * ns * 5);
  HIPCHK(hipMemcpyAsync(out.data(), V.out, b_out, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  k = 0;
  for (HCluster& cl : L) {
    cl.viewpoints.clear();
    for (int s = 0; s < ns; ++s) {
      const double* o = &out[((size_t)k * ns + s) * 5];
      if (o[0] == 0.0) continue;
      const int visib = (int)o[4];
      if (visib > c.min_visib_num) cl.viewpoints.push_back(HViewpoint{{o[1], o[2], cl.avg[2] + 0.0}, o[3], visib});
    }
    ++k;
  }
  return FUELMI_OK;
}

// computeFrontiersToVisit (:392-423)
extern "C" int fuelmi_frontier_compute_to_visit(fuelmi_frontier* f, int* n_active_new, int* n_dormant_new) {
  ARGCHK(f);
  FRONTIER_NOT_SEARCHING(f, "fuelmi_frontier_compute_to_visit");
  if (!f->have_vcfg) {
    fuelmi_set_error("fuelmi_frontier_compute_to_visit: no viewpoint configuration set");
    return FUELMI_EINVAL;
  }
  HIPCHK(hipSetDevice(f->map->device));
  int rc = sample_viewpoints(f, f->tmp);
  if (rc) return rc;
  int na = 0, nd = 0;
  rc = frontier_keep_clusters(f, f->tmp);
  if (rc) return rc;
  while (!f->tmp.empty()) {
    HCluster& c = f->tmp.front();
    if (!c.viewpoints.empty()) {
      // sort by coverage, best first -- std::sort with the reference's comparator (:403-405)
      std::sort(c.viewpoints.begin(), c.viewpoints.end(),
                [](const HViewpoint& a, const HViewpoint& b) { return a.visib_num > b.visib_num; });
      f->frontiers.splice(f->frontiers.end(), f->tmp, f->tmp.begin());
      ++na;
    } else {
      f->dormant.splice(f->dormant.end(), f->tmp, f->tmp.begin());
      ++nd;
    }
  }
  if (n_active_new) *n_active_new = na;
  if (n_dormant_new) *n_dormant_new = nd;
  return FUELMI_OK;
}

static const HCluster* view_nth(const fuelmi_frontier* f, int which, int k) {
  const std::list<HCluster>* L = which == 0 ? &f->tmp : (which == 1 ? &f->frontiers : (which == 2 ? &f->dormant : nullptr));
  if (!L || k < 0 || k >= (int)L->size()) return nullptr;
  auto it = L->begin();
  std::advance(it, k);
  return &*it;
}
extern "C" int fuelmi_frontier_viewpoint_count(const fuelmi_frontier* f, int which, int k) {
  ARGCHK(f);
  const HCluster* c = view_nth(f, which, k);
  ARGCHK(c);
  return (int)c->viewpoints.size();
}
extern "C" int fuelmi_frontier_viewpoints(const fuelmi_frontier* f, int which, int k, double* pos_yaw4, int* visib) {
  ARGCHK(f && pos_yaw4 && visib);
  const HCluster* c = view_nth(f, which, k);
  ARGCHK(c);
  for (size_t i = 0; i < c->viewpoints.size(); ++i) {
    for (int q = 0; q < 3; ++q) pos_yaw4[4 * i + q] = c->viewpoints[i].pos[q];
    pos_yaw4[4 * i + 3] = c->viewpoints[i].yaw;
    visib[i] = c->viewpoints[i].visib_num;
  }
  return FUELMI_OK;
}

// isFrontierCovered (:697-719) against the map's accumulated updated box (not consumed)
extern "C" int fuelmi_frontier_is_covered(fuelmi_frontier* f, int* covered) {
  ARGCHK(f && covered);
  FRONTIER_NOT_SEARCHING(f, "fuelmi_frontier_is_covered");
  *covered = 0;
  if (!f->have_vcfg) {
    fuelmi_set_error("fuelmi_frontier_is_covered: no viewpoint configuration set");
    return FUELMI_EINVAL;
  }
  fuelmi_map* m = f->map;
  HIPCHK(hipSetDevice(m->device));
  double umin[3], umax[3];
  fuelmi_map_get_updated_box(m, umin, umax, 0);
  std::vector<const HCluster*> cand;
  size_t ncell = 0;
  for (const std::list<HCluster>* L : {&f->frontiers, &f->dormant})
    for (const HCluster& c : *L) {
      bool ov = true;  // haveOverlap (:353-363)
      for (int i = 0; i < 3; ++i)
        if (std::max(c.bmin[i], umin[i]) > std::min(c.bmax[i], umax[i]) + 1e-3) ov = false;
      if (!ov) continue;
      cand.push_back(&c);
      ncell += c.size();
    }
  if (cand.empty()) return FUELMI_OK;
  const size_t nc = cand.size();
  std::vector<u64> off(nc);
  std::vector<u32> start(nc);
  u32 total = 0;
  for (size_t k = 0; k < nc; ++k) {
    off[k] = cand[k]->pool_off;
    start[k] = total;
    total += (u32)cand[k]->size();
  }
  (void)ncell;
  unsigned char* d;
  const size_t b_off = nc * sizeof(u64), b_start = ((nc * sizeof(u32) + 7) / 8) * 8;
  int rc = view_stage(f, b_off + b_start + nc * sizeof(u32) + 64, &d);
  if (rc) return rc;
  hipStream_t st = f->stream;
  HIPCHK(hipStreamWaitEvent(st, m->ev_planes, 0));
  u64* d_off = reinterpret_cast<u64*>(d);
  u32* d_start = reinterpret_cast<u32*>(d + b_off);
  u32* d_changed = reinterpret_cast<u32*>(d + b_off + b_start);
  HIPCHK(hipMemcpyAsync(d_off, off.data(), b_off, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_start, start.data(), nc * sizeof(u32), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(d_changed, 0, nc * sizeof(u32), st));
  k_vp_changed<<<(int)((total + 255) / 256), 256, 0, st>>>(m->g, m->occ_bits.p, m->unk_bits.p, f->pool, d_off, d_start,
                                                           (int)nc, total, d_changed);
  HIPCHK(hipGetLastError());
  std::vector<u32> changed(nc);
  HIPCHK(hipMemcpyAsync(changed.data(), d_changed, nc * sizeof(u32), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  for (size_t k = 0; k < cand.size(); ++k) {
    // "++change_num >= change_thresh" inside the loop over changed cells: true iff at least one cell
    // changed and the count reaches int(min_view_finish_fraction * cells.size())
    const int thresh = (int)(f->vcfg.min_view_finish_fraction * cand[k]->size());
    if (changed[k] >= 1u && (int)changed[k] >= thresh) {
      *covered = 1;
      break;
    }
  }
  return FUELMI_OK;
}
