// insert.hip -- depth-point fusion (SDFMap::inputPointCloud, plan_env/src/sdf_map.cpp:259-345;
// setCacheOccupancy :243-257; closetPointInMap :347-362; RayCaster plan_env/src/raycast.cpp:323-407).
//
// The reference is sequential: per point classify/clip, count hit|miss at the end voxel, and the
// FIRST point that lands in an end voxel this frame casts a ray back to the camera marking misses;
// afterwards every touched voxel gets one clamped log-odds update (hit wins iff any hit, because
// count_miss_ is set to 1, not incremented).  Order only matters for "first point per end voxel",
// which is the lowest point index -> atomicMin.  Everything else is a set union:
//   A  per point : classify, atomicOr hit/miss bit of the end voxel, atomicMin owner, bbox
//   B  per point : if owner and flag_rayend_ != raycast_num_: walk the ray, atomicOr miss bits
//   C  per word  : touched = hit|miss -> log-odds update in f64 (bit-exact), refresh state planes
// All geometry is f64 with the reference's exact expressions (compiled with -ffp-contract=off).
#include <cfloat>
#include <cstring>
#include <cmath>

#include "fuelmi_internal.h"

struct DepthArgs {
  const unsigned short* img;
  int rows, cols, margin, skip, nu, nslots;
  double fx, fy, cx, cy, maxdist, mindist, inv_factor;
  double R[9], t[3];
  float* out;  // [nslots][4]
  u64* count;  // valid points
};
// one sampled pixel -> world point (MapROS::proessDepthImage, plan_env/src/map_ros.cpp:176-215): slot s = iv*nu + iu
// keeps the reference's point order; false = dropped by the min-distance filter.  f64 in the reference's order,
// result rounded to float like pcl::PointXYZ.  Quirk kept: the depth comes from pixel u, the zero test reads the
// pixel `skip` further (:190-198); past the end of the image it counts as 0.
__device__ __forceinline__ bool project_pixel(const DepthArgs& D, int s, float o[3]) {
  const int iv = s / D.nu, iu = s - iv * D.nu;
  const int v = D.margin + iv * D.skip, u = D.margin + iu * D.skip;
  const long at = (long)v * D.cols + u;
  double depth = (double)D.img[at] * D.inv_factor;
  const long nxt = at + D.skip;
  const unsigned short ztest = nxt < (long)D.rows * D.cols ? D.img[nxt] : (unsigned short)0;
  if (ztest == 0 || depth > D.maxdist)
    depth = D.maxdist;
  else if (depth < D.mindist)
    return false;
  const double c0 = (u - D.cx) * depth / D.fx, c1 = (v - D.cy) * depth / D.fy, c2 = depth;
  o[0] = (float)(D.R[0] * c0 + D.R[1] * c1 + D.R[2] * c2 + D.t[0]);
  o[1] = (float)(D.R[3] * c0 + D.R[4] * c1 + D.R[5] * c2 + D.t[1]);
  o[2] = (float)(D.R[6] * c0 + D.R[7] * c1 + D.R[8] * c2 + D.t[2]);
  return true;
}

struct InsertArgs {
  const unsigned char* pts;  // device copy of the records (from_depth == 0)
  int stride, n;
  int from_depth;            // 1: point i is pixel slot i of D, projected on the fly (no point buffer in between)
  DepthArgs D;
  double cam[3];
  double max_ray;
  signed char num;  // raycast_num_ after increment
  u64* hit;
  u64* miss;
  u32* owner;
  unsigned char* flag_rayend;
  u64* h_out;    // pinned host: [0..5] sortable-encoded min xyz, max xyz of camera + end points, [6] projected
                 // points of the frame, [7] frame stamp (written last)
  u64* head;     // device: [6] projected points (depth front end), re-zeroed here for the next frame
  u64 epoch;
  u64* partial;  // [classify blocks][8] per-block boxes + projected-pixel counts, folded by block 0 of k_insert_raycast
  struct InsRec* rec;  // [n] what k_insert_classify found for every slot (k_insert_raycast does not redo it)
  int nblk;      // classify blocks
};

struct InsRec {
  double pt[3];  // the (clamped) end point
  long a;        // its voxel address, -1: the slot adds nothing
};
__device__ __forceinline__ u64 enc_f64(double d) {
  u64 u = (u64)__double_as_longlong(d);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec_f64_dev(u64 e) {
  const u64 u = (e >> 63) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e;
  return __longlong_as_double((long long)u);
}
static inline double dec_f64(u64 e) {
  u64 u = (e >> 63) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e;
  double d;
  memcpy(&d, &u, sizeof(d));
  return d;
}
__device__ __forceinline__ bool in_map_pos(const Geo& g, const double p[3]) {
  for (int i = 0; i < 3; ++i)
    if (p[i] < g.minb[i] + 1e-4) return false;
  for (int i = 0; i < 3; ++i)
    if (p[i] > g.maxb[i] - 1e-4) return false;
  return true;
}

// classify one point exactly like sdf_map.cpp:276-303; returns false if the point is dropped
__device__ __forceinline__ bool classify(const Geo& g, const InsertArgs& A, int i, double pt[3], int& flag) {
  if (A.from_depth) {
    float q[3];
    if (!project_pixel(A.D, i, q)) return false;  // dropped by the min-distance filter: no point in this slot
    pt[0] = q[0], pt[1] = q[1], pt[2] = q[2];
  } else {
    const float* p = reinterpret_cast<const float*>(A.pts + (size_t)i * A.stride);
    if (isnan(p[0])) return false;
    pt[0] = p[0], pt[1] = p[1], pt[2] = p[2];
  }
  double length;
  if (!in_map_pos(g, pt)) {
    // closetPointInMap (:347-362)
    double diff[3], min_t = 1000000;
    for (int k = 0; k < 3; ++k) diff[k] = pt[k] - A.cam[k];
    for (int k = 0; k < 3; ++k) {
      if (fabs(diff[k]) > 0) {
        double t1 = (g.maxb[k] - A.cam[k]) / diff[k];
        if (t1 > 0 && t1 < min_t) min_t = t1;
        double t2 = (g.minb[k] - A.cam[k]) / diff[k];
        if (t2 > 0 && t2 < min_t) min_t = t2;
      }
    }
    for (int k = 0; k < 3; ++k) pt[k] = A.cam[k] + (min_t - 1e-3) * diff[k];
    double d0 = pt[0] - A.cam[0], d1 = pt[1] - A.cam[1], d2 = pt[2] - A.cam[2];
    length = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    if (length > A.max_ray) {
      pt[0] = d0 / length * A.max_ray + A.cam[0];
      pt[1] = d1 / length * A.max_ray + A.cam[1];
      pt[2] = d2 / length * A.max_ray + A.cam[2];
    }
    if (pt[2] < 0.2) return false;
    flag = 0;
  } else {
    double d0 = pt[0] - A.cam[0], d1 = pt[1] - A.cam[1], d2 = pt[2] - A.cam[2];
    length = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    if (length > A.max_ray) {
      pt[0] = d0 / length * A.max_ray + A.cam[0];
      pt[1] = d1 / length * A.max_ray + A.cam[1];
      pt[2] = d2 / length * A.max_ray + A.cam[2];
      if (pt[2] < 0.2) return false;
      flag = 0;
    } else
      flag = 1;
  }
  return true;
}

__device__ __forceinline__ long pos_adr(const Geo& g, const double p[3]) {
  int id[3];
  for (int k = 0; k < 3; ++k) id[k] = (int)floor((p[k] - g.org[k]) * g.res_inv);
  return (long)id[0] * g.nyz + (long)id[1] * g.nz + id[2];
}

__global__ void __launch_bounds__(256) k_insert_classify(Geo g, InsertArgs A) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  double pt[3];
  int flag = 0;
  bool ok = (i < A.n) && classify(g, A, i, pt, flag);
  u32 n_proj = 0u;  // proj_points_cnt: pixels that survive the projection (whatever the fusion does with them)
  if (A.from_depth) {
    float q[3];
    n_proj = (u32)__popcll(__ballot(i < A.n && project_pixel(A.D, i, q)));
  }
  long a = -1;
  if (ok) {
    a = pos_adr(g, pt);
    if (!(a >= 0 && a < g.N)) ok = false;
  }
  if (i < A.n) {
    InsRec r;
    r.pt[0] = pt[0], r.pt[1] = pt[1], r.pt[2] = pt[2];
    r.a = ok ? a : -1L;
    A.rec[i] = r;
  }
  {
    // neighbouring pixels end in the same voxel more often than not: a lane whose predecessor holds the
    // same (voxel, hit/miss) pair has nothing to add -- same bit, and the predecessor's index is lower
    const int lane = threadIdx.x & 63;
    const long key = ok ? 2 * a + flag : -1L - lane;
    const long prev = __shfl_up(key, 1, 64);
    if (ok && !(lane > 0 && prev == key)) {
      u64 bit = 1ull << (a & 63);
      if (flag)
        atomicOr(&A.hit[a >> 6], bit);
      else
        atomicOr(&A.miss[a >> 6], bit);
      atomicMin(&A.owner[a], (u32)i);
    }
  }
  // bounding box of the kept end points (update_min/max, :303-306): wave reduce, block reduce in LDS,
  // one record per block; the next kernel folds the records (per-wave atomics on six words were
  // ~1200 same-address operations per frame and made this the longest kernel of the fusion)
  __shared__ u64 s_lo[16][3], s_hi[16][3];
  __shared__ u32 s_proj[16];
  u64 lo[3], hi[3];
  for (int k = 0; k < 3; ++k) {
    lo[k] = ok ? enc_f64(pt[k]) : ~0ull;
    hi[k] = ok ? enc_f64(pt[k]) : 0ull;
    for (int off = 32; off > 0; off >>= 1) {
      u64 t = __shfl_down(lo[k], off, 64);
      lo[k] = min(lo[k], t);
      t = __shfl_down(hi[k], off, 64);
      hi[k] = max(hi[k], t);
    }
  }
  const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    for (int k = 0; k < 3; ++k) s_lo[wave][k] = lo[k], s_hi[wave][k] = hi[k];
    s_proj[wave] = n_proj;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    u64 l = ~0ull, h = 0ull;
    for (int w = 0; w < nwave; ++w) l = min(l, s_lo[w][k]), h = max(h, s_hi[w][k]);
    A.partial[(size_t)blockIdx.x * 8 + k] = l;
    A.partial[(size_t)blockIdx.x * 8 + 3 + k] = h;
  }
  if (threadIdx.x == 3) {  // (one same-address atomic per wave on a frame counter made this an 18 us kernel)
    u64 c = 0ull;
    for (int w = 0; w < nwave; ++w) c += s_proj[w];
    A.partial[(size_t)blockIdx.x * 8 + 6] = c;
  }
}

// RayCaster helpers (raycast.cpp:6-23)
// fmod(x, 1) == x - trunc(x) bit for bit (the difference is exactly representable; only the sign of a zero result
// can differ, and the "+ modulus" that follows erases it) -- the library fmod is a long loop
__device__ __forceinline__ double rc_fmod1(double x) { return x - trunc(x); }
__device__ __forceinline__ double rc_mod(double value, double modulus) {  // only ever called with modulus 1
  (void)modulus;
  return rc_fmod1(rc_fmod1(value) + 1.0);
}
__device__ __forceinline__ double rc_intbound(double s, double ds) {
  if (ds < 0) {
    s = -s;
    ds = -ds;
  }
  s = rc_mod(s, 1);
  return (1 - s) / ds;
}

#ifndef RC_SPLIT
#define RC_SPLIT 4  // lanes per ray (k_insert_raycast)
#endif
#ifndef RC_SLOTS
#define RC_SLOTS 64  // point slots per workgroup of k_insert_raycast (256 threads = 64 rays x RC_SPLIT lanes at most)
#endif
#ifndef CUBE_XY
#define CUBE_XY 64  // lines per side of the workgroup's LDS bitmap (x 32 voxels in z)
#endif
__global__ void __launch_bounds__(256) k_insert_raycast(Geo g, InsertArgs A) {
  if (blockIdx.x == 0) {  // fold the per-block boxes of k_insert_classify (all 256 threads: a serial
                          // loop over ~300 records by six threads cost 40 us of dependent loads)
    __shared__ u64 s_red[4][6];
    u64 v[6] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull};
    u64 cnt = 0ull;
    for (int b = threadIdx.x; b < A.nblk; b += 256) {
      for (int k = 0; k < 6; ++k) {
        const u64 p = A.partial[(size_t)b * 8 + k];
        v[k] = k < 3 ? min(v[k], p) : max(v[k], p);
      }
      cnt += A.partial[(size_t)b * 8 + 6];
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    __shared__ u64 s_cntw[4];
    if ((threadIdx.x & 63) == 0) s_cntw[threadIdx.x >> 6] = cnt;
    for (int k = 0; k < 6; ++k)
      for (int off = 32; off > 0; off >>= 1) {
        const u64 t = __shfl_down(v[k], off, 64);
        v[k] = k < 3 ? min(v[k], t) : max(v[k], t);
      }
    if ((threadIdx.x & 63) == 0)
      for (int k = 0; k < 6; ++k) s_red[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 6) {
      const int k = threadIdx.x;
      u64 r = enc_f64(A.cam[k < 3 ? k : k - 3]);  // update_min = update_max = camera_pos (:265-266)
      for (int w = 0; w < 4; ++w) r = k < 3 ? min(r, s_red[w][k]) : max(r, s_red[w][k]);
      A.h_out[k] = r;
    }
    if (threadIdx.x == 6) A.h_out[6] = s_cntw[0] + s_cntw[1] + s_cntw[2] + s_cntw[3];
    // the host is waiting for exactly these eight words (it sizes the next launches by the box): publish them
    // now, the ray walks of this block follow
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&A.h_out[7], A.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // Only the first point of every end voxel casts a ray (~1 point in 4): compact the casters of the
  // block into LDS first, so that the walk runs with full waves and the other waves retire at once
  __shared__ double s_pt[256][3];
  __shared__ u32 s_cnt;
  // The rays of a workgroup (neighbouring pixels) form a thin fan that converges on the camera and revisits the
  // same voxels over and over: the miss marks inside a CUBE_XY x CUBE_XY x 32 voxel box laid over the fan (its
  // end voxels and the camera voxel; when the fan is larger, the camera's end of it) are collected in an LDS
  // bitmap (one word = 32 z-neighbours of a line) and flushed once per workgroup as whole words -- a few
  // hundred atomics instead of one per ray step.  With the walk split over four lanes per ray the global
  // atomics of the cells outside a small camera-centred cube had become the kernel's bottleneck.
  __shared__ u32 s_seen[CUBE_XY * CUBE_XY];
  for (int t = threadIdx.x; t < CUBE_XY * CUBE_XY; t += 256) s_seen[t] = 0u;
  if (threadIdx.x == 0) s_cnt = 0u;
  __syncthreads();
  {
    const int i = blockIdx.x * RC_SLOTS + threadIdx.x;
    double p0[3] = {0.0, 0.0, 0.0};
    const long a = (threadIdx.x < RC_SLOTS && i < A.n) ? A.rec[i].a : -1L;
    bool cast = a >= 0;
    if (cast) {
      cast = A.owner[a] == (u32)i;  // the first point of this end voxel
      if (cast) {
        p0[0] = A.rec[i].pt[0], p0[1] = A.rec[i].pt[1], p0[2] = A.rec[i].pt[2];
        A.owner[a] = 0xFFFFFFFFu;  // leave the owner table clean for the next frame
        cast = (signed char)A.flag_rayend[a] != A.num;
        if (cast) A.flag_rayend[a] = (unsigned char)A.num;
      }
    }
    if (cast) {
      const u32 slot = atomicAdd(&s_cnt, 1u);
      s_pt[slot][0] = p0[0], s_pt[slot][1] = p0[1], s_pt[slot][2] = p0[2];
    }
  }
  __syncthreads();
  int cv[3];  // one LDS word = the 32 z-neighbours cv[2] .. cv[2]+31 of the line (cv[0] + ux, cv[1] + uy)
  for (int k = 0; k < 3; ++k) {
    // the end points of these slots: the box k_insert_classify recorded for its workgroup (256 slots: a superset)
    const u64* rec = A.partial + (size_t)((blockIdx.x * RC_SLOTS) >> 8) * 8;
    const int cam_c = (int)floor((A.cam[k] - g.org[k]) * g.res_inv), ext = k < 2 ? CUBE_XY : 32;
    int lo = cam_c, hi = cam_c;
    if (rec[k] <= rec[3 + k]) {  // (an empty record: min = ~0, max = 0)
      lo = min(lo, (int)floor((dec_f64_dev(rec[k]) - g.org[k]) * g.res_inv));
      hi = max(hi, (int)floor((dec_f64_dev(rec[3 + k]) - g.org[k]) * g.res_inv));
    }
    cv[k] = hi - lo < ext ? lo : (cam_c == lo ? lo : (cam_c == hi ? hi - ext + 1 : cam_c - ext / 2));
  }
  // Every ray is walked by RC_SPLIT lanes, each doing the cells whose crossing parameter lies in its quarter
  // [q, q+1) / RC_SPLIT of the ray.  A lane reaches its starting state without walking: the tMax of an axis after j
  // steps is tMax + tDelta added j times whatever the other axes did, so it advances every axis on its own while
  // tMax < q / RC_SPLIT (the same additions in the same order as the reference's walk: same bits), which is exactly
  // the state the sequential walk is in when it has consumed every crossing below that parameter -- the walk
  // takes crossings in increasing tMax order (ties z, y, x).  The sequential walk was the longest dependent chain
  // of the fusion (~80 steps x ~350 ns on a lone wave).
  for (u32 task = threadIdx.x; task < s_cnt * RC_SPLIT; task += 256) {
  const u32 ray = task / RC_SPLIT, part = task % RC_SPLIT;
  const double pt[3] = {s_pt[ray][0], s_pt[ray][1], s_pt[ray][2]};

  // RayCaster::input(pt_w, camera_pos) (raycast.cpp:329-372)
  double s[3], e[3];
  for (int k = 0; k < 3; ++k) {
    s[k] = pt[k] / g.res;
    e[k] = A.cam[k] / g.res;
  }
  int c[3], ec[3], st[3];
  double tmax[3], tdel[3];
  for (int k = 0; k < 3; ++k) {
    c[k] = (int)floor(s[k]);
    ec[k] = (int)floor(e[k]);
    double d = ec[k] - c[k];
    int di = (int)d;
    st[k] = di == 0 ? 0 : (di < 0 ? -1 : 1);
    tmax[k] = rc_intbound(s[k], d);
    tdel[k] = ((double)st[k]) / d;
  }
  const double off[3] = {0.5 - g.org[0] / g.res, 0.5 - g.org[1] / g.res, 0.5 - g.org[2] / g.res};
  int guard = abs(ec[0] - c[0]) + abs(ec[1] - c[1]) + abs(ec[2] - c[2]) + 4;
  // nextId (:374-407) reports the current cell ((tmp + offset_).cast<int>()), stops at the end cell, else steps
  // along the axis with the smallest tMax (ties: z before y before x).  The first reported cell -- the end voxel
  // itself -- is discarded (:314).  One wave walks alone on its SIMD, so the step is written branch-free: the
  // nested ifs of the reference cost ~150 instructions per cell as divergent code, the selects below ~60.
  // A ray whose first and last cell both index into the map stays inside it (every axis moves one way only):
  // its cells need no bounds tests.
  int c0 = c[0], c1 = c[1], c2 = c[2];
  double t0 = tmax[0], t1 = tmax[1], t2 = tmax[2];
  const double tau_lo = (double)part / RC_SPLIT;
  const double tau_hi = part + 1 < RC_SPLIT ? (double)(part + 1) / RC_SPLIT : INFINITY;
  if (part) {  // (an axis that does not move has tMax = inf: untouched)
    while (t0 < tau_lo) t0 += tdel[0], c0 += st[0];
    while (t1 < tau_lo) t1 += tdel[1], c1 += st[1];
    while (t2 < tau_lo) t2 += tdel[2], c2 += st[2];
  }
  const int is0 = (int)((double)c0 + off[0]), is1 = (int)((double)c1 + off[1]), is2 = (int)((double)c2 + off[2]);
  const int ie0 = (int)((double)ec[0] + off[0]), ie1 = (int)((double)ec[1] + off[1]), ie2 = (int)((double)ec[2] + off[2]);
  const bool inside = min(is0, ie0) >= 0 && max(is0, ie0) < g.nx && min(is1, ie1) >= 0 && max(is1, ie1) < g.ny &&
                      min(is2, ie2) >= 0 && max(is2, ie2) < g.nz;
#define RC_STEP()                                                         \
  {                                                                       \
    const bool xy = t0 < t1, xz = t0 < t2, yz = t1 < t2;                  \
    const bool sx = xy && xz, sy = !xy && yz, sz = !(sx || sy);           \
    c0 += sx ? st[0] : 0, c1 += sy ? st[1] : 0, c2 += sz ? st[2] : 0;     \
    t0 = sx ? t0 + tdel[0] : t0;                                          \
    t1 = sy ? t1 + tdel[1] : t1;                                          \
    t2 = sz ? t2 + tdel[2] : t2;                                          \
  }
  bool more = !(c0 == ec[0] && c1 == ec[1] && c2 == ec[2]) && fmin(fmin(t0, t1), t2) < tau_hi;
  if (more && c0 == c[0] && c1 == c[1] && c2 == c[2]) {  // the ray's first cell (the end voxel itself) is reported
                                                           // first and discarded; it falls to the lane that owns
                                                           // the first crossing
    RC_STEP()
    more = --guard >= 0;
  }
  while (more) {
    const int ix = (int)((double)c0 + off[0]), iy = (int)((double)c1 + off[1]), iz = (int)((double)c2 + off[2]);
    if (((c0 ^ ec[0]) | (c1 ^ ec[1]) | (c2 ^ ec[2])) == 0) break;
    if (!(t0 < tau_hi || t1 < tau_hi || t2 < tau_hi)) break;  // the next lane's cells start here
    const long av = (long)ix * g.nyz + (long)iy * g.nz + iz;
    bool send = true, raw = true;
    if (!inside) {
      send = av >= 0 && av < g.N && ix >= 0 && ix < g.nx && iy >= 0 && iy < g.ny && iz >= 0 && iz < g.nz;
      raw = av >= 0 && av < g.N;  // a cell outside the index box still addresses a voxel (aliased rows); outside
                                  // [0, N) the reference is undefined behaviour: dropped
    }
    const u32 ux = (u32)(ix - cv[0]), uy = (u32)(iy - cv[1]), uz = (u32)(iz - cv[2]);
    if (send && (ux | uy) < (u32)CUBE_XY && uz < 32u) {
      atomicOr(&s_seen[ux * CUBE_XY + uy], 1u << uz);  // flushed as whole words when the block is done
    } else if (raw) {
      atomicOr(&A.miss[av >> 6], 1ull << (av & 63));
    }
    RC_STEP()
    if (--guard < 0) break;
  }
#undef RC_STEP
  }  // walkers
  // flush the cube: every non-empty LDS word is 32 z-consecutive voxels of one line, i.e. one or two
  // words of the miss plane
  __syncthreads();
  for (int t = threadIdx.x; t < CUBE_XY * CUBE_XY; t += 256) {
    const u32 bits = s_seen[t];
    if (!bits) continue;
    const int x = cv[0] + t / CUBE_XY, y = cv[1] + t % CUBE_XY;
    const long a0 = (long)x * g.nyz + (long)y * g.nz + cv[2];  // address of bit 0 (its voxel may lie below z = 0:
    const long w0 = a0 >> 6;                                    // then the low bits are clear, see the walk)
    const int sh = (int)(a0 & 63);
    const u64 lo = (u64)bits << sh;
    if (lo) atomicOr(&A.miss[w0], lo);
    if (sh > 32) {
      const u64 hi = (u64)bits >> (64 - sh);
      if (hi) atomicOr(&A.miss[w0 + 1], hi);
    }
  }
}

// log-odds update of every touched voxel (:332-344) + refresh of the state planes.  One WAVE per 64-voxel word,
// one lane per voxel: the f64 read-modify-write of a word's voxels is one coalesced 512-byte access instead of a
// serial loop over its set bits, and the new state bits of the word are two ballots.
__global__ void __launch_bounds__(256)
k_insert_update(Geo g, u64* __restrict__ hit, u64* __restrict__ miss, double* __restrict__ occ,
                u64* __restrict__ occ_bits, u64* __restrict__ unk_bits, int w_lo, int w_hi, double l_hit,
                double l_miss, double l_min, double l_max, double l_occ) {
  const int lane = threadIdx.x & 63;
  const int w = w_lo + (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (w > w_hi) return;
  const u64 h = hit[w], ms = miss[w];
  const u64 t = h | ms;
  if (t == 0ull) return;  // (wave-uniform)
  const double thr_unk = l_min - 1e-3;
  const bool touched = (t >> lane) & 1ull;
  bool is_occ = (occ_bits[w] >> lane) & 1ull, is_unk = (unk_bits[w] >> lane) & 1ull;
  if (touched) {
    const long a = 64L * w + lane;
    const double upd = ((h >> lane) & 1ull) ? l_hit : l_miss;
    double o = occ[a];
    if (o < thr_unk) o = l_occ;
    o = fmin(fmax(o + upd, l_min), l_max);
    occ[a] = o;
    is_occ = o > l_occ;
    is_unk = o < thr_unk;
  }
  const u64 ob = __ballot(is_occ), ub = __ballot(is_unk);
  if (lane == 0) {
    hit[w] = 0ull;
    miss[w] = 0ull;
    occ_bits[w] = ob;
    unk_bits[w] = ub;
  }
}

// fusion of n point records already resident on the device.  d_head: 64-byte device scratch
// ([0..5] sortable-encoded bbox, [6] number of valid points when `counted`).  Frames whose slots are
// all empty leave the map untouched, like `if (point_num == 0) return;` (:260).
static int insert_points_dev(fuelmi_map* m, const unsigned char* d_pts, int stride, int n, const double cam[3],
                             const DepthArgs* depth, bool counted, int* n_valid) {
  HIPCHK(map_wait_plane_readers(m));  // (a frontier search in flight still reads the planes this fusion rewrites)
  ++m->fusion_count;
  const Geo& g = m->g;
  const fuelmi_map_info& I = m->info;
  const signed char num_before = m->raycast_num;
  m->raycast_num = (signed char)(m->raycast_num + 1);  // char wrap like the reference
  {
    const size_t need = (size_t)((n + 255) / 256) * 8;
    if (need > m->ins_partial_cap) {
      if (m->ins_partial) HIPCHK(hipFree(m->ins_partial));
      m->ins_partial = nullptr;
      m->ins_partial_cap = 0;
      HIPCHK(hipMalloc(reinterpret_cast<void**>(&m->ins_partial), (need + 64) * sizeof(u64)));
      m->ins_partial_cap = need + 64;
    }
  }
  InsertArgs A;
  A.pts = d_pts;
  A.stride = stride;
  A.n = n;
  A.from_depth = depth ? 1 : 0;
  if (depth)
    A.D = *depth;
  else
    memset(&A.D, 0, sizeof(A.D));
  for (int k = 0; k < 3; ++k) A.cam[k] = cam[k];
  A.max_ray = m->cfg.max_ray_length;
  A.num = m->raycast_num;
  A.hit = m->hit_bits.p;
  A.miss = m->miss_bits.p;
  A.owner = m->ray_owner;
  A.flag_rayend = m->flag_rayend;
  A.h_out = m->h_ins;
  A.head = m->ins_head;
  A.epoch = ++m->ins_epoch;
  int nb = (n + 255) / 256;
  A.nblk = nb;
  A.partial = m->ins_partial;
  if ((size_t)n > m->ins_rec_cap) {
    if (m->ins_rec) HIPCHK(hipFree(m->ins_rec));
    m->ins_rec = nullptr;
    m->ins_rec_cap = 0;
    HIPCHK(hipMalloc(&m->ins_rec, ((size_t)n + 1024) * sizeof(InsRec)));
    m->ins_rec_cap = (size_t)n + 1024;
  }
  A.rec = reinterpret_cast<InsRec*>(m->ins_rec);
  k_insert_classify<<<nb, 256, 0, m->stream>>>(g, A);
  k_insert_raycast<<<(n + RC_SLOTS - 1) / RC_SLOTS, 256, 0, m->stream>>>(g, A);
  HIPCHK(hipGetLastError());
  // the end-point box sizes the next launches: poll the stamp the fusion's second kernel writes into pinned memory
  // (a blocking stream synchronisation costs ~40 us of wake-up latency per frame)
  u64 h_bbox[8];
  {
    volatile u64* hv = m->h_ins;
    unsigned spins = 0;
    while (hv[7] != A.epoch) {
      if ((++spins & 0x3FFFu) == 0u) {
        const hipError_t q = hipStreamQuery(m->stream);
        if (q != hipErrorNotReady && q != hipSuccess) HIPCHK(q);
        if (q == hipSuccess && hv[7] != A.epoch) {
          fuelmi_set_error("fusion: the kernels finished without publishing the end-point box");
          return FUELMI_EHIP;
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int k = 0; k < 8; ++k) h_bbox[k] = hv[k];
  }
  if (counted) {
    if (n_valid) *n_valid = (int)h_bbox[6];
    if (h_bbox[6] == 0) {  // nothing projected: the reference returns before touching any state
      m->raycast_num = num_before;
      return FUELMI_OK;
    }
  }
  if (m->reset_updated_box) {
    for (int k = 0; k < 3; ++k) m->upd_min[k] = m->upd_max[k] = cam[k];
    m->reset_updated_box = false;
  }
  double umin[3], umax[3];
  for (int k = 0; k < 3; ++k) {
    umin[k] = dec_f64(h_bbox[k]);
    umax[k] = dec_f64(h_bbox[3 + k]);
  }
  // local bound (:319-324) and accumulated update box (:326-330)
  const int nv[3] = {g.nx, g.ny, g.nz};
  const double linf = std::max(m->cfg.resolution, m->cfg.local_bound_inflate);
  for (int k = 0; k < 3; ++k) {
    double inf = (k < 2) ? linf : 0.0;
    int hi = (int)std::floor((umax[k] + inf - g.org[k]) * g.res_inv);
    int lo = (int)std::floor((umin[k] - inf - g.org[k]) * g.res_inv);
    m->local_bound.lo[k] = std::max(std::min(lo, nv[k] - 1), 0);
    m->local_bound.hi[k] = std::max(std::min(hi, nv[k] - 1), 0);
    m->upd_min[k] = std::min(umin[k], m->upd_min[k]);
    m->upd_max[k] = std::max(umax[k], m->upd_max[k]);
  }
  // touched voxels lie in the index bbox of camera and end points (+1 voxel: the ray walker
  // and posToIndex floor differently at cell faces)
  int lo3[3], hi3[3];
  for (int k = 0; k < 3; ++k) {
    lo3[k] = std::max((int)std::floor((umin[k] - g.org[k]) * g.res_inv) - 1, 0);
    hi3[k] = std::min((int)std::floor((umax[k] - g.org[k]) * g.res_inv) + 1, nv[k] - 1);
  }
  long a_lo = (long)lo3[0] * g.nyz + (long)lo3[1] * g.nz + lo3[2];
  long a_hi = (long)hi3[0] * g.nyz + (long)hi3[1] * g.nz + hi3[2];
  int w_lo = (int)(a_lo >> 6), w_hi = (int)(a_hi >> 6);
  // A camera outside the map (possible through inputPointCloud, which has no isInMap(camera) test): ray cells
  // whose y / z index leaves the map alias voxels of neighbouring rows (setCacheOccupancy, :243-257, indexes
  // with the linear address and has no bounds test; addresses outside [0, N) are undefined behaviour there and
  // dropped here) -- anywhere in the grid.  They are applied in this frame like the reference applies them:
  // sweep every word (words without marks cost one 16-byte read).
  bool cam_in = true;
  for (int k = 0; k < 3; ++k)
    if (cam[k] < g.minb[k] + 1e-4 || cam[k] > g.maxb[k] - 1e-4) cam_in = false;
  if (!cam_in) w_lo = 0, w_hi = g.W - 1;
  k_insert_update<<<(w_hi - w_lo + 4) / 4, 256, 0, m->stream>>>(
      g, m->hit_bits.p, m->miss_bits.p, m->occ, m->occ_bits.p, m->unk_bits.p, w_lo, w_hi, I.prob_hit_log,
      I.prob_miss_log, I.clamp_min_log, I.clamp_max_log, I.min_occupancy_log);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(m->ev_planes, m->stream));
  ++m->planes_ver;
  return FUELMI_OK;
}

int insert_points(fuelmi_map* m, const float* xyz, int stride, int n, const double cam[3]) {
  size_t pbytes = (size_t)n * stride;
  int rc = map_ensure_stage(m, pbytes + 64, 0);
  if (rc) return rc;
  unsigned char* d_pts = reinterpret_cast<unsigned char*>(m->d_stage) + 64;
  StageScope sc(m, FUELMI_K_INSERT);
  HIPCHK(hipMemcpyAsync(d_pts, xyz, pbytes, hipMemcpyHostToDevice, m->stream));
  return insert_points_dev(m, d_pts, stride, n, cam, nullptr, false, nullptr);
}

// ---- depth image -> world points on the device (MapROS::proessDepthImage, plan_env/src/map_ros.cpp:176-215)
// One lane per sampled pixel (v, u) = margin + k*skip.  Arithmetic in f64 in the reference's order
// (depth = px * (1/scale); pt = R * [(u-cx)*d/fx, (v-cy)*d/fy, d] + t; stored as float).  Quirk kept:
// the depth comes from pixel u, the zero test reads the pixel `skip` further (the row pointer is
// advanced in between, :190-198); past the end of the image (reference: out-of-bounds read) it
// counts as 0.  Output slot s = iv*nu + iu keeps the reference's point order; a pixel dropped by the
// min-distance filter leaves x = NaN in its slot, which the fusion kernels skip.
__global__ void __launch_bounds__(256) k_project_depth(DepthArgs D) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  bool valid = false;
  if (s < D.nslots) {
    float q[3];
    valid = project_pixel(D, s, q);
    reinterpret_cast<float4*>(D.out)[s] = valid ? make_float4(q[0], q[1], q[2], 1.f) : make_float4(NAN, 0.f, 0.f, 1.f);
  }
  const u64 m = __ballot(valid);
  if (D.count && (threadIdx.x & 63) == 0 && m) atomicAdd(D.count, (u64)__popcll(m));
}

static void quat_to_rot(const double q[4], double R[9]) {  // Eigen's toRotationMatrix(), q = (w,x,y,z)
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1.0 - (tyy + tzz), R[1] = txy - twz, R[2] = txz + twy;
  R[3] = txy + twz, R[4] = 1.0 - (txx + tzz), R[5] = tyz - twx;
  R[6] = txz - twy, R[7] = tyz + twx, R[8] = 1.0 - (txx + tyy);
}

// uploads the image, projects it; *d_pts_out / *d_head_out point into the map's device staging area
// uploads the image and fills the projection arguments; launch: also write the projected points to the staging
// area (fuelmi_map_project_depth) -- the fusion projects on the fly instead
static int project_depth_dev(fuelmi_map* m, const unsigned short* depth, int rows, int cols,
                             const fuelmi_depth_cfg* c, const double pos[3], const double q[4], float** d_pts_out,
                             DepthArgs* D_out, int* nslots_out, bool launch) {
  const int margin = c->depth_filter_margin, skip = c->skip_pixel;
  const int nu = cols - 2 * margin > 0 ? (cols - 2 * margin + skip - 1) / skip : 0;
  const int nvv = rows - 2 * margin > 0 ? (rows - 2 * margin + skip - 1) / skip : 0;
  const int nslots = nu * nvv;
  *nslots_out = nslots;
  const size_t img_bytes = ((size_t)rows * cols * 2 + 255) & ~(size_t)255;
  // the caller's image is ordinary pageable memory (a cv::Mat): stage it through the map's pinned
  // buffer so the upload is one DMA instead of the runtime's chunked pageable path
  int rc = map_ensure_stage(m, 64 + img_bytes + (size_t)nslots * 16 + 256, (size_t)rows * cols * 2);
  if (rc) return rc;
  u64* d_head = reinterpret_cast<u64*>(m->d_stage);
  unsigned short* d_img = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(m->d_stage) + 256);
  float* d_pts = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(m->d_stage) + 256 + img_bytes);
  *d_pts_out = d_pts;
  (void)d_head;
  if (nslots == 0) return FUELMI_OK;
  // An image the device can address -- device memory, or host memory that is pinned / registered (hipHostMalloc,
  // hipHostRegister, fuelmi_host_register) -- is read where it lies: no staging copy, and for device memory no
  // PCIe traffic inside the cycle either.  The caller keeps it unchanged until the next call on this map.
  const unsigned short* direct = nullptr;
  bool foreign = false;  // device memory of ANOTHER GPU (a fleet in one process): copied, never dereferenced here
  {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, depth) == hipSuccess &&
        (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged) &&
        at.devicePointer) {
      if (at.type == hipMemoryTypeDevice && at.device != m->device)
        foreign = true;
      else
        direct = reinterpret_cast<const unsigned short*>(at.devicePointer);
    } else
      (void)hipGetLastError();  // ordinary pageable memory (a cv::Mat): not an error
  }
  if (foreign) {
    HIPCHK(hipMemcpyAsync(d_img, depth, (size_t)rows * cols * 2, hipMemcpyDefault, m->stream));
  } else if (!direct) {
    // (every user of the pinned staging buffer synchronises before it returns, so it is free here)
    memcpy(m->h_stage, depth, (size_t)rows * cols * 2);
    // the fusion kernels read the (pinned) staged image over PCIe themselves -- ~0.6 MB per 640 x 480 frame, read
    // by two kernels -- instead of waiting for a DMA copy in front of them; the stand-alone projection keeps the copy
    if (launch)
      HIPCHK(hipMemcpyAsync(d_img, m->h_stage, (size_t)rows * cols * 2, hipMemcpyHostToDevice, m->stream));
  }
  DepthArgs D;
  D.img = direct ? direct : ((!launch && !foreign) ? reinterpret_cast<const unsigned short*>(m->h_stage) : d_img);
  D.rows = rows, D.cols = cols, D.margin = margin, D.skip = skip, D.nu = nu, D.nslots = nslots;
  D.fx = c->fx, D.fy = c->fy, D.cx = c->cx, D.cy = c->cy;
  D.maxdist = c->depth_filter_maxdist, D.mindist = c->depth_filter_mindist;
  D.inv_factor = 1.0 / c->k_depth_scaling_factor;
  quat_to_rot(q, D.R);
  for (int k = 0; k < 3; ++k) D.t[k] = pos[k];
  D.out = d_pts;
  D.count = nullptr;
  *D_out = D;
  if (launch) k_project_depth<<<(nslots + 255) / 256, 256, 0, m->stream>>>(D);
  HIPCHK(hipGetLastError());
  return FUELMI_OK;
}

static bool depth_args_ok(const fuelmi_depth_cfg* c, int rows, int cols) {
  return c && rows > 0 && cols > 0 && c->skip_pixel > 0 && c->depth_filter_margin >= 0 && c->fx != 0.0 &&
         c->fy != 0.0 && c->k_depth_scaling_factor != 0.0;
}

extern "C" int fuelmi_map_project_depth(fuelmi_map* m, const unsigned short* depth, int rows, int cols,
                                        const fuelmi_depth_cfg* cfg, const double cam_pos[3],
                                        const double cam_q_wxyz[4], float* xyz, int cap, int* n_points) {
  ARGCHK(m && depth && cam_pos && cam_q_wxyz && n_points && depth_args_ok(cfg, rows, cols) && (cap == 0 || xyz));
  HIPCHK(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lk(m->qmu);  // (staging buffers: see fuelmi_map_input_depth)
  float* d_pts;
  DepthArgs D;
  int nslots;
  int rc = project_depth_dev(m, depth, rows, cols, cfg, cam_pos, cam_q_wxyz, &d_pts, &D, &nslots, true);
  if (rc) return rc;
  std::vector<float> h((size_t)nslots * 4);
  if (nslots) HIPCHK(hipMemcpyAsync(h.data(), d_pts, h.size() * sizeof(float), hipMemcpyDeviceToHost, m->stream));
  HIPCHK(hipStreamSynchronize(m->stream));
  int n = 0;
  for (int s = 0; s < nslots; ++s) {
    if (std::isnan(h[4 * (size_t)s])) continue;
    if (n < cap) memcpy(xyz + 3 * (size_t)n, &h[4 * (size_t)s], 3 * sizeof(float));
    ++n;
  }
  *n_points = n;
  return FUELMI_OK;
}

// MapROS::depthPoseCallback (:121-150) minus the ROS plumbing: projection + inputPointCloud without
// the points ever leaving the device.  The caller follows with fuelmi_map_inflate_local like the
// reference does when local_updated_ is set.
extern "C" int fuelmi_map_input_depth(fuelmi_map* m, const unsigned short* depth, int rows, int cols,
                                      const fuelmi_depth_cfg* cfg, const double cam_pos[3],
                                      const double cam_q_wxyz[4], int* n_points) {
  ARGCHK(m && depth && cam_pos && cam_q_wxyz && depth_args_ok(cfg, rows, cols));
  if (n_points) *n_points = 0;
  const Geo& g = m->g;
  for (int k = 0; k < 3; ++k)  // if (!map_->isInMap(camera_pos_)) return;
    if (cam_pos[k] < g.minb[k] + 1e-4 || cam_pos[k] > g.maxb[k] - 1e-4) return FUELMI_OK;
  HIPCHK(hipSetDevice(m->device));
  // the staging buffers (size, host part) are shared with the host-staged queries other threads may be running
  std::lock_guard<std::mutex> lk(m->qmu);
  StageScope sc(m, FUELMI_K_INSERT);
  float* d_pts;
  DepthArgs D;
  int nslots;
  int rc = project_depth_dev(m, depth, rows, cols, cfg, cam_pos, cam_q_wxyz, &d_pts, &D, &nslots, false);
  if (rc || nslots == 0) return rc;
  return insert_points_dev(m, nullptr, 16, nslots, cam_pos, &D, true, n_points);
}

/* Pin a host buffer the caller will hand to fuelmi_map_input_depth again and again (a camera driver's frame ring):
 * frames inside it are read by the kernels directly, without the staging copy. */
extern "C" int fuelmi_host_register(void* ptr, size_t bytes) {
  ARGCHK(ptr && bytes);
  HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterMapped));
  return FUELMI_OK;
}
extern "C" int fuelmi_host_unregister(void* ptr) {
  ARGCHK(ptr);
  HIPCHK(hipHostUnregister(ptr));
  return FUELMI_OK;
}
/* Plain device buffers for callers without a HIP runtime of their own (the tests, bench.py: frames resident in
 * HBM before the timed region). */
extern "C" int fuelmi_device_alloc(int device, size_t bytes, void** out) {
  ARGCHK(out && bytes);
  *out = nullptr;
  HIPCHK(hipSetDevice(device));
  HIPCHK(hipMalloc(out, bytes));
  return FUELMI_OK;
}
extern "C" int fuelmi_device_upload(void* dst, const void* src, size_t bytes) {
  ARGCHK(dst && src);
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return FUELMI_OK;
}
/* STREAM-triad on this device, a[i] = b[i] + s * c[i] over three arrays of `bytes` each (choose >> 256 MiB so the
 * Infinity Cache cannot hold them): the HBM bandwidth a simple kernel actually reaches, to put beside the 8 TB/s
 * vendor peak the rooflines are quoted against. */
__global__ void __launch_bounds__(256) k_triad(float4* __restrict__ a, const float4* __restrict__ b,
                                                const float4* __restrict__ c, float s, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 x = b[i], y = c[i];
    a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
  }
}
extern "C" int fuelmi_hbm_triad(int device, size_t bytes, int reps, double* gb_per_s) {
  ARGCHK(gb_per_s && bytes >= 4096 && reps >= 1);
  *gb_per_s = 0.0;
  HIPCHK(hipSetDevice(device));
  bytes &= ~(size_t)15;
  struct Scratch {  // released on every way out
    float4 *a = nullptr, *b = nullptr, *c = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      for (float4* p : {a, b, c})
        if (p) (void)hipFree(p);
    }
  } S;
  if (hipMalloc(&S.a, bytes) != hipSuccess || hipMalloc(&S.b, bytes) != hipSuccess || hipMalloc(&S.c, bytes) != hipSuccess) {
    (void)hipGetLastError();
    fuelmi_set_error("fuelmi_hbm_triad: cannot allocate 3 x %zu bytes", bytes);
    return FUELMI_EHIP;
  }
  HIPCHK(hipMemset(S.b, 0, bytes));
  HIPCHK(hipMemset(S.c, 0, bytes));
  HIPCHK(hipEventCreate(&S.e0));
  HIPCHK(hipEventCreate(&S.e1));
  const size_t n4 = bytes / 16;
  for (int grid : {256 * 8, 256 * 16, 256 * 32, 256 * 64, 256 * 256}) {  // the best of a few launch shapes
    k_triad<<<grid, 256>>>(S.a, S.b, S.c, 0.5f, n4);  // warm-up
    HIPCHK(hipEventRecord(S.e0, nullptr));
    for (int r = 0; r < reps; ++r) k_triad<<<grid, 256>>>(S.a, S.b, S.c, 0.5f, n4);
    HIPCHK(hipEventRecord(S.e1, nullptr));
    HIPCHK(hipEventSynchronize(S.e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, S.e0, S.e1));
    *gb_per_s = std::max(*gb_per_s, 3.0 * (double)bytes * reps / (ms * 1e-3) / 1e9);
  }
  return FUELMI_OK;
}
/* The x pass's traffic mix as a plain streaming kernel: read n 16-bit values, write n floats (1 byte in : 2 bytes out).
 * What the device reaches on THAT mix is the ceiling to hold k_esdf_x_pk2 against (writes cost more than reads: the
 * triad's 2 : 1 read : write mix overstates it). */
__global__ void __launch_bounds__(256) k_expand(float4* __restrict__ out, const uint2* __restrict__ in, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const uint2 v = in[i];
    out[i] = make_float4((float)(v.x & 0xffffu), (float)(v.x >> 16), (float)(v.y & 0xffffu), (float)(v.y >> 16));
  }
}
extern "C" int fuelmi_hbm_expand(int device, size_t bytes_in, int reps, double* gb_per_s) {
  ARGCHK(gb_per_s && bytes_in >= 4096 && reps >= 1);
  *gb_per_s = 0.0;
  HIPCHK(hipSetDevice(device));
  bytes_in &= ~(size_t)7;
  struct Scratch {
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~Scratch() {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      if (a) (void)hipFree(a);
      if (b) (void)hipFree(b);
    }
  } S;
  if (hipMalloc(&S.a, 2 * bytes_in) != hipSuccess || hipMalloc(&S.b, bytes_in) != hipSuccess) {
    (void)hipGetLastError();
    fuelmi_set_error("fuelmi_hbm_expand: cannot allocate 3 x %zu bytes", bytes_in);
    return FUELMI_EHIP;
  }
  HIPCHK(hipMemset(S.b, 0, bytes_in));
  HIPCHK(hipEventCreate(&S.e0));
  HIPCHK(hipEventCreate(&S.e1));
  const size_t n4 = bytes_in / 8;
  for (int grid : {256 * 8, 256 * 16, 256 * 32, 256 * 64, 256 * 256}) {
    k_expand<<<grid, 256>>>(static_cast<float4*>(S.a), static_cast<const uint2*>(S.b), n4);
    HIPCHK(hipEventRecord(S.e0, nullptr));
    for (int r = 0; r < reps; ++r) k_expand<<<grid, 256>>>(static_cast<float4*>(S.a), static_cast<const uint2*>(S.b), n4);
    HIPCHK(hipEventRecord(S.e1, nullptr));
    HIPCHK(hipEventSynchronize(S.e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, S.e0, S.e1));
    *gb_per_s = std::max(*gb_per_s, 3.0 * (double)bytes_in * reps / (ms * 1e-3) / 1e9);
  }
  return FUELMI_OK;
}
extern "C" int fuelmi_device_free(void* ptr) {
  if (ptr) HIPCHK(hipFree(ptr));
  return FUELMI_OK;
}

extern "C" int fuelmi_map_input_points(fuelmi_map* m, const float* xyz, int stride_bytes, int n,
                                       const double camera_pos[3]) {
  ARGCHK(m && camera_pos && n >= 0 && stride_bytes >= 12 && (n == 0 || xyz));
  if (n == 0) return FUELMI_OK;  // reference: if (point_num == 0) return;
  HIPCHK(hipSetDevice(m->device));
  std::lock_guard<std::mutex> lk(m->qmu);  // (staging buffers: see fuelmi_map_input_depth)
  return insert_points(m, xyz, stride_bytes, n, camera_pos);
}
