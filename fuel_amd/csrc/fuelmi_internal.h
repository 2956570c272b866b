// fuelmi_internal.h -- shared host/device declarations of libfuelmi (gfx950 only).
#ifndef FUELMI_INTERNAL_H_
#define FUELMI_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdio>
#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/fuelmi.h"

typedef unsigned long long u64;
typedef unsigned int u32;

// ---- error plumbing -------------------------------------------------------------------------
void fuelmi_set_error(const char* fmt, ...);
#define HIPCHK(expr)                                                                        \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      fuelmi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__,    \
                       __LINE__);                                                           \
      return FUELMI_EHIP;                                                                   \
    }                                                                                       \
  } while (0)
#define ARGCHK(cond)                                                        \
  do {                                                                      \
    if (!(cond)) {                                                          \
      fuelmi_set_error("invalid argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
      return FUELMI_EINVAL;                                                 \
    }                                                                       \
  } while (0)

// ---- grid geometry passed to kernels by value -------------------------------------------------
struct Geo {
  int nx, ny, nz;
  int nyz;    // ny*nz
  int N;      // nx*ny*nz  (< 2^31)
  int W;      // number of 64-bit words covering N bits
  double res, res_inv;
  double org[3];
  double minb[3], maxb[3];
};

struct Box3 {  // inclusive voxel index box
  int lo[3], hi[3];
};

// A bit-plane over the linear voxel address space: bit (adr & 63) of word (adr >> 6).
// Allocated with `margin` zero words on both sides so shifted window loads need no bounds tests.
struct Plane {
  u64* base = nullptr;  // allocation
  u64* p = nullptr;     // word 0
};

#define INF32 0x3FFFFFFFu
#define INF16 0xFFFFu

// chunks of the packed ESDF family's z/y pass and the column tiles of its x pass (esdf.hip, round 5)
#define PK2_MAXCH 16
struct Pk2Chunks {
  int n;                     // pieces of the aligned z range
  int n8;                    // ... of which the first n8 are 8 segments wide: their tiles are interleaved, tile (y, c) = y * n8 + c
  short seg0[PK2_MAXCH];     // first z-segment (4 voxels) of the aligned z range
  short g[PK2_MAXCH];        // segments (8, 4, 2, 1)
  int tile0[PK2_MAXCH + 1];  // first column tile of the narrower pieces (c >= n8; tile0[n8] = ylen * n8); [n] = tiles in all
};

#define PK2_MAXZCH 32
struct Pk2ZChunks {  // chunks of the z/y pass: pieces of the column tiles' z extent
  int n;
  short seg0[PK2_MAXZCH];       // first z-segment
  signed char g[PK2_MAXZCH];    // segments (8, 4, 2, 1)
  signed char gsh[PK2_MAXZCH];  // log2 of the width of the tile piece the chunk lies in
  signed char off[PK2_MAXZCH];  // first segment inside that piece
  int tile0[PK2_MAXZCH];        // column tile of y-row y: tile0 + (y / rows per tile) * tstride
  int tstride[PK2_MAXZCH];
};

// ---- the map object ---------------------------------------------------------------------------
struct ProfileSlot {
  std::vector<hipEvent_t> ev;  // pairs
  size_t used = 0;
};

struct fuelmi_map {
  fuelmi_map_cfg cfg;
  fuelmi_map_info info;
  Geo g;
  int device = 0;
  hipStream_t stream = nullptr;
  int margin_words = 0;

  // device state
  double* occ = nullptr;        // log-odds, exact reference values            8 B/voxel
  Plane occ_bits;               // occ > min_occupancy_log
  Plane unk_bits;               // occ < clamp_min_log - 1e-3
  Plane infl_bits;              // occupancy_buffer_inflate_ == 1
  Plane tmp_bits;               // scratch plane (occ & box)
  Plane tmp2_bits;              // scratch plane (y/z-dilated sources of the inflation)
  float* dist = nullptr;        // distance_buffer_ as f32                     4 B/voxel
  u32* esdf_tmp = nullptr;      // y-pass result (squared voxel units)         4 B/voxel (packed family: only the values above 16 bits)
  u32* esdf_tmp16 = nullptr;    // packed family: 16-bit y-pass result in the x pass's tile order   2 B/voxel (esdf.hip, round 5)
  size_t esdf_tmp16_bytes = 0;
  u32 esdf_serial = 0;          // stamps the "some slab holds a source" word of an update
  bool esdf_pk2_last = false;   // the last z/y pass wrote the 16-bit hand-over
  Pk2Chunks pk2_ch = {};        // ... in these column tiles
  Pk2ZChunks pk2_zch = {};      // ... written by these chunks of the z/y pass
  unsigned char* flag_rayend = nullptr;  // flag_rayend_                      1 B/voxel
  u32* ray_owner = nullptr;     // per-frame end-voxel owner (point index)     4 B/voxel
  Plane hit_bits, miss_bits;    // per-frame touched voxels
  u64* ins_partial = nullptr;   // per-block end-point boxes of the fusion's classify kernel
  size_t ins_partial_cap = 0;
  // objects created on this map (frontier finders, B-spline batches): if the map is destroyed first -- garbage
  // collectors and destructor orders do that -- each is told (its stream drained, its map pointer cleared) so that
  // its own destroy does not touch freed memory
  struct Dependent {
    void* obj;
    void (*orphan)(void*);
  };
  std::vector<Dependent> dependents;
  std::mutex dep_mu;  // finders / batches of a fleet are created and destroyed from different threads
  void* ins_rec = nullptr;      // per-slot records of the fusion's classify kernel (32 B each)
  size_t ins_rec_cap = 0;
  u64* ins_head = nullptr;      // [16] device; [8] = the ESDF far-output statistic (two u32, esdf.hip); [0..7]: [6] = points projected by the depth front end of the current frame
  u64* h_ins = nullptr;         // [16] pinned; [8] = the ESDF statistic as the x pass handed it over; [0..7]: end-point box [0..5], projected points [6], frame stamp [7] -- written
                                // by the fusion's second kernel, polled by the host (no blocking stream sync)
  u64 ins_epoch = 0;
  // ESDF far-output statistic (esdf.hip): device table [2 * 256 + 1] (pairs per group of 16 x-slabs, epoch), its
  // pinned copy, and the host's latest pair of every group
  u32* esdf_stat = nullptr;
  u32* h_esdf_stat = nullptr;
  u32 far_hist[256][2] = {};
  unsigned short far_tag[256] = {};  // tag of the entry of the pinned table each pair was taken from
  bool far_last = false;
  int esdf_family_pin = FUELMI_ESDF_AUTO;    // fuelmi_map_set_esdf_family
  int esdf_family_last = FUELMI_ESDF_PLAIN;  // family the z/y pass of the last update ran
  signed char raycast_num = 0;
  unsigned occ_epoch = 0;  // bumped by occupancy changes that bypass the updated box (upload, resetBuffer)

  // host-side bookkeeping the reference keeps in MapData
  Box3 local_bound;
  double upd_min[3], upd_max[3];
  bool reset_updated_box = true;

  // host-staged queries and mirror syncs of THIS map share its staging buffers: one lock per map (different
  // maps -- a fleet in one process -- never wait for each other)
  std::mutex qmu;
  // host mirrors registered for zero-copy refreshes (fuelmi_map_register_mirrors): host pointer, its device
  // alias, byte size; [0] occupancy f64, [1] inflate i8, [2] distance f64
  struct Mirror {
    void* host = nullptr;
    void* dev = nullptr;
  } mirror[3];
  // staging
  void* d_stage = nullptr;
  size_t d_stage_bytes = 0;
  void* h_stage = nullptr;  // pinned
  size_t h_stage_bytes = 0;

  // Read-only queries (getDistWithGrad, the one-shot B-spline calls: combineCost / optimize of ONE trajectory, the
  // spline glue) run on QUERY SLOTS: a side stream, two events and a pinned block each, taken from a pool -- so that
  // the <= 10 optimiser threads of topoReplan (plan_manage/src/planner_manager.cpp:446-453) and the visualisation
  // thread (exploration_manager/src/fast_exploration_fsm.cpp:122) neither queue behind the mutators on the map's
  // stream nor behind each other, and no call allocates or frees device memory (hipFree drains the device).  Inputs
  // are packed into the pinned block and read by the kernel in place; results are written there by the kernel; the
  // caller polls the slot's event.  A slot's stream is ordered behind what the map's stream held when the call began.
  struct QuerySlot {
    hipStream_t st = nullptr;
    hipEvent_t ev_dep = nullptr, ev_done = nullptr, ev_rd = nullptr;  // ev_rd: recorded by a WRITER of dist / infl (map_wait_query_readers)
    unsigned char* pin = nullptr;
    size_t pin_cap = 0;
    bool busy = false;
  };
  std::vector<std::unique_ptr<QuerySlot>> qslots;
  std::mutex qs_mu;
  std::condition_variable qs_cv;
  // readers (query slots: from "is the map's stream idle?" to "my kernel is launched") share it, a writer of the distance
  // field takes it alone while it makes its stream wait for the launched query kernels: no query can slip between a
  // writer's look at the slots and the writer's kernels, so every query sees the field before or after an update, never
  // a mix (with signed_dist: never the positive-only intermediate the negative pass merges into in place)
  std::shared_timed_mutex rw_mu;
  std::mutex prof_mu;  // the stage-profile log (StageScope) is shared by every thread that launches on this map

  // measurement
  double bench_host_us[7] = {0, 0, 0, 0, 0, 0, 0};  // fuelmi_bench_cycles: mean host microseconds per C-ABI call
  hipEvent_t t0 = nullptr, t1 = nullptr;  // fuelmi_timer_begin / _end
  int last_inflate_kernel = -1;           // 0: k_inflate_fused, 1: the factored pair (fuelmi_map_last_inflate_kernel)
  hipEvent_t t_prof0 = nullptr;           // origin of fuelmi_profile_get_timeline (recorded by fuelmi_profile_enable)
  hipEvent_t ev_planes = nullptr;  // recorded after every kernel that rewrites the occupancy state planes
  unsigned long long planes_ver = 0;  // ... and counted: a search stream that has already waited for this record does not queue the wait again
  // ... and the other direction: the last kernel of a running frontier search that READS those planes (set by
  // fuelmi_frontier_search_begin, owned by the finder).  A fusion queued while the search is in flight -- the next
  // depth frame of a streaming pipeline -- waits for it on the device instead of being forbidden
  // (one entry per finder with a search in flight: two finders on one map must not overwrite each other's event)
  std::vector<hipEvent_t> planes_read_evs;
  // ... and finders whose running search has NOT marked that point (round 5: the event between the chain's first two
  // kernels is a barrier packet on the search's critical path and three API calls' worth of host time -- a finder records
  // it only once it has seen a plane mutator arrive while one of its searches was in flight).  A mutator that finds such
  // an entry records the event NOW -- behind the whole chain queued so far: correct, just later -- and tells the finder
  // (`*sticky = true`) to mark the point itself from its next search on.
  struct LateReader {
    hipStream_t st;
    hipEvent_t ev;
    bool* sticky;
  };
  std::vector<LateReader> late_readers;
  unsigned long long fusion_count = 0;  // fusions / uploads queued so far (a search notices one queued behind its back)
  unsigned profile_mask = 0;
  ProfileSlot prof[FUELMI_K_COUNT];
  // event pair of a single-kernel stage being profiled: the launch site attaches it to the kernel itself
  // (hipExtLaunchKernelGGL), so the pair holds the kernel's own begin / end and not the times at which
  // the command processor got round to two marker packets
  hipEvent_t kev[2] = {nullptr, nullptr};
};

int map_ensure_stage(fuelmi_map* m, size_t dev_bytes, size_t host_bytes);

// Streams of the library.  `which` names the role ("MAP": a map's stream, "FR": a finder's search streams);
// priority: INT_MIN = default.
hipError_t fuelmi_stream_create(hipStream_t* s, int priority, const char* which);
// a query slot with at least `bytes` of pinned memory, its stream ordered behind the map's; released (and waited for)
// by the guard
struct QuerySlotGuard {
  fuelmi_map* m = nullptr;
  fuelmi_map::QuerySlot* s = nullptr;
  bool reading = false;  // holds m->rw_mu shared (acquire() .. the launch; finish() lets go)
  int acquire(fuelmi_map* m_, size_t bytes);
  hipError_t finish();  // record + poll the slot's completion event
  ~QuerySlotGuard();
};
int plane_alloc(fuelmi_map* m, Plane& pl);  // zeroed, with margins

// stage profiling helpers: bracket a launch sequence belonging to `stage`
struct StageScope {
  fuelmi_map* m;
  int stage;
  hipEvent_t e1 = nullptr;
  hipStream_t st = nullptr;
  // kernel_timed: the scope covers exactly one kernel launched through STAGE_LAUNCH
  StageScope(fuelmi_map* m_, int stage_, hipStream_t stream = nullptr, bool kernel_timed = false);
  ~StageScope();
};

// launch on the map's stream; inside a kernel-timed StageScope the profiling events ride on the kernel
#define STAGE_LAUNCH(m, kern, grid, block, lds, ...)                                                            \
  do {                                                                                                          \
    if ((m)->kev[0]) {                                                                                          \
      hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, (m)->stream, (m)->kev[0], (m)->kev[1], 0,       \
                            __VA_ARGS__);                                                                       \
      (m)->kev[0] = (m)->kev[1] = nullptr;                                                                      \
    } else {                                                                                                    \
      hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, (m)->stream, __VA_ARGS__);                         \
    }                                                                                                           \
  } while (0)

// Wait for a stream the caller is about to consume the results of: poll for a while (a blocking synchronisation
// pays ~15 us of wake-up latency even when the work is done within microseconds), then block.
// FUELMI_POLL_YIELD=1 (bench.py sets it when a fleet has fewer than two host cores per rank): polling loops hand the
// core back between looks instead of spinning on it
static inline bool poll_yields() {
  static const bool v = getenv("FUELMI_POLL_YIELD") != nullptr && atoi(getenv("FUELMI_POLL_YIELD")) != 0;
  return v;
}
static inline hipError_t stream_wait(hipStream_t s) {
  const bool yld = poll_yields();
  for (int spins = 0; spins < 200000; ++spins) {  // ~0.2 s of polling at most
    const hipError_t q = hipStreamQuery(s);
    if (q != hipErrorNotReady) return q;
    if (yld) std::this_thread::yield();
  }
  return hipStreamSynchronize(s);
}

// to be called by everything that rewrites the occupancy planes, before it queues its kernels
static inline hipError_t map_wait_plane_readers(fuelmi_map* m) {
  hipError_t e = hipSuccess;
  for (hipEvent_t ev : m->planes_read_evs) {
    const hipError_t e1 = hipStreamWaitEvent(m->stream, ev, 0);
    if (e1 != hipSuccess) e = e1;
  }
  m->planes_read_evs.clear();
  for (const fuelmi_map::LateReader& r : m->late_readers) {
    hipError_t e1 = hipEventRecord(r.ev, r.st);
    if (e1 == hipSuccess) e1 = hipStreamWaitEvent(m->stream, r.ev, 0);
    if (e1 != hipSuccess) e = e1;
    *r.sticky = true;
  }
  m->late_readers.clear();
  return e;
}
static inline void map_add_late_reader(fuelmi_map* m, hipStream_t st, hipEvent_t ev, bool* sticky) {
  for (fuelmi_map::LateReader& r : m->late_readers)
    if (r.ev == ev) {
      r.st = st;
      return;
    }
  m->late_readers.push_back({st, ev, sticky});
}
static inline void map_drop_late_reader(fuelmi_map* m, hipEvent_t ev) {
  for (size_t i = 0; i < m->late_readers.size(); ++i)
    if (m->late_readers[i].ev == ev) {
      m->late_readers.erase(m->late_readers.begin() + (long)i);
      return;
    }
}
// Writers of the distance field on the map's stream wait for the query kernels already LAUNCHED on busy query slots
// (write-after-read), under m->rw_mu so that no query is between its look at the stream and its launch.
int map_wait_query_readers(fuelmi_map* m);
// a finder has recorded `ev` behind the last kernel of its search that reads the planes
static inline void map_add_plane_reader(fuelmi_map* m, hipEvent_t ev) {
  for (hipEvent_t e : m->planes_read_evs)
    if (e == ev) return;
  m->planes_read_evs.push_back(ev);
}
static inline void map_drop_plane_reader(fuelmi_map* m, hipEvent_t ev) {
  for (size_t i = 0; i < m->planes_read_evs.size(); ++i)
    if (m->planes_read_evs[i] == ev) {
      m->planes_read_evs.erase(m->planes_read_evs.begin() + (long)i);
      return;
    }
}

// ---- device helpers ---------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ u64 plane_window(const u64* __restrict__ p, long bit) {
  // 64 bits of the plane starting at (signed) bit index `bit`
  long wi = bit >> 6;
  int sh = (int)(bit & 63);
  u64 lo = p[wi];
  if (sh == 0) return lo;
  u64 hi = p[wi + 1];
  return (lo >> sh) | (hi << (64 - sh));
}

__device__ __forceinline__ u64 bit_range(int first, int count) {
  // `count` ones starting at bit `first` (0 <= first, first+count <= 64)
  if (count <= 0) return 0ull;
  u64 ones = (count >= 64) ? ~0ull : ((1ull << count) - 1ull);
  return ones << first;
}

// mask of the bits of word w whose voxel lies inside the inclusive index box
__device__ __forceinline__ u64 box_mask_word(const Geo& g, int w, const Box3& b) {
  long a0 = 64L * w;
  if (a0 >= g.N) return 0ull;
  int line = (int)(a0 / g.nz);
  int z = (int)(a0 - (long)line * g.nz);
  int x = line / g.ny;
  int y = line - x * g.ny;
  u64 mask = 0ull;
  int bpos = 0;
  while (bpos < 64 && x < g.nx) {
    int len = min(g.nz - z, 64 - bpos);
    if (x >= b.lo[0] && x <= b.hi[0] && y >= b.lo[1] && y <= b.hi[1]) {
      int zlo = max(z, b.lo[2]);
      int zhi = min(z + len - 1, b.hi[2]);
      if (zlo <= zhi) mask |= bit_range(bpos + (zlo - z), zhi - zlo + 1);
    }
    bpos += len;
    z = 0;
    if (++y == g.ny) {
      y = 0;
      ++x;
    }
  }
  return mask;
}
__device__ __forceinline__ double dist_to_f64(float d, double res) {
  // "no source in the box": the reference stores res*sqrt(DBL_MAX) (sdf_map.cpp:196 with
  // fillESDF's DBL_MAX sentinel); the device keeps +inf in f32
  // (-inf appears only in signed mode: "+= -dist_neg + res" with dist_neg = res*sqrt(DBL_MAX))
  return isinf(d) ? copysign(res * sqrt(1.7976931348623157e308), (double)d) : (double)d;
}
// SDFMap::getDistWithGrad (sdf_map.cpp:497-536), one thread per query, f64 like the reference
__device__ __forceinline__ double get_distance_idx(const Geo& g, const float* dist, int x, int y, int z) {
  if (x < 0 || y < 0 || z < 0 || x > g.nx - 1 || y > g.ny - 1 || z > g.nz - 1) return -1.0;
  return dist_to_f64(dist[(long)x * g.nyz + (long)y * g.nz + z], g.res);
}
// getDistWithGrad split in two so that a caller can put independent work between the eight loads and
// their first use: _issue computes the cell and starts the loads, _finish interpolates.
struct DistGather {
  bool in_map;      // isInMap(pos) (:498-501)
  unsigned oob;     // corner c = 4x+2y+z lies outside the map: value -1 (getDistance)
  float raw[8];
  double diff[3];
};
__device__ __forceinline__ void dist_gather_issue(const Geo& g, const float* __restrict__ dist, const double pos[3],
                                                  DistGather& G) {
  G.in_map = true;
  G.oob = 0u;
  for (int k = 0; k < 3; ++k)
    if (pos[k] < g.minb[k] + 1e-4 || pos[k] > g.maxb[k] - 1e-4) G.in_map = false;
  if (!G.in_map) return;
  int idx[3];
  for (int k = 0; k < 3; ++k) {
    double pm = pos[k] - 0.5 * g.res * 1.0;
    idx[k] = (int)floor((pm - g.org[k]) * g.res_inv);
    double ip = (idx[k] + 0.5) * g.res + g.org[k];
    G.diff[k] = (pos[k] - ip) * g.res_inv;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int x = idx[0] + (c >> 2), y = idx[1] + ((c >> 1) & 1), z = idx[2] + (c & 1);
    const bool out = x < 0 || y < 0 || z < 0 || x > g.nx - 1 || y > g.ny - 1 || z > g.nz - 1;
    G.raw[c] = 0.0f;
    if (out)
      G.oob |= 1u << c;
    else
      G.raw[c] = dist[(long)x * g.nyz + (long)y * g.nz + z];
  }
}
__device__ __forceinline__ double dist_gather_finish(const Geo& g, const DistGather& G, double grad[3]) {
  if (!G.in_map) {
    grad[0] = grad[1] = grad[2] = 0.0;
    return 0.0;
  }
  double v[2][2][2];
#pragma unroll
  for (int c = 0; c < 8; ++c) v[c >> 2][(c >> 1) & 1][c & 1] = ((G.oob >> c) & 1u) ? -1.0 : dist_to_f64(G.raw[c], g.res);
  const double* diff = G.diff;
  double v00 = (1 - diff[0]) * v[0][0][0] + diff[0] * v[1][0][0];
  double v01 = (1 - diff[0]) * v[0][0][1] + diff[0] * v[1][0][1];
  double v10 = (1 - diff[0]) * v[0][1][0] + diff[0] * v[1][1][0];
  double v11 = (1 - diff[0]) * v[0][1][1] + diff[0] * v[1][1][1];
  double v0 = (1 - diff[1]) * v00 + diff[1] * v10;
  double v1 = (1 - diff[1]) * v01 + diff[1] * v11;
  double d = (1 - diff[2]) * v0 + diff[2] * v1;
  grad[2] = (v1 - v0) * g.res_inv;
  grad[1] = ((1 - diff[2]) * (v10 - v00) + diff[2] * (v11 - v01)) * g.res_inv;
  double g0 = (1 - diff[2]) * (1 - diff[1]) * (v[1][0][0] - v[0][0][0]);
  g0 += (1 - diff[2]) * diff[1] * (v[1][1][0] - v[0][1][0]);
  g0 += diff[2] * (1 - diff[1]) * (v[1][0][1] - v[0][0][1]);
  g0 += diff[2] * diff[1] * (v[1][1][1] - v[0][1][1]);
  grad[0] = g0 * g.res_inv;
  return d;
}
__device__ __forceinline__ double dist_with_grad_dev(const Geo& g, const float* __restrict__ dist, const double pos[3],
                                     double grad[3]) {
  DistGather G;
  dist_gather_issue(g, dist, pos, G);
  return dist_gather_finish(g, G, grad);
}
#endif  // __HIPCC__

// ---- kernels' host launchers (defined in the .hip files) ---------------------------------------
int esdf_update(fuelmi_map* m);
int insert_points(fuelmi_map* m, const float* xyz, int stride_bytes, int n, const double cam[3]);

#endif
